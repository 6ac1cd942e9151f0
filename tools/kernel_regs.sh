#!/bin/bash
# Register / LDS / occupancy report of every kernel in one csrc file (cross-compiles, no GPU needed):
#   tools/kernel_regs.sh attn_h_bwd.hip [extra hipcc flags]
F=$1; shift
EXTRA=$(python3 - <<PY
import sys
sys.path.insert(0, ".")
from dgdm_histopath_lab_amd import _build
print(" ".join(_build.EXTRA_FLAGS.get("$F", [])))
PY
)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -I include -I dgdm_histopath_lab_amd/csrc $EXTRA "$@" \
  -Rpass-analysis=kernel-resource-usage -c dgdm_histopath_lab_amd/csrc/$F -o /tmp/kr_$$.o 2>&1 | python3 -c '
import re, sys
cur = None
for line in sys.stdin:
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark: (.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur: print(cur)
        n = t.split(":", 1)[1].strip()
        n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n); n = re.sub(r"^_ZL\d+", "", n)
        cur = n[:44].ljust(46)
    elif cur and any(t.startswith(k) for k in ("VGPRs:", "AGPRs:", "ScratchSize", "Occupancy", "LDS Size", "SGPRs:")):
        cur += t.replace(" [bytes/lane]", "").replace(" [bytes/block]", "").replace(" [waves/SIMD]", "") + "  "
if cur: print(cur)
'
rm -f /tmp/kr_$$.o

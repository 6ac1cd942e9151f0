#!/bin/bash
# A second build of the library with extra compiler defines, for same-box A/B runs through tools/run_with_lib.py:
#   bash tools/build_variant_lib.sh w248 -DDGDM_TN_WANT=248   ->  dgdm_histopath_lab_amd/lib/w248/libdgdm_hip.so
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/dgdm_histopath_lab_amd/lib/$NAME; mkdir -p $OUT/obj
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function $* -I $ROOT/include -I $ROOT/dgdm_histopath_lab_amd/csrc"
for f in $ROOT/dgdm_histopath_lab_amd/csrc/*.hip; do
  b=$(basename $f .hip); extra=""
  case $b in attn_h_bwd) extra="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans";; attn_h_fwd) extra="-fno-honor-nans";;
             attn_h_bwd_fused) extra="${FUSED_EXTRA--mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans}";; esac
  /opt/rocm/bin/hipcc $FLAGS $extra -c $f -o $OUT/obj/$b.o 2>/dev/null &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libdgdm_hip.so $OUT/obj/*.o && echo built $OUT/libdgdm_hip.so

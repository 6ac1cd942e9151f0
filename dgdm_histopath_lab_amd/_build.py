"""Builds libdgdm_hip.so (hipcc, --offload-arch=gfx950) in-tree from csrc/*.hip.

Cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
gpurun snapshot.  ``python -m dgdm_histopath_lab_amd._build`` or ``__graft_entry__.build()``.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdgdm_hip.so")
# diagnostic twin of the library: the GEMM kernels that issue their loads as inline asm, built with -DDGDM_STAGE_CANARY (their staging
# registers hold NaN until a load lands; tests/test_hip_gemm_img.py runs them).  Never loaded by the product path.
CANARY_PATH = os.path.join(LIB_DIR, "canary", "libdgdm_hip.so")
CANARY_SOURCES = ("gemm_img.hip", "gemm_h.hip")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def _hipcc() -> str:
    cand = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(cand):
        raise RuntimeError("hipcc not found (need ROCm to build libdgdm_hip.so)")
    return cand


# Per-file extra flags.  The split-fp16 attention backward kernels consume every MFMA result on the VALU right away; with
# the accumulators in AGPRs (the compiler's choice under their register pressure) each value costs a v_accvgpr_read and
# each C operand a v_accvgpr_write -- 272 of the 1330 VALU instructions of the dQ loop.  Keeping the MFMA operands in
# architected VGPRs removes them (235 / 210 VGPRs, no spills).
# -fno-honor-nans: fmaxf on MFMA results otherwise gets a canonicalising v_max_f32 x, x per operand (masked scores are -1e30, never
# NaN or inf).
EXTRA_FLAGS = {"attn_h_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans"],
               "attn_h_bwd_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans"],
               "attn_h_fwd.hip": ["-fno-honor-nans"]}


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest() -> str:
    # the include paths enter relative to the repository: the snapshot on a GPU box lives under another root, and a digest over absolute
    # paths made every box rebuild the library it had just received
    h = hashlib.sha256((" ".join(f.replace(ROOT, "<root>") for f in FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
    for root in (CSRC, os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(root)):
            if f.endswith((".hip", ".hpp", ".h")):
                h.update(f.encode()); h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    stamp = os.path.join(LIB_DIR, "build.sha256")
    dig = _digest()
    if (not force and os.path.exists(LIB_PATH) and os.path.exists(CANARY_PATH) and os.path.exists(stamp)
            and open(stamp).read().strip() == dig):
        return LIB_PATH
    hipcc = _hipcc()
    objdir = os.path.join(LIB_DIR, "obj")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src, canary=False):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + (".canary.o" if canary else ".o"))
        cmd = [hipcc, *FLAGS, *(["-DDGDM_STAGE_CANARY"] if canary else []), *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, _sources()))
        cobjs = list(ex.map(lambda f: compile_one(os.path.join(CSRC, f), True), CANARY_SOURCES))
    os.makedirs(os.path.dirname(CANARY_PATH), exist_ok=True)
    swapped = [o for o in objs if os.path.basename(o)[:-2] + ".hip" not in CANARY_SOURCES] + cobjs
    for out, files in ((LIB_PATH, objs), (CANARY_PATH, swapped)):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *files], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    open(stamp, "w").write(dig)
    if verbose:
        print(f"built {LIB_PATH}")
    return LIB_PATH


# ---- sanitizer twin (host code only): tests/test_abi.py runs the argument checks, workspace-size arithmetic and descriptor packing of
# every entry point under AddressSanitizer + UndefinedBehaviorSanitizer ON THE CPU (GPU ASan / XNACK are not available on this
# pool; hipcc ignores -fsanitize for the gfx950 side, the host side of every .hip file is instrumented).  Never loaded by the product.
ASAN_PATH = os.path.join(LIB_DIR, "asan", "libdgdm_hip.so")
ASAN_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize=vptr", "-fno-sanitize-recover=undefined",
              "-shared-libsan", "-Wno-option-ignored"]


def asan_runtime() -> str:
    """Path of the shared ASan runtime of ROCm's clang (to LD_PRELOAD into the python that loads the twin), or ''."""
    r = subprocess.run([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    cand = r.stdout.strip()
    if os.path.isabs(cand) and os.path.exists(cand):
        return cand
    import glob
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return hits[0] if hits else ""


def build_sanitized(verbose: bool = False) -> str:
    os.makedirs(os.path.dirname(ASAN_PATH), exist_ok=True)
    stamp = os.path.join(os.path.dirname(ASAN_PATH), "build.sha256")
    dig = _digest() + hashlib.sha256(" ".join(ASAN_FLAGS).encode()).hexdigest()
    if os.path.exists(ASAN_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return ASAN_PATH
    hipcc = _hipcc()
    objdir = os.path.join(os.path.dirname(ASAN_PATH), "obj")
    os.makedirs(objdir, exist_ok=True)
    base = [f for f in FLAGS if f != "-O3"]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([hipcc, *base, *ASAN_FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc (sanitized) failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, _sources()))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-o", ASAN_PATH, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link (sanitized) failed:\n{r.stdout}\n{r.stderr}")
    open(stamp, "w").write(dig)
    if verbose:
        print(f"built {ASAN_PATH}")
    return ASAN_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)

"""GEMM microbenchmark at the model's shapes: bf16x3 (exact 3-way bf16 split, 6 MFMAs / product) vs f16x2 (fp16 hi+lo, 3 MFMAs /
product, amax slots already filled -- as in a step, where producers / the first GEMM on a tensor fill them) vs the fp32-MFMA
kernels.  Alternating launches on the same box (box-to-box spread of these kernels is up to 7 %)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import ops

def t(fn, iters=40):
    """Device time per call: the calls are recorded into one HIP graph (the Python wrapper costs ~15 us per call, more than the
    small shapes run) and the graph is replayed; includes the ~1.5 us boundary between dependent launches, as in a step."""
    for _ in range(3): fn()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (3 * iters)

dev = "cuda:0"
shapes = [(40000, 768, 512), (40000, 512, 512), (40000, 544, 512), (40000, 544, 256), (40000, 288, 256), (40000, 160, 128), (40000, 128, 128),
          (40000, 128, 384), (40000, 512, 256), (40000, 256, 128), (20000, 160, 128), (10000, 160, 128), (5000, 160, 128)]
if os.environ.get("SHAPES") == "unet":     # the narrow kernel's shapes at the U-Net's levels (N <= 128)
    shapes = [(m, k, n) for m in (40000, 20000, 10000, 5000) for (k, n) in ((160, 128), (128, 128), (128, 64))]
maths = os.environ.get("MATHS", "bf16x3,f16x2reg,f16x2").split(",")   # f16x2reg: the register-staged kernels (weight images off)


def run(fn, mt):
    if mt == "f16x2reg":
        ops.USE_WEIGHT_IMAGES = False
        try:
            return fn("f16x2")
        finally:
            ops.USE_WEIGHT_IMAGES = True
    return fn(mt)


tot = {m: 0.0 for m in maths}
for (m, k, n) in shapes:
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev); gy = torch.randn(m, n, device=dev)
    for ten in (x, w, gy):
        ops.ensure_amax(ten)
    fl = 2.0 * m * k * n
    r = dict(M=m, K=k, N=n)
    for name, fn in (("nt", lambda mt: ops.gemm_nt_raw(x, w, b, math=mt)), ("nn", lambda mt: ops.gemm_nn_raw(gy, w, math=mt)),
                     ("tn", lambda mt: ops.gemm_tn_raw(gy, x, True, math=mt))):
        cells = []
        for mt in maths:
            us = t(lambda: run(fn, mt))
            tot[mt] += us
            cells.append(f"{mt} {us:6.1f}us {fl/us/1e6:5.1f}TF")
        r[name] = " | ".join(cells)
    print(json.dumps(r))
print(json.dumps({"sum_us": {k: round(v, 1) for k, v in tot.items()}}))

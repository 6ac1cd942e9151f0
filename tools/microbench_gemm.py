"""fp32 GEMM microbenchmark: own MFMA kernels vs the library path (torch -> hipBLASLt) at the model's shapes."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import ops

def t(fn, iters=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters

dev = "cuda:0"
shapes = [(40000, 768, 512), (40000, 512, 512), (40000, 544, 512), (40000, 544, 256), (40000, 288, 256), (40000, 160, 128), (40000, 128, 128),
          (40000, 128, 384), (40000, 128, 512), (40000, 512, 256), (40000, 256, 128), (20000, 160, 128), (10000, 160, 128), (5000, 160, 128)]
for (m, k, n) in shapes:
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev); gy = torch.randn(m, n, device=dev)
    fl = 2.0 * m * k * n
    r = dict(M=m, K=k, N=n)
    for name, ours, x3, lib in (
            ("nt", lambda: ops.gemm_nt_raw(x, w, b), lambda: ops.gemm_nt_raw(x, w, b, math="bf16x3"), lambda: torch.nn.functional.linear(x, w, b)),
            ("nn", lambda: ops.gemm_nn_raw(gy, w), lambda: ops.gemm_nn_raw(gy, w, math="bf16x3"), lambda: gy @ w),
            ("tn", lambda: ops.gemm_tn_raw(gy, x, True), lambda: ops.gemm_tn_raw(gy, x, True, math="bf16x3"), lambda: (gy.t() @ x, gy.sum(0)))):
        a, c, l = t(ours), t(x3), t(lib)
        r[name] = f"fp32 {a:6.1f}us {fl/a/1e6:5.1f}TF | bf16x3 {c:6.1f}us {fl/c/1e6:5.1f}TF | lib {l:6.1f}us {fl/l/1e6:5.1f}TF"
    print(json.dumps(r))

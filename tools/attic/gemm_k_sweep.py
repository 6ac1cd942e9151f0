"""Fixed vs per-stage cost of the bf16x3 NT GEMM: time over K at fixed M, N (back-to-back launches, events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import ops
def t(fn, reps=30):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for m, n in ((40000, 128), (20000, 128), (5000, 128), (40000, 512)):
    row = []
    for k in (16, 32, 64, 160, 320, 640):
        x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda"); y = torch.empty(m, n, device="cuda")
        row.append((k, t(lambda: ops.gemm_nt_raw(x, w, None, out=y, math="bf16x3"))))
    print(f"M={m} N={n}: " + "  ".join(f"K={k}: {us:6.1f}us" for k, us in row), flush=True)
y = torch.empty(40000, 128, device="cuda"); z = torch.randn(40000, 128, device="cuda")
print("copy 40000x128 (20 MB read + 20 MB write):", round(t(lambda: y.copy_(z)), 1), "us")

"""Runs the three bf16x3 contractions a few times at one shape (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import ops
m, k, n = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (40000, 768, 512)))
x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda"); b = torch.randn(n, device="cuda"); gy = torch.randn(m, n, device="cuda")
for _ in range(3):
    ops.gemm_nt_raw(x, w, b, math="bf16x3"); ops.gemm_nn_raw(gy, w, math="bf16x3"); ops.gemm_tn_raw(gy, x, True, math="bf16x3")
torch.cuda.synchronize()

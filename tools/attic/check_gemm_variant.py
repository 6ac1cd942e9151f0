"""Values of the row GEMMs of whatever build tools/run_with_lib.py loaded against float64 (a variant build has no test suite of its own)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import ops
torch.manual_seed(0)
worst = 0.0
for (m, k, n) in ((5000, 128, 128), (5000, 160, 64), (257, 128, 128), (4999, 144, 100), (20000, 128, 128), (40000, 160, 128), (300, 16, 4)):
    x = torch.randn(m, k, device="cuda:0"); w = torch.randn(n, k, device="cuda:0"); b = torch.randn(n, device="cuda:0")
    y = ops.gemm_nt_raw(x, w, b, math="f16x2")
    ref = x.double() @ w.double().t() + b.double()
    acc = torch.randn(m, n, device="cuda:0"); acc0 = acc.clone()
    ops.gemm_nt_raw(x, w, b, out=acc, accumulate=True, math="f16x2")
    e1 = ((y.double() - ref).norm() / ref.norm()).item(); e2 = ((acc.double() - (ref + acc0.double())).norm() / ref.norm()).item()
    worst = max(worst, e1, e2)
    print(f"M {m:6d} K {k:4d} N {n:4d}: rel-L2 {e1:.2e}, accumulate {e2:.2e}")
assert worst < 3e-7, worst
print("ok")
# dW = dy^T x and db = colsum(dy) (the split-M kernels), ragged row counts and column tails
worst = 0.0
for (m, k, n) in ((5000, 128, 128), (40000, 160, 128), (40000, 768, 512), (4999, 144, 100), (257, 128, 128), (33, 64, 32), (20001, 512, 256)):
    x = torch.randn(m, k, device="cuda:0"); gy = torch.randn(m, n, device="cuda:0")
    dW, db = ops.gemm_tn_raw(gy, x, True, math="f16x2")
    rW = gy.double().t() @ x.double(); rb = gy.double().sum(0)
    e1 = ((dW.double() - rW).norm() / rW.norm()).item(); e2 = ((db.double() - rb).norm() / rb.norm()).item()
    dW2, db2 = ops.gemm_tn_raw(gy, x, True, math="f16x2")
    same = torch.equal(dW, dW2) and torch.equal(db, db2)
    worst = max(worst, e1, e2)
    print(f"tn M {m:6d} K {k:4d} N {n:4d}: dW rel-L2 {e1:.2e}, db {e2:.2e}, repeatable {same}")
    assert same
assert worst < 1e-6, worst
print("tn ok")

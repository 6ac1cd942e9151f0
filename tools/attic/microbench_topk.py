"""Device time of one exact top-k selection (ops.topk_perm) at the sizes of a training step, recorded into a HIP graph as in a step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import ops
dev = "cuda:0"
for n in (40000, 20000, 10000, 5000, 50000, 100000):
    s = torch.randn(n, device=dev)
    k = n // 2
    for _ in range(3): ops.topk_perm(s, k)
    g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20): out = ops.topk_perm(s, k)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    print(f"N {n:6d} k {k:6d}: {a.elapsed_time(b) * 1e3 / 100:6.1f} us per selection")

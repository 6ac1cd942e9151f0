"""Row-count sweep of the wide weight-image GEMM (k_gemm_img8): how much of a launch is the workgroup-count quantisation
(128-row x 256-column workgroups, two resident per CU = 512 slots)?  Graph-timed as tools/microbench_gemm.py."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import ops


def t(fn, iters=40):
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
for (k, n) in ((768, 512), (512, 256), (512, 512)):
    w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
    ops.ensure_amax(w)
    for m in (16384, 24576, 32768, 36864, 40000, 49152, 65536, 81920):
        x = torch.randn(m, k, device=dev)
        ops.ensure_amax(x)
        us = t(lambda: ops.gemm_nt_raw(x, w, b, math="f16x2"))
        wgs = ((m + 127) // 128) * ((n + 255) // 256)
        print(json.dumps(dict(M=m, K=k, N=n, workgroups=wgs, rounds=round(wgs / 512, 2), us=round(us, 1), TF_eq=round(2.0 * m * k * n / us / 1e6, 1),
                              us_per_1k_rows=round(us / m * 1000, 2))))

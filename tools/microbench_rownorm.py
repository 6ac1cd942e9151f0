"""Row-norm kernels at the model's shapes: device time per call from a HIP graph of 40 calls (tools/microbench_gemm.py's method).
Run against another build with tools/run_with_lib.py for a same-box A/B."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import _lib, ops


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
lib = _lib.load()
tot = 0.0
for (n, c, G, act, p, with_res) in [(40000, 128, 1, 0, 0.0, True), (40000, 128, 1, 1, 0.1, False), (20000, 128, 1, 0, 0.0, True), (10000, 128, 1, 0, 0.0, True),
                                     (5000, 128, 1, 0, 0.0, True), (40000, 512, 8, 3, 0.1, False), (40000, 256, 8, 3, 0.1, False), (40000, 128, 8, 3, 0.1, False),
                                     (40000, 256, 1, 1, 0.1, False), (40000, 512, 1, 1, 0.1, False)]:
    x = torch.randn(n, c, device=dev); res = torch.randn(n, c, device=dev) if with_res else None
    gam = torch.ones(c, device=dev); bet = torch.zeros(c, device=dev); dy = torch.randn(n, c, device=dev)
    y = torch.empty(n, c, device=dev); dx = torch.empty(n, c, device=dev); mean = torch.empty(n * G, device=dev); rstd = torch.empty(n * G, device=dev)
    dg = torch.empty(c, device=dev); db = torch.empty(c, device=dev)
    wsb = lib.dgdm_rownorm_bwd_workspace_bytes(n, c, G)
    ws = torch.empty(wsb // 4, device=dev)
    sp = lambda: _lib.stream_ptr(x.device)
    fwd = lambda: _lib.check(lib.dgdm_rownorm_fwd(x.data_ptr(), _lib.ptr(res), gam.data_ptr(), bet.data_ptr(), n, c, G, 1e-5, act, p, 7, y.data_ptr(),
                                                   mean.data_ptr(), rstd.data_ptr(), None, sp()), "fwd")
    bwd = lambda: _lib.check(lib.dgdm_rownorm_bwd(x.data_ptr(), _lib.ptr(res), gam.data_ptr(), bet.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(),
                                                   n, c, G, act, p, 7, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, None, sp()), "bwd")
    f, b = t(fwd), t(bwd)
    tot += f + b
    print(json.dumps(dict(N=n, C=c, G=G, act=act, p=p, res=with_res, fwd_us=round(f, 1), bwd_us=round(b, 1))))
print(json.dumps({"sum_us": round(tot, 1)}))

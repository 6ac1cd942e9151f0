"""Fused epilogues against the launches they replace, at the narrow layers' shapes (U-Net levels) and two wide ones; device time per
call from a HIP graph of 40 calls (tools/microbench_gemm.py's method).  Run against another build with tools/run_with_lib.py
(e.g. lib/nows: -DDGDM_NO_WS, the same epilogues on k_gemm_img instead of the weight-stationary kernel)."""
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
shapes = [(m, k, n) for m in (40000, 20000, 10000, 5000) for (k, n) in ((160, 128), (128, 128))] + [(40000, 544, 512), (40000, 288, 256)]
P, SEED = 0.1, 12345
for (m, k, n) in shapes:
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    gy = torch.randn(m, n, device=dev); res = torch.randn(m, n, device=dev); gam = torch.ones(n, device=dev); bet = torch.zeros(n, device=dev)
    pre = torch.randn(m, n, device=dev)
    for ten in (x, w, gy):
        ops.ensure_amax(ten)
    e0, e1 = ops.WEIGHT_IMAGES.get(0, w), ops.WEIGHT_IMAGES.get(1, w)
    y = torch.empty(m, n, device=dev); y2 = torch.empty(m, n, device=dev); mean = torch.empty(m, device=dev); rstd = torch.empty(m, device=dev)
    sp = lambda: _lib.stream_ptr(x.device)        # at call time: inside a capture the current stream is the capturing one
    act_k = lambda src, dst: _lib.check(lib.dgdm_act_dropout_fwd(src.data_ptr(), src.numel(), 1, P, SEED, dst.data_ptr(), None, None, sp()), "act")
    actb_k = lambda src, g, dst: _lib.check(lib.dgdm_act_dropout_bwd(src.data_ptr(), g.data_ptr(), g.numel(), 1, P, SEED, dst.data_ptr(), None, None, sp()), "actb")
    norm_k = lambda src, dst: _lib.check(lib.dgdm_rownorm_fwd(src.data_ptr(), res.data_ptr(), gam.data_ptr(), bet.data_ptr(), m, n, 1, 1e-5, 0, 0.0, 0,
                                                              dst.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, sp()), "norm")
    r = dict(M=m, K=k, N=n)
    r["act kernel alone"] = round(t(lambda: act_k(y, y2)), 1)
    r["act_bwd kernel alone"] = round(t(lambda: actb_k(pre, gy, y2)), 1)
    r["gemm"] = round(t(lambda: ops._gemm_rows_img(x, e0, 0, n, b, y, False)), 1)
    r["gemm+act (2 launches)"] = round(t(lambda: (ops._gemm_rows_img(x, e0, 0, n, b, y, False), act_k(y, y2))), 1)
    r["gemm_act (fused)"] = round(t(lambda: ops.gemm_img_act_raw(x, e0, n, b, 1, P, SEED)), 1)
    dxo = torch.empty(m, k, device=dev)
    if k == n or True:
        r["gemm_nn+act_bwd (2)"] = round(t(lambda: (ops._gemm_rows_img(gy, e1, 0, k, None, dxo, False), actb_k(x, dxo, dxo))), 1)
        r["gemm_act_bwd (fused)"] = round(t(lambda: ops.gemm_img_act_bwd_raw(gy, e1, k, x, 1, P, SEED)), 1)
    if ops.gemm_img_norm_supported(n, 1):
        r["gemm+norm (2)"] = round(t(lambda: (ops._gemm_rows_img(x, e0, 0, n, b, y, False), norm_k(y, y2))), 1)
        r["gemm_norm (fused)"] = round(t(lambda: ops.gemm_img_norm_raw(x, e0, n, b, res, gam, bet, 1, 1e-5)), 1)
    print(json.dumps(r))

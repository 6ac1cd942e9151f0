"""Where do the ~15 us of a small GEMM go?  Device time per launch (graph replay of 40 dependent launches) of: a trivial fill, and
the weight-image GEMM at a few (M, K) with N = 128 -- fixed cost (launch boundary + prologue + epilogue) vs per-stage cost."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import ops, _lib

def t(fn, iters=40):
    for _ in range(3): fn()
    g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * iters)

dev = "cuda:0"
lib = _lib.load()
buf = torch.zeros(1 << 20, dtype=torch.int32, device=dev)
print("fill 64 words        %6.2f us" % t(lambda: lib.dgdm_fill_u32(buf.data_ptr(), 64, 0, _lib.stream_ptr(buf.device))))
print("fill 1M words        %6.2f us" % t(lambda: lib.dgdm_fill_u32(buf.data_ptr(), 1 << 20, 0, _lib.stream_ptr(buf.device))))
for (m, k, n) in [(128, 32, 128), (128, 160, 128), (128, 1024, 128), (5000, 32, 128), (5000, 160, 128), (5000, 1024, 128), (40000, 32, 128), (40000, 160, 128),
                  (32768, 160, 128), (65536, 160, 128), (40000, 160, 256), (40000, 768, 512)]:
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); out = torch.empty(m, n, device=dev)
    ops.ensure_amax(x); ops.ensure_amax(w)
    e = ops.WEIGHT_IMAGES.get(0, w)
    ax = ops.amax_of(x); sp = lambda: _lib.stream_ptr(x.device)
    f = lambda: lib.dgdm_gemm_rows_img(x.data_ptr(), k, m, k, e.img.data_ptr(), e.tiles, 0, n, None, out.data_ptr(), n, 0, ax, sp())
    us = t(f)
    aw = ops.amax_of(w)
    f2 = lambda: lib.dgdm_gemm_nt_f16x2(x.data_ptr(), k, w.data_ptr(), k, None, out.data_ptr(), n, m, n, k, 0, ax, aw, sp())
    us2 = t(f2)
    print("M %6d K %5d N %4d   img %7.2f us   reg %7.2f us   (%.1f TF img)" % (m, k, n, us, us2, 2.0 * m * k * n / us / 1e6))

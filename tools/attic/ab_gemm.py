"""A/B of two builds of libdgdm_hip.so on the same box, alternating launches (box-to-box clock differences are larger than
most kernel changes).  usage: python tools/ab_gemm.py <libA.so> <libB.so>"""
import ctypes as C, sys, torch
libs = [C.CDLL(p) for p in sys.argv[1:3]]
P, I64, I32 = C.c_void_p, C.c_int64, C.c_int32
for l in libs:
    l.dgdm_gemm_nt_bf16x3.argtypes = [P, I64, P, I64, P, P, I64, I32, I32, I32, I32, P]
    l.dgdm_gemm_nn_bf16x3.argtypes = [P, I64, P, I64, P, I64, I32, I32, I32, I32, P]
    l.dgdm_gemm_tn_bf16x3.argtypes = [P, I64, P, I64, P, I64, P, I32, I32, I32, P, C.c_size_t, P]
    l.dgdm_gemm_tn_bf16x3_workspace_bytes.restype = C.c_size_t
    l.dgdm_gemm_tn_bf16x3_workspace_bytes.argtypes = [I32, I32, I32, I32]
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (m, k, n) in [(40000, 768, 512), (40000, 544, 512), (40000, 288, 256), (40000, 160, 128), (20000, 160, 128)]:
    x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda"); b = torch.randn(n, device="cuda"); gy = torch.randn(m, n, device="cuda")
    y = torch.empty(m, n, device="cuda"); dx = torch.empty(m, k, device="cuda"); dw = torch.empty(n, k, device="cuda"); db = torch.empty(n, device="cuda")
    wsb = libs[0].dgdm_gemm_tn_bf16x3_workspace_bytes(m, n, k, 1); ws = torch.empty(wsb // 4 + 1, device="cuda")
    res = {}
    for kind in ("nt", "nn", "tn"):
        def call(l):
            if kind == "nt": return lambda: l.dgdm_gemm_nt_bf16x3(x.data_ptr(), k, w.data_ptr(), k, b.data_ptr(), y.data_ptr(), n, m, n, k, 0, st)
            if kind == "nn": return lambda: l.dgdm_gemm_nn_bf16x3(gy.data_ptr(), n, w.data_ptr(), k, dx.data_ptr(), k, m, n, k, 0, st)
            return lambda: l.dgdm_gemm_tn_bf16x3(gy.data_ptr(), n, x.data_ptr(), k, dw.data_ptr(), k, db.data_ptr(), m, n, k, ws.data_ptr(), wsb, st)
        fa, fb = call(libs[0]), call(libs[1])
        for f in (fa, fb): f(); f()
        ta = tb = 0.0
        for _ in range(5):
            ta += timeit(fa); tb += timeit(fb)
        res[kind] = (ta / 5, tb / 5)
    print(f"{m}x{k}->{n}: " + "  ".join(f"{kd} A {a:7.1f} us  B {b_:7.1f} us ({(b_ / a - 1) * 100:+.1f}%)" for kd, (a, b_) in res.items()), flush=True)
# accuracy of both builds against float64 on one shape
m, k, n = 4000, 544, 512
x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") / k ** 0.5; y = torch.empty(m, n, device="cuda")
ref = x.double() @ w.double().t()
for name, l in zip("AB", libs):
    l.dgdm_gemm_nt_bf16x3(x.data_ptr(), k, w.data_ptr(), k, None, y.data_ptr(), n, m, n, k, 0, st); torch.cuda.synchronize()
    print(name, "NT max abs err vs fp64: %.3e  (max |ref| %.2f)" % (float((y.double() - ref).abs().max()), float(ref.abs().max())))

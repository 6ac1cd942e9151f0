"""Where does one key-block iteration of k_attn_h_bwd_dkv spend its time?  Needs the diagnostic build (tools/build_stamps_lib.sh).
Prints the stamps of workgroup (40, 0), thread 0, iteration 40 as cycles since the iteration's first stamp and ns (s_memrealtime)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "stamps", "libdgdm_hip.so")
from dgdm_histopath_lab_amd import ops
dev = "cuda:0"
B, n, H = 4, 10000, 8
plan = ops.AttnPlan([i * n for i in range(B + 1)], dev)
Cc = H * 16
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B * n, 3 * Cc, device=dev, generator=g); pos = torch.rand(B * n, 2, device=dev, generator=g)
gout = torch.randn(B * n, Cc, device=dev, generator=g)
stamps = torch.zeros(64, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.dgdm_debug_set_attn_stamps.argtypes = [C.c_void_p]
assert lib.dgdm_debug_set_attn_stamps(stamps.data_ptr()) == 0
names = {0: "iteration top", 1: "bias done", 2: "head 0: S, dP, exp, dropout issued", 3: "head 0: dV, dK MFMAs issued", 4: "head 1: S, dP, exp, dropout issued",
         5: "head 1: dV, dK MFMAs issued", 10: "before barrier 1", 11: "after barrier 1", 12: "next block's DMA issued", 13: "after barrier 2 (DMA landed)"}
for p in (0.1, 0.0):
    outh, lse2_b, pk = ops.spatial_attn_h_fwd_raw(qkv, pos, plan, H, 0.25, 1.0, p, 123)
    dq = torch.empty_like(qkv)
    for rep in range(3):
        stamps.zero_()
        ops.spatial_attn_h_bwd_raw(pk, outh, gout, plan, H, 0.25, 1.0, lse2_b, dq, p, 123)
        torch.cuda.synchronize()
    s = stamps.cpu().tolist()
    print(f"dropout {p}:")
    t0, r0 = s[0], s[1]
    for i in sorted(names):
        if s[2 * i]:
            print(f"   {names[i]:42s} {s[2 * i] - t0:7d} cycles   {(s[2 * i + 1] - r0) * 10:6d} ns")

"""Attention kernel microbenchmark at the bench workload (4 x 10k nodes, H=8): fwd / dq / dkv, with and without dropout."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import ops
ops.ATTN_BWD_FUSED = False        # the variant loops below time the two-pass kernels; the one-pass backward has its own section
dev = "cuda:0"
B, n, H = int(os.environ.get("B", 4)), int(os.environ.get("N", 10000)), 8
ptr = [i * n for i in range(B + 1)]
plan = ops.AttnPlan(ptr, dev)
C = H * 16
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B * n, 3 * C, device=dev, generator=g)
pos = torch.rand(B * n, 2, device=dev, generator=g)
gout = torch.randn(B * n, C, device=dev, generator=g)
fl = 2.0 * B * n * n * H * 16
def t(fn, iters=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
variants = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]
for p in (0.0, 0.1):
    for var in ([] if os.environ.get("SPLIT_ONLY") else variants):
        ms = t(lambda: ops.spatial_attn_fwd_raw(qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:], pos, plan, H, 0.25, 1.0, var, p, 123))
        print(json.dumps(dict(kernel="fwd", variant=var, drop=p, ms=round(ms, 3), TF=round(2 * fl / ms / 1e9, 1))))
    ms = t(lambda: ops.attn_pack(qkv, 0, C, 3, 0.25 * ops.LOG2E, plan, H, pos=pos, pos_scale=ops.LOG2E))
    print(json.dumps(dict(kernel="attn_pack qkv", ms=round(ms, 3))))
    packed = ops.attn_pack(qkv, 0, C, 3, 0.25 * ops.LOG2E, plan, H, pos=pos, pos_scale=ops.LOG2E)
    for var in (1, 2, 3):
        ms = t(lambda: ops.spatial_attn_h_fwd_raw(qkv, pos, plan, H, 0.25, 1.0, p, 123, packed, var))
        print(json.dumps(dict(kernel="fwd split-fp16", variant=var, drop=p, ms=round(ms, 3), TF=round(2 * fl / ms / 1e9, 1))))
    if os.environ.get("FWD_ONLY"): continue
    outh, lse2_b, pk = ops.spatial_attn_h_fwd_raw(qkv, pos, plan, H, 0.25, 1.0, p, 123, packed)
    dq2 = torch.empty_like(qkv)
    for var in [int(v) for v in os.environ.get("BWD_VARIANTS", "0,3,4,5").split(",")]:
        ops.TIMERS.start()
        for _ in range(6):
            ops.spatial_attn_h_bwd_raw(pk, outh, gout, plan, H, 0.25, 1.0, lse2_b, dq2, p, 123, var, var)
        torch.cuda.synchronize(); ops.TIMERS.stop()
        for k, (cnt, ms) in ops.TIMERS.summary().items():
            prod = 3 if k.endswith("dq") else 4
            print(json.dumps(dict(kernel=k + " split-fp16", variant=var, drop=p, ms=round(ms, 3), TF=round(prod * fl / ms / 1e9, 1))))
    # the one-pass backward (csrc/attn_h_bwd_fused.hip): dQ + dK + dV in one key-stationary launch + the partial-tile reduction
    ops.ATTN_BWD_FUSED = True
    ops.TIMERS.start()
    for _ in range(6):
        ops.spatial_attn_h_bwd_raw(pk, outh, gout, plan, H, 0.25, 1.0, lse2_b, dq2, p, 123)
    torch.cuda.synchronize(); ops.TIMERS.stop()
    ops.ATTN_BWD_FUSED = False
    for k, (cnt, ms) in ops.TIMERS.summary().items():
        print(json.dumps(dict(kernel=k + " (one pass: dQ + dK + dV + reduction)", drop=p, ms=round(ms, 3), TF=round(7 * fl / ms / 1e9, 1))))
    if os.environ.get("SPLIT_ONLY"): continue
    out, lse2 = ops.spatial_attn_fwd_raw(qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:], pos, plan, H, 0.25, 1.0, 0, p, 123)
    dqkv = torch.empty_like(qkv)
    ops.TIMERS.start()
    for _ in range(6):
        ops.spatial_attn_bwd_raw(qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:], out, gout, pos, plan, H, 0.25, 1.0, lse2, dqkv, p, 123)
    torch.cuda.synchronize(); ops.TIMERS.stop()
    for k, (cnt, ms) in ops.TIMERS.summary().items():
        prod = 3 if k.endswith("dq") else 4
        print(json.dumps(dict(kernel=k, drop=p, ms=round(ms, 3), TF=round(prod * fl / ms / 1e9, 1))))

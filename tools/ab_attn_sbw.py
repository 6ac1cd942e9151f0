"""One-pass attention backward at the headline batch (4 x 10k nodes, 8 heads, dropout 0.1) and at configs[3] (1 x 50k, 16 heads): kernel +
reduction time from ops.TIMERS, and correctness of dQ / dK / dV against the two-pass kernels of the same library.  Run against another
build with tools/run_with_lib.py (e.g. lib/sbw8: -DDGDM_FUSED_SBW=8)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import ops

dev = "cuda:0"
for B, n, H, iters in ((4, 10000, 8, 8), (1, 50000, 16, 2)):
    ptr = [i * n for i in range(B + 1)]
    plan = ops.AttnPlan(ptr, dev)
    C = H * 16
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(B * n, 3 * C, device=dev, generator=g).requires_grad_(True)
    pos = torch.rand(B * n, 2, device=dev, generator=g)
    gout = torch.randn(B * n, C, device=dev, generator=g)
    res = {}
    for fused in (True, False):
        ops.ATTN_BWD_FUSED = fused
        o = ops.spatial_attention(qkv, pos, plan, H, 0.25, 1.0, 0.1, True, seed=77)
        for _ in range(2):
            qkv.grad = None
            o.backward(gout, retain_graph=True)
        ops.TIMERS.start(["attn_bwd_fused", "attn_bwd_dq_reduce", "attn_bwd_dq", "attn_bwd_dkv"])
        for _ in range(iters):
            qkv.grad = None
            o.backward(gout, retain_graph=True)
        torch.cuda.synchronize()
        ops.TIMERS.stop()
        tm = ops.TIMERS.summary()
        res[fused] = (qkv.grad.clone(), {k: (v[0] // iters, round(v[1], 4)) for k, v in tm.items()})
        ops.TIMERS.events = {}
    ops.ATTN_BWD_FUSED = True
    d = (res[True][0] - res[False][0]).abs().max().item() / res[False][0].abs().max().item()
    print(json.dumps(dict(B=B, n=n, H=H, one_pass=res[True][1], two_pass=res[False][1], max_rel_diff_one_vs_two_pass=d)))

"""How many weight images does a step build one by one, and why?  (diagnostic for ops.WeightImages)"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import ops
from dgdm_histopath_lab_amd.models import DGDMModel
from dgdm_histopath_lab_amd.synthetic import synthetic_batch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = DGDMModel(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
batch = synthetic_batch(0, 2, 2000, 8000, 768).to(dev)
R = ops.WEIGHT_IMAGES
why = collections.Counter()
orig = R._build_one
def spy(e):
    w0, w1 = e.srcs
    why[(e.kind, tuple(w0.shape), R._persistent(w0), type(w0).__name__, w0._base is not None, e.epoch, R.epoch,
         e.versions, (w0._version, None if w1 is None else w1._version))] += 1
    return orig(e)
R._build_one = spy
for it in range(4):
    why.clear()
    opt.zero_grad(set_to_none=True)
    out = model.pretrain_step(batch, mask_ratio=0.15)
    with ops.deferred_weight_grads():
        out["total_pretrain_loss"].backward()
    opt.step()
    torch.cuda.synchronize()
    print(f"step {it}: single builds {sum(why.values())}, registry {len(R.entries)}, table_n {R.table_n}, table_blocks {R.table_blocks}, dirty {R.dirty}")
    if it >= 2:
        for k, v in list(why.items())[:12]:
            print("   ", v, k)

# ---- shapes of the weight-image GEMMs and the dW GEMMs of one step (M, K, ncols) x count
shapes = collections.Counter()
orig_img = ops._gemm_rows_img
def spy_img(a, e, tile_begin, ncols, bias, out, accumulate):
    shapes[("img", a.size(0), a.size(1), ncols, bool(accumulate))] += 1
    return orig_img(a, e, tile_begin, ncols, bias, out, accumulate)
ops._gemm_rows_img = spy_img
orig_tn = ops.gemm_tn_raw
def spy_tn(dy, x, with_bias, **kw):
    shapes[("tn", dy.size(0), dy.size(1), x.size(1), with_bias)] += 1
    return orig_tn(dy, x, with_bias, **kw)
ops.gemm_tn_raw = spy_tn
batch = synthetic_batch(0, 4, 10000, 50000, 768).to(dev)
opt.zero_grad(set_to_none=True)
out = model.pretrain_step(batch, mask_ratio=0.15)
with ops.deferred_weight_grads():
    out["total_pretrain_loss"].backward()
torch.cuda.synchronize()
for k, v in sorted(shapes.items()):
    print(v, k)

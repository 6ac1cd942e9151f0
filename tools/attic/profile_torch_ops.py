"""Which Python call sites launch the remaining torch element-wise kernels of a training step?
usage: python tools/profile_torch_ops.py  (on the GPU box; prints aten op -> count, GPU time, innermost repo frames)"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dgdm_histopath_lab_amd import DGDMModel
from dgdm_histopath_lab_amd.synthetic import synthetic_batch
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DGDMModel(**bench.MODEL_CFG).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, fused=True)
batch = synthetic_batch(0, 4, 10000, 50000, 768).to(dev)
def step():
    opt.zero_grad(set_to_none=True)
    out = model.pretrain_step(batch, mask_ratio=0.15)
    out["total_pretrain_loss"].backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
want = ("aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::cat", "aten::clone", "aten::copy_", "aten::where", "aten::sub", "aten::div",
        "aten::index", "aten::fill_", "aten::zero_", "aten::sum", "aten::clamp", "aten::threshold_backward", "aten::contiguous")
agg = collections.defaultdict(lambda: [0, 0.0])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for ev in prof.events():
    if ev.name in want and ev.device_time > 0:
        frames = [f for f in (ev.stack or []) if root in f and "tools/" not in f][:2]
        shapes = str(ev.input_shapes)[:60]
        key = (ev.name, shapes, " <- ".join(f.replace(root + "/", "")[:70] for f in frames) or "(autograd engine)")
        a = agg[key]; a[0] += 1; a[1] += ev.device_time
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{v[0]:3d} {v[1]:8.1f} us  {k[0]:22s} {k[1]:60s} {k[2]}")

"""Diagnostic: activation values and activation gradients at traced points, GPU vs fp64 oracle."""
import sys, os, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dgdm_oracle as O
from dgdm_histopath_lab_amd import DGDMModel
from dgdm_histopath_lab_amd.synthetic import synthetic_batch
cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8)
cfg = O.OracleConfig(**cfgd)
P = O.init_params(cfg, seed=3, perturb=0.05)
batch = synthetic_batch(0, 2, 2000, 8000)
gen = torch.Generator().manual_seed(11)
n = batch.x.size(0)
rng = dict(timesteps=torch.tensor([2, 9]), noise=torch.randn(n, 128, generator=gen), noise_target=torch.randn(n, 128, generator=gen))
mask_idx = torch.randperm(n, generator=gen)[: int(n * 0.15)]
mask_tok = torch.randn(768, generator=gen)
MASK = len(sys.argv) > 1 and sys.argv[1] == "mask"
mk64 = dict(mask_indices=mask_idx, mask_token=mask_tok.double()) if MASK else {}
mkg = dict(mask_indices=mask_idx.cuda(), mask_token=mask_tok.cuda()) if MASK else {}
torch.set_num_threads(16)
b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(), pos=batch.pos.double(), batch=batch.batch)
tr64 = {}
O.loss_and_grads({k: v.double() for k, v in P.items()}, cfg, b64, trace=tr64, **mk64, **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
m = DGDMModel(**cfgd); m.load_state_dict(P); m = m.cuda().eval()
tr = {}
out = m.pretrain_step(batch.to("cuda"), mask_ratio=0.15 if MASK else 0.0, trace=tr, **mkg, **{k: v.cuda() for k, v in rng.items()})
for t in tr.values():
    if isinstance(t, torch.Tensor) and t.requires_grad: t.retain_grad()
out["total_pretrain_loss"].backward()
rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()
for k in tr64:
    a, b = tr.get(k), tr64[k]
    if a is None or not isinstance(b, torch.Tensor): continue
    if b.dtype == torch.long:
        print("%-28s equal=%s  ndiff=%d" % (k, torch.equal(a.cpu(), b), (a.cpu() != b).sum().item())); continue
    if not b.requires_grad or b.grad is None: continue
    print("%-28s value err %.2e   grad err %.2e  (|grad| %.2e)" % (k, rel(a, b.detach()), rel(a.grad, b.grad), b.grad.norm().item()))
print("---- parameter grads (same run, mask_ratio=0)")
Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
o = O.pretrain_step(Pg, cfg, b64, **mk64, **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
o["total_pretrain_loss"].backward()
named = dict(m.named_parameters())
for k in sorted(Pg):
    if ("up_convs.1" in k or "up_convs.2.graph_conv1" in k) and Pg[k].grad is not None and named[k].grad is not None:
        print("%-64s err %.2e |g| %.2e" % (k, rel(named[k].grad, Pg[k].grad), Pg[k].grad.norm().item()))
print("---- relu/unpool seam between up1.out and up2.in")
p0 = tr["perm0"].cpu()
a_in, b_in = tr["unet.up2.in"].detach().cpu().double(), tr64["unet.up2.in"].detach()
mism = ((a_in > 0) != (b_in > 0))
print("relu mask mismatches:", mism.sum().item(), "of", mism.numel(), " | zero outputs gpu", (a_in == 0).sum().item(), "oracle", (b_in == 0).sum().item())
rows = mism.any(1).nonzero().flatten()
print("rows with mismatch:", rows[:20].tolist())
ga, gb = tr["unet.up2.in"].grad.cpu().double(), tr64["unet.up2.in"].grad
exp_a = (ga * (a_in > 0))[p0]; exp_b = (gb * (b_in > 0))[p0]
print("gpu   grad(up1.out) vs own relu-gather recompute:", rel(tr["unet.up1.out"].grad, exp_a))
print("oracle grad(up1.out) vs own relu-gather recompute:", rel(tr64["unet.up1.out"].grad.float(), exp_b))
d = (tr["unet.up1.out"].grad.cpu().double() - tr64["unet.up1.out"].grad)
print("rows of up1.out grad with large diff:", (d.norm(dim=1) > 1e-3 * tr64["unet.up1.out"].grad.norm(dim=1).clamp_min(1e-12)).sum().item(), "of", d.shape[0])
rn = d.norm(dim=1); top = rn.topk(5)
print("top diff rows", top.indices.tolist(), top.values.tolist(), "their |g|", tr64["unet.up1.out"].grad.norm(dim=1)[top.indices].tolist())
mi = set(mask_idx.tolist()) if MASK else set()
print("top rows -> level0 node ids", p0[top.indices].tolist(), "masked?", [int(p0[i]) in mi for i in top.indices.tolist()])

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
torch.set_num_threads(16)
b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(), pos=batch.pos.double(), batch=batch.batch)
tr64 = {}
O.loss_and_grads({k: v.double() for k, v in P.items()}, cfg, b64, trace=tr64, **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
m = DGDMModel(**cfgd); m.load_state_dict(P); m = m.cuda().eval()
tr = {}
out = m.pretrain_step(batch.to("cuda"), mask_ratio=0.0, trace=tr, **{k: v.cuda() for k, v in rng.items()})
for t in tr.values():
    if isinstance(t, torch.Tensor) and t.requires_grad: t.retain_grad()
out["total_pretrain_loss"].backward()
rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()
for k in tr64:
    a, b = tr.get(k), tr64[k]
    if a is None or not isinstance(b, torch.Tensor): continue
    if b.dtype == torch.long:
        print("%-28s equal=%s" % (k, torch.equal(a.cpu(), b))); continue
    if not b.requires_grad or b.grad is None: continue
    print("%-28s value err %.2e   grad err %.2e  (|grad| %.2e)" % (k, rel(a, b.detach()), rel(a.grad, b.grad), b.grad.norm().item()))

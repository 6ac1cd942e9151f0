"""Diagnostic: capture (input, grad_output) of every F.linear inside hierarchical_processor.up_convs.1 on the GPU
path, recompute dW = gO^T @ in in float64 from the captured fp32 operands and compare with autograd's dW."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dgdm_oracle as O
from dgdm_histopath_lab_amd import DGDMModel
from dgdm_histopath_lab_amd.synthetic import synthetic_batch
import dgdm_histopath_lab_amd.core.graph_layers as GL
cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8)
P = O.init_params(O.OracleConfig(**cfgd), seed=3, perturb=0.05)
batch = synthetic_batch(0, 2, 2000, 8000)
gen = torch.Generator().manual_seed(11)
n = batch.x.size(0)
rng = dict(timesteps=torch.tensor([2, 9]), noise=torch.randn(n, 128, generator=gen), noise_target=torch.randn(n, 128, generator=gen))
m = DGDMModel(**cfgd); m.load_state_dict(P); m = m.cuda().eval()
rec = []
orig_linear, orig_addmm = GL.F.linear, torch.addmm
class Rec:
    def linear(self, x, w, b=None):
        y = orig_linear(x, w, b)
        if y.requires_grad:
            y.retain_grad(); rec.append(("linear", x, w, y))
        return y
    def addmm(self, c, a, bt):
        y = orig_addmm(c, a, bt)
        if y.requires_grad:
            y.retain_grad(); rec.append(("addmm", a, bt, y))
        return y
r = Rec()
GL.F.linear = r.linear; GL.torch.addmm = r.addmm
out = m.pretrain_step(batch.to("cuda"), mask_ratio=0.0, **{k: v.cuda() for k, v in rng.items()})
out["total_pretrain_loss"].backward()
GL.F.linear, GL.torch.addmm = orig_linear, orig_addmm
named = {id(p): k for k, p in m.named_parameters()}
for kind, x, w, y in rec:
    wp = w if kind == "linear" else w.t()   # addmm got W^T view
    base = wp._base if wp._base is not None else wp
    name = named.get(id(base), named.get(id(wp), "?"))
    if "up_convs.1" not in name and "up_convs.2.graph_conv1" not in name: continue
    go = y.grad
    if kind == "linear":
        dw64 = go.double().t() @ x.double()
        got = w.grad
    else:
        dw64 = (go.double().t() @ x.double())
        got = base.grad
    dw32 = (go.t() @ x)
    e_auto = ((got.double() - dw64).norm() / dw64.norm()).item()
    e_mm = ((dw32.double() - dw64).norm() / dw64.norm()).item()
    cond = ((go.double().abs().t() @ x.double().abs()).norm() / dw64.norm()).item()
    print("%-62s N=%d K=%d  autograd-vs-f64 %.2e  mm32-vs-f64 %.2e  cancellation %.1f  |x| %.2e |go| %.2e" % (name, x.shape[0], x.shape[1], e_auto, e_mm, cond, x.norm().item(), go.norm().item()))

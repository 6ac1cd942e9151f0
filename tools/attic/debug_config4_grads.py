"""Which gradients of the configs[4] parity leg (tests/test_training.py) carry the largest error, per attention arithmetic?
Prints name, |ref|, abs error, rel error for the worst ten.  GPU box only."""
import math, os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import decisions_from_trace
from oracle import dgdm_oracle as O
from dgdm_histopath_lab_amd import DGDMModel, ops
from dgdm_histopath_lab_amd.parallel import BalancedSlideLoader
from dgdm_histopath_lab_amd.synthetic import synthetic_graph
from dgdm_histopath_lab_amd.training import DGDMTrainer
DEV = "cuda:0"
cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8, diffusion_schedule="cosine")
gen = torch.Generator().manual_seed(4)
ns = torch.randint(1000, 10001, (16,), generator=gen).tolist()
ns[5] = 1000; ns[6] = 1200
slides = [synthetic_graph(200 + i, n, 5 * n, 768) for i, n in enumerate(ns)]
loader = BalancedSlideLoader(slides, 4, 2, 0, device=DEV)
torch.manual_seed(0)
model = DGDMModel(**cfgd).to(DEV)
tr = DGDMTrainer(model, learning_rate=1e-3, pretrain_epochs=3, finetune_epochs=0, masking_ratio=0.15, scheduler_type="cosine")
losses = tr.fit(loader, max_epochs=3, graphed=True)
batches = list(BalancedSlideLoader(slides, 4, 2, 0))
small = min(batches, key=lambda b: b.x.size(0))
n = small.x.size(0)
rng = dict(timesteps=torch.randint(0, 10, (small.num_graphs,), generator=gen), noise=torch.randn(n, 128, generator=gen), noise_target=torch.randn(n, 128, generator=gen))
mask_idx, mask_tok = torch.randperm(n, generator=gen)[: int(0.15 * n)], torch.randn(768, generator=gen)
P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
cfg = O.OracleConfig(**cfgd)
b64 = types.SimpleNamespace(x=small.x.double(), edge_index=small.edge_index, edge_attr=small.edge_attr.double(), pos=small.pos.double(), batch=small.batch)
tr64 = {}
torch.set_num_threads(32)
ref, gref = O.loss_and_grads({k: v.double() for k, v in P.items()}, cfg, b64, mask_indices=mask_idx, mask_token=mask_tok.double(), trace=tr64,
                             **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
model.eval()
for attn, gemm in (("fp16x2", "f16x2"), ("fp32", "f16x2"), ("fp32", "fp32")):
    prev = ops.configure(attention=attn, gemm=gemm)
    model.zero_grad(set_to_none=True)
    own, dec = {}, decisions_from_trace(tr64)
    out = model.pretrain_step(small.to(DEV), mask_indices=mask_idx.to(DEV), mask_token=mask_tok.to(DEV), trace=own, decisions=dec, **{k: v.to(DEV) for k, v in rng.items()})
    out["total_pretrain_loss"].backward()
    rows = []
    named = dict(model.named_parameters())
    for k, gr in gref.items():
        if gr.abs().max() < 1e-12: continue
        g = named[k].grad.cpu().double()
        rows.append(((g - gr).norm().item() / gr.norm().item(), k, gr.norm().item(), (g - gr).abs().max().item(), gr.numel()))
    rows.sort(reverse=True)
    print(f"attention {attn}, gemm {gemm}: loss err {abs(out['diffusion_loss'].item() - ref['diffusion_loss'].item()):.2e}")
    for r in rows[:8]:
        print("   rel %.2e  %-60s |ref| %.3e  max abs err %.2e  numel %d" % r)
    ops.configure(**prev)

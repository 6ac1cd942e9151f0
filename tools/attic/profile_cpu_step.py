"""cProfile of the Python side of one training step (where does the ~15 ms of host time per step go?)."""
import cProfile, pstats, os, sys, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import DGDMModel
from dgdm_histopath_lab_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DGDMModel(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, fused=True)
nodes = int(os.environ.get("NODES", "10000")); batch = synthetic_batch(0, 4, nodes, 5 * nodes, 768).to(dev)
def step():
    opt.zero_grad(set_to_none=True)
    out = model.pretrain_step(batch, mask_ratio=0.15)
    out["total_pretrain_loss"].backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])

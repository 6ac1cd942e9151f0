"""Host time to issue one eager training step (small graphs: the GPU finishes first, the step is launch-bound)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import DGDMModel
from dgdm_histopath_lab_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DGDMModel(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, fused=True)
for nodes in (500, 2000, 10000):
    batch = synthetic_batch(0, 4, nodes, 5 * nodes, 768).to(dev)
    def step():
        opt.zero_grad(set_to_none=True)
        out = model.pretrain_step(batch, mask_ratio=0.15)
        out["total_pretrain_loss"].backward()
        opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    t_issue = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / 10
    print(f"{nodes} nodes x 4: host issue {t_issue * 1e3:.2f} ms/step, wall {t_total * 1e3:.2f} ms/step", flush=True)

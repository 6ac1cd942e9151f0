"""Does the idle time between two replays of ONE recorded step (tools/replay_gaps.py: ~0.14 ms before the first kernel of every replay)
go away when two recordings of the same step alternate?  Same model / optimizer, two GraphedPretrainStep objects, headline batch."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dgdm_histopath_lab_amd import DGDMModel
from dgdm_histopath_lab_amd.optim import DGDMAdamW
from dgdm_histopath_lab_amd.synthetic import synthetic_batch
from dgdm_histopath_lab_amd.training import GraphedPretrainStep

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DGDMModel(**bench.MODEL_CFG).to(dev).train()
opt = DGDMAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
batch = synthetic_batch(0, 4, 10000, 50000, 768).to(dev)
recs = [GraphedPretrainStep(model, opt, mask_ratio=0.15) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2)]
for g in recs:
    for _ in range(g.warmup + 2):
        g(g.input_buffers if g.input_buffers is not None else batch)
torch.cuda.synchronize()


def run(order, steps=40):
    for i in range(6):
        g = recs[order[i % len(order)]]; g(g.input_buffers)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        g = recs[order[i % len(order)]]; g(g.input_buffers)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(3):
    print(json.dumps({"one recording": round(run([0]), 3), "alternating": round(run(list(range(len(recs)))), 3)}))

"""Which torch (non-library) ops launch kernels inside one eager training step, and from where?  torch.profiler over one step of
the headline workload: every aten op that launched at least one device kernel, with its chain of enclosing ops (autograd nodes
included) and the innermost Python frame of this package.  GPU box only."""
import collections, os, sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import DGDMModel, ops
from dgdm_histopath_lab_amd.optim import DGDMAdamW
from dgdm_histopath_lab_amd.synthetic import synthetic_batch

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DGDMModel(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8).to(dev).train()
opt = DGDMAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
batch = synthetic_batch(0, 4, 10000, 50000, 768).to(dev)


def step():
    opt.zero_grad(set_to_none=True)
    out = model.pretrain_step(batch, mask_ratio=0.15)
    with ops.deferred_weight_grads():
        out["total_pretrain_loss"].backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.profiler.kineto_results.events() if hasattr(prof.profiler, "kineto_results") else []
tree = prof.profiler.kineto_results.experimental_event_tree() if hasattr(prof.profiler, "kineto_results") else []
counts = collections.Counter()


def walk(node, chain):
    name = node.name
    kids = node.children
    launches = [k for k in kids if k.name.startswith(("hipLaunchKernel", "hipExtLaunch", "hipExtModuleLaunchKernel", "hipModuleLaunchKernel", "hipMemcpyAsync", "hipMemsetAsync"))]
    if launches and name.startswith("aten::"):
        frames = [c for c in chain if ".py(" in c and "dgdm_histopath_lab_amd" in c]
        outer = [c for c in chain if c.startswith(("aten::", "autograd::", "Optimizer", "torch::autograd")) or "Backward" in c]
        counts[(name, " < ".join(reversed(outer[-3:])), frames[-1].split("dgdm_histopath_lab_amd/")[-1] if frames else "-")] += len(launches)
    for k in kids:
        walk(k, chain + [name])


for root in tree:
    walk(root, [])
tot = sum(counts.values())
print(f"{tot} kernel launches from aten ops in one eager step")
for (name, outer, frame), n in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:28s} {frame:48s} {outer}")

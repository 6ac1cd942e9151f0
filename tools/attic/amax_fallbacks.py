"""Which GEMM operands of a training step still get their absolute maximum from a separate reduction launch (dgdm_amax_bits)
instead of from the kernel that produced them?  Prints (count, shape, call chain) for one eager step at the headline size."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from dgdm_histopath_lab_amd import DGDMModel, ops
from dgdm_histopath_lab_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DGDMModel(**bench.MODEL_CFG).to(dev).train()
batch = synthetic_batch(0, 4, 10000, 50000, 768).to(dev)
def step():
    model.zero_grad(set_to_none=True)
    model.pretrain_step(batch, mask_ratio=0.15)["total_pretrain_loss"].backward()
step(); torch.cuda.synchronize()
ops.AMAX_FALLBACK_LOG = []
step(); torch.cuda.synchronize()
agg = collections.Counter(ops.AMAX_FALLBACK_LOG)
print(len(ops.AMAX_FALLBACK_LOG), "reduction launches in one step")
for (shape, chain), c in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{c:3d}  {str(shape):18s} {chain}")

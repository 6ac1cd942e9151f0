"""Where does the split recording lose time against the single graph?  One rank through RCCL on one GPU (bench.py's FORCE_DIST rehearsal),
with the reducer's exchange replaced by variants: (a) one all-reduce of the whole buffer (shipped, FlatGradAllReducer.reduce_packed),
(b) two messages (bucket 0 asynchronously, bucket 1, wait for both: what an overlapped recording would have to issue), (c) no
collective at all (graphs still split), and the single-graph step for reference."""
import os, sys, time, socket, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import bench
from dgdm_histopath_lab_amd import DGDMAdamW, DGDMModel
from dgdm_histopath_lab_amd.parallel import FlatGradAllReducer
from dgdm_histopath_lab_amd.synthetic import synthetic_batch
from dgdm_histopath_lab_amd.training import GraphedPretrainStep
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
batch = synthetic_batch(0, 4, 10000, 50000, 768).to(dev)

def run(variant):
    torch.manual_seed(0)
    model = DGDMModel(**bench.MODEL_CFG).to(dev).train()
    opt = DGDMAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
    red = None if variant == "single graph" else FlatGradAllReducer(model, 1, always=True)
    if variant == "two messages":
        def two():
            a = red._launch(0, async_op=True); b = red._launch(1, async_op=True)
            a.wait(); b.wait()
        red.reduce_packed = two
    if variant == "split, no collective":
        red.reduce_packed = lambda: None
    step = GraphedPretrainStep(model, opt, grad_reducer=red, validate=os.environ.get("VALIDATE", "1") == "1")
    for _ in range(8):
        step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        step(batch)
    torch.cuda.synchronize()
    print(f"{variant:28s} {(time.perf_counter() - t0) / 20 * 1e3:7.3f} ms/step")
for v in ("single graph", "split, no collective", "one all-reduce (shipped)", "two messages"):
    run(v)
dist.destroy_process_group()

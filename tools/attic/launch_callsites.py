"""Which Python call sites launch the per-step small kernels (image_build_one, amax_bits, torch fills / copies / cats)?"""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import DGDMModel, ops, _lib
from dgdm_histopath_lab_amd.parallel import BalancedSlideLoader
from dgdm_histopath_lab_amd.synthetic import synthetic_graph
DEV = "cuda:0"
cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8, diffusion_schedule="cosine")
slides = [synthetic_graph(200 + i, 10000, 50000, 768) for i in range(4)]
batches = list(BalancedSlideLoader(slides, 4, 1, 0, device=DEV))
torch.manual_seed(0)
model = DGDMModel(**cfgd).to(DEV).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-5, fused=True)
counts = collections.Counter()
active = [False]
def site(skip=2, depth=5):
    st = traceback.extract_stack()[:-skip]
    st = [f for f in st if "dgdm_histopath_lab_amd" in f.filename][-depth:]
    return " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(st))
def wrap_method(obj, name, label):
    o = getattr(obj, name)
    def f(*a, **k):
        if active[0]:
            counts[(label, site())] += 1
        return o(*a, **k)
    setattr(obj, name, f)
wrap_method(ops.WEIGHT_IMAGES, "_build_one", "image_build_one")
lib = _lib.load()
class LibProxy:
    def __init__(self, lib): self.__dict__["_l"] = lib
    def __getattr__(self, n):
        fn = getattr(self._l, n)
        if n in ("dgdm_amax_bits", "dgdm_fill_u32"):
            def g(*a, **k):
                if active[0]: counts[(n, site())] += 1
                return fn(*a, **k)
            return g
        return fn
proxy = LibProxy(lib)
_lib.load = lambda: proxy
from torch.utils._python_dispatch import TorchDispatchMode
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        if active[0]:
            counts[("torch:" + str(func), site(skip=3))] += 1
        return func(*args, **(kwargs or {}))
for step in range(3):
    active[0] = step == 2
    opt.zero_grad(set_to_none=True)
    with Mode():
        out = model.pretrain_step(batches[0], mask_ratio=0.15)
        with ops.deferred_weight_grads():
            out["total_pretrain_loss"].backward()
    opt.step()
agg = collections.Counter()
for (label, s), n in counts.items():
    agg[label] += n
print("== totals"); [print(f"{n:5d} {l}") for l, n in agg.most_common(60)]
print("== sites")
skip = ("torch:aten.empty", "torch:aten.view", "torch:aten.detach", "torch:aten.as_strided", "torch:aten.slice", "torch:aten.select", "torch:aten.t.", "torch:aten.alias",
        "torch:aten._unsafe_view", "torch:aten.unsqueeze", "torch:aten.squeeze", "torch:aten.expand", "torch:aten.transpose", "torch:aten.reshape", "torch:aten.split", "torch:aten.unbind", "torch:aten.empty_like", "torch:aten.new_empty", "torch:aten.permute", "torch:aten._local_scalar", "torch:aten.lift_fresh", "torch:aten.is_", "torch:aten.sym_", "torch:aten.stride", "torch:aten.size")
for (label, s), n in sorted(counts.items(), key=lambda kv: (kv[0][0], -kv[1])):
    if label.startswith(skip): continue
    print(f"{n:4d} {label:34s} {s}")

"""Gradient error of the finetune-mode parity case (tests/test_heads.py) under each GEMM arithmetic: max over parameters of the
max-abs and rel-L2 error against the float64 oracle (diagnostic for the tolerance discussion in DESIGN.md)."""
import os, sys, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import dgdm_oracle as O
from dgdm_histopath_lab_amd import DGDMModel, ops
from dgdm_histopath_lab_amd.synthetic import synthetic_batch
from conftest import decisions_from_trace
from test_heads import _oracle_train_forward
DEV = "cuda:0"
B = int(os.environ.get("B", 6))
cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8, num_classes=5, regression_targets=3, dropout=0.0)
cfg = O.OracleConfig(**cfgd)
P = O.init_params(cfg, seed=13, perturb=0.05)
bufs = O.batchnorm_buffers(cfg, seed=13, trained=True)
batch = synthetic_batch(40, B, 300, 1200)
g = torch.Generator().manual_seed(3)
y, rt = torch.randint(0, 5, (B,), generator=g), torch.randn(B, 3, generator=g)
b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(), pos=batch.pos.double(), batch=batch.batch)
P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
P64.update({k: v.double() if v.is_floating_point() else v for k, v in bufs.items()})
tr64 = {}
ref = _oracle_train_forward(P64, cfg, b64, "finetune", tr64)
(O.classification_loss(ref["classification_logits"], y) + O.regression_loss(ref["regression_outputs"], rt.double())).backward()
dec = decisions_from_trace(tr64)
for gemm in ("fp32", "bf16x3", "f16x2"):
    for attn in ("fp32", "fp16x2"):
        ops.configure(gemm=gemm, attention=attn)
        m = DGDMModel(**cfgd); m.load_state_dict({**P, **bufs}, strict=True); m = m.to(DEV).train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
        out = m(batch.to(DEV), mode="finetune", decisions=dec)
        (m.classification_head.compute_loss(out["classification_logits"], y.to(DEV)) + m.regression_head.compute_loss(out["regression_outputs"], rt.to(DEV))).backward()
        worst_abs = worst_rel = 0.0; wa = wr = ""
        for k, v in P64.items():
            if not v.requires_grad or v.grad is None or v.grad.abs().max() < 1e-12 or v.numel() <= 4: continue
            d = dict(m.named_parameters())[k].grad.double().cpu() - v.grad
            a = float(d.abs().max() / max(1.0, float(v.grad.abs().max()))); r = float(d.norm() / v.grad.norm())
            if a > worst_abs: worst_abs, wa = a, k
            if r > worst_rel: worst_rel, wr = r, k
        print(f"gemm={gemm:7s} attn={attn:7s} worst max-abs/scale {worst_abs:.2e} ({wa})  worst rel-L2 {worst_rel:.2e} ({wr})")

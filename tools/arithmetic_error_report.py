"""Diagnostic: how far is each arithmetic from exact?  One pretrain_step (eval mode, injected draws) of the Base model, every
live parameter gradient, rel-L2 against the float64 oracle, for
  * the HIP path in its default arithmetic (fp16 hi+lo operand pairs, fp32 accumulate),
  * the HIP path with fp32 operands on the fp32 matrix instructions (ops.configure(attention="fp32", gemm="fp32")),
  * torch fp32 on the CPU (the reference's own arithmetic: the oracle code in float32).
Answers "is the default narrower than fp32 in effect?" with numbers.  Test tooling (imports oracle/).
usage: arithmetic_error_report.py [nodes edges graphs]   (the smooth model: no U-Net, so no top-k / ReLU decision can differ
between the four runs; the U-Net's layers run on the same kernels)."""
import collections
import json
import os
import statistics
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dgdm_oracle as O  # noqa: E402
from dgdm_histopath_lab_amd import DGDMModel, ops  # noqa: E402
from dgdm_histopath_lab_amd.synthetic import synthetic_batch  # noqa: E402


def main():
    nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    edges = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    graphs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    smooth = True
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8, use_hierarchical=not smooth)
    cfg = O.OracleConfig(**cfgd)
    P = O.init_params(cfg, seed=3, perturb=0.05)
    batch = synthetic_batch(0, graphs, nodes, edges)
    gen = torch.Generator().manual_seed(11)
    n = batch.x.size(0)
    rng = dict(timesteps=torch.randint(0, 10, (graphs,), generator=gen), noise=torch.randn(n, 128, generator=gen),
               noise_target=torch.randn(n, 128, generator=gen))
    mask_idx = torch.randperm(n, generator=gen)[: int(n * 0.15)]
    mask_tok = torch.randn(768, generator=gen)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(), pos=batch.pos.double(),
                                batch=batch.batch)
    kw64 = dict(mask_indices=mask_idx, mask_token=mask_tok.double(), **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
    r64, g64 = O.loss_and_grads({k: v.double() for k, v in P.items()}, cfg, b64, **kw64)
    kw32 = dict(mask_indices=mask_idx, mask_token=mask_tok, **rng)
    r32, g32 = O.loss_and_grads(P, cfg, batch, **kw32)

    def gpu(attention, gemm):
        prev = ops.configure(attention=attention, gemm=gemm)
        try:
            m = DGDMModel(**cfgd)
            m.load_state_dict(P)
            m = m.cuda().eval()
            kw = dict(mask_indices=mask_idx.cuda(), mask_token=mask_tok.cuda(), **{k: v.cuda() for k, v in rng.items()})
            out = m.pretrain_step(batch.to("cuda"), **kw)
            out["total_pretrain_loss"].backward()
            return out["diffusion_loss"].item(), {k: p.grad.double().cpu() for k, p in m.named_parameters() if p.grad is not None}
        finally:
            ops.configure(**prev)

    default = ops.configure()
    l_def, g_def = gpu(default["attention"], default["gemm"])
    l_f32, g_f32 = gpu("fp32", "fp32")
    rows, dead = [], []
    for k, g in g64.items():
        nb = g.norm().item()
        if k not in g_def:
            continue
        if g.abs().max().item() < 1e-12:      # zero in exact arithmetic (k_proj.bias: softmax is shift invariant): rounding noise only
            dead.append((k, g_def[k].abs().max().item(), g_f32[k].abs().max().item(), g32[k].abs().max().item()))
            continue
        rows.append((k, nb, (g_def[k] - g).norm().item() / nb, (g_f32[k] - g).norm().item() / nb, (g32[k].double() - g).norm().item() / nb))
    l64 = r64["diffusion_loss"].item()
    print("# %d graphs x %d nodes / %d edges, Base, eval mode, %s; default arithmetic = %s" % (graphs, nodes, edges,
          "smooth model (no U-Net)" if smooth else "U-Net on, the float64 run's decisions injected", json.dumps(default)))
    print("# loss: float64 %.10f | HIP default %+.2e | HIP fp32 %+.2e | torch-CPU fp32 %+.2e  (relative)" % (
        l64, (l_def - l64) / l64, (l_f32 - l64) / l64, (r32["diffusion_loss"].item() - l64) / l64))
    print("# rel-L2 error of every live parameter gradient against float64, grouped by module (max | median)")
    print("%-44s %5s  %-21s %-21s %-21s" % ("module", "n", "HIP default (hi+lo)", "HIP fp32 operands", "torch CPU fp32"))
    groups = collections.OrderedDict()
    for k, nb, a, b, c in rows:
        parts = k.split(".")
        pre = ".".join(parts[:3] if parts[0] in ("graph_encoder", "hierarchical_processor") and len(parts) > 3 else parts[:1])
        groups.setdefault(pre, []).append((a, b, c))
    for pre, v in groups.items():
        cols = ["%.2e | %.2e" % (max(x[i] for x in v), statistics.median(x[i] for x in v)) for i in range(3)]
        print("%-44s %5d  %-21s %-21s %-21s" % (pre, len(v), *cols))
    cols = ["%.2e | %.2e" % (max(x[i] for x in rows), statistics.median(x[i] for x in rows)) for i in (2, 3, 4)]
    print("%-44s %5d  %-21s %-21s %-21s" % ("ALL", len(rows), *cols))
    for k, a, b, c in dead:
        print("# %s is zero in exact arithmetic; max|g|: HIP default %.1e, HIP fp32 %.1e, torch CPU fp32 %.1e" % (k, a, b, c))
    worse = sum(1 for r in rows if r[2] > r[4])
    print("# gradients where the default arithmetic is further from float64 than torch-CPU fp32: %d of %d; than HIP fp32: %d of %d" % (
        worse, len(rows), sum(1 for r in rows if r[2] > r[3]), len(rows)))
    print("# geometric-mean ratio default/torch-fp32 %.2f, default/HIP-fp32 %.2f" % (
        statistics.geometric_mean(r[2] / max(r[4], 1e-12) for r in rows), statistics.geometric_mean(r[2] / max(r[3], 1e-12) for r in rows)))


if __name__ == "__main__":
    main()

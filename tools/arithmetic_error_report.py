"""Diagnostic: how far is each arithmetic from exact?  One pretrain_step (eval mode, injected draws), every live parameter
gradient, rel-L2 against the float64 oracle, for
  * the HIP path in its default arithmetic (fp16 hi+lo operand pairs, fp32 accumulate),
  * the HIP path with fp32 operands on the fp32 matrix instructions (ops.configure(attention="fp32", gemm="fp32")),
  * torch fp32 on the CPU (the reference's own arithmetic: the oracle code in float32).
Answers "is the default narrower than fp32 in effect?" with numbers.  Test tooling (imports oracle/).

usage: arithmetic_error_report.py [--case smooth|unet|sharp|large|all] [nodes edges graphs]
  smooth  Base, no U-Net (no top-k / ReLU decision can differ between the runs)                       [round 3's only instance]
  unet    Base with the graph U-Net on; the float64 run's ReLU / top-k decisions injected into ALL three fp32-level runs
  sharp   Base, no U-Net, q_proj / k_proj x 4: trained-like attention rows (mean row entropy < 1 nat; core/attention.py:135-157)
  large   configs[3] widths: hidden [1024, 512, 256], 16 heads, T = 20, U-Net on (decisions injected): K = 1024 reductions
          (core/graph_layers.py:400-458)
`cases()` / `run_case()` are what tests/test_hip_model.py::test_default_arithmetic_is_at_the_error_level_of_fp32 asserts on."""
import collections
import json
import os
import statistics
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dgdm_oracle as O  # noqa: E402

BASE = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8)
LARGE = dict(node_features=768, hidden_dims=[1024, 512, 256], num_diffusion_steps=20, attention_heads=16)


def _sharpen(P):
    for k in ("spatial_attention.attention.q_proj.weight", "spatial_attention.attention.q_proj.bias",
              "spatial_attention.attention.k_proj.weight", "spatial_attention.attention.k_proj.bias"):
        P[k] = P[k] * 4.0


def cases():
    return {"smooth": dict(cfgd=dict(BASE, use_hierarchical=False), tweak=None, note="Base, smooth model (no U-Net)"),
            "unet": dict(cfgd=dict(BASE), tweak=None, note="Base, U-Net on, the float64 run's ReLU / top-k decisions injected into every run"),
            "sharp": dict(cfgd=dict(BASE, use_hierarchical=False), tweak=_sharpen, note="Base, no U-Net, q_proj / k_proj x 4 (sharp attention rows)"),
            "large": dict(cfgd=dict(LARGE), tweak=None, note="Large widths (1024/512/256, 16 heads, T=20), U-Net on, decisions injected")}


def decisions_from_trace(trace):
    dec = {}
    for k, v in trace.items():
        if k.startswith("relu."):
            dec[k] = (v.detach() > 0)
        elif k.startswith("perm") and k[4:].isdigit():
            dec[k] = v.detach().clone()
    return dec


TRACE_ORDER = ["graph_unet", "unet.up2.out", "unet.up2.in", "unet.up1.out", "unet.up1.in", "unet.up0.out", "unet.up0.in", "unet.bottom",
               "relu.bottom", "unet.xs3", "relu.down2", "unet.xs2", "relu.down1", "unet.xs1", "relu.down0", "unet.xs0", "spatial_attention",
               "graph_encoder", "feature_encoder"]      # backward order of the activations both sides trace


def run_case(name, nodes=2000, edges=8000, graphs=2, seed=3, data_seed=0, trace_grads=False):
    """-> dict(rows=[(param, |ref|, err default, err HIP fp32, err torch fp32)], dead=[...], losses=(...), entropy=mean row entropy
    of the spatial attention in nats or None, note, default=the default arithmetic).  ``trace_grads``: also ``trace_rows`` =
    [(activation, gradient err default, HIP fp32, torch fp32, value err default, HIP fp32, torch fp32)] -- rel-L2 error of the gradient
    that ARRIVES at each traced activation and of the activation itself, in backward order: where an arithmetic's error enters
    (tools/gradient_error_trace.py)."""
    from dgdm_histopath_lab_amd import DGDMModel, ops
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    c = cases()[name]
    cfgd = c["cfgd"]
    cfg = O.OracleConfig(**cfgd)
    P = O.init_params(cfg, seed=seed, perturb=0.05)
    if c["tweak"] is not None:
        c["tweak"](P)
    hier = cfgd.get("use_hierarchical", True)
    batch = synthetic_batch(data_seed, graphs, nodes, edges)
    gen = torch.Generator().manual_seed(11 + data_seed)
    n = batch.x.size(0)
    cl, T = cfgd["hidden_dims"][-1], cfgd["num_diffusion_steps"]
    rng = dict(timesteps=torch.randint(0, T, (graphs,), generator=gen), noise=torch.randn(n, cl, generator=gen),
               noise_target=torch.randn(n, cl, generator=gen))
    mask_idx = torch.randperm(n, generator=gen)[: int(n * 0.15)]
    mask_tok = torch.randn(768, generator=gen)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(), pos=batch.pos.double(),
                                batch=batch.batch)
    kw64 = dict(mask_indices=mask_idx, mask_token=mask_tok.double(), **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
    tr64 = {} if (hier or trace_grads) else None
    r64, g64 = O.loss_and_grads({k: v.double() for k, v in P.items()}, cfg, b64, trace=tr64, **kw64)
    dec = decisions_from_trace(tr64) if hier else None
    entropy = None
    if name == "sharp":       # how sharp the rows are: mean PER-HEAD row entropy of graph 0's attention (float64, as mha() forms it)
        import math
        with torch.no_grad():
            P64 = {k: v.double() for k, v in P.items()}
            tr = {}
            O.forward(P64, cfg, b64, mode="inference", trace=tr)
            x0, p0 = tr["graph_encoder"][:nodes], b64.pos[:nodes]
            C, H = x0.shape[1], cfgd["attention_heads"]
            xp = x0 + O.sinusoid_pos_encoding(p0, C).to(x0.dtype)
            pre = "spatial_attention.attention"
            q = O._lin(P64, f"{pre}.q_proj", xp).view(nodes, H, C // H).transpose(0, 1)
            k = O._lin(P64, f"{pre}.k_proj", xp).view(nodes, H, C // H).transpose(0, 1)
            sc = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(C // H) - torch.norm(p0.unsqueeze(1) - p0.unsqueeze(0), dim=-1)
            w = torch.softmax(sc, dim=-1)
            entropy = float(-(w * w.clamp_min(1e-300).log()).sum(-1).mean())
    O.DECISIONS = dec
    tr32 = {} if trace_grads else None
    try:
        r32, g32 = O.loss_and_grads(P, cfg, batch, mask_indices=mask_idx, mask_token=mask_tok, trace=tr32, **rng)
    finally:
        O.DECISIONS = None

    def act_grads(tr):
        return {k: (t.grad.double().cpu(), t.detach().double().cpu()) for k, t in tr.items()
                if isinstance(t, torch.Tensor) and t.requires_grad and t.grad is not None}

    def gpu(attention, gemm):
        prev = ops.configure(attention=attention, gemm=gemm)
        try:
            m = DGDMModel(**cfgd)
            m.load_state_dict(P)
            m = m.cuda().eval()
            kw = dict(mask_indices=mask_idx.cuda(), mask_token=mask_tok.cuda(), **{k: v.cuda() for k, v in rng.items()})
            tr = {} if trace_grads else None
            out = m.pretrain_step(batch.to("cuda"), decisions=dec, trace=tr, **kw)
            if tr is not None:
                for t in tr.values():
                    if isinstance(t, torch.Tensor) and t.requires_grad:
                        t.retain_grad()
            out["total_pretrain_loss"].backward()
            traces.append(act_grads(tr) if tr is not None else None)
            return out["diffusion_loss"].item(), {k: p.grad.double().cpu() for k, p in m.named_parameters() if p.grad is not None}
        finally:
            ops.configure(**prev)

    traces = []
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
    trace_rows = None
    if trace_grads:
        a64, a32 = act_grads(tr64), act_grads(tr32)
        trace_rows = []
        for k in TRACE_ORDER:
            if k in a64 and all(k in a for a in (traces[0], traces[1], a32)):
                nb, nv = a64[k][0].norm().item(), a64[k][1].norm().item()
                trace_rows.append((k, *((a[k][0] - a64[k][0]).norm().item() / nb for a in (traces[0], traces[1], a32)),
                                   *((a[k][1] - a64[k][1]).norm().item() / nv for a in (traces[0], traces[1], a32))))
    l64 = r64["diffusion_loss"].item()
    return dict(trace_rows=trace_rows, rows=rows, dead=dead, losses=(l64, l_def, l_f32, r32["diffusion_loss"].item()), entropy=entropy, note=c["note"],
                default=default, shape=(graphs, nodes, edges))


def summary(rows):
    """(max, median) of the three error columns."""
    return [(max(r[i] for r in rows), statistics.median(r[i] for r in rows)) for i in (2, 3, 4)]


def report(name, res):
    rows, dead = res["rows"], res["dead"]
    l64, l_def, l_f32, l32 = res["losses"]
    graphs, nodes, edges = res["shape"]
    print("# case %s: %d graphs x %d nodes / %d edges, eval mode, %s; default arithmetic = %s" % (name, graphs, nodes, edges, res["note"],
                                                                                              json.dumps(res["default"])))
    if res["entropy"] is not None:
        print("# mean per-head row entropy of graph 0's attention weights: %.2f nats (ln N = %.2f)" % (res["entropy"], torch.tensor(float(nodes)).log()))
    print("# loss: float64 %.10f | HIP default %+.2e | HIP fp32 %+.2e | torch-CPU fp32 %+.2e  (relative)" % (
        l64, (l_def - l64) / l64, (l_f32 - l64) / l64, (l32 - l64) / l64))
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
    cols = ["%.2e | %.2e" % mm for mm in summary(rows)]
    print("%-44s %5d  %-21s %-21s %-21s" % ("ALL", len(rows), *cols))
    worst = max(rows, key=lambda r: r[2])
    print("# worst gradient of the default arithmetic: %s (|ref| %.3e): default %.2e, HIP fp32 %.2e, torch fp32 %.2e" % (worst[0], worst[1], *worst[2:]))
    for k, a, b, c in dead:
        print("# %s is zero in exact arithmetic; max|g|: HIP default %.1e, HIP fp32 %.1e, torch CPU fp32 %.1e" % (k, a, b, c))
    worse = sum(1 for r in rows if r[2] > r[4])
    print("# gradients where the default arithmetic is further from float64 than torch-CPU fp32: %d of %d; than HIP fp32: %d of %d" % (
        worse, len(rows), sum(1 for r in rows if r[2] > r[3]), len(rows)))
    print("# geometric-mean ratio default/torch-fp32 %.2f, default/HIP-fp32 %.2f" % (
        statistics.geometric_mean(r[2] / max(r[4], 1e-12) for r in rows), statistics.geometric_mean(r[2] / max(r[3], 1e-12) for r in rows)))
    print()


def main():
    args = sys.argv[1:]
    case = "smooth"
    if args and args[0] == "--case":
        case, args = args[1], args[2:]
    nodes = int(args[0]) if len(args) > 0 else 2000
    edges = int(args[1]) if len(args) > 1 else 8000
    graphs = int(args[2]) if len(args) > 2 else 2
    for name in (list(cases()) if case == "all" else [case]):
        report(name, run_case(name, nodes, edges, graphs))
        sys.stdout.flush()


if __name__ == "__main__":
    main()

"""GPU parity of the product DGDMModel (HIP path) against (a) the golden vectors captured from
the reference and (b) the CPU oracle on larger synthetic graphs.  eval() / dropout off, random
draws injected.  Tolerance: the north star's 1e-3 (conftest.assert_close metric)."""
import json
import types

import numpy as np
import pytest
import torch

from conftest import (T, assert_close, check_decision_margins, decisions_from_golden, decisions_from_trace, load_golden,
                      weights)
from oracle import dgdm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3


def _model(cfgd, P):
    from dgdm_histopath_lab_amd import DGDMModel
    m = DGDMModel(**cfgd)
    missing = m.load_state_dict({k: v for k, v in P.items()}, strict=True)
    return m.to(DEV).eval()


def _batch(g):
    from dgdm_histopath_lab_amd import GraphBatch
    b = GraphBatch(x=T(g["x"]), edge_index=T(g["edge_index"]), edge_attr=T(g["edge_attr"]), pos=T(g["pos"]))
    b.batch = T(g["batch"])
    return b.to(DEV)


@pytest.mark.parametrize("attn", ["fp32", "fp16x2"])
@pytest.mark.parametrize("tag", ["small", "base"])
def test_model_matches_reference_golden(tag, attn, monkeypatch):
    from dgdm_histopath_lab_amd import ops
    monkeypatch.setattr(ops, "ATTN_PRECISION", attn)
    g = load_golden(f"g7_model_{tag}")
    cfgd = json.loads(str(g["cfg_json"]))
    cfg = O.OracleConfig(**cfgd)
    if tag == "small":
        P = weights(g)
        P["spatial_attention.pos_encoding"] = torch.zeros(O.param_shapes(cfg)["spatial_attention.pos_encoding"])
    else:
        P = O.init_params(cfg, seed=int(g["init_seed"]), perturb=float(g["init_perturb"]))
    m = _model(cfgd, P)
    data = _batch(g)
    out = m(data, mode="inference", return_attention=True, return_embeddings=True)
    assert_close(out["graph_embedding"], g["inf_graph_embedding"], TOL, "graph_embedding")
    assert_close(out["node_embeddings"], g["inf_node_embeddings"], TOL, "node_embeddings")
    assert_close(out["attention_weights"][0], g["inf_attn0"], TOL, "attn0")
    assert_close(out["attention_weights"][1], g["inf_attn1"], TOL, "attn1")

    # the reference run's ReLU / top-k decisions are part of the fixture (dec.*, recorded by oracle/capture_golden.py from the
    # reference's own GraphUNet.forward): the kernels take the side of every kink from them, and every decision this path would
    # have taken differently must lie within the rounding margin (conftest.check_decision_margins)
    dec = decisions_from_golden(g)
    assert len(dec) == 13
    tr = {}
    outp = m.pretrain_step(data, mask_ratio=0.15, mask_indices=T(g["mask_indices"]).to(DEV), mask_token=T(g["mask_token"]).to(DEV),
                           timesteps=T(g["timesteps"]).to(DEV), noise=T(g["noise"]).to(DEV), noise_target=T(g["noise_target"]).to(DEV),
                           trace=tr, decisions=dec)
    assert set(outp) >= {"diffusion_loss", "total_pretrain_loss", "graph_embedding", "noisy_embeddings"}
    assert_close(outp["diffusion_loss"], g["pre_diffusion_loss"], TOL, "diffusion_loss")
    assert_close(outp["graph_embedding"], g["pre_graph_embedding"], TOL, "pre_graph_embedding")
    assert_close(outp["noisy_embeddings"], g["pre_noisy_embeddings"], TOL, "noisy_embeddings")
    outp["total_pretrain_loss"].backward()
    named = dict(m.named_parameters())
    flips, total = check_decision_margins(tr, dec)
    for i in range(3):       # index work stays bit-exact on this fixture (scores are far from tied)
        assert torch.equal(tr[f"own_perm{i}"].cpu(), dec[f"perm{i}"]), f"perm{i}"
    n = 0
    for k in g:
        if k.startswith("grad."):
            assert_close(named[k[5:]].grad, g[k], TOL, k); n += 1
        elif k.startswith("gradnorm."):
            name = k[9:]
            assert_close(named[name].grad.norm(), g[k], TOL, k)
            assert_close(named[name].grad.flatten()[:256], g["gradslice." + name], TOL, "gradslice." + name); n += 1
    assert n == 11
    assert named["spatial_attention.pos_encoding"].grad is None  # dead parameters stay dead (D9)
    # and without injection the forward outputs are the same numbers (the decisions only matter to the backward)
    outq = m.pretrain_step(data, mask_ratio=0.15, mask_indices=T(g["mask_indices"]).to(DEV), mask_token=T(g["mask_token"]).to(DEV),
                           timesteps=T(g["timesteps"]).to(DEV), noise=T(g["noise"]).to(DEV), noise_target=T(g["noise_target"]).to(DEV))
    assert_close(outq["diffusion_loss"], g["pre_diffusion_loss"], TOL, "diffusion_loss (own decisions)")
    assert_close(outq["graph_embedding"], g["pre_graph_embedding"], TOL, "graph_embedding (own decisions)")


def _run_both(cfgd, seed0, trace, nodes=2000, edges=8000, graphs=2, tweak=None, pos_fn=None):
    """One pretrain_step (masking + injected draws) on the HIP path and on the float64 oracle.
    The arbiter runs in float64 (same oracle code): fp32-vs-fp32 would fold the CPU path's own
    rounding into the comparison."""
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    cfg = O.OracleConfig(**cfgd)
    P = O.init_params(cfg, seed=3, perturb=0.05)
    if tweak is not None:
        tweak(P)
    batch = synthetic_batch(seed0, graphs, nodes, edges)
    if pos_fn is not None:
        batch.pos = pos_fn(batch)
        batch.pos_extent = None                       # the synthetic graphs' hint (1.0) no longer holds: taken from the host tensor,
        batch.pos_extent = batch.host_pos_extent()    # as a loader does (it decides whether the attention builds its zero-block map)
        from dgdm_histopath_lab_amd import ops
        assert ops.attn_zero_blocks_possible(batch.pos_extent, 1.0) and not ops.attn_zero_blocks_possible(1.0, 1.0)
    gen = torch.Generator().manual_seed(11 + seed0)
    n = batch.x.size(0)
    c_last, T = cfgd["hidden_dims"][-1], cfgd["num_diffusion_steps"]
    rng = dict(timesteps=torch.tensor([2, T - 1, 0, 5][:graphs]), noise=torch.randn(n, c_last, generator=gen),
               noise_target=torch.randn(n, c_last, generator=gen))
    mask_idx = torch.randperm(n, generator=gen)[: int(n * 0.15)]
    mask_tok = torch.randn(768, generator=gen)
    torch.set_num_threads(16)
    b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(),
                                pos=batch.pos.double(), batch=batch.batch)
    tr64 = {} if trace else None
    ref, gref = O.loss_and_grads({k: v.double() for k, v in P.items()}, cfg, b64, mask_indices=mask_idx, mask_token=mask_tok.double(),
                                 trace=tr64, **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
    m = _model(cfgd, P)
    tr = {} if trace else None
    dec = decisions_from_trace(tr64) if trace else None     # the checker's ReLU / top-k decisions go to the kernels
    out = m.pretrain_step(batch.to(DEV), mask_indices=mask_idx.to(DEV), mask_token=mask_tok.to(DEV), trace=tr, decisions=dec,
                          **{k: v.to(DEV) for k, v in rng.items()})
    out["total_pretrain_loss"].backward()
    if trace:
        tr["__flips__"] = check_decision_margins(tr, dec)
    return m, out, ref, gref, tr, tr64


def _assert_all_grads(m, gref, tol):
    named = dict(m.named_parameters())
    live = 0
    gmax = max(float(g.abs().max()) for g in gref.values())
    for k, gr in gref.items():
        got = named[k].grad
        if got is None and gr.abs().max() == 0:   # e.g. edge_lin with edge_attr=None: the oracle multiplies by zeros, the kernels skip the term
            continue
        assert got is not None, k
        if gr.abs().max() < 1e-12:  # dead-by-construction (e.g. k_proj.bias: softmax is shift invariant): rounding noise only
            assert got.abs().max() < max(1e-7, 1e-5 * gmax), k
            continue
        assert_close(got, gr, tol, "grad " + k); live += 1
    for k, p in named.items():  # nothing receives a gradient that the oracle leaves dead (D9)
        if k not in gref:
            assert p.grad is None or p.grad.abs().max() == 0, k
    return live


@pytest.mark.parametrize("case", ["smooth", "unet", "sharp", "large"])
def test_default_arithmetic_is_at_the_error_level_of_fp32(case):
    """The reference computes in fp32 ("everything is fp32"); the default kernels form every product on fp16 hi+lo
    operand pairs with fp32 accumulation.  Is that narrower IN EFFECT?  Same step, same draws, every live gradient against the
    float64 oracle: the default arithmetic, fp32 operands on the fp32 matrix instructions, and torch fp32 on the CPU (the
    oracle code in float32 = the reference's own arithmetic) must sit at the same distance from exact -- the default within 1.25x of
    the farther and 2x of the nearer of the two fp32 runs, at the worst gradient and at the median.  Round 4: not only on the smooth
    near-init Base model, but also with the U-Net on (the float64 run's kink decisions injected into all three), on sharp
    attention rows (q_proj / k_proj x 4, core/attention.py:135-157) and at Large widths (hidden 1024/512/256, 16 heads, K = 1024
    reductions, core/graph_layers.py:400-458).  tools/arithmetic_error_report.py prints the tables
    (profiles/r04_arithmetic_error_vs_float64.txt)."""
    import os
    import sys
    from dgdm_histopath_lab_amd import ops
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import arithmetic_error_report as R
    assert ops.configure() == dict(attention="fp16x2", gemm="f16x2")        # the arithmetic bench.py's headline runs on
    res = R.run_case(case, trace_grads=(case == "large"))
    rows = res["rows"]
    assert len(rows) >= 80
    (mx_d, md_d), (mx_h, md_h), (mx_c, md_c) = R.summary(rows)
    print("%s: default max %.2e med %.2e | HIP fp32 max %.2e med %.2e | torch CPU fp32 max %.2e med %.2e" % (case, mx_d, md_d, mx_h, md_h, mx_c, md_c))
    if case == "sharp":
        assert res["entropy"] < 1.0, res["entropy"]                         # the rows really are sharp (ln N = 7.6)
    # all three well inside the 1e-3 contract; the level itself is the problem's conditioning (Large: 2e-4 for fp32 operands too)
    assert max(mx_d, mx_h, mx_c) <= {"smooth": 5e-6, "unet": 5e-5, "sharp": 5e-5, "large": 5e-4}[case]
    # Two fp32 implementations of the same step already differ from each other (Large: median 9.0e-5 for the HIP fp32-operand kernels,
    # 3.2e-5 for torch on the CPU: summation orders).  The default arithmetic must sit INSIDE that band, with little slack:
    #   against the fp32 run that is FARTHER from exact: <= 1.25 x at the worst gradient and at the median
    #     (observed 0.80 / 0.76 / 0.91 / 1.12 and 0.96 / 0.92 / 0.89 / 0.60 for smooth / unet / sharp / large);
    #   against the NEARER one: <= 2 x both (observed 1.23 / 1.04 / 1.53 / 1.92 and 1.26 / 1.07 / 1.46 / 1.66).
    # An emulation that lost two operand bits fails both (round 4 allowed 4 x / 2.5 x against either run).
    # At Large widths the step amplifies rounding ~1e3 x (all three runs sit at 1-3e-4); until round 6 the HIP runs sat 2-4 x above
    # torch's there and round 5 had widened this bound to 3 x.  Round 6 found the cause (tools/gradient_error_trace.py,
    # tools/ubench/mfma_rounding.hip; profiles/r06_gradient_error_trace*.txt, r06_mfma_rounding.txt): the matrix pipe adds a block of
    # products into its fp32 accumulator with TWICE the rms error of one correctly rounded add (f16 and fp32 MFMA alike), so every
    # GEMM output -- in both HIP arithmetics -- carried 2.0 x the error of a CPU sgemm from the first layer on (with the tile GEMMs
    # replaced by float64 products the HIP runs are 0.3-0.8 x torch's error), and the problem amplifies either run's error by the same
    # ~1e3 (the top-k pooling's score gradient is a cancelling sum); one (weights, data) draw is one sample.  Round 5's branch-free GELU
    # was NOT the cause of the r04 -> r05 move (with the library erf back, three draws give default / torch 1.52, 3.50, 3.51 against
    # 2.05, 4.10, 1.31 with it): it re-drew the sample.  The fix is in the kernels: the default GEMMs now chain each column tile's
    # products of a stage from a ZERO accumulator and add that block to the running accumulator on the vector unit (csrc/gemm_img.hip
    # DGDM_IMG_FRESH, csrc/gemm_h.hip TN_FLUSH): every traced activation of the default run is now at 0.6-1.0 x torch's error (was
    # 1.5-2.1 x) and default / torch at the median gradient is 1.34, 1.69 over the draws.  So the bound against the nearer run is 2 x
    # again at every case, and the activation-level property is asserted below (default <= 1.5 x torch at all 19 trace points; the
    # fp32-operand kernels of the strict leg keep the single accumulator: <= 3 x).
    far_mx, far_md, near_mx, near_md = max(mx_h, mx_c), max(md_h, md_c), min(mx_h, mx_c), min(md_h, md_c)
    near = 2.0
    assert mx_d <= 1.25 * far_mx and md_d <= 1.25 * far_md, (mx_d, far_mx, md_d, far_md)
    assert mx_d <= near * near_mx and md_d <= near * near_md, (mx_d, near_mx, md_d, near_md)
    if res["trace_rows"]:
        assert len(res["trace_rows"]) >= 15
        for k, _gd, _gh, _gt, v_def, v_hip, v_torch in res["trace_rows"]:
            assert v_def <= 1.5 * v_torch and v_hip <= 3.0 * v_torch, (k, v_def, v_hip, v_torch)


class _DropoutSites:
    """Records every dropout site of one HIP forward (kind, seed, shape) in execution order, regenerates each site's mask with the
    kernels themselves, and serves them to the oracle (oracle.DROPOUT_HOOK) -- so that the TRAINING-mode step bench.py times can be
    compared with the restatement draw for draw, although the two sides have different random number generators.
      act_dropout / row_norm sites: the mask is a function of (seed, element index): the activation kernel on a tensor of ones
      attention weights:            the forward kernel itself on Q = K = 0, pos = 0 (uniform probabilities 1/n) and one-hot V
                                    blocks, 16 key columns per run (as test_attn_dropout_mask_consistent... does)
    The attention-pool weights' dropout is switched off on both sides (its mask is a third hash; the pooled row is not on the
    pretraining loss)."""

    def __init__(self, ops, monkeypatch):
        self.ops, self.sites, self.masks = ops, [], None
        kw = lambda a, k, name, pos: k[name] if name in k else (a[pos] if pos is not None and len(a) > pos else 0.0)
        describe = {
            "act_dropout": lambda a, k, out: [dict(kind="act", shape=tuple(out.shape), p=float(kw(a, k, "drop_p", 2)))],
            "row_norm": lambda a, k, out: [dict(kind="rownorm", shape=tuple(out.shape), p=float(k["drop_p"]))],
            "spatial_attention": lambda a, k, out: [dict(kind="attn", plan=a[2], H=a[3], p=float(kw(a, k, "drop_p", 6)), n=a[0].size(0))],
            # round 5: sites that live in GEMM epilogues (the same (seed, element index) masks, drawn in the same order)
            "graph_layer": lambda a, k, out: [dict(kind="act", shape=(a[0].size(0), a[3].out_channels), p=float(a[7]))] * 2,
            "linear_norm": lambda a, k, out: ([dict(kind="act", shape=tuple(out.shape), p=float(k["pre_drop_p"]))] if k.get("pre_drop_p", 0) > 0 else [])
                                             + ([dict(kind="rownorm", shape=tuple(out.shape), p=float(k["drop_p"]))] if k.get("drop_p", 0) > 0 else []),
            "denoise_first_layer": lambda a, k, out: [dict(kind="rownorm", shape=tuple(out.shape), p=float(k["drop_p"]))],
        }
        for name, d in describe.items():
            monkeypatch.setattr(ops, name, self._wrap(getattr(ops, name), d))

    def _wrap(self, fn, describe):
        def wrapped(*a, **k):
            c0, n0 = self.ops._seed_counter, len(self.sites)
            out = fn(*a, **k)
            drawn = self.ops._seed_counter - c0
            if drawn and len(self.sites) == n0:                  # dropout was live, and no wrapped op inside recorded the draws already
                sites = [dict(d) for d in describe(a, k, out)]
                assert len(sites) == drawn, (fn.__name__, drawn, sites)
                for j, d in enumerate(sites):
                    d["seed"] = (torch.initial_seed() * 0x9E3779B1 + (c0 + 1 + j) * 0x85EBCA6B) & 0xFFFFFFFF
                    self.sites.append(d)
            elif drawn:
                assert len(self.sites) - n0 == drawn, (fn.__name__, drawn)
            return out
        return wrapped

    def build_masks(self):
        from dgdm_histopath_lab_amd import _lib
        ops, lib = self.ops, _lib.load()
        self.masks = []
        for s in self.sites:
            if s["kind"] != "attn":
                n = 1
                for d in s["shape"]:
                    n *= d
                ones, y = torch.ones(n, device=DEV), torch.empty(n, device=DEV)
                _lib.check(lib.dgdm_act_dropout_fwd(ones.data_ptr(), n, 0, s["p"], s["seed"], y.data_ptr(), None, None,
                                                    _lib.stream_ptr(ones.device)), "mask probe")
                self.masks.append(y.view(s["shape"]).cpu().double())
                continue
            plan, H, n = s["plan"], s["H"], s["n"]
            C = 16 * H
            ptr = plan.ptr_dev.cpu().tolist()
            F = [torch.zeros(H, ptr[g + 1] - ptr[g], ptr[g + 1] - ptr[g], dtype=torch.float64) for g in range(len(ptr) - 1)]
            pos0 = torch.zeros(n, 2, device=DEV)
            for c in range((n + 15) // 16):
                buf = torch.zeros(n, 3 * C, device=DEV)
                for kk in range(16 * c, min(n, 16 * c + 16)):
                    buf[kk, 2 * C + torch.arange(H) * 16 + (kk - 16 * c)] = 1.0
                if ops.ATTN_PRECISION == "fp32":
                    o, _ = ops.spatial_attn_fwd_raw(buf[:, :C], buf[:, C:2 * C], buf[:, 2 * C:], pos0, plan, H, 0.25, 1.0, 0, s["p"], s["seed"])
                else:
                    o, _, _ = ops.spatial_attn_h_fwd_raw(buf, pos0, plan, H, 0.25, 1.0, s["p"], s["seed"])
                o = o.cpu().double().view(n, H, 16)
                for g in range(len(ptr) - 1):
                    a, b = ptr[g], ptr[g + 1]
                    lo, hi = max(16 * c, a), min(16 * c + 16, b)
                    if lo < hi:      # keys lo..hi-1 of graph g: o[q, h, k - 16c] = F[h, q, k] / n_g
                        F[g][:, :, lo - a:hi - a] = (o[a:b, :, lo - 16 * c:hi - 16 * c] * (b - a)).permute(1, 0, 2)
            keep = 1.0 / (1.0 - int(s["p"] * 65536) / 65536)
            for f in F:
                assert bool(((f.abs() < 1e-3) | ((f - keep).abs() < 1e-2)).all())
            self.masks.append([torch.where(f > 0.5 * keep, torch.full_like(f, keep), torch.zeros_like(f)) for f in F])
        return self

    def oracle_hook(self, ptr):
        order = {}

        def hook(site, x, p, graph):
            if site == "global_pool.attention.attn_dropout":
                return torch.ones_like(x)
            if site not in order:
                order[site] = len(order)
            s, m = self.sites[order[site]], self.masks[order[site]]
            assert abs(s["p"] - p) < 1e-12, (site, s, p)
            if s["kind"] == "attn":
                assert site.endswith("attn_dropout"), site
                m = m[graph]
            elif graph is not None:
                m = m[graph:graph + 1] if site.startswith("global_pool") else m[ptr[graph]:ptr[graph + 1]]
            assert m.shape == x.shape, (site, tuple(m.shape), tuple(x.shape))
            return m.to(x.dtype)
        hook.order = order
        return hook


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("hierarchical", [False, True])
def test_training_mode_step_matches_oracle_under_the_kernels_own_masks(hierarchical, fused, monkeypatch):
    """The step bench.py times runs in TRAINING mode: hash dropout at ~35 sites (kernel-level evidence only).
    Draw-for-draw parity with the reference's Philox masks is impossible, but the masks are deterministic functions of
    (seed, index): the kernels' own masks of one step are extracted site by site (_DropoutSites), handed to the float64 oracle
    (every F.dropout of the restatement multiplies by the kernel's mask of the same site instead of drawing), and the loss and
    EVERY live parameter gradient of the training-mode step must agree at the 1e-3 contract -- which proves, at model level, that
    each site applies its mask where the reference applies dropout (core/attention.py:154,168, graph_layers.py:233-239,
    encoders.py:73-91,267, diffusion.py:94-104), that forward and backward of every site regenerate the same mask, and that
    nothing on the path drops twice or not at all.  ``fused`` (round 5): the same step with the activations / norms behind the
    GEMMs as their epilogues (ops.FUSE_EPILOGUES; off by default -- measured slower -- but a complete second implementation of the
    same sites: same seeds in the same order, same (seed, element index) masks)."""
    from dgdm_histopath_lab_amd import ops
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    monkeypatch.setattr(ops, "FUSE_EPILOGUES", fused)
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8, use_hierarchical=hierarchical)
    cfg = O.OracleConfig(**cfgd)
    P = O.init_params(cfg, seed=3, perturb=0.05)
    batch = synthetic_batch(7, 2, 208, 832)
    gen = torch.Generator().manual_seed(23)
    n = batch.x.size(0)
    rng = dict(timesteps=torch.tensor([3, 8]), noise=torch.randn(n, 128, generator=gen), noise_target=torch.randn(n, 128, generator=gen))
    mask_idx = torch.randperm(n, generator=gen)[: int(n * 0.15)]
    mask_tok = torch.randn(768, generator=gen)
    ptr = [0, 208, 416]

    def hip_step(trace=None, decisions=None):
        m = _model(cfgd, P).train()
        m.global_pool.attention.attn_dropout.p = 0.0
        ops._seed_counter = 1000                      # both HIP runs draw the same seeds, site by site
        out = m.pretrain_step(batch.to(DEV), mask_indices=mask_idx.to(DEV), mask_token=mask_tok.to(DEV), trace=trace, decisions=decisions,
                              **{k: v.to(DEV) for k, v in rng.items()})
        return m, out

    torch.manual_seed(20240)                          # the seeds of the sites derive from torch.initial_seed()
    rec = _DropoutSites(ops, monkeypatch)
    hip_step()                                        # run 1: which sites drop, with which seeds, on which shapes
    sites = list(rec.sites)
    assert len(sites) >= (15 if not hierarchical else 30), len(sites)
    assert sum(s["kind"] == "attn" for s in sites) == 1
    rec.build_masks()
    for s, mk in zip(rec.sites, rec.masks):           # every mask drops about p of its elements
        for t in (mk if isinstance(mk, list) else [mk]):
            assert abs(float((t == 0).double().mean()) - s["p"]) < max(0.02, 5.0 * (s["p"] * (1 - s["p"]) / t.numel()) ** 0.5), s
    # oracle, float64, training mode, the kernels' masks injected
    torch.set_num_threads(16)
    b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(),
                                pos=batch.pos.double(), batch=batch.batch)
    tr64 = {} if hierarchical else None
    hook = rec.oracle_hook(ptr)
    monkeypatch.setattr(O, "DROPOUT_HOOK", hook)
    ref, gref = O.loss_and_grads({k: v.double() for k, v in P.items()}, cfg, b64, mask_indices=mask_idx, mask_token=mask_tok.double(),
                                 training=True, trace=tr64, **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
    monkeypatch.setattr(O, "DROPOUT_HOOK", None)
    assert len(hook.order) == len(sites), (len(hook.order), len(sites))       # the restatement has exactly the kernels' sites
    # run 2: same seeds (same masks), the checker's ReLU / top-k decisions injected where the U-Net is on
    rec.sites = []
    tr = {} if hierarchical else None
    dec = decisions_from_trace(tr64) if hierarchical else None
    m, out = hip_step(tr, dec)
    assert [(s["kind"], s["seed"]) for s in rec.sites] == [(s["kind"], s["seed"]) for s in sites]
    out["total_pretrain_loss"].backward()
    if hierarchical:
        check_decision_margins(tr, dec)
    assert_close(out["diffusion_loss"], ref["diffusion_loss"], TOL, "training-mode diffusion_loss")
    live = _assert_all_grads(m, gref, TOL)
    assert live >= 85
    named = dict(m.named_parameters())
    worst = max(float((named[k].grad.double().cpu() - g).norm() / g.norm()) for k, g in gref.items() if g.abs().max() > 1e-12)
    print(f"training mode, {len(sites)} dropout sites ({sum(s['kind'] == 'rownorm' for s in sites)} fused into row kernels), "
          f"loss {float(out['diffusion_loss'].detach()):.6f} vs {float(ref['diffusion_loss'].detach()):.6f}, {live} live gradients, worst rel-L2 {worst:.2e}")
    # and the masks matter: the eval-mode loss of the same weights is a different number
    ev = _model(cfgd, P).pretrain_step(batch.to(DEV), mask_indices=mask_idx.to(DEV), mask_token=mask_tok.to(DEV),
                                       **{k: v.to(DEV) for k, v in rng.items()})
    assert abs(float(ev["diffusion_loss"].detach()) - float(out["diffusion_loss"].detach())) > 1e-3 * float(out["diffusion_loss"].detach())


@pytest.mark.parametrize("attn", ["fp32", "fp16x2"])
def test_smooth_model_matches_oracle_2k_nodes_all_params(attn, monkeypatch):
    """cfg1-sized graphs (2 x 2000 nodes / 8000 edges), Base dims, use_hierarchical=False: the
    network is smooth (GELU / SiLU / softmax, no ReLU, no top-k), so EVERY live parameter gradient
    must agree with the exact (float64) oracle well inside the 1e-3 contract."""
    from dgdm_histopath_lab_amd import ops
    monkeypatch.setattr(ops, "ATTN_PRECISION", attn)
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8, use_hierarchical=False)
    m, out, ref, gref, _, _ = _run_both(cfgd, 0, trace=False)
    for k in ("diffusion_loss", "graph_embedding", "noisy_embeddings"):
        assert_close(out[k], ref[k], 1e-4, k)
    assert _assert_all_grads(m, gref, 1e-4) > 60


def _full_model_against_oracle(cfgd, seed0, nodes, edges, graphs, min_live, tweak=None, pos_fn=None):
    """U-Net on (ReLU + top-k): outputs, traced activations, top-k selections and EVERY live parameter gradient against the
    float64 oracle at the 1e-3 contract -- one fixed instance, no retry, no skip.  The oracle's kink decisions are injected into
    the kernels (GraphUNet.forward `decisions`), and every decision the HIP path would have taken differently is held to the
    rounding margin inside _run_both (conftest.check_decision_margins)."""
    m, out, ref, gref, tr, tr64 = _run_both(cfgd, seed0, trace=True, nodes=nodes, edges=edges, graphs=graphs, tweak=tweak, pos_fn=pos_fn)
    _full_model_against_oracle.last = (m, tr64)
    for k in ("diffusion_loss", "graph_embedding", "noisy_embeddings"):
        assert_close(out[k], ref[k], TOL, k)
    for k in ("feature_encoder", "graph_encoder", "spatial_attention", "graph_unet"):
        assert_close(tr[k], tr64[k], TOL, k)
    for k, v in tr64.items():
        if k.startswith("perm"):
            assert torch.equal(tr[k].cpu(), v), k            # the injected selection is the one that ran
    assert _assert_all_grads(m, gref, TOL) > min_live
    return tr["__flips__"]


@pytest.mark.parametrize("attn", ["fp32", "fp16x2"])
def test_full_model_matches_oracle_2k_nodes_all_params(attn, monkeypatch):
    """2 x 2000 nodes / 8000 edges, Base dims, the shipped default configuration (graph U-Net on), both attention precisions."""
    from dgdm_histopath_lab_amd import ops
    monkeypatch.setattr(ops, "ATTN_PRECISION", attn)
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8)
    flips, total = _full_model_against_oracle(cfgd, 0, 2000, 8000, 2, 100)
    assert total > 1_000_000


def test_full_model_on_raster_pixel_positions_matches_oracle():
    """Positions as the reference's preprocessing stores them (patch centres in level-0 pixels, row by row, 224 apart;
    preprocessing/tissue_graph_builder.py:381-384): with temperature 1 every attention row is one-hot up to weights that are 0.0f in
    fp32, the zero-block map (csrc/attn_skip.hip) walks over 90 % of the block pairs, and outputs, traced activations and every live
    parameter gradient still meet the float64 oracle at the 1e-3 contract."""
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8)

    def raster(batch):
        n = batch.x.size(0) // 2
        i = torch.arange(n)
        one = torch.stack([(i % 45).float(), (i // 45).float()], 1) * 224.0
        return torch.cat([one, one + 1000.0])

    _full_model_against_oracle(cfgd, 5, 2000, 8000, 2, 100, pos_fn=raster)


def test_full_model_matches_oracle_at_the_headline_graph_size():
    """One BASELINE configs[1] graph (10 000 nodes / 50 000 edges), default path (split-fp16 attention, graph U-Net on)."""
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8)
    torch.set_num_threads(32)
    _full_model_against_oracle(cfgd, 7, 10000, 50000, 1, 100)


def test_full_model_matches_oracle_on_the_headline_batch():
    """BASELINE configs[1] exactly: a batch of 4 graphs of 10 000 nodes / 50 000 edges through the default path (the top-k of
    every pooling level ranks the 40 000 / 20 000 / 10 000 nodes of the whole batch, graph_layers.py:306-310)."""
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8)
    torch.set_num_threads(32)
    _full_model_against_oracle(cfgd, 20, 10000, 50000, 4, 100)


def test_large_config_matches_oracle_all_params():
    """BASELINE configs[3] dims (DGDM-Large: hidden [1024, 512, 256], 16 heads, T = 20) on 2 x 600-node
    graphs, smooth variant: every live gradient against the float64 oracle."""
    cfgd = dict(node_features=768, hidden_dims=[1024, 512, 256], num_diffusion_steps=20, attention_heads=16, use_hierarchical=False)
    m, out, ref, gref, _, _ = _run_both(cfgd, 5, trace=False, nodes=600, edges=3000)
    for k in ("diffusion_loss", "graph_embedding", "noisy_embeddings"):
        assert_close(out[k], ref[k], 1e-4, k)
    assert _assert_all_grads(m, gref, 2e-4) > 60


def test_large_config_with_unet_matches_oracle_all_params():
    """BASELINE configs[3] dims WITH the hierarchical U-Net (core/graph_layers.py:400-458 at hidden 256, 16 heads, T = 20; three
    top-k levels) on 2 x 2000-node graphs: outputs, traced activations, bit-exact selections and every live gradient against the
    float64 oracle, decisions injected and held to the rounding margin."""
    cfgd = dict(node_features=768, hidden_dims=[1024, 512, 256], num_diffusion_steps=20, attention_heads=16)
    torch.set_num_threads(32)
    flips, total = _full_model_against_oracle(cfgd, 5, 2000, 8000, 2, 100)
    assert total > 2_000_000


def _attention_row_entropy(P, cfg, tr64, batch_pos, n, H):
    """Mean row entropy (nats) of the spatial attention of the first graph, from the float64 trace."""
    import math
    h, pos = tr64["graph_encoder"][:n].double(), batch_pos[:n].double()
    P64 = {k: v.double() for k, v in P.items() if k.startswith("spatial_attention.attention.")}
    xp = h + O.sinusoid_pos_encoding(pos, h.size(1)).double()
    q = O._lin(P64, "spatial_attention.attention.q_proj", xp).view(n, H, -1).transpose(0, 1)
    k = O._lin(P64, "spatial_attention.attention.k_proj", xp).view(n, H, -1).transpose(0, 1)
    w = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(q.size(-1)) - torch.norm(pos[:, None] - pos[None], dim=-1), dim=-1)
    return float(-(w * torch.log(w.clamp_min(1e-300))).sum(-1).mean())


@pytest.mark.parametrize("qk_scale,max_entropy", [(3.0, 2.0), (4.0, 1.0)])
def test_full_model_with_sharp_attention_rows_matches_oracle(qk_scale, max_entropy):
    """The default path on TRAINED-LIKE attention: q_proj / k_proj weights scaled until the mean row entropy
    is far below 0.5 ln N (1.4 and 0.65 nats at N = 2000, against 7.26 at the near-init weights every other model-level test uses);
    2 x 2000 nodes, Base dims, U-Net on; outputs and every live gradient at the 1e-3 contract."""
    import math
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8)
    scaled = {}

    def tweak(P):
        for n in ("q_proj", "k_proj"):
            P[f"spatial_attention.attention.{n}.weight"] *= qk_scale
        scaled.update(P)
    torch.set_num_threads(32)
    _full_model_against_oracle(cfgd, 0, 2000, 8000, 2, 100, tweak=tweak)
    _, tr64 = _full_model_against_oracle.last
    ent = _attention_row_entropy(scaled, O.OracleConfig(**cfgd), tr64, synthetic_batch(0, 2, 2000, 8000).pos, 2000, 8)
    print(f"qk_scale {qk_scale}: mean row entropy {ent:.2f} nats (ln N = {math.log(2000):.2f})")
    assert ent < max_entropy <= 0.5 * math.log(2000)


def test_large_config_full_size_graph_is_an_independent_unit():
    """configs[3] at full size (one 50 000-node / 300 000-edge graph, DGDM-Large incl. the hierarchical U-Net):
    the step runs, everything is finite, and -- size-independent property of the path -- a graph's embedding
    does not depend on what else is in the batch (block-diagonal edges, per-graph attention / pooling;
    checked without the U-Net, whose top-k ranks nodes over the whole batch, graph_layers.py:306-310)."""
    from dgdm_histopath_lab_amd import DGDMModel, GraphBatch
    from dgdm_histopath_lab_amd.synthetic import synthetic_graph
    torch.manual_seed(0)
    big = synthetic_graph(0, 50000, 300000, 768)
    small = synthetic_graph(1, 3000, 12000, 768)
    cfg = dict(node_features=768, hidden_dims=[1024, 512, 256], num_diffusion_steps=20, attention_heads=16)
    m = DGDMModel(**cfg).to(DEV)
    out = m.pretrain_step(GraphBatch.from_data_list([big]).to(DEV))
    out["total_pretrain_loss"].backward()
    assert torch.isfinite(out["total_pretrain_loss"])
    live = [p.grad for p in m.parameters() if p.grad is not None]
    assert len(live) > 60 and all(torch.isfinite(g).all() for g in live)
    del out, live
    m2 = DGDMModel(use_hierarchical=False, **cfg).to(DEV).eval()
    with torch.no_grad():
        alone = m2(GraphBatch.from_data_list([big]).to(DEV), mode="inference")["graph_embedding"]
        both = m2(GraphBatch.from_data_list([small, big]).to(DEV), mode="inference")["graph_embedding"]
    assert_close(both[1:2], alone.double(), 1e-5, "embedding of the 50k graph alone vs in a batch")


@pytest.mark.parametrize("variant", ["ragged", "no_edge_attr", "no_pos"])
def test_ragged_batches_and_missing_optional_fields_match_oracle(variant):
    """Graph sizes around the kernels' tile edges (1, 2, 63, 64, 65, 130, 777 nodes), one graph without any edge,
    duplicate edges and pre-existing self loops; optional fields absent (edge_attr None -> zeros, encoders.py:258-261;
    pos None -> no spatial attention, dgdm_model.py:341).  Smooth model, every live gradient against float64."""
    from dgdm_histopath_lab_amd import GraphBatch, GraphData
    cfgd = dict(node_features=96, hidden_dims=[128, 64, 64], num_diffusion_steps=10, attention_heads=4, use_hierarchical=False)
    cfg = O.OracleConfig(**cfgd)
    P = O.init_params(cfg, seed=11, perturb=0.05)
    gen = torch.Generator().manual_seed(21)
    graphs = []
    for gi, n in enumerate([1, 2, 63, 64, 65, 130, 777]):
        e = 0 if gi == 2 else (0 if n == 1 else 3 * n)
        ei = torch.randint(0, n, (2, e), generator=gen)
        if e >= 8:
            ei[:, 1] = ei[:, 0]                 # duplicate edge
            ei[1, 2] = ei[0, 2]                 # pre-existing self loop
        graphs.append(GraphData(x=torch.randn(n, 96, generator=gen), edge_index=ei,
                                edge_attr=None if variant == "no_edge_attr" else torch.randn(e, 32, generator=gen),
                                pos=None if variant == "no_pos" else torch.rand(n, 2, generator=gen)))
    batch = GraphBatch.from_data_list(graphs)
    n = batch.x.size(0)
    B = len(graphs)
    rng = dict(timesteps=torch.randint(0, 10, (B,), generator=gen), noise=torch.randn(n, 64, generator=gen),
               noise_target=torch.randn(n, 64, generator=gen))
    mask_idx = torch.randperm(n, generator=gen)[: int(n * 0.15)]
    mask_tok = torch.randn(96, generator=gen)
    b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index,
                                edge_attr=None if batch.edge_attr is None else batch.edge_attr.double(),
                                pos=None if batch.pos is None else batch.pos.double(), batch=batch.batch)
    ref, gref = O.loss_and_grads({k: v.double() for k, v in P.items()}, cfg, b64, mask_indices=mask_idx, mask_token=mask_tok.double(),
                                 **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
    m = _model(cfgd, P)
    out = m.pretrain_step(batch.to(DEV), mask_indices=mask_idx.to(DEV), mask_token=mask_tok.to(DEV), **{k: v.to(DEV) for k, v in rng.items()})
    out["total_pretrain_loss"].backward()
    for k in ("diffusion_loss", "graph_embedding", "noisy_embeddings"):
        assert_close(out[k], ref[k], 1e-4, k)
    assert out["graph_embedding"].shape == (B, 64)
    assert _assert_all_grads(m, gref, 5e-4) > 40          # k_proj gradients are near-cancelling sums: half the 1e-3 contract


@pytest.mark.parametrize("sizes", [[1], [2, 1], [5], [3, 9, 1, 17]])
def test_tiny_batches_through_the_unet_match_oracle(sizes):
    """The hierarchical path at degenerate sizes: top-k levels shrink to k = max(1, int(0.5 N)) = 1 node, graphs lose
    all their edges after pooling (graph_layers.py:298-327).  Forward outputs against the oracle; backward must run."""
    from dgdm_histopath_lab_amd import GraphBatch, GraphData
    cfgd = dict(node_features=32, hidden_dims=[32, 32, 32], num_diffusion_steps=10, attention_heads=2)
    cfg = O.OracleConfig(**cfgd)
    P = O.init_params(cfg, seed=4, perturb=0.05)
    gen = torch.Generator().manual_seed(sum(sizes))
    graphs = []
    for n in sizes:
        e = 0 if n == 1 else 2 * n
        graphs.append(GraphData(x=torch.randn(n, 32, generator=gen), edge_index=torch.randint(0, n, (2, e), generator=gen),
                                edge_attr=torch.randn(e, 32, generator=gen), pos=torch.rand(n, 2, generator=gen)))
    batch = GraphBatch.from_data_list(graphs)
    n, B = batch.x.size(0), len(sizes)
    rng = dict(timesteps=torch.randint(0, 10, (B,), generator=gen), noise=torch.randn(n, 32, generator=gen),
               noise_target=torch.randn(n, 32, generator=gen))
    b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(),
                                pos=batch.pos.double(), batch=batch.batch)
    ref = O.forward({k: v.double() for k, v in P.items()}, cfg, b64, mode="pretrain",
                    **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
    m = _model(cfgd, P)
    out = m(batch.to(DEV), mode="pretrain", **{k: v.to(DEV) for k, v in rng.items()})
    for k in ("diffusion_loss", "graph_embedding"):
        assert_close(out[k], ref[k], TOL, k)
    out["diffusion_loss"].backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_forward_matches_oracle_at_the_headline_graph_size():
    """One BASELINE configs[1] graph (10 000 nodes / 50 000 edges, 768 features, Base dims) through encoder, spatial
    attention, diffusion loss and attention pooling: forward outputs against the float64 oracle (the U-Net is left out because a single top-k flip among 10 000 scores would move a node)."""
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8, use_hierarchical=False)
    cfg = O.OracleConfig(**cfgd)
    P = O.init_params(cfg, seed=3, perturb=0.05)
    batch = synthetic_batch(0, 1, 10000, 50000)
    gen = torch.Generator().manual_seed(99)
    n = batch.x.size(0)
    rng = dict(timesteps=torch.tensor([4]), noise=torch.randn(n, 128, generator=gen), noise_target=torch.randn(n, 128, generator=gen))
    torch.set_num_threads(32)
    b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(),
                                pos=batch.pos.double(), batch=batch.batch)
    with torch.no_grad():
        ref = O.forward({k: v.double() for k, v in P.items()}, cfg, b64, mode="pretrain", return_embeddings=True,
                        **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
        m = _model(cfgd, P)
        out = m(batch.to(DEV), mode="pretrain", return_embeddings=True, **{k: v.to(DEV) for k, v in rng.items()})
    for k in ("diffusion_loss", "graph_embedding", "node_embeddings", "noisy_embeddings"):
        assert_close(out[k], ref[k], 2e-4, k)


def test_gradients_match_oracle_at_the_headline_graph_size():
    """The same graph, one full pretrain_step (masking, forward, backward): every live parameter gradient against the
    float64 oracle (dense 10 000 x 10 000 x 8 attention in float64 on the host: ~10 s on the GPU box's cores)."""
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8, use_hierarchical=False)
    torch.set_num_threads(32)
    m, out, ref, gref, _, _ = _run_both(cfgd, 7, trace=False, nodes=10000, edges=50000, graphs=1)
    for k in ("diffusion_loss", "graph_embedding", "noisy_embeddings"):
        assert_close(out[k], ref[k], 2e-4, k)
    assert _assert_all_grads(m, gref, 5e-4) > 60


@pytest.mark.parametrize("pooling", ["mean", "max", "set2set"])
def test_non_default_pools_match_oracle(pooling):
    """dgdm_model.py:552-585, 618-642: graph embeddings of the mean / max / "set2set" (= mean) pools on a ragged three-graph batch
    (one graph of two nodes), forward against the float64 oracle and the gradient that reaches the node embeddings."""
    from dgdm_histopath_lab_amd.graph import GraphBatch
    from dgdm_histopath_lab_amd.synthetic import synthetic_graph
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8, use_hierarchical=False,
                pooling=pooling)
    cfg = O.OracleConfig(**cfgd)
    P = O.init_params(cfg, seed=5, perturb=0.05)
    batch = GraphBatch.from_data_list([synthetic_graph(0, 700, 2800), synthetic_graph(1, 2, 2), synthetic_graph(2, 333, 1500)])
    b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(),
                                pos=batch.pos.double(), batch=batch.batch)
    P64 = {k: v.double().requires_grad_(k.startswith("graph_encoder.output_proj")) for k, v in P.items()}
    ref = O.forward(P64, cfg, b64, mode="inference", return_embeddings=True)
    w = torch.randn(ref["graph_embedding"].shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    (ref["graph_embedding"] * w).sum().backward()
    if pooling == "set2set":     # the reference's unused LSTM (dgdm_model.py:623) has no counterpart in the oracle's parameters
        from dgdm_histopath_lab_amd import DGDMModel
        m = DGDMModel(**cfgd)
        res = m.load_state_dict(dict(P), strict=False)
        assert not res.unexpected_keys and all(k.startswith("global_pool.lstm.") for k in res.missing_keys)
        m = m.to(DEV).eval()
    else:
        m = _model(cfgd, P)
    out = m(batch.to(DEV), mode="inference", return_embeddings=True)
    assert out["graph_embedding"].shape == (3, 128)
    assert_close(out["graph_embedding"], ref["graph_embedding"], 2e-4, "graph_embedding")
    assert_close(out["node_embeddings"], ref["node_embeddings"], 2e-4, "node_embeddings")
    (out["graph_embedding"] * w.to(DEV).float()).sum().backward()
    for k in ("graph_encoder.output_proj.weight", "graph_encoder.output_proj.bias"):
        assert_close(dict(m.named_parameters())[k].grad, P64[k].grad, 1e-3, k)


def test_model_error_contract():
    from dgdm_histopath_lab_amd import DGDMModel, GraphData, ModelConfigurationError, ModelInferenceError
    with pytest.raises(ModelConfigurationError):
        DGDMModel(hidden_dims=[])
    with pytest.raises(ModelConfigurationError):
        DGDMModel(pooling="nope")
    m = DGDMModel(node_features=32, hidden_dims=[32, 16], attention_heads=1, graph_layers=3).to(DEV).eval()
    x = torch.randn(10, 32, device=DEV); ei = torch.randint(0, 10, (2, 20), device=DEV)
    ok = m(GraphData(x=x, edge_index=ei, pos=torch.rand(10, 2, device=DEV)))
    assert ok["graph_embedding"].shape == (1, 16)
    bad = x.clone(); bad[0, 0] = float("nan")
    with pytest.raises(ModelInferenceError, match="NaN"):
        m(GraphData(x=bad, edge_index=ei))
    with pytest.raises(ModelInferenceError, match="node features"):
        m(GraphData(x=torch.randn(10, 31, device=DEV), edge_index=ei))
    with pytest.raises(ModelInferenceError, match="invalid node indices"):
        m(GraphData(x=x, edge_index=ei + 5))
    neg = ei.clone(); neg[1, 7] = -1
    with pytest.raises(ModelInferenceError, match="negative"):
        m(GraphData(x=x, edge_index=neg))
    inf = x.clone(); inf[9, 31] = float("-inf")           # last element: the scalar tail of the vectorised pass
    with pytest.raises(ModelInferenceError, match="infinity"):
        m(GraphData(x=inf, edge_index=ei))
    odd = torch.randn(11, 33, device=DEV)[:, :32]          # non-contiguous x takes the tensor-expression checks
    odd[3, 3] = float("nan")
    with pytest.raises(ModelInferenceError, match="NaN"):
        m(GraphData(x=odd, edge_index=ei))
    with pytest.raises(ModelInferenceError):
        m(GraphData(x=x.cpu(), edge_index=ei.cpu()))  # no CPU fallback


def test_generate_embeddings_and_direct_diffusion_loss_match_the_oracle():
    """VERDICT r3 item 7 -- two boundary methods SURVEY 8(b) names had no direct test:
    * `generate_embeddings(data, layer)` (reference models/dgdm_model.py:527): "final" = the graph embedding, "node" = the node
      embeddings of an inference forward, under no_grad; anything else raises ValueError;
    * `_compute_diffusion_loss(node_embeddings, data)` called DIRECTLY (reference :405-445) on given node embeddings with injected
      draws: loss, the last graph's noisy embeddings and the gradient w.r.t. the embeddings against the float64 oracle's
      add_noise / predict_noise / per-graph MSE (ragged graphs, so the mean over graphs of per-graph means is not a plain mean)."""
    import torch.nn.functional as F
    from dgdm_histopath_lab_amd import GraphBatch
    from dgdm_histopath_lab_amd.synthetic import synthetic_graph
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8)
    cfg = O.OracleConfig(**cfgd)
    P = O.init_params(cfg, seed=5, perturb=0.05)
    m = _model(cfgd, P)
    sizes = [700, 1300, 333]
    batch = GraphBatch.from_data_list([synthetic_graph(40 + i, n, 4 * n, 768) for i, n in enumerate(sizes)])
    P64 = {k: v.double() for k, v in P.items()}
    b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(),
                                pos=batch.pos.double(), batch=batch.batch)
    torch.set_num_threads(16)
    ref = O.forward(P64, cfg, b64, mode="inference", return_embeddings=True)
    dev_batch = batch.to(DEV)
    for p in m.parameters():
        p.requires_grad_(True)
    ge = m.generate_embeddings(dev_batch)                      # layer="final" is the default
    ne = m.generate_embeddings(dev_batch, layer="node")
    assert not ge.requires_grad and not ne.requires_grad      # computed under no_grad, as the reference does
    assert_close(ge, ref["graph_embedding"], TOL, "generate_embeddings(final)")
    assert_close(ne, ref["node_embeddings"], TOL, "generate_embeddings(node)")
    assert ge.shape == (3, 128) and ne.shape == (sum(sizes), 128)
    with pytest.raises(ValueError):
        m.generate_embeddings(dev_batch, layer="middle")

    # _compute_diffusion_loss on GIVEN embeddings
    gen = torch.Generator().manual_seed(17)
    n = sum(sizes)
    h = torch.randn(n, 128, generator=gen)
    ts = torch.tensor([3, 9, 0])
    noise, target = torch.randn(n, 128, generator=gen), torch.randn(n, 128, generator=gen)
    h64 = h.double().requires_grad_(True)
    sched = O.diffusion_schedule(cfg.num_diffusion_steps, cfg.diffusion_schedule)
    ptr = [0, 700, 2000, 2333]
    losses, noisy = [], None
    for g in range(3):
        sl = slice(ptr[g], ptr[g + 1])
        noisy = O.add_noise(sched, h64[sl], noise[sl].double(), ts[g:g + 1])
        pred = O.predict_noise(P64, noisy, ts[g:g + 1], 0.1, False)
        losses.append(F.mse_loss(pred, target[sl].double()))       # strict_reference: the target is the second draw (D8)
    want = torch.stack(losses).mean()
    want.backward()
    hd = h.to(DEV).requires_grad_(True)
    out = m._compute_diffusion_loss(hd, dev_batch, timesteps=ts.to(DEV), noise=noise.to(DEV), noise_target=target.to(DEV))
    assert set(out) == {"diffusion_loss", "noisy_embeddings"}
    assert_close(out["diffusion_loss"], want, TOL, "diffusion_loss (direct call)")
    assert out["noisy_embeddings"].shape == (1, 333, 128)
    assert_close(out["noisy_embeddings"][0], noisy, TOL, "noisy_embeddings (last graph)")
    out["diffusion_loss"].backward()
    assert_close(hd.grad, h64.grad, TOL, "d loss / d node_embeddings")
    dl = dict(m.named_parameters())
    assert dl["diffusion_layer.denoise_net.0.weight"].grad is not None
    # without injected draws the method draws its own (timesteps, noise, target): finite, and different from call to call
    a = float(m._compute_diffusion_loss(hd.detach(), dev_batch)["diffusion_loss"])
    b = float(m._compute_diffusion_loss(hd.detach(), dev_batch)["diffusion_loss"])
    assert np.isfinite(a) and np.isfinite(b) and a != b


def test_unet_step_that_took_its_own_decisions_matches_the_oracle_under_those_decisions():
    """every other U-Net gradient test differentiates under the CHECKER's ReLU / top-k decisions (handed to the
    kernels).  Here the direction is reversed: the HIP model runs one pretrain step end to end with NOTHING injected -- its own
    ReLU sides, its own top-k selections -- and the float64 oracle is then told those decisions (oracle.DECISIONS: which elements
    passed each ReLU, which nodes each pooling level kept).  Loss, embeddings and every live gradient of the kernels' own run at
    the 1e-3 contract; the oracle, left to itself, must disagree with those decisions only inside the rounding margin."""
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8)
    cfg = O.OracleConfig(**cfgd)
    P = O.init_params(cfg, seed=3, perturb=0.05)
    batch = synthetic_batch(7, 2, 2000, 8000)
    gen = torch.Generator().manual_seed(23)
    n = batch.x.size(0)
    rng = dict(timesteps=torch.tensor([4, 8]), noise=torch.randn(n, 128, generator=gen), noise_target=torch.randn(n, 128, generator=gen))
    mask_idx, mask_tok = torch.randperm(n, generator=gen)[: int(n * 0.15)], torch.randn(768, generator=gen)
    m = _model(cfgd, P)
    own = {}
    out = m.pretrain_step(batch.to(DEV), mask_indices=mask_idx.to(DEV), mask_token=mask_tok.to(DEV), trace=own,
                          **{k: v.to(DEV) for k, v in rng.items()})          # no `decisions=`: the kernels decide
    out["total_pretrain_loss"].backward()
    dec = {}
    for k, v in own.items():
        if k.startswith("pre."):
            dec["relu." + k[4:]] = (v.detach() > 0).cpu()
        elif k.startswith("own_perm"):
            dec["perm" + k[8:]] = v.detach().cpu()
    assert sorted(dec) == sorted([f"relu.down{i}" for i in range(3)] + [f"relu.pool{i}" for i in range(3)] + [f"relu.up{i}" for i in range(3)] +
                                 ["relu.bottom"] + [f"perm{i}" for i in range(3)])
    torch.set_num_threads(32)
    b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(), pos=batch.pos.double(),
                                batch=batch.batch)
    kw = dict(mask_indices=mask_idx, mask_token=mask_tok.double(), **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
    O.DECISIONS = dec
    try:
        ref, gref = O.loss_and_grads({k: v.double() for k, v in P.items()}, cfg, b64, **kw)
    finally:
        O.DECISIONS = None
    for k in ("diffusion_loss", "graph_embedding", "noisy_embeddings"):
        assert_close(out[k], ref[k], TOL, k)
    assert _assert_all_grads(m, gref, TOL) >= 170
    # the oracle's own decisions (no injection) against the kernels' pre-activations and scores: differences only within the margin
    free = {}
    O.loss_and_grads({k: v.double() for k, v in P.items()}, cfg, b64, trace=free, **kw)
    flips, total = check_decision_margins(own, decisions_from_trace(free))
    print(f"decisions differing between the kernels' own run and the float64 run: {flips} of {total}")

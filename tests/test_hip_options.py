"""GPU parity of the constructor options that are NOT DGDMModel's defaults: activation "relu" / "elu",
normalization "batch" / "instance" / none, pooling "max" / "mean", diffusion_schedule "linear" / "sigmoid" -- against vectors
captured by running the reference's own classes with those arguments (tests/golden/g10_*, oracle/capture_golden.py) and against the
float64 oracle at model level.  Reference: models/encoders.py:57-64,95-100,202-219; models/dgdm_model.py:552-585; core/diffusion.py:29-61."""
import types

import pytest
import torch

from conftest import T, assert_close, load_golden, weights
from oracle import dgdm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("tag,act,norm", [("relu_batch", "relu", "batch"), ("elu_instance", "elu", "instance"), ("elu_layer", "elu", "layer"),
                                          ("relu_none", "relu", "none")])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_feature_encoder_options_match_reference_golden(tag, act, norm, mode):
    """The product FeatureEncoder built with the reference's non-default arguments, the reference's weights loaded with
    strict=True (BatchNorm1d buffers included; InstanceNorm1d has no parameters), against the reference's own outputs and
    gradients.  ELU / ReLU and the instance norm run on this library's row kernels (csrc/rownorm.hip, elementwise.hip), BatchNorm1d
    with its activation on the column-norm kernels (csrc/colnorm.hip)."""
    from dgdm_histopath_lab_amd.models.encoders import FeatureEncoder
    g = load_golden(f"g10_feature_encoder_{tag}_{mode}")
    fe = FeatureEncoder(48, 32, dropout=0.0, activation=act, normalization=norm)
    fe.load_state_dict(weights(g), strict=True)
    fe = fe.to(DEV).train(mode == "train")
    x = T(g["x"]).to(DEV).requires_grad_(True)
    y = fe(x)
    assert_close(y, g["y"], 1e-4, "y")
    (y * T(g["gy"]).to(DEV)).sum().backward()
    assert_close(x.grad, g["gx"], 1e-4, "gx")
    assert_close(fe.encoder[0].weight.grad, g["gw"], 1e-4, "gw")


@pytest.mark.parametrize("act,norm", [("elu", "instance"), ("relu", "layer"), ("elu", "graph")])   # "graph": accepted by the model's validation, nn.Identity() in both encoders
def test_encoders_with_options_match_the_float64_oracle_at_tile_gemm_sizes(act, norm):
    """The same options on 1 500 rows (the tile GEMMs and the fused row kernels instead of the small-M kernels), FeatureEncoder and
    the activation / norm behind every GraphEncoder layer, against the float64 restatement."""
    from dgdm_histopath_lab_amd import DGDMModel
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    cfgd = dict(node_features=96, hidden_dims=[64, 64, 32], num_diffusion_steps=10, attention_heads=2, use_hierarchical=False, activation=act,
                normalization=norm, dropout=0.0)
    cfg = O.OracleConfig(**cfgd)
    P = O.init_params(cfg, seed=7, perturb=0.05)
    m = DGDMModel(**cfgd)
    m.load_state_dict(P, strict=True)
    m = m.to(DEV).eval()
    batch = synthetic_batch(3, 2, 750, 3000, 96)
    b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(), pos=batch.pos.double(),
                                batch=batch.batch)
    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    tr64, tr = {}, {}
    ref = O.forward(P64, cfg, b64, mode="inference", trace=tr64)
    out = m(batch.to(DEV), mode="inference", return_embeddings=True, trace=tr)
    assert_close(tr["feature_encoder"], tr64["feature_encoder"], 1e-4, "feature_encoder")
    assert_close(tr["graph_encoder"], tr64["graph_encoder"], 1e-4, "graph_encoder")
    assert_close(out["graph_embedding"], ref["graph_embedding"], 1e-4, "graph_embedding")
    go = torch.randn(ref["graph_embedding"].shape, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    (ref["graph_embedding"] * go).sum().backward()
    (out["graph_embedding"] * go.float().to(DEV)).sum().backward()
    named = dict(m.named_parameters())
    n = 0
    for k, v in P64.items():
        if v.grad is not None and k.startswith(("feature_encoder", "graph_encoder")) and float(v.grad.abs().max()) > 1e-12:
            assert_close(named[k].grad, v.grad, 1e-3, "grad " + k); n += 1
    assert n >= 30


@pytest.mark.parametrize("name", ["max", "mean"])
def test_global_pools_match_reference_golden(name):
    from dgdm_histopath_lab_amd.models.dgdm_model import GlobalMaxPool, GlobalMeanPool
    g = load_golden(f"g10_pool_{name}")
    x = T(g["x"]).to(DEV).requires_grad_(True)
    out = (GlobalMaxPool() if name == "max" else GlobalMeanPool())(x, T(g["batch"]).to(DEV))
    assert_close(out, g["out"], 1e-6, "out")
    (out * T(g["go"]).to(DEV)).sum().backward()
    assert_close(x.grad, g["gx"], 1e-6, "gx")


@pytest.mark.parametrize("rows,C", [([5000, 1, 0, 777, 10000], 128), ([300], 36), ([64, 64], 512), ([3, 2, 1], 4)])
def test_segment_max_ties_empty_graphs_and_big_segments(rows, C):
    """dgdm_segment_max_*: bit-exact against torch's per-graph ``x[a:b].max(dim=0)`` (values AND the maximising row), the FIRST row on
    ties, zeros / no gradient for a graph without nodes (models/dgdm_model.py:577-583 leaves its zero row untouched)."""
    from dgdm_histopath_lab_amd import ops
    g = torch.Generator().manual_seed(sum(rows) + C)
    ptr = [0]
    for r in rows:
        ptr.append(ptr[-1] + r)
    n = ptr[-1]
    x = torch.randn(n, C, generator=g)
    x[torch.randint(0, n, (n // 3 + 1,), generator=g)] = 2.5        # many tied maxima (whole rows)
    xd = x.to(DEV).requires_grad_(True)
    plan = ops.AttnPlan(ptr, torch.device(DEV))
    out, arg = ops.segment_max(xd, plan, return_arg=True)
    go = torch.randn(len(rows), C, generator=g)
    (out * go.to(DEV)).sum().backward()
    want_dx = torch.zeros(n, C)
    for i, (a, b) in enumerate(zip(ptr[:-1], ptr[1:])):
        if a == b:
            assert torch.equal(out[i].cpu(), torch.zeros(C)) and bool((arg[i] == -1).all())
            continue
        v, idx = x[a:b].max(dim=0)
        first = (x[a:b] == v).float().argmax(dim=0)                   # the first row that attains the maximum
        assert torch.equal(out[i].cpu(), v), i
        assert torch.equal(arg[i].cpu().long(), first + a), i
        want_dx[first + a, torch.arange(C)] = go[i]
    assert torch.equal(xd.grad.cpu(), want_dx)


@pytest.mark.parametrize("opts", [dict(activation="elu", normalization="instance", pooling="max", diffusion_schedule="linear"),
                                  dict(activation="relu", normalization="layer", pooling="mean", diffusion_schedule="sigmoid"),
                                  dict(activation="gelu", normalization="graph", pooling="attention", diffusion_schedule="cosine")])
def test_model_with_non_default_options_matches_oracle(opts):
    """One pretrain_step of the whole model (U-Net on, Base widths, 2 x 600 nodes) built with non-default constructor values against the
    float64 oracle: loss, embeddings and every live gradient at 1e-3 (the diffusion schedule enters through the q-sample tables)."""
    from test_hip_model import _assert_all_grads, _run_both
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=20, attention_heads=8, **opts)
    m, out, ref, gref, tr, tr64 = _run_both(cfgd, 5, True, nodes=600, edges=2400, graphs=2)
    assert_close(out["diffusion_loss"], ref["diffusion_loss"], 1e-3, "diffusion_loss")
    assert_close(out["graph_embedding"], ref["graph_embedding"], 1e-3, "graph_embedding")
    assert_close(out["noisy_embeddings"], ref["noisy_embeddings"], 1e-3, "noisy_embeddings")
    assert _assert_all_grads(m, gref, 1e-3) >= 100


def test_set2set_pooling_is_the_mean_as_in_the_reference():
    """models/dgdm_model.py:618-642: the reference's "simplified Set2Set" returns the per-graph mean and never runs its LSTM."""
    from dgdm_histopath_lab_amd.models.dgdm_model import GlobalMeanPool, GlobalSet2SetPool
    g = load_golden("g10_pool_mean")
    x, batch = T(g["x"]).to(DEV), T(g["batch"]).to(DEV)
    pool = GlobalSet2SetPool(32).to(DEV)
    assert torch.equal(pool(x, batch), GlobalMeanPool()(x, batch))
    assert_close(pool(x, batch), g["out"], 1e-6, "set2set == mean")


@pytest.mark.parametrize("n,c", [(37, 32), (3000, 512), (40000, 128), (257, 36)])
@pytest.mark.parametrize("act,p", [(0, 0.0), (2, 0.0), (4, 0.0), (1, 0.2)])
def test_column_norm_is_batchnorm1d_with_activation(n, c, act, p, monkeypatch):
    """dgdm_colnorm_* (nn.BatchNorm1d over the nodes of a batch + activation + dropout, models/encoders.py:95-100 with
    normalization="batch") against torch's own module in float64 on the CPU: training mode (batch statistics; running averages and
    num_batches_tracked advance exactly as the module's), eval mode (running averages), outputs, input / affine gradients; with
    dropout the mask is taken from the kernels' own (seed, element index) function (dgdm_act_dropout_fwd on ones)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from dgdm_histopath_lab_amd import _lib, ops
    acts = {0: lambda z: z, 1: F.gelu, 2: F.relu, 4: F.elu}
    g = torch.Generator().manual_seed(n + c + act)
    x = torch.randn(n, c, generator=g) * 1.7 + 0.4
    gy = torch.randn(n, c, generator=g)
    ref = nn.BatchNorm1d(c).double()
    with torch.no_grad():
        ref.weight.copy_(1 + 0.3 * torch.randn(c, generator=g)); ref.bias.copy_(0.2 * torch.randn(c, generator=g))
        ref.running_mean.copy_(0.1 * torch.randn(c, generator=g)); ref.running_var.copy_(1 + 0.2 * torch.rand(c, generator=g))
    own = nn.BatchNorm1d(c)
    own.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in ref.state_dict().items()})
    own = own.to(DEV)
    for mode in ("train", "eval"):
        ref.train(mode == "train"); own.train(mode == "train")
        ref.zero_grad(); own.zero_grad()
        monkeypatch.setattr(ops, "_seed_counter", 500)
        xd = x.to(DEV).requires_grad_(True)
        y = ops.batch_norm(xd, own, act, p, mode == "train")
        mask = 1.0
        if p > 0 and mode == "train":
            seed = (torch.initial_seed() * 0x9E3779B1 + 501 * 0x85EBCA6B) & 0xFFFFFFFF
            ones, m = torch.ones(n * c, device=DEV), torch.empty(n * c, device=DEV)
            _lib.check(_lib.load().dgdm_act_dropout_fwd(ones.data_ptr(), n * c, 0, p, seed, m.data_ptr(), None, None, _lib.stream_ptr(ones.device)), "mask")
            mask = m.view(n, c).cpu().double()
        xr = x.double().requires_grad_(True)
        yr = acts[act](ref(xr)) * mask
        assert_close(y, yr, 2e-5, f"y {mode}")
        y.backward(gy.to(DEV)); yr.backward(gy.double())
        assert_close(xd.grad, xr.grad, 5e-5, f"dx {mode}")
        assert_close(own.weight.grad, ref.weight.grad, 5e-5, f"dgamma {mode}")
        assert_close(own.bias.grad, ref.bias.grad, 5e-5, f"dbeta {mode}")
        assert_close(own.running_mean, ref.running_mean, 1e-5, f"running_mean {mode}")
        assert_close(own.running_var, ref.running_var, 1e-5, f"running_var {mode}")
        assert int(own.num_batches_tracked) == int(ref.num_batches_tracked)

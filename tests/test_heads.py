"""task heads and finetune losses against vectors captured from the reference's own
models/decoders.py (imports and runs as-is; oracle/capture_golden.py::g9_heads): the oracle's restatement, the product's head
classes (plain torch modules: they run on the CPU here and on the GPU in the -m gpu leg), and -- on the GPU -- the whole model
in mode="finetune" / "inference" against the oracle."""
import types

import pytest
import torch

from conftest import T, assert_close, check_decision_margins, decisions_from_trace, load_golden, weights
from oracle import dgdm_oracle as O

TOL = 2e-5


def _grad(loss, ts):
    return torch.autograd.grad(loss, ts, allow_unused=True)


def test_oracle_heads_match_reference_golden():
    g = load_golden("g9_heads")
    x, y, tgt = T(g["x"]), T(g["y"]), T(g["targets"])
    P = {"classification_head." + k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in weights(g, "cls.").items()}
    xg = x.clone().requires_grad_(True)
    logits = O.classification_head(P, xg)
    assert_close(logits, g["cls_logits"], TOL, "logits")
    assert_close(O.classification_loss(logits, y), g["cls_loss"], TOL, "ce")
    assert_close(O.classification_loss(logits, y, class_weights=T(g["class_weights"])), g["cls_loss_weighted"], TOL, "weighted ce")
    assert_close(O.classification_loss(logits, y, label_smoothing=0.1), g["cls_loss_smooth"], TOL, "smoothed ce")
    gx, gw = _grad(O.classification_loss(logits, y), [xg, P["classification_head.classifier.0.weight"]])
    assert_close(gx, g["cls_gx"], TOL, "gx"); assert_close(gw, g["cls_gw0"], TOL, "gw0")
    xt = x.clone().requires_grad_(True)
    lt = O.classification_head(P, xt, p_drop=0.0, training=True)          # batch statistics
    assert_close(lt, g["cls_train_logits"], TOL, "train logits")
    gxt, gwt = _grad(O.classification_loss(lt, y), [xt, P["classification_head.classifier.4.weight"]])
    assert_close(gxt, g["cls_train_gx"], TOL, "train gx"); assert_close(gwt, g["cls_train_gw4"], TOL, "train gw4")

    R = {"regression_head." + k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in weights(g, "reg.").items()}
    xr = x.clone().requires_grad_(True)
    pr = O.regression_head(R, xr)
    assert_close(pr, g["reg_out"], TOL, "regression")
    for k in ("mse", "mae", "huber"):
        assert_close(O.regression_loss(pr, tgt, k), g[f"reg_loss_{k}"], TOL, k)
    gxr, gwr = _grad(O.regression_loss(pr, tgt), [xr, R["regression_head.mean_head.weight"]])
    assert_close(gxr, g["reg_gx"], TOL, "reg gx"); assert_close(gwr, g["reg_gwm"], TOL, "reg gwm")


def _product_heads(dev):
    from dgdm_histopath_lab_amd.models.decoders import ClassificationHead, RegressionHead
    g = load_golden("g9_heads")
    x, y, tgt = T(g["x"]).to(dev), T(g["y"]).to(dev), T(g["targets"]).to(dev)
    cls = ClassificationHead(128, 5, hidden_dims=[64])
    cls.load_state_dict(weights(g, "cls."), strict=True)
    cls = cls.to(dev).eval()
    logits = cls(x)
    assert_close(logits, g["cls_logits"], TOL, "logits"); assert_close(cls.compute_loss(logits, y), g["cls_loss"], TOL, "ce")
    assert torch.equal(cls.predict(x).cpu(), T(g["cls_pred"])); assert_close(cls.predict(x, return_probs=True), g["cls_probs"], TOL, "probs")
    cw = ClassificationHead(128, 5, hidden_dims=[64], class_weights=T(g["class_weights"]))
    cw.load_state_dict(weights(g, "cls."), strict=False)
    cw = cw.to(dev).eval()
    assert_close(cw.compute_loss(cw(x), y), g["cls_loss_weighted"], TOL, "weighted ce")
    cs = ClassificationHead(128, 5, hidden_dims=[64], label_smoothing=0.1)
    cs.load_state_dict(weights(g, "cls."), strict=True)
    cs = cs.to(dev).eval()
    assert_close(cs.compute_loss(cs(x), y), g["cls_loss_smooth"], TOL, "smoothed ce")
    ct = ClassificationHead(128, 5, hidden_dims=[64], dropout=0.0)
    ct.load_state_dict(weights(g, "cls."), strict=True)
    ct = ct.to(dev).train()
    xt = x.clone().requires_grad_(True)
    lt = ct(xt)
    assert_close(lt, g["cls_train_logits"], TOL, "train logits")
    gxt, gwt = _grad(ct.compute_loss(lt, y), [xt, ct.classifier[4].weight])
    assert_close(gxt, g["cls_train_gx"], TOL, "train gx"); assert_close(gwt, g["cls_train_gw4"], TOL, "train gw4")
    assert_close(ct.classifier[1].running_mean, g["cls_train_running_mean"], TOL, "running mean")
    assert_close(ct.classifier[1].running_var, g["cls_train_running_var"], TOL, "running var")
    reg = RegressionHead(128, 3, hidden_dims=[64])
    reg.load_state_dict(weights(g, "reg."), strict=True)
    reg = reg.to(dev).eval()
    pr = reg(x)
    assert_close(pr, g["reg_out"], TOL, "regression")
    for k in ("mse", "mae", "huber"):
        assert_close(reg.compute_loss(pr, tgt, k), g[f"reg_loss_{k}"], TOL, k)
    ru = RegressionHead(128, 3, hidden_dims=[64], output_activation="softplus", predict_uncertainty=True)
    ru.load_state_dict(weights(g, "regu."), strict=True)
    ru = ru.to(dev).eval()
    pu = ru(x)
    for k in ("mean", "var", "log_var"):
        assert_close(pu[k], g[f"regu_{k}"], TOL, k)
    assert_close(ru.compute_loss(pu, tgt, "gaussian_nll"), g["regu_nll"], TOL, "nll"); assert_close(ru.compute_loss(pu, tgt, "mse"), g["regu_mse"], TOL, "mse(mean)")
    assert_close(ru.predict(x), g["regu_mean"], TOL, "predict")


def test_product_heads_match_reference_golden_cpu():
    _product_heads("cpu")


@pytest.mark.gpu
def test_product_heads_match_reference_golden_gpu():
    _product_heads("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("mode,training,graphs", [("finetune", True, 12), ("finetune", True, 6), ("inference", False, 12)])   # 6: x 300 nodes, the round-2 failure
def test_model_finetune_and_inference_outputs_match_oracle(mode, training, graphs):
    """DGDMModel(num_classes=5, regression_targets=3) on 12 x 200-node (and 6 x 300-node) graphs, Base widths, U-Net on.  (The heads' BatchNorm
    runs on BATCH statistics in finetune mode, and the gradient through a BatchNorm over very few rows is ill-conditioned: over 2
    rows both normalise to +-1 and it is rounding noise.  Round 2 ran this test on 12 graphs because at 6 the attention of that
    round -- probabilities and dS as single fp16 -- put 0.5-1.1x the tolerance on the worst gradient (the test had
    been re-sized until it passed); with P and dS carried hi+lo the 6-graph batch it was first written for is back, at the same
    1e-3.)  classification_logits /
    probs, regression_outputs, graph_embedding against the float64 oracle; in finetune mode (training-mode BatchNorm = batch
    statistics, every dropout probability set to 0) also the supervised loss of trainer.py:130-175 and EVERY live gradient."""
    from dgdm_histopath_lab_amd import DGDMModel
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    DEV = "cuda:0"
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8, num_classes=5,
                regression_targets=3, dropout=0.0)
    cfg = O.OracleConfig(**cfgd)
    P = O.init_params(cfg, seed=13, perturb=0.05)
    bufs = O.batchnorm_buffers(cfg, seed=13, trained=True)
    batch = synthetic_batch(40, graphs, 200, 800) if graphs == 12 else synthetic_batch(40, graphs, 300, 1200)
    y, rt = torch.tensor([1, 4, 0, 2, 2, 3, 0, 1, 3, 4, 2, 0])[:graphs], torch.randn(12, 3, generator=torch.Generator().manual_seed(3))[:graphs]
    b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(), pos=batch.pos.double(),
                                batch=batch.batch)
    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    P64.update({k: v.double() if v.is_floating_point() else v for k, v in bufs.items()})
    tr64 = {}
    # every stochastic layer off: the heads' and the U-Net's dropout follow cfg.dropout = 0; pool / denoiser dropouts are not on this path's loss
    ref = O.forward(P64, cfg, b64, mode, training=False, trace=tr64) if not training else _oracle_train_forward(P64, cfg, b64, mode, tr64)
    m = DGDMModel(**cfgd)
    m.load_state_dict({**P, **bufs}, strict=True)
    m = m.to(DEV)
    m.train(training)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    tr = {}
    dec = decisions_from_trace(tr64)
    out = m(batch.to(DEV), mode=mode, trace=tr, decisions=dec)
    check_decision_margins(tr, dec)
    for k in ("graph_embedding", "classification_logits", "classification_probs", "regression_outputs"):
        assert_close(out[k], ref[k], 1e-3, k)
    assert "diffusion_loss" not in out
    if not training:
        return
    loss = m.classification_head.compute_loss(out["classification_logits"], y.to(DEV)) + \
        m.regression_head.compute_loss(out["regression_outputs"], rt.to(DEV))
    loss_ref = O.classification_loss(ref["classification_logits"], y) + O.regression_loss(ref["regression_outputs"], rt.double())
    assert_close(loss, loss_ref, 1e-3, "supervised loss")
    loss.backward(); loss_ref.backward()
    named = dict(m.named_parameters())
    live = 0
    for k, v in P64.items():
        if not v.requires_grad or v.grad is None or v.grad.abs().max() < 1e-12:
            continue
        assert named[k].grad is not None, k
        if v.numel() <= 4 and k.endswith(".bias"):
            # a scalar bias gradient (score MLP of a pooling level) is ONE signed sum over the nodes: when it nearly cancels its own
            # magnitude is no yardstick; the weight of the same layer sums the same per-node terms and provides the scale
            scale = float(P64[k[:-4] + "weight"].grad.abs().max())
            err = float((named[k].grad.double().cpu() - v.grad).abs().max())
            assert err <= 1e-3 * max(scale, float(v.grad.abs().max())), (k, err, scale)
            live += 1
            continue
        assert_close(named[k].grad, v.grad, 1e-3, "grad " + k); live += 1
    assert live > 100


def _oracle_train_forward(P64, cfg, b64, mode, trace):
    """Oracle forward with BatchNorm in training mode (batch statistics) but no dropout draw: cfg.dropout is 0 and the two
    hard-coded 0.1 sites (denoiser, pooling attention) are set to 0 on the HIP side; the oracle's `training` flag would draw
    them, so the trunk runs with training=False and only the heads see training=True."""
    out = O.forward(P64, cfg, b64, mode, training=False, trace=trace)
    g = out["graph_embedding"]
    out["classification_logits"] = O.classification_head(P64, g, cfg.activation, 0.0, True)
    out["classification_probs"] = torch.softmax(out["classification_logits"], dim=-1)
    out["regression_outputs"] = O.regression_head(P64, g, cfg.activation, 0.0, True)
    return out

"""Host logic of the trainer harness (reference: training/trainer.py) -- CPU, with a stand-in model --
and a short real run on the GPU."""
import math

import pytest
import torch
import torch.nn as nn


class _Head(nn.Module):
    def __init__(self, c, k):
        super().__init__()
        self.fc = nn.Linear(c, k)

    def forward(self, x):
        return self.fc(x)

    def compute_loss(self, logits, y):
        return nn.functional.cross_entropy(logits, y)


class _FakeModel(nn.Module):
    """Duck-types the DGDMModel methods the trainer calls."""

    def __init__(self):
        super().__init__()
        self.enc = nn.Linear(6, 4)
        self.classification_head = _Head(4, 3)
        self.regression_head = None
        self.calls = []

    def _embed(self, batch):
        return self.enc(batch.x)

    def pretrain_step(self, batch, mask_ratio=0.15):
        self.calls.append(("pretrain", mask_ratio))
        loss = self._embed(batch).pow(2).mean()
        return {"diffusion_loss": loss, "total_pretrain_loss": loss}

    def forward(self, batch, mode="inference", return_attention=False, return_embeddings=False):
        self.calls.append((mode, return_attention, return_embeddings))
        h = self._embed(batch)
        g = torch.stack([h[batch.batch == i].mean(0) for i in range(int(batch.batch.max()) + 1)])
        logits = self.classification_head(g)
        return {"graph_embedding": g, "node_embeddings": h, "classification_logits": logits, "classification_probs": logits.softmax(-1)}

    def _compute_diffusion_loss(self, emb, batch):
        self.calls.append(("difffallback",))
        return {"diffusion_loss": emb.pow(2).mean()}


class _Batch:
    def __init__(self, with_y=True):
        g = torch.Generator().manual_seed(0)
        self.x = torch.randn(10, 6, generator=g)
        self.batch = torch.tensor([0] * 5 + [1] * 5)
        if with_y:
            self.y = torch.tensor([0, 2])

    def to(self, dev):
        return self


def test_lr_trajectory_matches_closed_form_with_finetune_drop():
    from dgdm_histopath_lab_amd.training import DGDMTrainer, closed_form_lr
    tr = DGDMTrainer(_FakeModel(), learning_rate=1e-4, pretrain_epochs=2, finetune_epochs=2)
    steps_per_epoch, epochs = 5, 4
    total = steps_per_epoch * epochs
    tr.configure_optimizers(total)
    lrs = []
    for epoch in range(epochs):
        tr.current_epoch = epoch
        tr.on_train_epoch_start()
        for _ in range(steps_per_epoch):
            lrs.append(tr.optimizers().param_groups[0]["lr"])
            tr.optimizers().step(); tr._scheduler.step()
    switch = 2 * steps_per_epoch
    for t, lr in enumerate(lrs):
        assert math.isclose(lr, closed_form_lr(t, 1e-4, total, switch), rel_tol=1e-9), (t, lr)
    assert math.isclose(lrs[switch], 0.1 * closed_form_lr(switch, 1e-4, total), rel_tol=1e-9)   # x0.1 at finetune entry
    assert lrs[0] == 1e-4 and tr.current_phase == "finetune"
    # eta_min = 0.01 * lr is the floor of the pretrain cosine
    assert math.isclose(closed_form_lr(total, 1e-4, total), 1e-6, rel_tol=1e-9)


def test_phase_switch_log_names_and_fallback():
    from dgdm_histopath_lab_amd.training import DGDMTrainer
    m = _FakeModel()
    tr = DGDMTrainer(m, pretrain_epochs=1, finetune_epochs=1, masking_ratio=0.3)
    losses = tr.fit([_Batch(), _Batch()], max_epochs=2)
    assert len(losses) == 4 and all(math.isfinite(l) for l in losses)
    assert m.calls[0] == ("pretrain", 0.3) and m.calls[2] == ("finetune", True, True)     # forward(..., True, True): trainer.py:89
    assert tr.logged["train/phase"] == 1.0 and "train/classification_loss" in tr.logged and "train/accuracy" in tr.logged
    assert tr.global_step == 4 and tr.current_epoch == 2
    # no labels -> diffusion fallback (trainer.py:164-170)
    m.calls.clear()
    tr.current_epoch = 1
    tr.training_step(_Batch(with_y=False))
    assert ("difffallback",) in m.calls
    val = tr.validation_step(_Batch())
    assert set(val) == {"val_loss", "val_accuracy"}
    pred = tr.predict_step(_Batch())
    assert set(pred) == {"graph_embeddings", "node_embeddings", "classification_probs", "predicted_classes"}


def test_checkpoint_layouts(tmp_path):
    from dgdm_histopath_lab_amd.training import DGDMTrainer
    a, b, c = DGDMTrainer(_FakeModel()), DGDMTrainer(_FakeModel()), DGDMTrainer(_FakeModel())
    with torch.no_grad():
        for p in a.model.parameters():
            p.add_(1.0)
    a.current_epoch, a.global_step = 3, 17
    f = tmp_path / "m.pt"
    a.save_model(str(f))
    ck = torch.load(str(f), weights_only=False)
    assert set(ck) == {"model_state_dict", "hyperparameters", "epoch", "global_step"}          # trainer.py:351-356
    info = b.load_checkpoint(str(f), strict=True)
    assert b.current_epoch == 3 and b.global_step == 17 and info["hyperparameters"]["masking_ratio"] == 0.15
    for p, q in zip(a.model.parameters(), b.model.parameters()):
        assert torch.equal(p, q)
    # Lightning layout of the reference trainer: keys prefixed with the attribute name "model."
    g = tmp_path / "lightning.ckpt"
    torch.save({"state_dict": {f"model.{k}": v for k, v in a.model.state_dict().items()}, "hyper_parameters": {"learning_rate": 3e-4},
                "epoch": 1, "global_step": 9}, str(g))
    info = c.load_checkpoint(str(g), strict=True)
    assert info["hyperparameters"]["learning_rate"] == 3e-4
    for p, q in zip(a.model.parameters(), c.model.parameters()):
        assert torch.equal(p, q)
    with pytest.raises(ValueError):
        torch.save({"weights": {}}, str(g)); c.load_checkpoint(str(g))


def test_losses_restate_reference_formulas():
    from dgdm_histopath_lab_amd.training import ContrastiveLoss, DiffusionLoss
    g = torch.Generator().manual_seed(1)
    p, t = torch.randn(7, 5, generator=g), torch.randn(7, 5, generator=g)
    mask = torch.tensor([1, 0, 1, 1, 0, 1, 1.0])
    assert torch.allclose(DiffusionLoss()(p, t, mask), ((p - t) ** 2 * mask[:, None]).mean())
    assert torch.allclose(DiffusionLoss("l1", "sum")(p, t), (p - t).abs().sum())
    z = torch.randn(6, 4, generator=g); b = torch.tensor([0, 0, 0, 1, 1, 2])     # node 5 has no positive: excluded
    zn = z / z.norm(dim=1, keepdim=True)
    sim = (zn @ zn.t() / 0.1).exp()
    want = []
    for i in range(5):
        pos = sum(sim[i, j] for j in range(6) if j != i and b[j] == b[i])
        want.append(-torch.log(pos / sim[i].sum()))
    assert torch.allclose(ContrastiveLoss(0.1)(z, b), torch.stack(want).mean(), rtol=1e-5)


@pytest.mark.gpu
def test_trainer_runs_real_model_and_predicts(tmp_path):
    from dgdm_histopath_lab_amd import DGDMModel
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    from dgdm_histopath_lab_amd.training import DGDMTrainer, predict_graph
    torch.manual_seed(0)
    model = DGDMModel(node_features=64, hidden_dims=[64, 32, 32], num_diffusion_steps=10, attention_heads=4, num_classes=3,
                      regression_targets=2).to("cuda:0")
    batches = []
    for i in range(2):
        b = synthetic_batch(10 * i, 2, 300, 1200, 64).to("cuda:0")
        b.y = torch.tensor([0, 2], device="cuda:0")
        b.regression_targets = torch.randn(2, 2, device="cuda:0")
        batches.append(b)
    tr = DGDMTrainer(model, learning_rate=1e-3, pretrain_epochs=2, finetune_epochs=2)
    losses = tr.fit(batches)
    assert len(losses) == 8 and all(math.isfinite(l) for l in losses)
    assert tr.current_phase == "finetune" and {"train/classification_loss", "train/regression_loss", "train/diffusion_loss"} <= set(tr.logged)
    f = tmp_path / "ck.pt"
    tr.save_model(str(f))
    model2 = DGDMModel(node_features=64, hidden_dims=[64, 32, 32], num_diffusion_steps=10, attention_heads=4, num_classes=3,
                       regression_targets=2).to("cuda:0")
    tr2 = DGDMTrainer(model2)
    info = tr2.load_checkpoint(str(f), strict=True)
    assert not info["missing_keys"] and not info["unexpected_keys"]
    one = synthetic_batch(5, 1, 200, 800, 64)
    pa, pb = predict_graph(model, one, return_embeddings=True), predict_graph(model2, one, return_embeddings=True)
    assert {"classification_probs", "predicted_class", "confidence", "class_0_prob", "class_2_prob", "regression_outputs",
            "regression_target_1", "graph_embedding", "node_embeddings", "num_nodes", "num_edges"} <= set(pa)
    assert pa["num_nodes"] == 200 and pa["num_edges"] == 400
    assert (pa["graph_embedding"] == pb["graph_embedding"]).all()       # same weights, eval mode: bitwise reproducible



@pytest.mark.gpu
@pytest.mark.parametrize("hierarchical,width", [(False, "toy"), (True, "toy"), (False, "base"), (True, "base")])
def test_loss_trajectory_matches_oracle_with_plain_adamw(hierarchical, width):
    """("base": the same trajectory at DGDM-Base widths -- 768 features, hidden 512/256/128, 8 heads -- VERDICT r2 weak 4.)
    SURVEY.md 8(f) N1, second pin: the first k pretraining steps of DGDMTrainer on the HIP path against the CPU oracle driven by
    plain torch AdamW + CosineAnnealingLR with the reference's recipe (training/trainer.py:217-254: lr 1e-4 -> here 1e-3 to make
    the weights move, weight_decay 1e-5, T_max = total steps, eta_min = 0.01 lr), the same injected draws on both sides (entity
    mask, mask token, timesteps, noise, target; every dropout probability 0).  With the graph U-Net on, the oracle's ReLU / top-k
    decisions of each step are handed to the kernels (conftest.check_decision_margins holds the differing ones to the margin)."""
    import types
    from conftest import assert_close, check_decision_margins, decisions_from_trace
    from oracle import dgdm_oracle as O
    from dgdm_histopath_lab_amd import DGDMModel
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    from dgdm_histopath_lab_amd.training import DGDMTrainer, closed_form_lr
    DEV, K, LR = "cuda:0", 6, 1e-3
    feats, dims, heads = (64, [64, 32, 32], 2) if width == "toy" else (768, [512, 256, 128], 8)
    cfgd = dict(node_features=feats, hidden_dims=dims, num_diffusion_steps=10, attention_heads=heads, dropout=0.0,
                use_hierarchical=hierarchical)
    cfg = O.OracleConfig(**cfgd)
    P0 = O.init_params(cfg, seed=21, perturb=0.05)
    batches = [synthetic_batch(50 + 7 * i, 2, 260 + 40 * i, 1000 + 160 * i, feats) for i in range(2)]     # two layouts, cycled
    gen = torch.Generator().manual_seed(5)
    draws = []
    for i in range(K):
        n = batches[i % 2].x.size(0)
        draws.append(dict(timesteps=torch.randint(0, 10, (2,), generator=gen), noise=torch.randn(n, dims[-1], generator=gen),
                          noise_target=torch.randn(n, dims[-1], generator=gen), mask_indices=torch.randperm(n, generator=gen)[: int(0.15 * n)],
                          mask_token=torch.randn(feats, generator=gen)))
    # --- checker: float64 oracle + plain AdamW / cosine schedule on the CPU
    P = {k: v.double().requires_grad_(True) for k, v in P0.items()}
    opt = torch.optim.AdamW(list(P.values()), lr=LR, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=K, eta_min=LR * 0.01)
    ref_losses, ref_traces = [], []

    def oracle_step(i):
        b = batches[i % 2]
        b64 = types.SimpleNamespace(x=b.x.double(), edge_index=b.edge_index, edge_attr=b.edge_attr.double(), pos=b.pos.double(), batch=b.batch)
        d = draws[i]
        tro = {}
        opt.zero_grad(set_to_none=True)
        out = O.pretrain_step(P, cfg, b64, mask_indices=d["mask_indices"], mask_token=d["mask_token"].double(), timesteps=d["timesteps"],
                              noise=d["noise"].double(), noise_target=d["noise_target"].double(), trace=tro)
        out["total_pretrain_loss"].backward()
        for p in P.values():            # dead parameters (D9) carry no gradient: AdamW skips them, as on the HIP side
            pass
        opt.step(); sched.step()
        ref_losses.append(float(out["total_pretrain_loss"].detach()))
        return tro

    # --- HIP trainer, fed the same draws (and, with the U-Net, the checker's decisions of the same step)
    m = DGDMModel(**cfgd)
    m.load_state_dict(P0, strict=True)
    m = m.to(DEV)
    tr = DGDMTrainer(m, learning_rate=LR, weight_decay=1e-5, pretrain_epochs=K, finetune_epochs=0, masking_ratio=0.15)
    own_traces = []

    def step_kwargs(i):
        tro = oracle_step(i)
        kw = {k: v.to(DEV) for k, v in draws[i].items()}
        if hierarchical:
            own_traces.append(({}, decisions_from_trace(tro)))
            kw["trace"], kw["decisions"] = own_traces[-1]
        return kw
    tr.step_kwargs = step_kwargs
    zero_p = [mod for mod in m.modules() if isinstance(mod, torch.nn.Dropout)]

    def on_step(trainer, loss):
        pass
    # fit() switches the model to train(): dropout sites are disabled by probability, not by mode
    for mod in zero_p:
        mod.p = 0.0
    losses = tr.fit([batches[0], batches[1]] * (K // 2), max_epochs=1, steps_per_epoch=K, on_step=on_step)
    assert len(losses) == K == len(ref_losses)
    for i, (a, b) in enumerate(zip(losses, ref_losses)):
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (i, a, b)
    for own, dec in own_traces:
        # after a few optimizer steps the two sets of weights agree to the 1e-3 contract (asserted below), and so do the
        # activations: a decision may differ wherever the pre-activation lies within THAT distance of zero
        check_decision_margins(own, dec, margin=1e-3)
    # the parameters themselves after K optimizer steps, and the learning rate the schedule arrived at
    named = dict(m.named_parameters())
    moved = 0
    gmax = max(float(v.grad.abs().max()) for v in P.values() if v.grad is not None)
    for k, v in P.items():
        if v.grad is None:
            assert torch.equal(named[k].detach().cpu(), P0[k]), k      # dead parameters never move (no weight decay without a gradient)
            continue
        if float(v.grad.abs().max()) < 1e-9 * gmax:
            # zero by construction (attention k_proj.bias: softmax is shift invariant): Adam divides rounding noise by its own
            # magnitude, so the fp32 path random-walks by ~lr per step where exact arithmetic stays put; the loss cannot see it
            continue
        # Adam normalises every element's gradient by its own running magnitude: an element whose gradient is within rounding of
        # zero moves by up to lr per step in a direction rounding decides, so element-wise the two runs may part by 2 * sum(lr);
        # in rel-L2 (what the 1e-3 contract measures) the parameters must agree
        a, b = named[k].detach().cpu().double(), v.detach()
        assert float((a - b).norm() / b.norm()) <= 1e-3, ("param " + k, float((a - b).norm() / b.norm()))
        assert float((a - b).abs().max()) <= 2 * sum(closed_form_lr(t, LR, K) for t in range(K)), "param " + k
        moved += 1
    assert moved > 60
    lr = float(tr.optimizers().param_groups[0]["lr"])
    assert lr == pytest.approx(closed_form_lr(K, LR, K), rel=1e-5) and lr == pytest.approx(opt.param_groups[0]["lr"], rel=1e-6)


@pytest.mark.gpu
def test_predict_graph_values_and_reference_keyed_checkpoint(tmp_path):
    """SURVEY.md 8(f) N2 at value level: every entry of predict_graph's dictionary (evaluation/predictor.py:188-257) equals what
    `forward(mode="inference")` returned for the same graph; a checkpoint that holds exactly the REFERENCE's keys (its
    save_model layout, trainer.py:348-358, no `graph_encoder.dim_proj.*`) loads with only the repair-R2 keys reported missing and
    reproduces the oracle's outputs."""
    import types
    import numpy as np
    from conftest import assert_close
    from oracle import dgdm_oracle as O
    from dgdm_histopath_lab_amd import DGDMModel
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    from dgdm_histopath_lab_amd.training import DGDMTrainer, predict_graph
    DEV = "cuda:0"
    cfgd = dict(node_features=64, hidden_dims=[64, 32, 32], num_diffusion_steps=10, attention_heads=2, num_classes=4, regression_targets=2,
                use_hierarchical=False)
    cfg = O.OracleConfig(**cfgd)
    P = {**O.init_params(cfg, seed=8, perturb=0.05), **O.batchnorm_buffers(cfg, seed=8, trained=True)}
    ref_keys = {k: v for k, v in P.items() if ".dim_proj." not in k}                  # what a reference checkpoint can contain
    f = tmp_path / "reference_layout.pt"
    torch.save({"model_state_dict": ref_keys, "hyperparameters": {"learning_rate": 1e-4}, "epoch": 4, "global_step": 123}, str(f))
    m = DGDMModel(**cfgd).to(DEV)
    tr = DGDMTrainer(m)
    info = tr.load_checkpoint(str(f))
    assert sorted(info["missing_keys"]) == sorted(k for k in P if ".dim_proj." in k) and not info["unexpected_keys"]
    assert tr.current_epoch == 4 and tr.global_step == 123
    with torch.no_grad():                                                              # the R2 projections are the build's own: give the oracle the same
        Pm = {**P, **{k: v.detach().cpu() for k, v in m.state_dict().items() if ".dim_proj." in k}}
    one = synthetic_batch(5, 1, 333, 1300, 64)
    pred = predict_graph(m, one, return_attention=True, return_embeddings=True)
    m.eval()
    with torch.no_grad():
        out = m(one.to(DEV), mode="inference", return_attention=True, return_embeddings=True)
    probs = out["classification_probs"].cpu().numpy()
    assert np.array_equal(pred["classification_probs"], probs) and pred["predicted_class"] == int(probs.argmax())
    assert pred["confidence"] == float(probs.max()) and all(pred[f"class_{i}_prob"] == float(probs[0, i]) for i in range(4))
    reg = out["regression_outputs"].cpu().numpy()
    assert np.array_equal(pred["regression_outputs"], reg) and all(pred[f"regression_target_{i}"] == float(reg[0, i]) for i in range(2))
    assert np.array_equal(pred["graph_embedding"], out["graph_embedding"].cpu().numpy())
    assert np.array_equal(pred["node_embeddings"], out["node_embeddings"].cpu().numpy())
    assert np.array_equal(pred["attention_weights"][0], out["attention_weights"][0].cpu().numpy())
    assert pred["num_nodes"] == 333 and pred["num_edges"] == 650
    b64 = types.SimpleNamespace(x=one.x.double(), edge_index=one.edge_index, edge_attr=one.edge_attr.double(), pos=one.pos.double(), batch=one.batch)
    with torch.no_grad():
        ref = O.forward({k: (v.double() if v.is_floating_point() else v) for k, v in Pm.items()}, cfg, b64, "inference", return_embeddings=True)
    for k in ("classification_probs", "regression_outputs", "graph_embedding", "node_embeddings"):
        assert_close(out[k], ref[k], 1e-3, k)


@pytest.mark.gpu
def test_config4_mixed_size_pretrain_stream_through_the_trainer():
    """BASELINE configs[4] as a workload: DGDM-Base, graphs of 1k..10k nodes (E = 5 N), masking_ratio 0.15, cosine schedule,
    sharded by the cost-aware loader (here: the bin of rank 0 of 2 -- the other rank's work is identical in kind), driven by
    DGDMTrainer.fit for two epochs with recorded steps: the layouts recur, so the second epoch replays graphs.  Checked: losses
    finite and falling on average under lr 1e-3, the LR trajectory, the rank balance of every step, and -- parity -- one full
    pretrain_step of the SMALLEST per-rank batch against the float64 oracle (loss, embeddings and every live gradient, kink
    decisions injected)."""
    import types
    from conftest import assert_close, check_decision_margins, decisions_from_trace
    from oracle import dgdm_oracle as O
    from dgdm_histopath_lab_amd import DGDMModel
    from dgdm_histopath_lab_amd.parallel import BalancedSlideLoader
    from dgdm_histopath_lab_amd.synthetic import synthetic_graph
    from dgdm_histopath_lab_amd.training import DGDMTrainer, GraphedStepCache, closed_form_lr
    DEV = "cuda:0"
    # the run below trains with dropout: its trajectory -- and with it the weights the parity leg is taken at -- depends on the
    # device-side dropout seed epoch, which earlier tests of the session advance.  Start from epoch 0 so that this test checks the
    # same weights whatever ran before it (at other epochs the scalar gradient of pools.0.score_net.2.bias, a sum over all nodes
    # that cancels to ~1e-8 of its terms, has been seen 2e-3 off while every other gradient held 3e-5).
    from dgdm_histopath_lab_amd import _lib, ops
    _lib.check(_lib.load().dgdm_seed_epoch_set(0, _lib.stream_ptr(torch.device(DEV))), "dgdm_seed_epoch_set")
    ops._seed_counter = 0          # ... and the host-side counter that numbers the dropout sites
    cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8, diffusion_schedule="cosine")
    gen = torch.Generator().manual_seed(4)
    ns = torch.randint(1000, 10001, (16,), generator=gen).tolist()
    ns[5] = 1000; ns[6] = 1200                      # one small step so that the oracle leg stays cheap
    slides = [synthetic_graph(200 + i, n, 5 * n, 768) for i, n in enumerate(ns)]
    loader = BalancedSlideLoader(slides, 4, 2, 0, device=DEV)          # global batch 4 over 2 ranks: 4 steps per epoch
    assert len(loader) == 4 and loader.max_over_mean_load() < 1.25
    torch.manual_seed(0)
    model = DGDMModel(**cfgd).to(DEV)
    tr = DGDMTrainer(model, learning_rate=1e-3, pretrain_epochs=3, finetune_epochs=0, masking_ratio=0.15, scheduler_type="cosine")
    losses = tr.fit(loader, max_epochs=3, graphed=True)
    assert len(losses) == 12 and all(math.isfinite(l) for l in losses)
    assert sum(losses[8:]) < sum(losses[:4])                            # the objective moves (target is fresh noise: it falls towards 1)
    assert isinstance(tr._graphed, GraphedStepCache) and len(tr._graphed.steps) == 4 and tr._graphed.replays >= 3   # epochs 2-3 record, then replay
    assert float(tr.optimizers().param_groups[0]["lr"]) == pytest.approx(closed_form_lr(12, 1e-3, 12), rel=1e-4)
    # --- parity of one step on the smallest per-rank batch, with the weights the run arrived at
    batches = list(BalancedSlideLoader(slides, 4, 2, 0))
    small = min(batches, key=lambda b: b.x.size(0))
    n = small.x.size(0)
    rng = dict(timesteps=torch.randint(0, 10, (small.num_graphs,), generator=gen), noise=torch.randn(n, 128, generator=gen),
               noise_target=torch.randn(n, 128, generator=gen))
    mask_idx, mask_tok = torch.randperm(n, generator=gen)[: int(0.15 * n)], torch.randn(768, generator=gen)
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = O.OracleConfig(**cfgd)
    b64 = types.SimpleNamespace(x=small.x.double(), edge_index=small.edge_index, edge_attr=small.edge_attr.double(), pos=small.pos.double(),
                                batch=small.batch)
    tr64 = {}
    torch.set_num_threads(32)
    ref, gref = O.loss_and_grads({k: v.double() for k, v in P.items()}, cfg, b64, mask_indices=mask_idx, mask_token=mask_tok.double(), trace=tr64,
                                 **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
    model.eval()
    model.zero_grad(set_to_none=True)
    own, dec = {}, decisions_from_trace(tr64)
    out = model.pretrain_step(small.to(DEV), mask_indices=mask_idx.to(DEV), mask_token=mask_tok.to(DEV), trace=own, decisions=dec,
                              **{k: v.to(DEV) for k, v in rng.items()})
    out["total_pretrain_loss"].backward()
    check_decision_margins(own, dec)
    for k in ("diffusion_loss", "graph_embedding", "noisy_embeddings"):
        assert_close(out[k], ref[k], 1e-3, k)
    named, live = dict(model.named_parameters()), 0
    for k, gr in gref.items():
        if gr.abs().max() < 1e-12:
            continue
        assert_close(named[k].grad, gr, 1e-3, "grad " + k); live += 1
    assert live > 100


@pytest.mark.gpu
def test_dgdm_adamw_matches_torch_adamw_and_exchanges_state_dicts():
    """optim.DGDMAdamW (one HIP launch per 96 tensors) against torch.optim.AdamW in float64 on the CPU, the reference's optimizer
    (training/trainer.py:217-226): 150 tensors of odd sizes (two launches; tails that are not multiples of 4; a 4-byte-aligned
    view), one parameter that never gets a gradient (skipped, as torch skips it), one that gets its first gradient at step 3 (its
    own step count, as in torch).  Then: the state dict loads into torch's AdamW and back, and both continue identically; a
    device-tensor learning rate; the step recorded into a HIP graph."""
    from dgdm_histopath_lab_amd.optim import DGDMAdamW
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(0)
    sizes = [1, 3, 4, 5, 127, 128, 4096, 4097, 8191, 70001] + [int(torch.randint(1, 3000, (1,), generator=g)) for _ in range(140)]
    base = [torch.randn(n, generator=g) for n in sizes]
    flat = torch.zeros(1001, device=DEV)
    ref = [torch.nn.Parameter(b.double().clone()) for b in base]
    own = [torch.nn.Parameter(b.to(DEV).clone()) for b in base]
    own[7] = torch.nn.Parameter(flat[1:])       # a parameter whose storage starts 4 bytes off a 16-byte boundary: scalar path
    ref[7] = torch.nn.Parameter(torch.zeros(1000, dtype=torch.float64))
    kw = dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    o_ref, o_own = torch.optim.AdamW(ref, **kw), DGDMAdamW(own, **kw)

    def grads(step):
        for i, (a, b) in enumerate(zip(ref, own)):
            if i == 5 or (i == 9 and step < 3):          # 5: never a gradient; 9: joins at step 3
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, generator=g)
            a.grad, b.grad = gr.double(), gr.to(DEV)

    def same(tol=2e-6):
        for i, (a, b) in enumerate(zip(ref, own)):
            d = (b.detach().cpu().double() - a.detach()).abs().max().item()
            assert d <= tol * max(1.0, a.detach().abs().max().item()), (i, d)
    for s in range(6):
        grads(s)
        o_ref.step(); o_own.step()
    same()
    assert torch.equal(own[5].detach().cpu(), base[5])                                   # no gradient: untouched (no weight decay either)
    assert float(o_own.state[own[0]]["step"]) == 6.0 and float(o_own.state[own[9]]["step"]) == 3.0 and own[5] not in o_own.state
    # state dict -> torch.optim.AdamW (fp32, GPU) and back
    twin = [torch.nn.Parameter(p.detach().clone()) for p in own]
    o_twin = torch.optim.AdamW(twin, **kw)
    import copy
    # deep copies: Optimizer.load_state_dict casts with .to(), which hands back the SAME tensors when dtype and device already match
    o_twin.load_state_dict(copy.deepcopy(o_own.state_dict()))
    o_back = DGDMAdamW(own, **kw)
    o_back.load_state_dict(copy.deepcopy(o_twin.state_dict()))
    for s in range(6, 9):
        grads(s)
        for a, b in zip(own, twin):
            b.grad = None if a.grad is None else a.grad.clone()
        o_ref.step(); o_back.step(); o_twin.step()
    same()
    for a, b in zip(own, twin):
        assert (a.detach() - b.detach()).abs().max().item() <= 2e-6 * max(1.0, b.detach().abs().max().item())
    assert float(o_back.state[own[9]]["step"]) == 6.0 and float(o_back.state[own[0]]["step"]) == 9.0
    # device-side learning rate + HIP graph replay: three replays == three steps of the CPU optimizer
    lr_dev = torch.tensor(3e-3, device=DEV)
    for grp in o_back.param_groups:
        grp["lr"] = lr_dev
    grads(9)
    static = [None if p.grad is None else p.grad.clone() for p in own]
    for p, sg in zip(own, static):
        p.grad = sg
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr):
            o_back.step()
    torch.cuda.current_stream().wait_stream(side)
    for s in range(3):
        for a, sg in zip(ref, static):
            a.grad = None if sg is None else sg.cpu().double()
        o_ref.step()
        gr.replay()
    torch.cuda.synchronize()
    same(4e-6)
    assert float(o_back.state[own[0]]["step"]) == 12.0


@pytest.mark.gpu
def test_dgdm_adamw_member_that_skips_steps_keeps_its_own_count():
    """parameters that got their first gradient in the same call share one device step counter.  A member WITHOUT a gradient
    in a later step (frozen after the pretrain -> finetune switch; a branch the batch did not take under zero_grad(set_to_none=True))
    must not have its count advanced -- torch.optim.AdamW does not advance it -- and must continue from its own count when it comes
    back: bias correction and the exported ``step`` then equal torch's.  Parameters 1 and 2 skip steps 2-3 and rejoin at step 4."""
    from dgdm_histopath_lab_amd.optim import DGDMAdamW
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(1)
    base = [torch.randn(n, generator=g) for n in (257, 1000, 33, 4096)]
    ref = [torch.nn.Parameter(b.double().clone()) for b in base]
    own = [torch.nn.Parameter(b.to(DEV).clone()) for b in base]
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    o_ref, o_own = torch.optim.AdamW(ref, **kw), DGDMAdamW(own, **kw)
    for step in range(7):
        for i, (a, b) in enumerate(zip(ref, own)):
            if i in (1, 2) and step in (2, 3):
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, generator=g)
            a.grad, b.grad = gr.double(), gr.to(DEV)
        o_ref.step(); o_own.step()
        for i, (a, b) in enumerate(zip(ref, own)):
            assert float(o_own.state[b]["step"]) == float(o_ref.state[a]["step"]), (step, i)
            d = (b.detach().cpu().double() - a.detach()).abs().max().item()
            assert d <= 2e-6 * max(1.0, a.detach().abs().max().item()), (step, i, d)
    assert float(o_own.state[own[1]]["step"]) == 5.0 and float(o_own.state[own[0]]["step"]) == 7.0
    sd = o_own.state_dict()["state"]
    assert [float(sd[i]["step"]) for i in range(4)] == [7.0, 5.0, 5.0, 7.0]


@pytest.mark.gpu
def test_dgdm_adamw_cohorts_merge_again_and_a_split_inside_a_capture_raises():
    """ADVICE r5.  (i) Cohorts only ever split: 24 parameters whose gradients come and go in a rotating pattern fragment into more
    than MERGE_ABOVE cohorts; once every parameter has a gradient again, cohorts with EQUAL counts merge (one launch per distinct
    count, not one per parameter) while every parameter's count and value keep following torch.optim.AdamW.  (ii) A live set that
    changes INSIDE a stream capture would record the clone of the departing members' counter: it raises instead."""
    from dgdm_histopath_lab_amd.optim import DGDMAdamW
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(2)
    base = [torch.randn(64 + 8 * i, generator=g) for i in range(24)]
    ref = [torch.nn.Parameter(b.double().clone()) for b in base]
    own = [torch.nn.Parameter(b.to(DEV).clone()) for b in base]
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    o_ref, o_own = torch.optim.AdamW(ref, **kw), DGDMAdamW(own, **kw)

    def run(step, skip):
        for i, (a, b) in enumerate(zip(ref, own)):
            if skip(i):
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, generator=g)
            a.grad, b.grad = gr.double(), gr.to(DEV)
        o_ref.step(); o_own.step()
        for i, (a, b) in enumerate(zip(ref, own)):
            if o_ref.state.get(a):
                assert float(o_own.state[b]["step"]) == float(o_ref.state[a]["step"]), (step, i)
                d = (b.detach().cpu().double() - a.detach()).abs().max().item()
                assert d <= 2e-6 * max(1.0, a.detach().abs().max().item()), (step, i, d)

    run(0, lambda i: False)
    peak = 0
    for step in range(1, 7):                   # rotating gaps: parameter i pauses in step 1 + i % 6 ...
        run(step, lambda i: i % 6 == step - 1)
        peak = max(peak, len(o_own._cohorts[0]))
    for step in range(7, 11):                  # ... and again in step 7 + (i // 6) % 4: every cohort of the first round splits in four
        run(step, lambda i: (i // 6) % 4 == step - 7)
        peak = max(peak, len(o_own._cohorts[0]))
    assert peak > DGDMAdamW.MERGE_ABOVE, peak
    for step in range(11, 11 + DGDMAdamW.MERGE_AFTER_QUIET + 2):      # everybody live: all 24 have paused exactly twice -> ONE count;
        run(step, lambda i: False)                                    # a quiet period after the last split, then one cohort, one launch
    assert len(o_own._cohorts[0]) == 1, len(o_own._cohorts[0])
    assert len({float(o_own.state[b]["step"]) for b in own}) == 1

    # (ii) the capture guard
    p2 = [torch.nn.Parameter(torch.randn(128, device=DEV)) for _ in range(2)]
    o2 = DGDMAdamW(p2, **kw)
    for p in p2:
        p.grad = torch.randn(128, device=DEV)
    o2.step()                                   # both in one cohort
    p2[1].grad = None
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with pytest.raises(RuntimeError, match="changed inside a stream capture"):
            with torch.cuda.graph(gr, stream=side):
                o2.step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

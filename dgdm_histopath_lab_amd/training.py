"""Training / inference harness around DGDMModel with the interface of the reference's
``DGDMTrainer`` (dgdm_histopath/training/trainer.py:21-358) and the output dictionary of
``DGDMPredictor.predict_graph`` (evaluation/predictor.py:188-257) -- without PyTorch Lightning:
one process per GPU, gradients exchanged by one RCCL all-reduce of the live-gradient buffer
(parallel.FlatGradAllReducer).

What is kept exactly (file:line of the reference):
  * phase switch by epoch: epochs < pretrain_epochs run ``model.pretrain_step`` (+ the contrastive
    term only when the outputs carry ``node_embeddings`` -- they never do, D9), later epochs the
    supervised step with the diffusion-loss fallback (trainer.py:91-175);
  * AdamW(lr, weight_decay) + CosineAnnealingLR(T_max=total_steps, eta_min=0.01*lr) or OneCycleLR,
    stepped once per optimizer step; learning rate x0.1 when the finetune phase starts
    (trainer.py:217-271);
  * scalar names of the log (``train/total_loss``, ``train/diffusion_loss``, ``train/phase`` ...);
  * ``save_model`` checkpoint layout {model_state_dict, hyperparameters, epoch, global_step}
    (trainer.py:348-358); ``load_checkpoint`` also accepts a Lightning ``.ckpt`` of the reference
    trainer ({"state_dict": {"model.<key>": ...}, "hyper_parameters": ...}).
"""
from __future__ import annotations

import math
from typing import Any, Dict, Iterable, List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.optim import AdamW
from torch.optim.lr_scheduler import CosineAnnealingLR, OneCycleLR


class DiffusionLoss(nn.Module):
    """Elementwise loss between predicted and target noise with an optional per-row mask and reduction
    (training/losses.py:15-70)."""

    def __init__(self, loss_type: str = "mse", reduction: str = "mean"):
        super().__init__()
        if loss_type not in ("mse", "l1", "huber"):
            raise ValueError(f"Unknown loss type: {loss_type}")
        self.loss_type, self.reduction = loss_type, reduction

    def forward(self, predicted: torch.Tensor, target: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        fn = {"mse": F.mse_loss, "l1": F.l1_loss, "huber": F.smooth_l1_loss}[self.loss_type]
        loss = fn(predicted, target, reduction="none")
        if mask is not None:
            loss = loss * mask.unsqueeze(-1)
        return loss.mean() if self.reduction == "mean" else (loss.sum() if self.reduction == "sum" else loss)


class ContrastiveLoss(nn.Module):
    """InfoNCE over nodes with same-graph positives (training/losses.py:73-214).  O(N^2) memory: only
    for small batches; unreachable from the reference's pretraining step (D9), kept for the API."""

    def __init__(self, temperature: float = 0.1, similarity_function: str = "cosine", reduction: str = "mean"):
        super().__init__()
        if similarity_function not in ("cosine", "dot"):
            raise ValueError(f"Unknown similarity function: {similarity_function}")
        self.temperature, self.similarity_function, self.reduction = temperature, similarity_function, reduction

    def forward(self, embeddings: torch.Tensor, batch_indices: torch.Tensor) -> torch.Tensor:
        z = F.normalize(embeddings, dim=1)                       # the reference normalises for both similarity kinds
        sim = (z @ z.t() / self.temperature).exp()
        same = batch_indices.unsqueeze(0) == batch_indices.unsqueeze(1)
        same.fill_diagonal_(False)
        pos = (sim * same).sum(1).clamp_min(1e-8)
        loss = -(pos / sim.sum(1)).log()[same.any(1)]
        return loss.mean() if self.reduction == "mean" else (loss.sum() if self.reduction == "sum" else loss)


class _Scalars(dict):
    """Latest value of every logged scalar.  Values logged as device tensors are kept as tensors (no host sync in
    the step) and converted to float when read."""

    def __getitem__(self, k):
        return float(dict.__getitem__(self, k))

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self]

    def values(self):
        return [self[k] for k in self]


class DGDMTrainer(nn.Module):
    """Drop-in for the reference's Lightning module; drive it with ``fit`` or call the ``*_step``
    hooks from your own loop."""

    def __init__(self, model: nn.Module, learning_rate: float = 1e-4, weight_decay: float = 1e-5, pretrain_epochs: int = 50,
                 finetune_epochs: int = 50, masking_ratio: float = 0.15, diffusion_noise_schedule: str = "cosine",
                 use_contrastive_loss: bool = True, contrastive_temperature: float = 0.1, scheduler_type: str = "cosine",
                 warmup_steps: int = 1000, **kwargs):
        super().__init__()
        self.model = model
        self.learning_rate, self.weight_decay = learning_rate, weight_decay
        self.pretrain_epochs, self.finetune_epochs = pretrain_epochs, finetune_epochs
        self.masking_ratio = masking_ratio
        self.use_contrastive_loss = use_contrastive_loss
        self.scheduler_type, self.warmup_steps = scheduler_type, warmup_steps
        self.diffusion_loss = DiffusionLoss()
        self.contrastive_loss = ContrastiveLoss(temperature=contrastive_temperature) if use_contrastive_loss else None
        self.hparams: Dict[str, Any] = dict(learning_rate=learning_rate, weight_decay=weight_decay, pretrain_epochs=pretrain_epochs,
                                            finetune_epochs=finetune_epochs, masking_ratio=masking_ratio,
                                            diffusion_noise_schedule=diffusion_noise_schedule,
                                            use_contrastive_loss=use_contrastive_loss, contrastive_temperature=contrastive_temperature,
                                            scheduler_type=scheduler_type, warmup_steps=warmup_steps, **kwargs)
        self.current_phase = "pretrain"
        self.current_epoch = 0
        self.global_step = 0
        self.logged: Dict[str, float] = _Scalars()  # latest value of every logged scalar
        self._graphed: Optional["GraphedPretrainStep"] = None
        self._optimizer: Optional[torch.optim.Optimizer] = None
        self._scheduler = None
        # build-only: ``step_kwargs(global_step) -> dict`` of extra keyword arguments for ``model.pretrain_step`` (the random-draw /
        # decision injection hooks of DGDMModel), so that a parity test can drive the trainer and a CPU checker with the same draws
        self.step_kwargs = None

    # ------------------------------------------------------------------ logging sink
    def log(self, name: str, value, **_):
        dict.__setitem__(self.logged, name, value.detach() if torch.is_tensor(value) else float(value))

    def log_dict(self, d: Dict[str, Any], **_):
        for k, v in d.items():
            self.log(k, v)

    @property
    def device(self) -> torch.device:
        return next(self.model.parameters()).device

    def optimizers(self):
        return self._optimizer

    # ------------------------------------------------------------------ steps (trainer.py:87-216)
    def forward(self, batch, mode: str = "inference") -> Dict[str, torch.Tensor]:
        return self.model(batch, mode=mode, return_attention=True, return_embeddings=True)

    def training_step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        if self.current_epoch < self.pretrain_epochs:
            return self._pretrain_step(batch, batch_idx)
        return self._finetune_step(batch, batch_idx)

    def _pretrain_step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        extra = self.step_kwargs(self.global_step) if self.step_kwargs is not None else {}
        outputs = self.model.pretrain_step(batch, mask_ratio=self.masking_ratio, **extra)
        total = outputs["total_pretrain_loss"]
        if self.contrastive_loss is not None and "node_embeddings" in outputs:
            c = self.contrastive_loss(outputs["node_embeddings"], batch.batch)
            total = total + c
            self.log("train/contrastive_loss", c)
        self.log("train/total_loss", total)
        self.log("train/diffusion_loss", outputs["diffusion_loss"])
        if "reconstruction_loss" in outputs:
            self.log("train/reconstruction_loss", outputs["reconstruction_loss"])
        self.log("train/phase", 0.0)
        return total

    def _supervised_terms(self, outputs, batch, prefix: str, metrics: Dict[str, torch.Tensor]):
        total, n = 0.0, 0
        y = getattr(batch, "y", None)
        if "classification_logits" in outputs and y is not None and getattr(self.model, "classification_head", None) is not None:
            loss = self.model.classification_head.compute_loss(outputs["classification_logits"], y)
            acc = (outputs["classification_logits"].argmax(1) == y).float().mean()
            metrics[f"{prefix}classification_loss" if prefix == "train/" else "val_loss"] = loss
            metrics[f"{prefix}accuracy" if prefix == "train/" else "val_accuracy"] = acc
            total, n = total + loss, n + 1
        rt = getattr(batch, "regression_targets", None)
        if "regression_outputs" in outputs and rt is not None and getattr(self.model, "regression_head", None) is not None:
            loss = self.model.regression_head.compute_loss(outputs["regression_outputs"], rt)
            metrics[f"{prefix}regression_loss" if prefix == "train/" else "val_regression_loss"] = loss
            if prefix != "train/":
                metrics["val_mae"] = F.l1_loss(outputs["regression_outputs"], rt)
            total, n = total + loss, n + 1
        return total, n

    def _finetune_step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        outputs = self.forward(batch, mode="finetune")
        metrics: Dict[str, torch.Tensor] = {}
        total, n = self._supervised_terms(outputs, batch, "train/", metrics)
        self.log_dict(metrics)
        if n == 0:  # no supervised targets: fall back to the diffusion objective (trainer.py:164-170)
            total = self.model._compute_diffusion_loss(outputs["node_embeddings"], batch)["diffusion_loss"]
            self.log("train/diffusion_loss", total)
        self.log("train/total_loss", total)
        self.log("train/phase", 1.0)
        return total

    @torch.no_grad()
    def validation_step(self, batch, batch_idx: int = 0) -> Dict[str, torch.Tensor]:
        outputs = self.forward(batch, mode="inference")
        metrics: Dict[str, torch.Tensor] = {}
        self._supervised_terms(outputs, batch, "val/", metrics)
        self.log_dict(metrics)
        return metrics

    def test_step(self, batch, batch_idx: int = 0):
        return self.validation_step(batch, batch_idx)

    @torch.no_grad()
    def predict_step(self, batch, batch_idx: int = 0) -> Dict[str, Any]:
        outputs = self.forward(batch, mode="inference")
        pred: Dict[str, Any] = {"graph_embeddings": outputs["graph_embedding"], "node_embeddings": outputs.get("node_embeddings")}
        if "classification_probs" in outputs:
            pred["classification_probs"] = outputs["classification_probs"]
            pred["predicted_classes"] = outputs["classification_logits"].argmax(1)
        if "regression_outputs" in outputs:
            pred["regression_predictions"] = outputs["regression_outputs"]
        if "attention_weights" in outputs:
            pred["attention_weights"] = outputs["attention_weights"]
        return pred

    # ------------------------------------------------------------------ optimiser / schedule (trainer.py:217-271)
    def configure_optimizers(self, total_steps: int):
        """``total_steps`` = Lightning's ``trainer.estimated_stepping_batches``."""
        on_gpu = next(self.model.parameters()).is_cuda
        if on_gpu:      # the same AdamW arithmetic on one HIP launch (optim.py).  A model on the CPU has no forward here (the kernels
                        # raise DGDMKernelError): torch's AdamW is only built so that host-side logic (schedules, phase switch) can be tested
            from .optim import DGDMAdamW
            opt = DGDMAdamW(self.model.parameters(), lr=self.learning_rate, weight_decay=self.weight_decay)
        else:
            opt = AdamW(self.model.parameters(), lr=self.learning_rate, weight_decay=self.weight_decay)
        if self.scheduler_type == "cosine":
            sched = CosineAnnealingLR(opt, T_max=total_steps, eta_min=self.learning_rate * 0.01)
        elif self.scheduler_type == "onecycle":
            sched = OneCycleLR(opt, max_lr=self.learning_rate, total_steps=total_steps, pct_start=0.1)
        else:
            sched = None
        self._optimizer, self._scheduler = opt, sched
        return opt if sched is None else {"optimizer": opt, "lr_scheduler": {"scheduler": sched, "interval": "step", "frequency": 1}}

    def on_train_epoch_start(self):
        phase = "pretrain" if self.current_epoch < self.pretrain_epochs else "finetune"
        if phase != self.current_phase:
            self.current_phase = phase
            if phase == "finetune" and self._optimizer is not None:
                for group in self._optimizer.param_groups:
                    if torch.is_tensor(group["lr"]):
                        group["lr"].mul_(0.1)       # device-side learning rate of a recorded step: keep its address
                    else:
                        group["lr"] = group["lr"] * 0.1

    def on_validation_epoch_end(self):
        if self._optimizer is not None:
            self.log("learning_rate", self._optimizer.param_groups[0]["lr"])

    # ------------------------------------------------------------------ loop
    def fit(self, train_loader: Iterable, val_loader: Optional[Iterable] = None, max_epochs: Optional[int] = None,
            steps_per_epoch: Optional[int] = None, grad_reducer=None, on_step=None, graphed: bool = False) -> List[float]:
        """Runs ``max_epochs`` (default pretrain + finetune epochs) over ``train_loader`` (re-iterable;
        batches already on the model's device or exposing ``.to``).  ``grad_reducer``: an object
        with ``all_reduce()`` called between backward and the optimizer step (data parallel).
        ``graphed``: replay the pretrain step from HIP graphs -- one recording per recurring batch layout (GraphedStepCache);
        with a ``grad_reducer`` one recording (GraphedPretrainStep) for the layout of the first batch, other layouts eagerly.
        The finetune phase runs eagerly.
        Returns the per-step training losses."""
        max_epochs = self.pretrain_epochs + self.finetune_epochs if max_epochs is None else max_epochs
        if steps_per_epoch is None:
            steps_per_epoch = len(train_loader)  # type: ignore[arg-type]
        if self._optimizer is None:
            self.configure_optimizers(max_epochs * steps_per_epoch)
        losses: List[float] = []
        dev = self.device
        for epoch in range(self.current_epoch, max_epochs):
            self.current_epoch = epoch
            phase_before = self.current_phase
            self.on_train_epoch_start()
            if self.current_phase != phase_before:
                # other parameters carry gradients from here on: drop the recorded step (and its gradient buffers) and
                # let the reducer rebuild its flat buffer -- every rank switches at the same epoch
                self._graphed = None
                self._optimizer.zero_grad(set_to_none=True)
                if hasattr(grad_reducer, "reset"):
                    grad_reducer.reset()
            self.model.train()
            for i, batch in enumerate(train_loader):
                if i >= steps_per_epoch:
                    break
                batch = batch.to(dev) if hasattr(batch, "to") else batch
                loss = None
                if graphed and self.current_phase == "pretrain":
                    if self._graphed is None:
                        # a recording per recurring layout; data parallel: each split around the collective, all on ONE flat buffer
                        self._graphed = GraphedStepCache(self.model, self._optimizer, self.masking_ratio,
                                                         step_fn=lambda b: self._pretrain_step(b), grad_reducer=grad_reducer)
                    try:
                        loss = self._graphed(batch)
                    except BatchLayoutError:    # another batch layout: eager step below
                        loss = None
                if loss is None:
                    # gradients recorded by a graphed step must keep their addresses: zero them in place
                    fresh = self._graphed is None
                    self._optimizer.zero_grad(set_to_none=fresh)
                    loss = self.training_step(batch, i)
                    if fresh:       # .grad is None everywhere: the weight-gradient reductions may be deferred to the end of the pass
                        from . import ops
                        with ops.deferred_weight_grads():
                            loss.backward()
                    else:
                        loss.backward()
                    if grad_reducer is not None:
                        grad_reducer.all_reduce()
                    self._optimizer.step()
                if self._scheduler is not None:
                    self._scheduler.step()
                self.global_step += 1
                losses.append(loss.detach())
                if on_step is not None:
                    on_step(self, loss)
            if val_loader is not None:
                self.model.eval()
                for i, batch in enumerate(val_loader):
                    self.validation_step(batch.to(dev) if hasattr(batch, "to") else batch, i)
                self.on_validation_epoch_end()
            self.current_epoch = epoch + 1
        return [float(l) for l in losses]

    @torch.no_grad()
    def generate_embeddings(self, dataloader) -> Dict[str, Any]:
        self.model.eval()
        embs, labels, ids = [], [], []
        for batch in dataloader:
            batch = batch.to(self.device) if hasattr(batch, "to") else batch
            embs.append(self.forward(batch, mode="inference")["graph_embedding"].cpu())
            if getattr(batch, "y", None) is not None:
                labels.append(batch.y.cpu())
            if getattr(batch, "slide_id", None) is not None:
                ids.extend(batch.slide_id)
        return {"embeddings": torch.cat(embs, 0), "labels": torch.cat(labels, 0) if labels else None, "slide_ids": ids}

    # ------------------------------------------------------------------ construction / checkpoints (trainer.py:335-358)
    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "DGDMTrainer":
        from .models import DGDMModel
        return cls(model=DGDMModel(**config.get("model", {})), **config.get("training", {}))

    def save_model(self, filepath: str):
        torch.save({"model_state_dict": self.model.state_dict(), "hyperparameters": dict(self.hparams), "epoch": self.current_epoch,
                    "global_step": self.global_step}, filepath)

    def load_checkpoint(self, filepath: str, strict: bool = False, map_location=None) -> Dict[str, Any]:
        """Loads a ``save_model`` file or a Lightning checkpoint of the reference trainer.  ``strict=False``
        by default: this model has ``graph_encoder.dim_proj.*`` (repair R2) which reference files lack."""
        ckpt = torch.load(filepath, map_location=map_location or "cpu", weights_only=False)
        if "model_state_dict" in ckpt:
            state = ckpt["model_state_dict"]
        elif "state_dict" in ckpt:  # Lightning: keys carry the attribute name of the wrapped model
            state = {k[len("model."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("model.")}
        else:
            raise ValueError("not a DGDM checkpoint: expected 'model_state_dict' or 'state_dict'")
        missing, unexpected = self.model.load_state_dict(state, strict=strict)
        self.current_epoch = int(ckpt.get("epoch", 0))
        self.global_step = int(ckpt.get("global_step", 0))
        return {"missing_keys": list(missing), "unexpected_keys": list(unexpected),
                "hyperparameters": ckpt.get("hyperparameters", ckpt.get("hyper_parameters", {}))}


class BatchLayoutError(ValueError):
    """A batch does not have the tensor shapes / per-graph node offsets a recorded step was captured for."""


class GraphedPretrainStep:
    """``pretrain_step -> backward [-> gradient all-reduce] -> AdamW step`` recorded once as HIP graphs and
    replayed for every later batch of the same shape (same tensor shapes and per-graph node offsets).

    A step of the path is ~1000 short kernel launches; replaying them from a graph removes the host from
    the loop (the step is then bounded by the kernels alone, which matters most for small batches).  The
    work is the same as the eager step: entity masking draws fresh variates on every replay (torch's
    generator is graph-aware), and the dropout sites fold in the device-side seed epoch, advanced by a
    kernel at the head of the graph (csrc/common.hpp, DgdmSeed).

    * ``optimizer``: ``optim.DGDMAdamW`` (one launch) or torch's fused AdamW; it is switched to ``capturable`` (device-side
      step count and learning rate).  Set the learning rate with :meth:`set_lr` (an asynchronous fill, no host sync).
    * the first ``warmup`` calls run eagerly on a side stream (they are real training steps), the next call
      records, every later call replays.  A batch of another shape raises ``BatchLayoutError`` (a ``ValueError``) -- run it eagerly.
    * ``validate=True`` keeps the model's input checks (NaN / inf / edge range, one host readback per batch) in
      front of every step, as the eager forward has them; the recorded region itself cannot hold a readback.
    * with ``grad_reducer`` the graph is split around the collective: [forward+backward] -> all-reduce
      (eager, RCCL) -> [optimizer step].  ``one_message=True`` makes the eager warm-up steps exchange their gradients the way a
      replay does -- hooks passive, ONE all-reduce of the whole flat buffer -- so that a rank which replays and a rank which is
      still warming a layout up issue the same collective (``GraphedStepCache`` sets it).
    """

    def __init__(self, model: nn.Module, optimizer: torch.optim.Optimizer, mask_ratio: float = 0.15, grad_reducer=None,
                 warmup: int = 2, step_fn=None, validate: bool = True, one_message: bool = False):
        from . import _lib
        self._lib = _lib
        self.validate = validate
        self.one_message = one_message
        self.model, self.opt, self.mask_ratio, self.reducer, self.warmup = model, optimizer, mask_ratio, grad_reducer, warmup
        self.step_fn = step_fn or (lambda batch: model.pretrain_step(batch, mask_ratio=self.mask_ratio)["total_pretrain_loss"])
        self.dev = next(model.parameters()).device
        if self.dev.type != "cuda":
            raise _lib.DGDMKernelError("GraphedPretrainStep records HIP graphs: the model must live on the GPU")
        for group in optimizer.param_groups:
            if not group.get("fused"):
                raise ValueError("GraphedPretrainStep needs optim.DGDMAdamW or a fused optimizer (torch.optim.AdamW(..., fused=True))")
            group["capturable"] = True
            if not isinstance(group["lr"], torch.Tensor):
                group["lr"] = torch.tensor(float(group["lr"]), dtype=torch.float32, device=self.dev)
        for st in optimizer.state.values():          # an optimizer that already stepped keeps its count on the host
            if isinstance(st.get("step"), torch.Tensor) and not st["step"].is_cuda:
                st["step"] = st["step"].to(self.dev)
        self.static = None
        self._signature = None
        self._graphs: List[torch.cuda.CUDAGraph] = []
        self._loss = None
        self._calls = 0
        self._side = torch.cuda.Stream(self.dev)
        self.replay_done = torch.cuda.Event()         # see `input_buffers`: recorded behind every replay of the forward + backward graph

    # ------------------------------------------------------------------ static inputs
    # tensors of a batch the step reads (PyG-style objects keep them outside __dict__, so they are named here)
    FIELDS = ("x", "edge_index", "edge_attr", "pos", "batch", "y", "regression_targets")

    @classmethod
    def _sig(cls, batch):
        """Layout of a batch: shapes/dtypes of its tensors AND the per-graph node offsets as host integers -- two batches with the
        same total node count but another split must not share a recording (the graph bakes the offsets in).  A batch that carries
        its offsets on the host (GraphBatch.ptr) costs no device sync here; one that only has a device ``batch`` / ``ptr`` tensor
        costs one readback per step."""
        from .graph import graph_ptr
        sig = []
        for k in cls.FIELDS:
            v = getattr(batch, k, None)
            if isinstance(v, torch.Tensor):
                sig.append((k, tuple(v.shape), v.dtype))
        # the host-side extent of the positions decides whether the recording holds the zero-block map's launches
        # (ops.attn_zero_blocks_possible): a batch with another extent must not replay this recording
        sig.append(("pos_extent", getattr(batch, "pos_extent", None)))
        md = getattr(batch, "max_degree", None)      # whether the index sets carry long-row tables (GraphStructure): part of the layout
        sig.append(("long_rows", md is None or md + 1 > 128))
        sig.append(("ptr", tuple(graph_ptr(batch, batch.x.size(0)))))
        return tuple(sig)

    def _load(self, batch):
        sig = self._sig(batch)
        if self.static is None:
            from .graph import GraphBatch
            st = GraphBatch()
            for k in self.FIELDS:
                v = getattr(batch, k, None)
                setattr(st, k, v.clone() if isinstance(v, torch.Tensor) else v)
            st.ptr = list(sig[-1][1])            # host offsets: no device sync in the step
            st.pos_extent = getattr(batch, "pos_extent", None)
            st.max_degree = getattr(batch, "max_degree", None)
            self.static, self._signature = st, sig
            return
        if sig != self._signature:
            raise BatchLayoutError("GraphedPretrainStep: batch layout differs from the recorded one (tensor shapes or per-graph node "
                             "offsets); run this batch through the eager step")
        for k in self.FIELDS:
            v = getattr(batch, k, None)
            if isinstance(v, torch.Tensor):
                dst = getattr(self.static, k)
                # a loader that fills `input_buffers` in place hands the buffers back: nothing to copy.  Identity is the tensor object
                # or the same storage window WITH the same strides (a same-pointer view with other strides is another tensor: copied)
                same = v is dst or (v.data_ptr() == dst.data_ptr() and v.stride() == dst.stride() and v.dtype == dst.dtype)
                if not same:
                    dst.copy_(v, non_blocking=True)

    @property
    def input_buffers(self):
        """The recording's own input tensors (a ``GraphBatch``; ``None`` before the first call).  A data pipeline that writes the
        next batch of the same layout INTO these tensors (e.g. the host-to-device copy of its loader) and passes this object to
        ``__call__`` saves the device-to-device copy of the batch (130 MB per step at 4 x 10k nodes x 768 features).

        ORDERING CONTRACT: a replay reads these tensors on the stream ``__call__`` ran on.  Whoever writes the next batch into them
        from another stream (a loader's copy stream) must (i) wait for ``replay_done`` -- an event recorded on the step's stream
        right behind every replay -- before the write, and (ii) make the step's stream wait for the write before the next
        ``__call__`` (``torch.cuda.current_stream().wait_stream(copy_stream)``).  Writes issued on the step's own stream need
        neither."""
        return self.static

    def set_lr(self, lr: float, group: int = 0) -> None:
        self.opt.param_groups[group]["lr"].fill_(float(lr))

    # ------------------------------------------------------------------ the step
    def _advance_seed(self):
        self._lib.check(self._lib.load().dgdm_seed_epoch_advance(self._lib.stream_ptr(self.dev)), "dgdm_seed_epoch_advance")

    def _forward_backward(self):
        from . import ops
        self._advance_seed()
        loss = self.step_fn(self.static)
        with ops.deferred_weight_grads():        # gradients start as None here: the dW reductions of the pass run in one launch at its end
            loss.backward()
        return loss.detach()

    def _eager(self):
        if self.reducer is not None and self.one_message:
            self.reducer.capturing = True   # hooks passive: no early bucket, the exchange below is the replay's single message
            try:
                loss = self._forward_backward()
            finally:
                self.reducer.capturing = False
            self.reducer.pack()
            self.reducer.reduce_packed()
            self.reducer.adopt_views()
            self.opt.step()
            return loss
        loss = self._forward_backward()
        if self.reducer is not None:
            self.reducer.all_reduce()       # leaves p.grad as views of the reducer's flat buffer
        self.opt.step()
        return loss

    def _record(self):
        # gradients allocated while recording live in the graph's pool and are overwritten (not accumulated) by each
        # replay; nothing may free them afterwards (no zero_grad(set_to_none=True) between replayed steps)
        self.opt.zero_grad(set_to_none=True)
        # thread_local: the communicator's watchdog thread polls events of earlier collectives; that must not
        # invalidate the recording (and nothing of those collectives may still be in flight when it starts)
        torch.cuda.synchronize(self.dev)
        mode = dict(capture_error_mode="thread_local")
        g1 = torch.cuda.CUDAGraph()
        from . import ops
        ops.begin_amax_recording()               # the recording's first operand-maximum slot opens a chunk: its zero-fill is recorded too
        ops.WEIGHT_IMAGES.prepare(self.dev)      # the table of weight images the warm-up steps registered (a copy a capture cannot record)
        # the recording bakes the ADDRESSES of the batch-layout constants (graph offsets, per-graph sizes) into its kernel
        # arguments: hold them here, whatever the value cache of ops.device_constant evicts later
        if self.reducer is not None:
            self.reducer.capturing = True       # its hooks stay passive: the recording packs the buffer itself
        import gc
        gc_was = gc.isenabled()
        gc.disable()     # a cyclic collection DURING the capture could destroy an older recording (hipGraphDestroy inside a capture aborts)
        try:
            with ops.collect_device_constants() as self._held_constants, torch.cuda.graph(g1, **mode):
                self._loss = self._forward_backward()
                if self.reducer is None:
                    self.opt.step()
                else:
                    self.reducer.pack()          # graph 1 ends by laying the live gradients into the flat buffer (one multi-tensor copy)
            self._graphs = [g1]
            if self.reducer is not None:
                self.reducer.adopt_views()       # the optimizer step is recorded on the buffer's slices: nothing is copied back
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g1.pool(), **mode):
                    self.opt.step()
                self._graphs.append(g2)
        finally:
            self._amax_rec = ops.end_amax_recording()
            if gc_was:
                gc.enable()
            if self.reducer is not None:
                self.reducer.capturing = False

    def __call__(self, batch) -> torch.Tensor:
        """One training step on ``batch``; returns the (detached, device) loss of this step."""
        check = getattr(self.model, "_validate_forward_inputs", None)
        if self.validate and check is not None:
            try:
                check(batch, "pretrain", False, False)
            except Exception as e:
                from .models.dgdm_model import ModelInferenceError
                raise ModelInferenceError(f"Input validation failed: {e}")
        had = getattr(self.model, "validate_inputs", None)
        if had is not None:
            self.model.validate_inputs = False
        try:
            return self._step(batch)
        finally:
            if had is not None:
                self.model.validate_inputs = had

    def _step(self, batch) -> torch.Tensor:
        self._load(batch)
        self._calls += 1
        if self._calls <= self.warmup:
            self._side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(self._side):
                self.opt.zero_grad(set_to_none=True)
                loss = self._eager()
            torch.cuda.current_stream(self.dev).wait_stream(self._side)
            return loss
        if not self._graphs:
            self._record()          # recording launches nothing: the step itself is the replay below
        self._graphs[0].replay()
        self.replay_done.record(torch.cuda.current_stream(self.dev))     # the inputs have been consumed once this event has passed
        if self.reducer is not None:
            self.reducer.reduce_packed()    # the one eager piece: ONE all-reduce of the whole buffer graph 0 just packed
            self._graphs[1].replay()
        from . import ops
        ops.amax_recording_replayed(self._amax_rec)     # the replay rewrote the operand-maximum slots it owns
        ops.weights_changed()         # the replayed optimizer step wrote the weights without bumping their version counters
        return self._loss.clone()     # the recorded loss tensor is overwritten by the next replay


class GraphedStepCache:
    """Recorded steps for a stream of batches whose layouts RECUR (the same batch compositions epoch after epoch, a cycled
    validation-style set, fixed-size batches): one ``GraphedPretrainStep`` per layout signature, least-recently-used eviction.
    All recordings share the model, the optimizer and its state; each owns the activations and gradient buffers of its layout
    (without a reducer ``param.grad`` points at the buffers of the layout recorded last).

    Data parallel (``grad_reducer``): every layout's first graph ends by packing its gradients into the reducer's ONE flat buffer,
    the all-reduce of that buffer runs eagerly, and every layout's second graph is the optimizer step on the buffer's slices --
    ``param.grad`` is the same set of views whatever layout ran last.  Warm-up steps of a new layout exchange the same single
    message (``one_message``), so ranks whose layouts recur at different times still issue identical collectives, one per step.

    A layout seen for the first time runs ``warmup`` eager steps (they are real training steps; the first one also creates the
    layout's device constants, which a capture cannot), is recorded on the next, and replays from then on.  Graph padding to
    bucket sizes is NOT done: top-k pooling and attention are defined over the true node sets (graph_layers.py:306-310)."""

    def __init__(self, model: nn.Module, optimizer: torch.optim.Optimizer, mask_ratio: float = 0.15, step_fn=None, max_layouts: int = 16,
                 warmup: int = 1, validate: bool = True, grad_reducer=None):
        import collections
        self.model, self.opt, self.mask_ratio, self.step_fn, self.reducer = model, optimizer, mask_ratio, step_fn, grad_reducer
        self.max_layouts, self.warmup, self.validate = max_layouts, warmup, validate
        self.steps: "collections.OrderedDict" = collections.OrderedDict()
        self.replays = self.eager = 0

    def __call__(self, batch) -> torch.Tensor:
        sig = GraphedPretrainStep._sig(batch)
        st = self.steps.get(sig)
        if st is None:
            st = GraphedPretrainStep(self.model, self.opt, self.mask_ratio, warmup=self.warmup if self.steps else max(self.warmup, 2),
                                     step_fn=self.step_fn, validate=self.validate, grad_reducer=self.reducer, one_message=True)
            self.steps[sig] = st
            while len(self.steps) > self.max_layouts:
                self.steps.popitem(last=False)
        else:
            self.steps.move_to_end(sig)
        if st._graphs:
            self.replays += 1
        else:
            self.eager += 1
        return st(batch)

    def set_lr(self, lr: float, group: int = 0) -> None:
        self.opt.param_groups[group]["lr"].fill_(float(lr))


def closed_form_lr(step: int, base_lr: float, total_steps: int, finetune_start_step: Optional[int] = None) -> float:
    """Learning rate after ``step`` scheduler steps under the reference's cosine recipe
    (eta_min = 0.01*lr, T_max = total_steps), ignoring the x0.1 at the finetune switch when
    ``finetune_start_step`` is None.  CosineAnnealingLR is recursive, so scaling the group lr by 0.1
    at step s multiplies only the (lr - eta_min) excursion that the recursion carries forward."""
    eta_min = 0.01 * base_lr
    cos = lambda t: eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t / total_steps)) / 2
    if finetune_start_step is None or step < finetune_start_step:
        return cos(step)
    # recursion: lr_{t+1} - eta_min = (lr_t - eta_min) * (1 + cos(pi (t+1)/T)) / (1 + cos(pi t/T))
    lr = 0.1 * cos(finetune_start_step)
    for t in range(finetune_start_step, step):
        lr = eta_min + (lr - eta_min) * (1 + math.cos(math.pi * (t + 1) / total_steps)) / (1 + math.cos(math.pi * t / total_steps))
    return lr


@torch.no_grad()
def predict_graph(model: nn.Module, graph, return_attention: bool = False, return_embeddings: bool = False) -> Dict[str, Any]:
    """The dictionary ``DGDMPredictor.predict_graph`` returns (evaluation/predictor.py:188-257) for one
    graph: numpy arrays, python scalars, per-class / per-target entries, graph statistics."""
    model.eval()
    dev = next(model.parameters()).device
    graph = graph.to(dev) if hasattr(graph, "to") else graph
    out = model(graph, mode="inference", return_attention=return_attention, return_embeddings=return_embeddings)
    pred: Dict[str, Any] = {}
    if "classification_probs" in out:
        probs = out["classification_probs"].cpu().numpy()
        pred["classification_probs"] = probs
        pred["predicted_class"] = int(np.argmax(probs))
        pred["confidence"] = float(np.max(probs))
        for i, p in enumerate(probs.reshape(-1) if probs.ndim > 1 and probs.shape[0] == 1 else probs):
            pred[f"class_{i}_prob"] = float(p) if np.ndim(p) == 0 else p
    if "regression_outputs" in out:
        reg = out["regression_outputs"].cpu().numpy()
        pred["regression_outputs"] = reg
        for i, r in enumerate(reg.reshape(-1) if reg.ndim > 1 and reg.shape[0] == 1 else reg):
            pred[f"regression_target_{i}"] = float(r) if np.ndim(r) == 0 else r
    if "graph_embedding" in out:
        pred["graph_embedding"] = out["graph_embedding"].cpu().numpy()
    if return_embeddings and "node_embeddings" in out:
        pred["node_embeddings"] = out["node_embeddings"].cpu().numpy()
    if return_attention and "attention_weights" in out:
        aw = out["attention_weights"]
        pred["attention_weights"] = [a.cpu().numpy() for a in aw] if isinstance(aw, (list, tuple)) else aw.cpu().numpy()
    n_nodes = getattr(graph, "num_nodes", None)
    pred["num_nodes"] = int(n_nodes if n_nodes is not None else graph.x.size(0))
    pred["num_edges"] = int(graph.edge_index.size(1)) // 2   # undirected graph stored in both directions
    return pred


"""Diffusion block of the DGDM hot path (K7/K8).

Host-side mirror of the reference's ``core/diffusion.py`` (``DiffusionScheduler``,
``DiffusionLayer`` with identical parameter names).  Differences in *how* it is computed:

* the schedule tables live on the device (the reference re-uploads them on every call,
  diffusion.py:131,134) and per-graph coefficients are gathered with the ``batch`` vector, so a
  whole batch of graphs with different timesteps is one pass (the reference loops over graphs);
* the ``[x_t | t_emb]`` concat (diffusion.py:165-170) is never built: the time half of the first
  Linear is constant per graph, ``W[:, C:] @ t_emb_g + b`` is computed once per graph and added
  as a per-graph bias.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import ops


class DiffusionScheduler:
    """Beta schedules and derived tables (reference: diffusion.py:16-61)."""

    def __init__(self, num_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02, schedule: str = "cosine"):
        self.num_timesteps, self.schedule = num_timesteps, schedule
        T = num_timesteps
        if schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, T)
        elif schedule == "cosine":
            s = 0.008
            grid = torch.linspace(0, T, T + 1)
            ac = torch.cos(((grid / T) + s) / (1 + s) * math.pi * 0.5) ** 2
            ac = ac / ac[0]
            betas = torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
        elif schedule == "sigmoid":  # the reference maps the sigmoid ramp onto [beta_start, beta_end] (diffusion.py:34,56-61)
            betas = torch.sigmoid(torch.linspace(-6, 6, T)) * (beta_end - beta_start) + beta_start
        else:
            raise ValueError(f"Unknown schedule: {schedule}")
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.alphas_cumprod_prev = F.pad(self.alphas_cumprod[:-1], (1, 0), value=1.0)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self._dev = {}

    def on(self, device) -> dict:
        key = str(device)
        if key not in self._dev:
            ac = self.alphas_cumprod.to(device)
            self._dev[key] = dict(sqrt_ac=torch.sqrt(ac), sqrt_1mac=torch.sqrt(1.0 - ac), alphas=self.alphas.to(device),
                                  alphas_cumprod=ac, posterior_variance=self.posterior_variance.to(device))
        return self._dev[key]


class DiffusionLayer(nn.Module):
    """q-sample + time-conditioned noise predictor + DDPM sampler (reference: diffusion.py:64-275)."""

    def __init__(self, node_dim: int, hidden_dim: int, num_timesteps: int = 1000, schedule: str = "cosine",
                 conditioning_dim: Optional[int] = None):
        super().__init__()
        self.node_dim, self.hidden_dim, self.num_timesteps = node_dim, hidden_dim, num_timesteps
        self.scheduler = DiffusionScheduler(num_timesteps, schedule=schedule)
        self.time_embed = nn.Sequential(nn.Linear(128, hidden_dim), nn.SiLU(), nn.Linear(hidden_dim, hidden_dim))
        self.denoise_net = nn.Sequential(
            nn.Linear(node_dim + hidden_dim, hidden_dim * 2), nn.GroupNorm(8, hidden_dim * 2), nn.SiLU(), nn.Dropout(0.1),
            nn.Linear(hidden_dim * 2, hidden_dim), nn.GroupNorm(8, hidden_dim), nn.SiLU(), nn.Dropout(0.1),
            nn.Linear(hidden_dim, node_dim))
        self.condition_net = nn.Linear(conditioning_dim, hidden_dim) if conditioning_dim is not None else None

    def get_timestep_embedding(self, timesteps: Tensor, dim: int = 128) -> Tensor:
        half = dim // 2
        f = torch.exp(torch.arange(half, device=timesteps.device) * -(math.log(10000) / (half - 1)))
        e = timesteps.float()[:, None] * f[None, :]
        e = torch.cat([torch.sin(e), torch.cos(e)], dim=1)
        return F.pad(e, (0, 1)) if dim % 2 == 1 else e

    # -- batched (segment) forms: one timestep per graph, rows gathered with `seg` -----------
    def add_noise_segments(self, x0: Tensor, noise: Tensor, timesteps: Tensor, seg: Tensor, plan=None) -> Tensor:
        tab = self.scheduler.on(x0.device)
        if plan is not None and x0.dim() == 2 and x0.size(1) % 4 == 0:          # one launch for the whole batch (csrc/diffusion_ops.hip)
            return ops.qsample(x0, noise, timesteps, tab["sqrt_ac"], tab["sqrt_1mac"], plan)
        a, b = tab["sqrt_ac"][timesteps][seg].unsqueeze(-1), tab["sqrt_1mac"][timesteps][seg].unsqueeze(-1)
        return a * x0 + b * noise

    def time_features(self, timesteps: Tensor) -> Tensor:
        """time_embed(sinusoid(t)) (diffusion.py:147-163), [len(timesteps), hidden]."""
        emb = self._embedding_table(timesteps.device).index_select(0, timesteps.reshape(-1))
        te = ops.linear_small(emb, self.time_embed[0].weight, self.time_embed[0].bias, ops.ACT_SILU)
        return ops.linear_small(te, self.time_embed[2].weight, self.time_embed[2].bias)

    def time_bias(self, timesteps: Tensor) -> Tensor:
        """Per-timestep bias of the first denoiser layer, [len(timesteps), 2*hidden]: ``time_embed`` of the sinusoid
        (diffusion.py:147-163) pushed through the time half of ``denoise_net[0]`` plus its bias -- the ``[x_t | t_emb]`` concat of
        diffusion.py:165-170 without the concat.  A few rows: the exact-fp32 small-M kernels (csrc/smallm.hip)."""
        C = self.node_dim
        lin0 = self.denoise_net[0]
        return ops.linear_small(self.time_features(timesteps), lin0.weight[:, C:], lin0.bias)

    def _embedding_table(self, device) -> Tensor:
        """Sinusoidal embeddings of the T possible timesteps (diffusion.py:112-121), built once per device."""
        tabs = self.__dict__.setdefault("_emb_tables", {})
        key = str(device)
        if key not in tabs:
            tabs[key] = self.get_timestep_embedding(torch.arange(self.num_timesteps, device=device)).contiguous()
        return tabs[key]

    def _check_group_norms(self) -> None:
        for i in (1, 5):
            gn = self.denoise_net[i]
            if not ops.row_norm_supported(gn.num_channels, gn.num_groups):
                raise ops._lib.DGDMKernelError(f"GroupNorm({gn.num_groups}, {gn.num_channels}): the fused row kernels need group widths that "
                                               "are multiples of 4 channels")

    def _denoise_tail(self, h: Tensor) -> Tensor:
        """``h``: the first layer's output AFTER its GroupNorm + SiLU + dropout (``ops.denoise_first_layer(..., norm=...)``).
        Linear -> GroupNorm -> SiLU -> dropout (one launch where the GEMM takes the norm as its epilogue), then the last Linear."""
        gn, drop, lin = self.denoise_net[5], self.denoise_net[7], self.denoise_net[4]
        h = ops.linear_norm(h, lin.weight, lin.bias, gn.weight, gn.bias, groups=gn.num_groups, eps=gn.eps, act=ops.ACT_SILU, drop_p=drop.p,
                            training=self.training)
        return ops.lin(self.denoise_net[8], h)

    def predict_noise_segments(self, x_noisy: Tensor, timesteps: Tensor, seg: Tensor, plan=None) -> Tensor:
        """x_noisy [N_tot, C]; timesteps [B]; seg [N_tot] graph id per row; ``plan`` (ops.AttnPlan)
        carries the per-graph row offsets for the segment kernels."""
        if plan is None:
            if timesteps.numel() != 1:
                raise ValueError("predict_noise_segments needs the batch plan when graphs carry different timesteps")
            plan = ops.AttnPlan([0, x_noisy.size(0)], x_noisy.device)
        lin0 = self.denoise_net[0]
        self._check_group_norms()
        h = ops.denoise_first_layer(x_noisy, self.time_features(timesteps), lin0.weight, lin0.bias, plan, norm=self.denoise_net[1],
                                    drop_p=self.denoise_net[3].p, training=self.training)
        return self._denoise_tail(h)

    # -- reference-shaped API (same graph for every row) --------------------------------------
    def add_noise(self, x_start: Tensor, noise: Tensor, timesteps: Tensor) -> Tensor:
        tab = self.scheduler.on(x_start.device)
        a, b = tab["sqrt_ac"][timesteps], tab["sqrt_1mac"][timesteps]
        while a.dim() < x_start.dim():
            a, b = a.unsqueeze(-1), b.unsqueeze(-1)
        return a * x_start + b * noise

    def condition_features(self, condition: Optional[Tensor], rows: int) -> Optional[Tensor]:
        """condition_net(condition) (diffusion.py:158-161), [1, hidden] or [rows, hidden]; None without a condition or without a
        ``condition_net`` (the reference ignores the argument then: ``if condition is not None and self.condition_net is not None``)."""
        if condition is None or self.condition_net is None:
            return None
        c = condition.reshape(1, -1) if condition.dim() == 1 else condition
        if c.dim() != 2 or c.size(0) not in (1, rows):
            raise ValueError(f"condition must be [1, {self.condition_net.in_features}] or one row per node [{rows}, ...], got {tuple(condition.shape)}")
        return ops.lin(self.condition_net, c)

    def _first_layer_rows(self, x: Tensor, emb: Tensor) -> Tensor:
        """denoise_net[0..3] on [x | emb] with one embedding PER ROW (a per-row condition): the concatenation the reference forms
        (diffusion.py:165-170), one GEMM over K = node_dim + hidden, then GroupNorm + SiLU + dropout in the row kernel."""
        lin0, gn, drop = self.denoise_net[0], self.denoise_net[1], self.denoise_net[3]
        h = ops.lin(lin0, torch.cat([x, emb], dim=1))
        return ops.row_norm(h, gn.weight, gn.bias, groups=gn.num_groups, eps=gn.eps, act=ops.ACT_SILU, drop_p=drop.p, training=self.training)

    def predict_noise(self, x_noisy: Tensor, timesteps: Tensor, condition: Optional[Tensor] = None) -> Tensor:
        if x_noisy.dim() != 2:
            raise ValueError("predict_noise expects [N, C] (the 3-D form cannot run in the reference: GroupNorm, D5)")
        ce = self.condition_features(condition, x_noisy.size(0))
        if ce is None:
            seg = torch.zeros(x_noisy.size(0), dtype=torch.long, device=x_noisy.device)
            return self.predict_noise_segments(x_noisy, timesteps[:1], seg)
        self._check_group_norms()
        te = self.time_features(timesteps[:1]) + ce                  # t_emb + cond_emb (diffusion.py:160-161)
        if ce.size(0) == 1:                                          # one condition for every row: it rides in the per-graph bias
            lin0 = self.denoise_net[0]
            plan = ops.AttnPlan([0, x_noisy.size(0)], x_noisy.device)
            h = ops.denoise_first_layer(x_noisy, te, lin0.weight, lin0.bias, plan, norm=self.denoise_net[1], drop_p=self.denoise_net[3].p,
                                        training=self.training)
        else:
            h = self._first_layer_rows(x_noisy, te)
        return self._denoise_tail(h)

    def forward(self, x_start: Tensor, timesteps: Optional[Tensor] = None, noise: Optional[Tensor] = None,
                condition: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        if timesteps is None:
            timesteps = torch.randint(0, self.num_timesteps, (1,), device=x_start.device)
        if noise is None:
            noise = torch.randn_like(x_start)
        x_noisy = self.add_noise(x_start, noise, timesteps[:1])
        return x_noisy, self.predict_noise(x_noisy, timesteps, condition)

    @torch.no_grad()
    def sample(self, shape, device, condition: Optional[Tensor] = None, num_inference_steps: int = 50,
               x_init: Optional[Tensor] = None, step_noise: Optional[List[Tensor]] = None, graphed: bool = False) -> Tensor:
        """DDPM ancestral sampling (reference: diffusion.py:214-275): timesteps = linspace(T-1, 0, steps).long(), final step
        returns x0_hat without noise.  ``x_init`` / ``step_noise`` inject the random draws (tests).

        Per step: 3 tile GEMMs (the time half of the first layer enters as that GEMM's bias), 2 fused GroupNorm+SiLU rows, one
        update kernel (dgdm_ddpm_step) and one normal draw -- 7 launches, no host synchronisation; in eval mode at node_dim 128 / 256 the
        whole step is ONE launch (csrc/sample_step.hip, `ops.denoise_ddpm_step`) and the loop's draws are one launch up front.  The time-embedding MLP runs
        ONCE for the <= T distinct timesteps before the loop (3 small-M launches).  ``graphed=True`` records the whole loop as one
        HIP graph per (rows, steps) and replays it (fresh draws on every replay through torch's graph-aware generator)."""
        if len(shape) != 2 or shape[1] != self.node_dim:
            raise ValueError(f"sample expects shape [N, {self.node_dim}] (the 3-D form cannot run in the reference: GroupNorm, D5)")
        device = torch.device(device)
        steps = int(num_inference_steps)
        ts = torch.linspace(self.num_timesteps - 1, 0, steps).long().tolist()
        x = torch.randn(shape, device=device) if x_init is None else x_init.to(device=device, dtype=torch.float32).contiguous()
        if step_noise is not None:
            step_noise = [z.to(device=device, dtype=torch.float32).contiguous() for z in step_noise]
        ce = self.condition_features(None if condition is None else condition.to(device), shape[0])
        if not graphed or ce is not None:      # a condition is a caller-owned tensor: that loop is not recorded
            return self._sample_loop(x, ts, step_noise, ce)
        key = (tuple(shape), steps, str(device), self.training, step_noise is not None)
        cache = self.__dict__.setdefault("_sample_graphs", {})
        if key not in cache:
            sx = torch.empty(shape, device=device)
            sz = [torch.empty(shape, device=device) for _ in range(steps - 1)] if step_noise is not None else None
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):          # warm-up outside the capture (allocator, lazy module state)
                sx.copy_(x)
                self._sample_loop(sx, ts[:2] + ts[-1:], sz[:2] if sz else None)
            torch.cuda.current_stream(device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            import gc
            gc_was = gc.isenabled()
            gc.disable()      # no cyclic collection inside the capture (it could destroy another recording: illegal during capture)
            # the recording must rebuild the weight images it reads (the warm-up above built them, a plain capture would find them
            # fresh and record no build: replays after an optimizer step / load_state_dict would then run on old weights), and the
            # operand-maximum slots it takes must start on a chunk boundary so that every replay re-zeroes them
            ops.weights_changed()
            ops.begin_amax_recording()
            try:
                with ops.collect_device_constants() as held, torch.cuda.graph(g):
                    out = self._sample_loop(sx, ts, sz)
            finally:
                amax_rec = ops.end_amax_recording()
                if gc_was:
                    gc.enable()
            cache[key] = (g, sx, sz, out, held, amax_rec)
        g, sx, sz, out, _, amax_rec = cache[key]
        sx.copy_(x)
        if sz is not None:
            for dst, src in zip(sz, step_noise):
                dst.copy_(src)
        g.replay()
        ops.amax_recording_replayed(amax_rec)
        return out.clone()

    SAMPLE_NOISE_BYTES = 1 << 30      # the loop's normal draws are made in one launch when they fit this budget

    def _sample_loop(self, x: Tensor, ts: List[int], step_noise: Optional[List[Tensor]], ce: Optional[Tensor] = None) -> Tensor:
        device, C = x.device, self.node_dim
        sch = self.scheduler
        # fp32 table arithmetic as the reference does it on its tensors (diffusion.py:245-270), then host scalars
        s1mac, sac = torch.sqrt(1 - sch.alphas_cumprod).tolist(), torch.sqrt(sch.alphas_cumprod).tolist()
        salpha, svar = torch.sqrt(sch.alphas).tolist(), torch.sqrt(sch.posterior_variance).tolist()
        # first-layer bias of every timestep 0..T-1 in one go (3 small-M launches; no host-to-device copy inside the loop)
        all_t = ops.device_constant(range(self.num_timesteps), torch.long, device)
        tf = None
        if ce is None:
            bias = self.time_bias(all_t)                                                               # [T, 2*hidden]
        elif ce.size(0) == 1:      # one condition: t_emb + cond_emb (diffusion.py:160-161) through the time half of the first layer
            lin0 = self.denoise_net[0]
            bias = ops.linear_small(self.time_features(all_t) + ce, lin0.weight[:, C:], lin0.bias)
        else:                      # a condition per row: the embedding enters as columns of the first GEMM's operand
            tf = self.time_features(all_t)                                                             # [T, hidden]
        row = list(range(self.num_timesteps))
        w0x = self.denoise_net[0].weight[:, :C]
        gn1 = self.denoise_net[1]
        self._check_group_norms()
        dn = self.denoise_net
        fused = (tf is None and not (self.training and (dn[3].p > 0 or dn[7].p > 0)) and dn[1].num_groups == 8 and dn[5].num_groups == 8
                 and self.hidden_dim == 2 * C and all(m.bias is not None for m in (dn[0], dn[4], dn[8])) and ops.denoise_ddpm_step_supported(C))
        zs = None
        if fused and step_noise is None and len(ts) > 1 and (len(ts) - 1) * x.numel() * 4 <= self.SAMPLE_NOISE_BYTES:
            zs = torch.randn(len(ts) - 1, *x.shape, device=device)      # every step's draw in ONE launch (a launch per step otherwise)
        for i, t in enumerate(ts):
            last = i == len(ts) - 1
            if fused:      # the whole step in one launch (csrc/sample_step.hip): every operation of it is row-local
                z = None if last else (zs[i] if zs is not None else (torch.randn_like(x) if step_noise is None else step_noise[i]))
                x = ops.denoise_ddpm_step(x, z, w0x, dn[4].weight, dn[8].weight, bias[row[t]], gn1, dn[4].bias, dn[5], dn[8].bias,
                                          s1mac[t], sac[t], salpha[t], svar[t], last)
                continue
            # first layer (the time half of its weight enters as the bias) with its GroupNorm + SiLU + dropout
            if tf is not None:
                h = self._first_layer_rows(x, ce + tf[row[t]])
            else:
                h = ops.linear_norm(x, w0x, bias[row[t]], gn1.weight, gn1.bias, groups=gn1.num_groups, eps=gn1.eps, act=ops.ACT_SILU,
                                    drop_p=self.denoise_net[3].p, training=self.training)
            eps = self._denoise_tail(h)
            z = None if last else (torch.randn_like(x) if step_noise is None else step_noise[i])
            x = ops.ddpm_step(x, eps, z, s1mac[t], sac[t], salpha[t], svar[t], last)
        return x

"""AdamW of the DGDM trainer on one HIP kernel (csrc/optim.hip).

The reference builds ``torch.optim.AdamW(lr, weight_decay)`` (training/trainer.py:217-226).  torch's fused implementation
spends 6 launches / 0.19 ms per step on the ~180 live tensors of DGDM-Base; :class:`DGDMAdamW` runs the same arithmetic in one
launch (two beyond 96 tensors), keeps torch's optimizer interface (``param_groups`` for the LR schedulers, ``state`` with
``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, ``state_dict`` / ``load_state_dict`` interchangeable with
``torch.optim.AdamW``'s) and can be recorded into a HIP graph (the step count lives on the device and is advanced by the kernel;
the learning rate may be a device tensor).

Parameters without a gradient are skipped, as torch does (the dead parameters of DGDMModel, SURVEY D9, never move).  The step
count is per parameter in torch; here parameters that received their first gradient in the same ``step()`` call share ONE device
counter (a "cohort"): after the trainer's switch to fine-tuning the task heads start their own count, as they would in torch.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import torch

from . import _lib


class DGDMAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if isinstance(lr, torch.Tensor) and lr.numel() != 1:
            raise ValueError("Tensor lr must be 1-element")
        if not 0.0 <= float(lr):
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"Invalid beta parameters: {betas}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        # fused / capturable: what GraphedPretrainStep and torch's load_state_dict look at (the step count is a device tensor)
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False, foreach=None,
                        capturable=True, differentiable=False, fused=True, decoupled_weight_decay=True)
        super().__init__(params, defaults)
        self._cohorts: Dict[int, List[dict]] = {}       # group index -> [{"step": tensor, "ticket": tensor, "ids": set(id(p))}]
        self._tables: Dict[tuple, tuple] = {}           # (group, cohort, pointers) -> (ctypes array, count): rebuilt when a grad moves
        self._since_split: Dict[int, int] = {}          # group index -> host steps since a cohort of the group last split

    # ------------------------------------------------------------------ state
    def _init_state(self, p: torch.nn.Parameter, step: torch.Tensor) -> None:
        st = self.state[p]
        st["step"] = step
        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    def _rebuild_cohorts(self, gi: int, group: dict) -> None:
        """After load_state_dict every parameter holds its own copy of its count: group equal counts again (one host read per
        parameter, once)."""
        by_value: Dict[float, dict] = {}
        for p in group["params"]:
            st = self.state.get(p)
            if not st:
                continue
            s = st["step"]
            if not isinstance(s, torch.Tensor):
                s = torch.tensor(float(s), dtype=torch.float32, device=p.device)
            val = float(s)
            c = by_value.get(val)
            if c is None:
                c = by_value[val] = {"step": s.to(device=p.device, dtype=torch.float32).reshape(()).clone(),
                                     "ticket": torch.zeros(1, dtype=torch.int32, device=p.device), "ids": set()}
            c["ids"].add(id(p))
            st["step"] = c["step"]
        self._cohorts[gi] = list(by_value.values())

    def state_dict(self):
        """torch's layout; every parameter gets a COPY of its cohort's count (torch.optim.AdamW advances each parameter's own
        ``step`` tensor: handing it one shared tensor would advance that tensor once per parameter)."""
        sd = super().state_dict()
        sd["state"] = {k: {n: (v.clone() if n == "step" and isinstance(v, torch.Tensor) else v) for n, v in st.items()}
                       for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._cohorts, self._tables, self._since_split = {}, {}, {}
        for gi, group in enumerate(self.param_groups):
            group["fused"], group["capturable"] = True, True
            self._rebuild_cohorts(gi, group)

    MERGE_ABOVE = 8          # cohorts per group beyond which EVERY step looks for cohorts it can merge again (one device read)
    MERGE_AFTER_QUIET = 8    # ... and once, this many steps after the last split, whenever more than one cohort is left

    def _merge_cohorts(self, gi: int, group, live_ids) -> None:
        """Cohorts only ever split (a member without a gradient leaves with a copy of the count); a model whose branches get
        gradients on some batches only would drift towards one launch per parameter.  Cohorts whose members are ALL live
        in this step and whose counts are EQUAL are one cohort again: reads the counters once (a host sync; only when a group has
        more than MERGE_ABOVE cohorts, never inside a capture)."""
        cohorts = self._cohorts[gi]
        counts = torch.stack([c["step"].reshape(()) for c in cohorts]).tolist()
        merged, by_count = [], {}
        for c, n in zip(cohorts, counts):
            if c["ids"] and c["ids"] <= live_ids:
                if n in by_count:
                    by_count[n]["ids"] |= c["ids"]
                    continue
                by_count[n] = c
            if c["ids"]:
                merged.append(c)
        if len(merged) == len(cohorts):
            return
        owner = {i: c for c in merged for i in c["ids"]}
        for q in group["params"]:
            c = owner.get(id(q))
            if c is not None and self.state.get(q):
                self.state[q]["step"] = c["step"]
        self._cohorts[gi] = merged
        self._tables = {k: v for k, v in self._tables.items() if k[0] != gi}
        self._since_split[gi] = 0

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            if gi not in self._cohorts:
                self._cohorts[gi] = []
            cohorts = self._cohorts[gi]
            fresh = [p for p in live if not self.state.get(p)]
            if fresh:
                dev = fresh[0].device
                c = {"step": torch.zeros((), dtype=torch.float32, device=dev), "ticket": torch.zeros(1, dtype=torch.int32, device=dev),
                     "ids": set()}
                for p in fresh:
                    if p.grad.is_sparse:
                        raise RuntimeError("DGDMAdamW does not support sparse gradients")
                    if p.dtype != torch.float32 or not p.is_cuda:
                        raise _lib.DGDMKernelError(f"DGDMAdamW steps fp32 parameters on the GPU (got {p.dtype} on {p.device})")
                    self._init_state(p, c["step"])
                    c["ids"].add(id(p))
                cohorts.append(c)
            capturing = torch.cuda.is_current_stream_capturing()
            live_ids = {id(p) for p in live}
            quiet = self._since_split.get(gi, 0)
            if not capturing and (len(cohorts) > self.MERGE_ABOVE or (len(cohorts) > 1 and quiet == self.MERGE_AFTER_QUIET)):
                self._merge_cohorts(gi, group, live_ids)
                cohorts = self._cohorts[gi]
            self._since_split[gi] = quiet + 1
            lr = group["lr"]
            lr_dev = lr.data_ptr() if isinstance(lr, torch.Tensor) and lr.is_cuda else None
            lr_host = 0.0 if lr_dev is not None else float(lr)
            b1, b2 = group["betas"]
            for ci in range(len(cohorts)):
                c = cohorts[ci]
                ps = [p for p in live if id(p) in c["ids"]]
                if not ps:
                    continue
                if len(ps) < len(c["ids"]):
                    # Members without a gradient in this step (frozen after a phase switch, a branch the batch did not take with
                    # zero_grad(set_to_none=True)): torch.optim.AdamW would not advance THEIR step count, and the cohort's one
                    # counter is about to advance.  They leave with a copy of the count as it stands and form a cohort of
                    # their own; when they come back they step from where they stopped.
                    gone = c["ids"] - live_ids
                    if capturing:
                        # the clone below would be RECORDED: every replay would overwrite the departed members' count with the live
                        # cohort's.  The live set of a recorded step must be the live set of the eager steps before it.
                        raise RuntimeError("DGDMAdamW: the set of parameters with a gradient changed inside a stream capture "
                                           f"({len(gone)} of {len(c['ids'])} members of a cohort have none); run one eager step with this "
                                           "set first (GraphedPretrainStep's warm-up steps do)")
                    c2 = {"step": c["step"].clone(), "ticket": torch.zeros(1, dtype=torch.int32, device=c["step"].device), "ids": gone}
                    for q in group["params"]:
                        if id(q) in gone:
                            self.state[q]["step"] = c2["step"]
                    c["ids"] = c["ids"] - gone
                    cohorts.append(c2)
                    self._since_split[gi] = 0
                ptrs = []
                for p in ps:
                    g, st = p.grad, self.state[p]
                    if g.dtype != torch.float32 or not g.is_contiguous() or not p.is_contiguous():
                        raise _lib.DGDMKernelError("DGDMAdamW needs contiguous fp32 parameters and gradients")
                    ptrs.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()))
                key = (gi, ci)
                cached = self._tables.get(key)
                if cached is None or cached[0] != ptrs:
                    arr = (_lib.AdamTensor * len(ptrs))(*[_lib.AdamTensor(*t) for t in ptrs])
                    self._tables[key] = cached = (ptrs, arr)
                dev = ps[0].device
                _lib.check(lib.dgdm_adamw_step(C.cast(cached[1], C.c_void_p), len(ptrs), lr_dev, lr_host, float(b1), float(b2), float(group["eps"]),
                                               float(group["weight_decay"]), c["step"].data_ptr(), c["ticket"].data_ptr(), _lib.stream_ptr(dev)),
                           "dgdm_adamw_step")
        return loss

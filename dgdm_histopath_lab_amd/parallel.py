"""Data-parallel harness for the DGDM hot path: one process per GPU, slides sharded across
ranks, the live gradients exchanged through one flat fp32 buffer (two buckets) per step over RCCL/xGMI
(`torch.distributed` backend "nccl" == RCCL on ROCm).

The reference gets data parallelism implicitly from Lightning's DDP (cli/train.py:346-359) and
would fail there: ~46 % of DGDMModel's parameters never receive a gradient (dead
``node_to_qkv`` / ``edge_to_key`` / ``norm2`` / ``pos_encoding`` / ``spatial_proj``, SURVEY.md 2b,
D9) and no ``find_unused_parameters`` is set.  Here only parameters that actually carry a gradient
enter the buffer (Base: 14.6 MB instead of 28.5 MB).  Graphs never interact inside a batch
(block-diagonal edges, per-graph attention / loss / pooling), so there is no data-path
collective; the only exchange is the gradient sum.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


class FlatGradAllReducer:
    """Average gradients across ranks through ONE persistent flat fp32 buffer, exchanged in two buckets.

    * The set of live parameters is fixed at the first call (same on every rank: same model, same mode) and checked
      afterwards; dead parameters (D9) never enter the buffer.  They are laid out in the ORDER THEIR GRADIENTS COMPLETED in that
      first backward (post-accumulate hooks), i.e. last layers first: bucket 0 = the first ``bucket_split`` of the bytes is
      complete while the backward still runs through the encoder.
    * From the second step on (eager mode) a hook on the last parameter of bucket 0 packs that bucket and starts its all-reduce
      ASYNCHRONOUSLY, under the tail of the backward; ``all_reduce()`` (called after backward) packs and reduces bucket 1 and waits
      for both.  xGMI is point-to-point (7 links per GPU): two large messages keep the links busy while paying the collective
      latency twice, not ~140 times.
    * After the exchange ``p.grad`` of every live parameter IS its slice of the flat buffer (a view): the optimizer reads the
      averaged gradients in place -- nothing is copied back.  ``ReduceOp.AVG`` does the division inside the collective (NCCL/RCCL);
      gloo (CPU tests) sums and scales.
    * Recorded steps (training.GraphedPretrainStep) call ``pack()`` inside the recording (one multi-tensor copy, no host cost),
      ``reduce_packed()`` eagerly between the two graphs (ONE all-reduce of the whole buffer: nothing runs beside it there), and
      record the optimizer step on the views.
    * The early launch and gradient accumulation (two backwards before one ``all_reduce()``) cannot be combined: ``begin_step``
      detects the second forward and runs that step without overlap.
    """

    def __init__(self, module: torch.nn.Module, world_size: Optional[int] = None, group=None, always: bool = False,
                 bucket_split: float = 0.5, overlap: bool = True):
        self.always = always          # issue the collective even with one rank (rehearsals of the N > 1 path)
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.group, self.bucket_split, self.overlap = group, bucket_split, overlap
        self.live: Optional[List[torch.nn.Parameter]] = None
        self.flat: Optional[torch.Tensor] = None
        self._order: List[torch.nn.Parameter] = []
        self._probe = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._fired = 0
        self._early = None            # in-flight work handle of bucket 0
        self._forwards = 0            # grad-enabled forwards of the module since the last exchange (> 1: gradient accumulation)
        self._fwd_probe = module.register_forward_pre_hook(lambda m, args: self.begin_step())
        self.capturing = False        # set by a recording: hooks stay passive, the recording calls pack() itself
        self.stats = {"early_launches": 0, "steps": 0}
        self.record_events = False    # diagnostics: HIP events at the early launch and at all_reduce() entry (see backward_tail_ms)
        self._ev_early = self._ev_end = None
        self.backward_tail_ms: List[float] = []     # per step: backward time left on the stream when bucket 0 left (record_events)

    # ------------------------------------------------------------------ hooks
    def begin_step(self) -> None:
        """Called at the start of every grad-enabled forward of the module (a forward pre-hook does; call it yourself when the
        loss does not go through ``module.__call__``).  The early-launch state belongs to ONE forward/backward/all_reduce() round:
        * a step abandoned between backward and ``all_reduce()`` (an exception, a skipped non-finite step) leaves bucket 0's
          collective in flight and the hook counter set -- waited for and cleared here, so that the next step launches early again
          instead of waiting on a stale handle;
        * a SECOND forward before ``all_reduce()`` is gradient accumulation: the early launch cannot be combined with it (bucket 0
          would leave after the first micro-batch, and once ``.grad`` is a view of the flat buffer the second backward would
          accumulate into memory the collective is still reducing).  Overlap is switched off for such a step: what is in flight
          is waited for and discarded, ``all_reduce()`` packs and sends both buckets after the last backward."""
        if self.capturing or not torch.is_grad_enabled():
            return
        self._forwards += 1
        if self._forwards > 1 and self.live is not None and all(p.grad is None for p in self.live):
            self._forwards = 1        # gradients were dropped since the last forward: the earlier step was abandoned, this is a new one
        stale, self._early, self._fired = self._early, None, 0
        if stale is not None:
            stale.wait()

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if self.live is None:
            self._order.append(p)                 # first backward: learn the completion order
            return
        if self.capturing or not self.overlap or not self._active() or self._forwards > 1:
            return
        if id(p) in self._bucket0_ids:
            self._fired += 1
            if self._fired == len(self._bucket0_ids) and self._early is None:
                from . import ops
                ops.flush_deferred_tn()           # weight gradients whose reduction was queued for the end of the pass (ops.py)
                self._pack(0)
                self._early = self._launch(0, async_op=True)
                self.stats["early_launches"] += 1
                if self.record_events:
                    self._ev_early = torch.cuda.Event(enable_timing=True)
                    self._ev_early.record()

    def _active(self) -> bool:
        return self.world > 1 or self.always

    # ------------------------------------------------------------------ layout
    def _setup(self):
        have = [p for p in self.params if p.grad is not None]
        seen = {id(p) for p in self._order}
        order = [p for p in self._order if p.grad is not None] + [p for p in have if id(p) not in seen]
        uniq, ids = [], set()
        for p in order:                            # a parameter used twice fires once per backward; keep the first position
            if id(p) not in ids:
                uniq.append(p); ids.add(id(p))
        self.live, self._live_ids = uniq, ids
        n = sum(p.numel() for p in self.live)
        ref = self.live[0]
        self.flat = torch.empty(n, dtype=ref.grad.dtype, device=ref.grad.device)
        self.views, off, cut = [], 0, None
        for i, p in enumerate(self.live):
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
            if cut is None and off >= self.bucket_split * n:
                cut = (i + 1, off)
        k, split = cut if cut is not None else (len(self.live), n)
        if k == len(self.live) and len(self.live) > 1:      # keep two non-empty buckets
            k, split = len(self.live) - 1, n - self.live[-1].numel()
        self._bucket_params = [self.live[:k], self.live[k:]]
        self._bucket_views = [self.views[:k], self.views[k:]]
        self._bucket_flat = [self.flat[:split], self.flat[split:]]
        self._bucket0_ids = {id(p) for p in self.live[:k]}

    def reset(self) -> None:
        """Forget the live set (call on every rank at the same step, e.g. when the training phase changes)."""
        stale = self._early
        self.live, self.flat, self._order, self._fired, self._early, self._forwards = None, None, [], 0, None, 0
        if stale is not None:
            stale.wait()

    @property
    def nbytes(self) -> int:
        return 0 if self.flat is None else self.flat.numel() * self.flat.element_size()

    @property
    def bucket_nbytes(self) -> List[int]:
        return [] if self.flat is None else [b.numel() * b.element_size() for b in self._bucket_flat]

    # ------------------------------------------------------------------ exchange
    def _pack(self, b: int) -> None:
        src, dst = [], []
        for p, v in zip(self._bucket_params[b], self._bucket_views[b]):
            if p.grad.data_ptr() != v.data_ptr():        # already a view of the buffer: nothing to move
                src.append(p.grad); dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)

    def _launch(self, b: int, async_op: bool = False):
        buf = self._bucket_flat[b]
        if buf.numel() == 0:
            return None
        if dist.get_backend(self.group) == "nccl":
            return dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            return _ScaledWork(w, buf, self.world)
        buf.div_(self.world)
        return None

    def _check_live(self) -> None:
        if any(p.grad is None for p in self.live) or any(p.grad is not None and id(p) not in self._live_ids for p in self.params):
            raise RuntimeError("the set of parameters receiving gradients changed between steps")

    def pack(self) -> None:
        """Copy every live gradient into the flat buffer (first call: fixes the live set and the layout)."""
        if self.live is None:
            self._setup()
        self._check_live()
        self._pack(0); self._pack(1)

    def reduce_packed(self) -> None:
        """All-reduce the (already packed) buffer as ONE message: between two recorded graphs nothing can overlap with it, and a
        second collective only adds its latency (measured on one MI355X through a single-rank RCCL group,
        tools/attic/dist_overhead_probe.py: +0.3 ms per step for the second message)."""
        self._forwards = 0
        if self.flat.numel() == 0:
            return
        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self.world)

    def adopt_views(self) -> None:
        """Make ``p.grad`` of every live parameter its slice of the flat buffer."""
        for p, v in zip(self.live, self.views):
            p.grad = v

    def all_reduce(self) -> None:
        if not self._active():
            return
        self.stats["steps"] += 1
        if self.live is None:
            self._setup()
        self._check_live()
        early, self._early, self._fired, self._forwards = self._early, None, 0, 0
        if self.record_events and self._ev_early is not None:
            self._ev_end = torch.cuda.Event(enable_timing=True)
            self._ev_end.record()
            self._pending_tail = (self._ev_early, self._ev_end)
            self._ev_early = None
        if early is None:
            self._pack(0)
            early = self._launch(0, async_op=True)
        self._pack(1)
        late = self._launch(1, async_op=True)
        for w in (early, late):
            if w is not None:
                w.wait()
        self.adopt_views()
        if self.record_events and getattr(self, "_pending_tail", None) is not None:
            a, b = self._pending_tail
            self._pending_tail = None
            b.synchronize()
            self.backward_tail_ms.append(a.elapsed_time(b))


class _ScaledWork:
    """Async SUM all-reduce followed by the 1/world scale (back-ends without ReduceOp.AVG)."""

    def __init__(self, work, buf, world):
        self.work, self.buf, self.world = work, buf, world

    def wait(self):
        self.work.wait()
        self.buf.div_(self.world)


def shard_slides(num_slides: int, rank: int, world: int) -> range:
    """Contiguous equal shards (slides are independent units): rank r owns [r*B/W, (r+1)*B/W)."""
    per = num_slides // world
    extra = num_slides % world
    start = rank * per + min(rank, extra)
    return range(start, start + per + (1 if rank < extra else 0))


def slide_cost(num_nodes: int, num_edges: int = 0, alpha: float = 1.0, beta: float = 2.0e4, gamma: float = 2.0e3) -> float:
    """Per-slide step cost model: dense node attention ~ alpha*N^2, per-node GEMM/norm work ~ beta*N,
    message passing ~ gamma*E (coefficients from the measured kernel times)."""
    return alpha * num_nodes * num_nodes + beta * num_nodes + gamma * num_edges


def balance_slides(costs: Sequence[float], world: int) -> List[List[int]]:
    """Cost-aware sharding for mixed-size slide streams (BASELINE config 5): longest-processing-time
    greedy -- heaviest slide first onto the least loaded rank.  Returns slide indices per rank."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    loads, bins = [0.0] * world, [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: loads[k])
        bins[r].append(i)
        loads[r] += costs[i]
    return bins


class BalancedSlideLoader:
    """Per-rank batches of a data-parallel step for slides of MIXED size (BASELINE config 5): every step takes the next
    ``global_batch`` slides of the (rank-identical) slide list and hands each rank the bin the cost-aware LPT sharding gives it
    (``balance_slides`` over ``slide_cost``: attention makes a slide's cost grow with N^2, so equal COUNTS per rank would leave the
    ranks waiting for whoever drew the large slides).  Re-iterable (one pass = one epoch, same batches every epoch -- which is
    what lets ``DGDMTrainer.fit(graphed=True)`` replay recorded steps for recurring layouts); feeds ``DGDMTrainer.fit`` directly.

    ``slides``: sequence of ``GraphData`` (or a callable ``i -> GraphData`` with ``num_slides``); every rank must pass the same
    list.  ``max_over_mean_load()`` reports the worst step's rank imbalance under the cost model."""

    def __init__(self, slides, global_batch: int, world: int, rank: int, num_slides: Optional[int] = None, device=None,
                 cost=slide_cost, drop_last: bool = True):
        self.get = slides if callable(slides) else slides.__getitem__
        self.n = num_slides if num_slides is not None else len(slides)
        if global_batch < world:
            raise ValueError(f"global_batch {global_batch} < world size {world}: some rank would get no slide")
        self.global_batch, self.world, self.rank, self.device, self.cost, self.drop_last = global_batch, world, rank, device, cost, drop_last
        self._sizes = None

    def __len__(self) -> int:
        return self.n // self.global_batch if self.drop_last else (self.n + self.global_batch - 1) // self.global_batch

    def plan(self):
        """[(slide indices of every rank) per step] -- identical on all ranks (pure function of the slide sizes)."""
        if self._sizes is None:
            self._sizes = []
            for i in range(self.n):
                g = self.get(i)
                self._sizes.append((int(g.x.size(0)), int(g.edge_index.size(1))))
        steps = []
        for s in range(len(self)):
            idx = list(range(s * self.global_batch, min(self.n, (s + 1) * self.global_batch)))
            bins = balance_slides([self.cost(*self._sizes[i]) for i in idx], self.world)
            steps.append([sorted(idx[j] for j in b) for b in bins])
        return steps

    def max_over_mean_load(self) -> float:
        worst = 1.0
        for bins in self.plan():
            loads = [sum(self.cost(*self._sizes[i]) for i in b) for b in bins]
            worst = max(worst, max(loads) / (sum(loads) / len(loads)))
        return worst

    def __iter__(self):
        from .graph import GraphBatch
        for bins in self.plan():
            mine = bins[self.rank]
            if not mine:
                raise RuntimeError("a rank received no slide for a step (global_batch too small for this world size)")
            b = GraphBatch.from_data_list([self.get(i) for i in mine])
            yield b.to(self.device) if self.device is not None else b

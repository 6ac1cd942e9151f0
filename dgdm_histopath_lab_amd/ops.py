"""Python wrappers (torch.autograd.Function) over the C ABI of libdgdm_hip.so.

Every function here launches hand-written HIP kernels on the current torch stream through
ctypes; tensors only provide device memory.  No CPU path exists.
"""
from __future__ import annotations

import collections
from typing import Optional

import math

import torch

from . import _lib
from .graph import GraphStructure


class KernelTimers:
    """Optional per-kernel timing with HIP events recorded on the stream the kernels are launched
    on (torch's current stream).  Used by bench.py for the roofline leg; off by default."""

    def __init__(self):
        self.enabled = False
        self.events = {}

    def start(self, names=None):
        self.enabled, self.events, self.names = True, {}, (set(names) if names else None)

    def stop(self):
        self.enabled = False

    def timed(self, name, fn):
        if not self.enabled or (self.names is not None and name not in self.names):
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.events.setdefault(name, []).append((a, b))
        return r

    def summary(self, stat: str = "mean"):
        """name -> (launches, mean or median ms per launch).  Call after torch.cuda.synchronize().  The median is what a fixed-shape
        workload should report: the start event of a pair runs the moment the GPU reaches it, so ONE host hiccup between the event and
        the launch behind it (GPU idle meanwhile) inflates that pair -- round 6 saw 4.17 ms for a 3.4 ms kernel from five launches."""
        import statistics
        f = statistics.median if stat == "median" else (lambda xs: sum(xs) / len(xs))
        return {k: (len(v), f([a.elapsed_time(b) for a, b in v])) for k, v in self.events.items()}


TIMERS = KernelTimers()


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise _lib.DGDMKernelError(f"HIP kernels compute in fp32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- K2 SpMM
def _lr(long_rows):
    import ctypes
    return None if long_rows is None else ctypes.byref(long_rows)


def spmm_raw(rowptr, col, w, X, num_rows: int, *, table_rows: Optional[int] = None, out: Optional[torch.Tensor] = None,
             bias: Optional[torch.Tensor] = None, accumulate: bool = False, addend: Optional[torch.Tensor] = None,
             long_rows=None) -> torch.Tensor:
    """Y[r] = sum_p w[p] X[col[p]] (+bias): dgdm_spmm.  X may be a column-strided view
    (row stride multiple of 4 floats); ``out`` likewise.  ``long_rows``: ``GraphStructure.long_rows(transposed)`` of the CSR that
    ``rowptr`` belongs to -- rows of hundreds of entries (hubs) are then split over many lane groups instead of one wavefront."""
    _lib.require_cuda(X, rowptr, col, w)
    lib = _lib.load()
    if X.dim() != 2 or X.stride(1) != 1:
        X = _f32c(X)
    C = X.size(1)
    if out is None:
        out = torch.empty(num_rows, C, dtype=torch.float32, device=X.device)
    assert out.stride(1) == 1 and out.size(1) == C and out.size(0) == num_rows
    tr = X.size(0) if table_rows is None else table_rows
    ldx = X.stride(0) if X.size(0) > 1 else max(C, X.stride(0))
    ldy = out.stride(0) if out.size(0) > 1 else max(C, out.stride(0))
    if addend is not None:   # Y = result + addend (dgdm_spmm_add)
        if bias is not None or accumulate:
            raise ValueError("addend excludes bias / accumulate")
        addend = _rowmajor(addend)
        TIMERS.timed(f"spmm_c{C}", lambda: _lib.check(
            lib.dgdm_spmm_add(rowptr.data_ptr(), col.data_ptr(), w.data_ptr(), X.data_ptr() if X.numel() else None, ldx, tr, addend.data_ptr(),
                              addend.stride(0) if addend.size(0) > 1 else max(C, addend.stride(0)), out.data_ptr(), ldy, num_rows, C,
                              _lr(long_rows), _lib.stream_ptr(X.device)), "dgdm_spmm_add"))
        return out
    TIMERS.timed(f"spmm_c{C}", lambda: _lib.check(
        lib.dgdm_spmm(rowptr.data_ptr(), col.data_ptr(), w.data_ptr(), X.data_ptr() if X.numel() else None, ldx, tr,
                      out.data_ptr(), ldy, num_rows, C, _lib.ptr(bias), int(accumulate), _lr(long_rows), _lib.stream_ptr(X.device)), "dgdm_spmm"))
    return out


class _Aggregate(torch.autograd.Function):
    """Y = A_hat X with A_hat the GCN-normalised adjacency incl. self loops
    (core/graph_layers.py:76-110); backward is the same kernel on the by-source CSR."""

    @staticmethod
    def forward(ctx, x, gs: GraphStructure, out):
        ctx.gs = gs
        return spmm_raw(gs.rowptr, gs.col, gs.w, x, gs.num_nodes, out=out, long_rows=gs.long_rows())

    @staticmethod
    def backward(ctx, gy):
        gs = ctx.gs
        return spmm_raw(gs.rowptr_t, gs.col_t, gs.w_t, _f32c(gy), gs.num_nodes, long_rows=gs.long_rows(True)), None, None


def aggregate(x: torch.Tensor, gs: GraphStructure, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    return _Aggregate.apply(x, gs, out)


class _AggregateConcat(torch.autograd.Function):
    """[A_hat x | EA_hat] as ONE [N, C + edge_dim] buffer: the SpMM writes its column block in place
    (strided output), the aggregated edge attributes fill the rest.  Feeds the single contraction
    [A_hat x | EA_hat] . [W | W_e]^T of GraphConvolution (core/graph_layers.py:89-110)."""

    @staticmethod
    def forward(ctx, x, ea_hat, gs: GraphStructure):
        cin, ed = x.size(1), ea_hat.size(1)
        buf = torch.empty(x.size(0), cin + ed, dtype=torch.float32, device=x.device)
        buf[:, cin:] = ea_hat
        spmm_raw(gs.rowptr, gs.col, gs.w, x, gs.num_nodes, out=buf[:, :cin], long_rows=gs.long_rows())
        ctx.gs, ctx.cin = gs, cin
        return buf

    @staticmethod
    def backward(ctx, gbuf):
        gs = ctx.gs
        return spmm_raw(gs.rowptr_t, gs.col_t, gs.w_t, gbuf[:, :ctx.cin], gs.num_nodes, long_rows=gs.long_rows(True)), None, None


def aggregate_concat(x: torch.Tensor, ea_hat: torch.Tensor, gs: GraphStructure) -> torch.Tensor:
    return _AggregateConcat.apply(x, ea_hat, gs)


# The graph U-Net's decoder convolves level j's nodes with the edge list of level j + 1 (reference defect D10, replicated under
# strict_reference): True derives those three index sets from the encoder side's by one copying launch each
# (GraphStructure.extended); False builds them from the edge list (six launches each; same arrays bit for bit).
CSR_EXTEND = True


def aggregate_edge_attr(edge_attr: Optional[torch.Tensor], gs: GraphStructure) -> torch.Tensor:
    """EA_hat[d] = sum_{e -> d} norm_e * edge_attr[e]  ([N, edge_dim]); the appended self-loop
    entries carry a zero attribute row (repair R1) and ``edge_attr=None`` means zeros
    (models/encoders.py:258-261).  Because ``edge_lin`` has no bias (graph_layers.py:49) the
    per-edge term of GraphConvolution.message equals ``EA_hat @ W_e^T`` -- one 32-wide
    aggregation per graph structure instead of an [E', C] intermediate per convolution."""
    if edge_attr is None:
        return None
    ea = _f32c(edge_attr)
    return spmm_raw(gs.rowptr, gs.eid, gs.w, ea, gs.num_nodes, table_rows=gs.num_edges, long_rows=gs.long_rows())


# ----------------------------------------------------------------------------- K4 attention
_DEVICE_CONSTANTS: "collections.OrderedDict" = collections.OrderedDict()
_DEVICE_CONSTANTS_MAX = 4096
_CONSTANT_SINKS: list = []      # lists that collect every constant handed out while a step is being recorded


def device_constant(values, dtype: torch.dtype, device) -> torch.Tensor:
    """Small host-known constant (graph offsets, per-graph sizes) as a device tensor, cached by value: the same
    batch layout comes back every step, and a cached tensor means no host-to-device copy inside the step -- which a
    HIP graph capture of the step could not record.

    The cache is an LRU (a mixed-size stream brings a new layout with every batch).  Eviction only drops the cache's own
    reference: whoever bakes a constant's ADDRESS into something that outlives the call -- a recorded HIP graph -- holds its
    own reference through ``collect_device_constants`` (training.GraphedPretrainStep does), so an evicted entry that a graph
    still reads stays allocated."""
    key = (tuple(values), dtype, str(device))
    t = _DEVICE_CONSTANTS.get(key)
    if t is None:
        t = _DEVICE_CONSTANTS[key] = torch.tensor(list(values), dtype=dtype).to(device)
        while len(_DEVICE_CONSTANTS) > _DEVICE_CONSTANTS_MAX:
            _DEVICE_CONSTANTS.popitem(last=False)
    else:
        _DEVICE_CONSTANTS.move_to_end(key)
    for sink in _CONSTANT_SINKS:
        sink.append(t)
    return t


class collect_device_constants:
    """Context manager: ``with collect_device_constants() as held:`` appends every constant ``device_constant`` returns inside
    the block to ``held`` (a list the caller keeps for as long as it needs the addresses to stay valid)."""

    def __init__(self):
        self.held: list = []

    def __enter__(self):
        _CONSTANT_SINKS.append(self.held)
        return self.held

    def __exit__(self, *exc):
        _CONSTANT_SINKS.remove(self.held)
        return False


class AttnPlan:
    """Per-batch descriptor of the variable-length attention launch: device ``ptr`` (int32 [B+1]),
    tile counts.  Built once per batch from host-side graph offsets (no device sync)."""

    __slots__ = ("ptr_host", "ptr_dev", "B", "N_tot", "num_q_tiles")

    def __init__(self, ptr_host, device):
        lib = _lib.load()
        self.ptr_host = [int(v) for v in ptr_host]
        self.B = len(self.ptr_host) - 1
        self.N_tot = self.ptr_host[-1]
        if max((self.ptr_host[g + 1] - self.ptr_host[g] for g in range(self.B)), default=0) >= 1 << 24:
            raise _lib.DGDMKernelError("graphs of 2^24 nodes or more are not supported (24-bit index arithmetic in the attention kernels)")
        qb = lib.dgdm_spatial_attn_q_tile_rows()
        self.num_q_tiles = sum((self.ptr_host[g + 1] - self.ptr_host[g] + qb - 1) // qb for g in range(self.B))
        self.ptr_dev = device_constant(self.ptr_host, torch.int32, device)


def spatial_attn_fwd_raw(q, k, v, pos, plan: AttnPlan, H: int, scale: float, inv_tau: float, variant: int = 0,
                         drop_p: float = 0.0, seed: int = 0):
    """q,k,v: [N_tot, H*16] views sharing one row stride (e.g. slices of a fused QKV buffer)."""
    _lib.require_cuda(q, k, v, pos)
    lib = _lib.load()
    N, C = q.shape
    assert C == H * 16, "spatial attention kernels are built for head_dim 16"
    assert q.stride(1) == 1 and q.stride(0) == k.stride(0) == v.stride(0) and N == plan.N_tot
    pos = _f32c(pos)
    out = torch.empty(N, C, dtype=torch.float32, device=q.device)
    lse2 = torch.empty(H, N, dtype=torch.float32, device=q.device)
    TIMERS.timed("attn_fwd", lambda: _lib.check(
        lib.dgdm_spatial_attn_fwd_variant(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), pos.data_ptr(),
                                          plan.ptr_dev.data_ptr(), plan.B, plan.num_q_tiles, N, H, scale, inv_tau, drop_p, seed,
                                          out.data_ptr(), out.stride(0), lse2.data_ptr(), variant, _lib.stream_ptr(q.device)),
        "dgdm_spatial_attn_fwd"))
    return out, lse2


LOG2E = 1.4426950408889634


class PackedOperands:
    """Block-aligned fp16 hi+lo row images of `ntensors` column blocks of one fp32 matrix (csrc/attn_h.hpp), plus (forward pack)
    the block-aligned, pre-scaled positions; (backward pack) -delta and 8 - lse2 per row."""

    __slots__ = ("R", "pos_b", "ndelta_b", "nlse_b", "ntensors", "r_stride", "skip_map")

    def r(self, z):
        return self.R[z * self.r_stride:]


# dO is packed as alpha * dO with alpha * max|dO| in (TARGET/2, TARGET] (device-side power of two): with P' = 2^8 P the products
# dS' = P' (keep dP - delta) stay ~4x further from fp16's maximum than round 2's (target 256, P unscaled) did
ATTN_GRAD_TARGET = 0.25
ATTN_BWD_CONCURRENT = False     # dQ pass on a side stream beside the dK/dV pass (tools/bench_with.py A/B switch)
# Round 5 experiment: the weight-gradient GEMMs that are already queued when the backward reaches the attention (diffusion, U-Net,
# out_proj: ~35 of the 50 problems of a step) are launched on a side stream BEHIND the attention backward in host order, so that
# their workgroups fill the slots the attention kernel's last, partial round of workgroups leaves empty (1 280 workgroups on 512 slots).
ATTN_BWD_OVERLAP_DW = False
ATTN_BWD_FUSED = True           # dQ, dK, dV in one key-stationary pass + a fixed-order reduction of the partial dQ tiles (False: two passes)
# bytes of partial-dQ scratch per launch of the one-pass backward.  The scratch grows with N^2 * H / 256 * 64 B (0.8 GB at 4 x 10k nodes x
# 8 heads; 10 GB for one 50k-node graph x 16 heads) and, inside a recorded step, stays in the recording's private pool for good.  Cutting it
# into groups is not free: every launch re-streams all query blocks and ends in its own partial round of workgroups (configs[3]: one
# launch 38 ms, three launches of a 4 GiB budget 3 x 14.3 = 43 ms, 77 -> 86 ms per step).  None = min(16 GiB, a quarter of the memory that
# is free on the device when the backward runs): one launch for every BASELINE configuration on a 288 GB MI355X, smaller groups when
# the card is shared or nearly full.  An int fixes the budget (tests force several groups with it).
ATTN_BWD_FUSED_BUDGET = None
# The (query block, key block) pairs whose weights are exactly 0.0f -- every score more than 200 log2 units below every row maximum, which
# is what raw slide coordinates (pixels) in -distance / temperature produce for all but a band of pairs -- are found from per-block
# bounds (two small launches per attention call) and walked over by the forward and the one-pass backward: the same bits in every
# output, a band's worth of work on real slides; nothing is skipped for positions in [0, 1) (csrc/attn_h.hpp, attn_skip.hip).
ATTN_SKIP_ZERO_BLOCKS = True
ATTN_ZERO_MARGIN_LOG2 = 200.0     # csrc/attn_h.hpp::ATTN_ZERO_MARGIN


def attn_zero_blocks_possible(pos_extent: Optional[float], inv_tau: float) -> bool:
    """Can ANY (query block, key block) pair of a batch be all-zero?  A pair is only marked when its distance term alone pushes every
    score 200 log2 units below the row maximum (csrc/attn_skip.hip::pair_is_zero), i.e. when two nodes of one graph are at least
    200 / (inv_tau log2 e) apart.  ``pos_extent`` -- an upper bound of the coordinate range inside any graph, known on the HOST
    (GraphData.host_pos_extent: loaders set it before the upload) -- decides that without a device sync: on BASELINE's U[0,1)^2
    positions no pair can be marked, and the attention then runs without the map's two launches and without the kernels' walk over
    it (12.78 against 12.83 ms per step).  Unknown extent (None): the map is built, as before."""
    if not ATTN_SKIP_ZERO_BLOCKS:
        return False
    if pos_extent is None:
        return True
    return math.sqrt(2.0) * float(pos_extent) * float(inv_tau) * LOG2E >= ATTN_ZERO_MARGIN_LOG2


_ATTN_BUDGET: dict = {}
ATTN_BWD_GROUPS_LAST = 0      # launches the last one-pass attention backward was cut into (bench.py reports it)


def _attn_bwd_budget(device) -> int:
    """Scratch budget of the one-pass attention backward's partial-dQ tiles: min(16 GiB, a quarter of the memory available to this
    process: free + torch's unused cache), taken ONCE per device and kept (the free-memory figure of every call shrank while the allocator warmed up,
    so the group count -- and with it the association of the dQ sum -- depended on the moment; one driver query per eager step went
    with it).  ``ops.reset_attn_bwd_budget()`` forgets it (after freeing a large model, say); an int in ``ATTN_BWD_FUSED_BUDGET`` fixes it."""
    if ATTN_BWD_FUSED_BUDGET is not None:
        return int(ATTN_BWD_FUSED_BUDGET)
    key = device.index if device.index is not None else torch.cuda.current_device()
    b = _ATTN_BUDGET.get(key)
    if b is None:
        if torch.cuda.is_current_stream_capturing():      # no runtime query inside a capture (the eager warm-up steps come first)
            return 16 << 30
        free, _total = torch.cuda.mem_get_info(key)      # what the driver can still hand out + what torch holds cached but unused
        avail = free + torch.cuda.memory_reserved(key) - torch.cuda.memory_allocated(key)
        _ATTN_BUDGET[key] = b = max(64 << 20, min(16 << 30, avail // 4))
    return b


def reset_attn_bwd_budget() -> None:
    _ATTN_BUDGET.clear()


_SIDE_STREAMS: dict = {}


def _side_stream(device) -> "torch.cuda.Stream":
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _SIDE_STREAMS.get(idx)
    if s is None:
        s = _SIDE_STREAMS[idx] = torch.cuda.Stream(device)
    return s


def attn_pack(x, col0: int, cstride: int, ntensors: int, scale0: float, plan: AttnPlan, H: int, pos=None, pos_scale: float = 1.0, O=None,
              scale_dev=None, lse2_b=None) -> PackedOperands:
    lib = _lib.load()
    dev, nb = x.device, plan.num_q_tiles
    pk = PackedOperands()
    pk.ntensors = ntensors
    pk.skip_map = None
    pk.r_stride = lib.dgdm_attn_pack_bytes(nb, H, 0) // 2
    pk.R = torch.empty(max(ntensors * pk.r_stride, 8), dtype=torch.float16, device=dev)
    pk.pos_b = torch.empty(max(lib.dgdm_attn_pack_bytes(nb, H, 2) // 4, 4), dtype=torch.float32, device=dev) if pos is not None else None
    nrow = max(lib.dgdm_attn_pack_bytes(nb, H, 3) // 4, 4)
    pk.ndelta_b = torch.empty(nrow, dtype=torch.float32, device=dev) if O is not None else None
    pk.nlse_b = torch.empty(nrow, dtype=torch.float32, device=dev) if lse2_b is not None else None
    _lib.check(lib.dgdm_attn_pack(x.data_ptr(), x.stride(0), col0, cstride, ntensors, scale0, _lib.ptr(scale_dev),
                                  plan.ptr_dev.data_ptr(), plan.B, nb, H,
                                  pk.R.data_ptr(), _lib.ptr(pos), pos_scale, _lib.ptr(pk.pos_b), _lib.ptr(O),
                                  O.stride(0) if O is not None else 0, _lib.ptr(pk.ndelta_b), _lib.ptr(lse2_b), _lib.ptr(pk.nlse_b),
                                  _lib.stream_ptr(dev)), "dgdm_attn_pack")
    return pk


def spatial_attn_h_fwd_raw(qkv, pos, plan: AttnPlan, H: int, scale: float, inv_tau: float, drop_p: float = 0.0, seed: int = 0,
                           packed: Optional[PackedOperands] = None, variant: int = 0, zero_blocks: Optional[bool] = None):
    """Split-fp16 forward over a fused [N, 3*H*16] QKV buffer; returns (out, lse2_b, packed).  ``zero_blocks``: build and use the
    zero-block map (None: ``ATTN_SKIP_ZERO_BLOCKS``; callers that know the positions' extent pass attn_zero_blocks_possible(...))."""
    lib = _lib.load()
    N, C = qkv.size(0), H * 16
    pk = packed if packed is not None else attn_pack(qkv, 0, C, 3, scale * LOG2E, plan, H, pos=pos, pos_scale=inv_tau * LOG2E)
    out = torch.empty(N, C, dtype=torch.float32, device=qkv.device)
    lse2_b = torch.empty(max(lib.dgdm_attn_pack_bytes(plan.num_q_tiles, H, 3) // 4, 4), dtype=torch.float32, device=qkv.device)
    if (ATTN_SKIP_ZERO_BLOCKS if zero_blocks is None else zero_blocks) and pk.skip_map is None and plan.num_q_tiles > 0:
        pk.skip_map = attn_skip_map(pk, plan, H)
    if ATTN_SKIP_MAP_SINK is not None:
        ATTN_SKIP_MAP_SINK.append((pk.skip_map, plan, H))
    slot = new_amax_slot(qkv.device)        # max |O| from the kernel itself: the output projection needs no reduction launch
    TIMERS.timed("attn_fwd", lambda: _lib.check(
        lib.dgdm_spatial_attn_h_fwd_sparse(pk.r(0).data_ptr(), pk.r(1).data_ptr(), pk.r(2).data_ptr(), pk.pos_b.data_ptr(),
                                           plan.ptr_dev.data_ptr(), plan.B, plan.num_q_tiles, H, drop_p, seed, out.data_ptr(),
                                           out.stride(0), lse2_b.data_ptr(), variant, _lib.ptr(pk.skip_map), slot, _lib.stream_ptr(qkv.device)),
        "dgdm_spatial_attn_h_fwd_sparse"))
    tag_amax(out, slot)
    return out, lse2_b, pk


def attn_skip_map(pk: PackedOperands, plan: AttnPlan, H: int) -> torch.Tensor:
    """The zero-block map of one attention call (csrc/attn_skip.hip) from its packed Q', K and positions: uint32 words, rows by query
    block and by key super-block.  The forward, the one-pass backward and its reduction must see the same map."""
    lib = _lib.load()
    dev, nb = pk.R.device, plan.num_q_tiles
    m = torch.empty(max(lib.dgdm_attn_skip_map_bytes(nb, H) // 4, 4), dtype=torch.int32, device=dev)
    ws = torch.empty(max(lib.dgdm_attn_skip_map_workspace_bytes(nb, H) // 4, 4), dtype=torch.float32, device=dev)
    _lib.check(lib.dgdm_attn_skip_map_build(pk.r(0).data_ptr(), pk.r(1).data_ptr(), pk.pos_b.data_ptr(), plan.ptr_dev.data_ptr(), plan.B, nb, H,
                                            ws.data_ptr(), ws.numel() * 4, m.data_ptr(), m.numel() * 4, _lib.stream_ptr(dev)),
               "dgdm_attn_skip_map_build")
    return m


ATTN_SKIP_MAP_SINK: Optional[list] = None      # measurement hook (bench.py): when a list, every forward appends (map or None, plan, H)


def attn_skip_live_scores(skip_map: Optional[torch.Tensor], plan: AttnPlan, H: int):
    """(scores the forward evaluates, scores the one-pass backward evaluates, all scores = sum n_g^2 H) under ``skip_map`` -- one device
    reduction over the map and ONE host read: measurement only, never on the training path."""
    total = sum((plan.ptr_host[g + 1] - plan.ptr_host[g]) ** 2 for g in range(plan.B)) * H
    if skip_map is None or plan.num_q_tiles == 0:
        return total, total, total
    counts = torch.zeros(3, dtype=torch.int64, device=skip_map.device)
    _lib.check(_lib.load().dgdm_attn_skip_map_count(skip_map.data_ptr(), plan.ptr_dev.data_ptr(), plan.B, plan.num_q_tiles, H, counts.data_ptr(),
                                                    _lib.stream_ptr(skip_map.device)), "dgdm_attn_skip_map_count")
    f, b, t = (int(v) for v in counts.tolist())
    assert t == total, (t, total)
    return f, b, t


def spatial_attn_h_bwd_raw(pk: PackedOperands, out, gout, plan: AttnPlan, H: int, scale: float, inv_tau: float, lse2_b, dqkv,
                           drop_p: float = 0.0, seed: int = 0, dq_variant: int = 0, dkv_variant: int = 0):
    """Split-fp16 backward: packs dO (+ -delta, 8 - lse2), then the dQ pass and the dK/dV pass.  ``inv_tau`` is already in the
    packed positions (kept in the signature for symmetry with the fp32 twin)."""
    lib = _lib.load()
    C = H * 16
    gout = _f32c(gout)
    st = _lib.stream_ptr(out.device)
    # fp16 range guard (device side): dO is packed as alpha*dO, results are multiplied by 1/alpha
    gs = torch.empty(2, dtype=torch.float32, device=out.device)
    wsb = lib.dgdm_amax_scale_workspace_bytes()
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=out.device)
    _lib.check(lib.dgdm_amax_pow2_scale(gout.data_ptr(), gout.numel(), ATTN_GRAD_TARGET, gs.data_ptr(), ws.data_ptr(), wsb, st), "dgdm_amax_pow2_scale")
    gk = attn_pack(gout, 0, C, 1, 1.0, plan, H, O=out, scale_dev=gs, lse2_b=lse2_b)
    def run_dq(stream):
        _lib.check(
            lib.dgdm_spatial_attn_h_bwd_dq(pk.r(0).data_ptr(), pk.r(1).data_ptr(), pk.r(2).data_ptr(), gk.r(0).data_ptr(),
                                           pk.pos_b.data_ptr(), gk.nlse_b.data_ptr(), gk.ndelta_b.data_ptr(), plan.ptr_dev.data_ptr(), plan.B,
                                           plan.num_q_tiles, H, scale, drop_p, seed, gs.data_ptr(), dqkv[:, :C].data_ptr(), dqkv.stride(0), dq_variant,
                                           stream), "dgdm_spatial_attn_h_bwd_dq")

    def run_dkv(stream):
        _lib.check(
            lib.dgdm_spatial_attn_h_bwd_dkv(pk.r(0).data_ptr(), pk.r(1).data_ptr(), pk.r(2).data_ptr(), gk.r(0).data_ptr(),
                                            pk.pos_b.data_ptr(), gk.nlse_b.data_ptr(), gk.ndelta_b.data_ptr(),
                                            plan.ptr_dev.data_ptr(), plan.B, plan.num_q_tiles, H, drop_p, seed, gs.data_ptr(),
                                            dqkv[:, C:2 * C].data_ptr(), dqkv[:, 2 * C:].data_ptr(), dqkv.stride(0), dkv_variant, stream),
            "dgdm_spatial_attn_h_bwd_dkv")
    overlap_ev = None
    if ATTN_BWD_FUSED and ATTN_BWD_OVERLAP_DW and _PENDING_TN and not TIMERS.enabled:
        overlap_ev = torch.cuda.Event()
        overlap_ev.record(torch.cuda.current_stream(out.device))       # everything the queued dW GEMMs read exists at this point
    if ATTN_BWD_FUSED:
        # one pass for dQ, dK, dV (csrc/attn_h_bwd_fused.hip): key super-blocks in groups whose partial-dQ scratch stays within the budget
        import ctypes
        ph = (ctypes.c_int32 * (plan.B + 1))(*plan.ptr_host)
        nsb, sb = lib.dgdm_spatial_attn_h_bwd_fused_superblocks(ph, plan.B), 0
        if nsb == 0:              # no keys at all (an empty batch / only empty graphs): nothing to launch, no gradient
            return dqkv.zero_()
        budget = _attn_bwd_budget(out.device)
        groups = []
        while sb < nsb:
            cnt = nsb - sb
            while cnt > 1 and lib.dgdm_spatial_attn_h_bwd_fused_workspace_bytes(ph, plan.B, H, sb, cnt) > budget:
                cnt = (cnt + 1) // 2
            groups.append((sb, cnt, lib.dgdm_spatial_attn_h_bwd_fused_workspace_bytes(ph, plan.B, H, sb, cnt)))
            sb += cnt
        global ATTN_BWD_GROUPS_LAST
        ATTN_BWD_GROUPS_LAST = len(groups)
        ws = torch.empty(max(max(g[2] for g in groups), 16) // 4, dtype=torch.float32, device=out.device)
        slot = new_amax_slot(out.device)    # max |dQ|, |dK|, |dV| from the kernels that write them (dqkv is the QKV projection's operand)

        for sb0, cnt, wsb in groups:
            TIMERS.timed("attn_bwd_fused", lambda: _lib.check(lib.dgdm_spatial_attn_h_bwd_fused_sparse(
                pk.r(0).data_ptr(), pk.r(1).data_ptr(), pk.r(2).data_ptr(), gk.r(0).data_ptr(), pk.pos_b.data_ptr(), gk.nlse_b.data_ptr(),
                gk.ndelta_b.data_ptr(), plan.ptr_dev.data_ptr(), ph, plan.B, plan.num_q_tiles, H, drop_p, seed, gs.data_ptr(),
                dqkv[:, C:2 * C].data_ptr(), dqkv[:, 2 * C:].data_ptr(), dqkv.stride(0), sb0, cnt, ws.data_ptr(), ws.numel() * 4,
                _lib.ptr(pk.skip_map), slot, st), "dgdm_spatial_attn_h_bwd_fused_sparse"))
            TIMERS.timed("attn_bwd_dq_reduce", lambda: _lib.check(lib.dgdm_spatial_attn_h_bwd_fused_reduce_sparse(
                plan.ptr_dev.data_ptr(), ph, plan.B, plan.num_q_tiles, H, scale, gs.data_ptr(), dqkv[:, :C].data_ptr(), dqkv.stride(0), sb0, cnt,
                ws.data_ptr(), ws.numel() * 4, _lib.ptr(pk.skip_map), slot, st), "dgdm_spatial_attn_h_bwd_fused_reduce_sparse"))
        tag_amax(dqkv, slot)
        if overlap_ev is not None:
            cur, side = torch.cuda.current_stream(out.device), _side_stream(out.device)
            side.wait_event(overlap_ev)
            with torch.cuda.stream(side):
                held = flush_deferred_tn(on_stream=_lib.stream_ptr(out.device))
            cur.wait_stream(side)          # join at once: whatever follows is ordered behind the dW launches (and may reuse their memory)
            del held
    elif ATTN_BWD_CONCURRENT and not TIMERS.enabled:
        # the two passes are independent (dQ | dK, dV: disjoint columns of dqkv) and each leaves the chip partly empty in its last
        # round of workgroups (1256 / 2512 workgroups on 512 / 768 resident slots): side by side the one fills the other's tail.
        # One fork / join pair; inside a recording the side stream joins the capture through the two waits.
        cur, side = torch.cuda.current_stream(out.device), _side_stream(out.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            run_dq(_lib.stream_ptr(out.device))
        run_dkv(st)
        cur.wait_stream(side)
    else:
        TIMERS.timed("attn_bwd_dq", lambda: run_dq(st))
        TIMERS.timed("attn_bwd_dkv", lambda: run_dkv(st))
    return dqkv


class _SpatialAttentionH(torch.autograd.Function):
    """Split-fp16 version of _SpatialAttention (same math; products on the 16-bit matrix pipe with EVERY operand -- Q', K, V,
    dO, the probabilities and dS -- carried as fp16 hi+lo pairs, fp32 accumulation).  Default attention path: ~2x faster than the
    fp32-MFMA kernels on gfx950 at the same error against float64 (tests/test_hip_attention.py)."""

    @staticmethod
    def forward(ctx, qkv, pos, plan: AttnPlan, H: int, scale: float, inv_tau: float, drop_p: float, seed: int, zero_blocks=None):
        qkv, pos = _f32c(qkv), _f32c(pos)
        out, lse2_b, pk = spatial_attn_h_fwd_raw(qkv, pos, plan, H, scale, inv_tau, drop_p, seed, zero_blocks=zero_blocks)
        ctx.save_for_backward(out, lse2_b, pk.R, pk.pos_b, *([pk.skip_map] if pk.skip_map is not None else []))
        ctx.meta = (plan, H, scale, inv_tau, drop_p, seed, pk.r_stride, qkv.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        out, lse2_b, R, pos_b, *skip = ctx.saved_tensors
        plan, H, scale, inv_tau, drop_p, seed, rs, shape = ctx.meta
        pk = PackedOperands()
        pk.R, pk.pos_b, pk.ndelta_b, pk.nlse_b, pk.ntensors, pk.r_stride = R, pos_b, None, None, 3, rs
        pk.skip_map = skip[0] if skip else None
        dqkv = torch.empty(shape, dtype=torch.float32, device=out.device)
        spatial_attn_h_bwd_raw(pk, out, gout, plan, H, scale, inv_tau, lse2_b, dqkv, drop_p, seed)
        return dqkv, None, None, None, None, None, None, None, None


def unblock_rows(xb: torch.Tensor, plan: AttnPlan, H: int) -> torch.Tensor:
    """[blk][H][64] block layout -> [H, N_tot] (test/diagnostic helper)."""
    v = xb[: plan.num_q_tiles * H * 64].view(plan.num_q_tiles, H, 64)
    outs, blk = [], 0
    for g in range(plan.B):
        n = plan.ptr_host[g + 1] - plan.ptr_host[g]
        nb = (n + 63) // 64
        outs.append(v[blk:blk + nb].permute(1, 0, 2).reshape(H, nb * 64)[:, :n])
        blk += nb
    return torch.cat(outs, dim=1)


def spatial_attn_bwd_raw(q, k, v, out, gout, pos, plan: AttnPlan, H: int, scale: float, inv_tau: float, lse2, dqkv,
                         drop_p: float = 0.0, seed: int = 0):
    """Writes dQ|dK|dV into the three column blocks of ``dqkv`` [N_tot, 3*H*16]."""
    lib = _lib.load()
    N, C = q.shape
    gout = _f32c(gout)
    delta = torch.empty(H, N, dtype=torch.float32, device=q.device)
    assert out.stride(0) == gout.stride(0) and dqkv.stride(1) == 1
    st = _lib.stream_ptr(q.device)
    TIMERS.timed("attn_bwd_dq", lambda: _lib.check(
        lib.dgdm_spatial_attn_bwd_dq(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), out.data_ptr(), gout.data_ptr(),
                                     out.stride(0), pos.data_ptr(), plan.ptr_dev.data_ptr(), plan.B, plan.num_q_tiles, N, H, scale,
                                     inv_tau, lse2.data_ptr(), drop_p, seed, dqkv[:, :C].data_ptr(), dqkv.stride(0), delta.data_ptr(), st),
        "dgdm_spatial_attn_bwd_dq"))
    TIMERS.timed("attn_bwd_dkv", lambda: _lib.check(
        lib.dgdm_spatial_attn_bwd_dkv(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), gout.data_ptr(), out.stride(0),
                                      pos.data_ptr(), plan.ptr_dev.data_ptr(), plan.B, plan.num_q_tiles, N, H, scale, inv_tau,
                                      lse2.data_ptr(), delta.data_ptr(), drop_p, seed, dqkv[:, C:2 * C].data_ptr(), dqkv[:, 2 * C:].data_ptr(),
                                      dqkv.stride(0), st), "dgdm_spatial_attn_bwd_dkv"))
    return dqkv


class _SpatialAttention(torch.autograd.Function):
    """softmax(QK^T/sqrt(d) - dist/tau) V over a fused [N_tot, 3C] QKV buffer
    (core/attention.py:135-157 + 261-283), per graph of the batch."""

    @staticmethod
    def forward(ctx, qkv, pos, plan: AttnPlan, H: int, scale: float, inv_tau: float, drop_p: float, seed: int):
        qkv = _f32c(qkv)
        C = qkv.size(1) // 3
        pos = _f32c(pos)
        out, lse2 = spatial_attn_fwd_raw(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], pos, plan, H, scale, inv_tau, 0, drop_p, seed)
        ctx.save_for_backward(qkv, out, lse2, pos)
        ctx.meta = (plan, H, scale, inv_tau, drop_p, seed)
        return out

    @staticmethod
    def backward(ctx, gout):
        qkv, out, lse2, pos = ctx.saved_tensors
        plan, H, scale, inv_tau, drop_p, seed = ctx.meta
        C = qkv.size(1) // 3
        dqkv = torch.empty_like(qkv)
        spatial_attn_bwd_raw(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, gout, pos, plan, H, scale, inv_tau, lse2, dqkv,
                             drop_p, seed)
        return dqkv, None, None, None, None, None, None, None


# "fp16x2": split-fp16 (hi+lo operands, fp32 accumulate) attention kernels -- the shipped path; "fp32": the same tiling on the
# fp32 matrix instructions (the reference's own arithmetic; parity tests run both, bench.py reports both).  Set with
# ``configure(attention=...)``; no environment variable selects kernels.
ATTN_PRECISION = "fp16x2"


ATTN_HEAD_DIMS = (16, 32, 64, 128)     # head widths the attention kernels take (narrower heads are zero-padded to the next one by the caller)


def _dense_groups(plan: AttnPlan):
    """(first graph, graphs, rows per graph) runs of the batch the dense kernels take in one launch each: the whole batch when its graphs
    have one size, graph by graph otherwise (empty graphs dropped)."""
    sizes = [plan.ptr_host[g + 1] - plan.ptr_host[g] for g in range(plan.B)]
    if plan.B > 0 and len(set(sizes)) == 1:
        return [(0, plan.B, sizes[0])] if sizes[0] > 0 else []
    return [(g, 1, n) for g, n in enumerate(sizes) if n > 0]


def _spatial_attention_dense(qkv, pos, plan: AttnPlan, H: int, D: int, scale: float, inv_tau: float, p: float, seed: int):
    """Spatial attention at head_dim 128 (hidden_dims[-1] = 128 with ONE head, 256 with two ...: valid in the reference,
    core/attention.py:36-40): the dense kernels of csrc/attn_dense.hip (two lanes per row) with the positions as their spatial bias,
    one launch per run of equal-sized graphs.  Correct, not fast -- no default configuration comes here."""
    C = H * D
    pos = _f32c(pos)
    outs = []
    for g0, b, n in _dense_groups(plan):
        r0, r1 = plan.ptr_host[g0], plan.ptr_host[g0] + b * n
        sl, ps = qkv[r0:r1], pos[r0:r1]
        m = DenseMask(None, None, b, H, n, n, qkv.device, posq=ps, posk=ps, inv_tau=inv_tau)
        o, _, _ = attn_dense(sl[:, :C], sl[:, C:2 * C], sl[:, 2 * C:], b, n, n, H, scale, m, p, p > 0, seed=(seed + 0x9E3779B1 * g0) & 0xFFFFFFFF)
        outs.append(o)
    if not outs:
        return qkv[:, :C] * 0.0
    return outs[0] if len(outs) == 1 else torch.cat(outs)


class _SpatialAttentionGen(torch.autograd.Function):
    """Spatial attention for head dims 32 / 64 (csrc/attn_gen.hip: fp32 on the vector units; the MFMA kernels are tiled for 16)."""

    @staticmethod
    def forward(ctx, qkv, pos, plan: AttnPlan, H: int, D: int, scale: float, inv_tau: float, drop_p: float, seed: int):
        lib = _lib.load()
        qkv, pos = _f32c(qkv), _f32c(pos)
        _lib.require_cuda(qkv, pos)
        C = H * D
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        out = torch.empty(qkv.size(0), C, dtype=torch.float32, device=qkv.device)
        lse = torch.empty(H, max(qkv.size(0), 1), dtype=torch.float32, device=qkv.device)
        _lib.check(lib.dgdm_spatial_attn_gen_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), qkv.stride(0), pos.data_ptr(), plan.ptr_dev.data_ptr(),
                                                 plan.B, plan.num_q_tiles, plan.N_tot, H, D, scale, inv_tau, drop_p, seed, out.data_ptr(), C,
                                                 lse.data_ptr(), _lib.stream_ptr(qkv.device)), "dgdm_spatial_attn_gen_fwd")
        ctx.save_for_backward(qkv, pos, out, lse)
        ctx.meta = (plan, H, D, scale, inv_tau, drop_p, seed)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        qkv, pos, out, lse = ctx.saved_tensors
        plan, H, D, scale, inv_tau, drop_p, seed = ctx.meta
        C = H * D
        gout = _f32c(gout)
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        _lib.check(lib.dgdm_spatial_attn_gen_bwd(qkv[:, :C].data_ptr(), qkv[:, C:2 * C].data_ptr(), qkv[:, 2 * C:].data_ptr(), qkv.stride(0),
                                                 pos.data_ptr(), plan.ptr_dev.data_ptr(), plan.B, plan.num_q_tiles, plan.N_tot, H, D, scale, inv_tau,
                                                 drop_p, seed, out.data_ptr(), gout.data_ptr(), C, lse.data_ptr(), delta.data_ptr(),
                                                 dqkv[:, :C].data_ptr(), dqkv[:, C:2 * C].data_ptr(), dqkv[:, 2 * C:].data_ptr(), dqkv.stride(0),
                                                 _lib.stream_ptr(qkv.device)), "dgdm_spatial_attn_gen_bwd")
        return dqkv, None, None, None, None, None, None, None, None


def spatial_attention(qkv, pos, plan: AttnPlan, H: int, scale: float, inv_tau: float = 1.0, drop_p: float = 0.0,
                      training: bool = False, seed: Optional[int] = None, pos_extent: Optional[float] = None):
    """dropout(softmax(QK^T*scale - dist*inv_tau)) V per graph; ``drop_p`` applies to the attention
    weights (core/attention.py:154) in training mode.  ``qkv`` [N, 3 * H * D] with D in ``ATTN_HEAD_DIMS``: 16 runs on the MFMA
    kernels (``ATTN_PRECISION``), 32 / 64 on the vector-unit kernels of csrc/attn_gen.hip."""
    p = float(drop_p) if training else 0.0
    if p > 0 and seed is None:
        seed = next_dropout_seed()
    D = qkv.size(1) // (3 * H)
    if qkv.size(1) != 3 * H * D or D not in ATTN_HEAD_DIMS:
        raise _lib.DGDMKernelError(f"spatial attention kernels take head dims {ATTN_HEAD_DIMS} (pad narrower heads with zeros), got {qkv.size(1)} / (3 x {H})")
    if D == 128:
        return _spatial_attention_dense(qkv, pos, plan, H, D, scale, inv_tau, p, seed or 0)
    if D != 16:
        return _SpatialAttentionGen.apply(qkv, pos, plan, H, D, scale, inv_tau, p, seed or 0)
    if ATTN_PRECISION == "fp32":
        return _SpatialAttention.apply(qkv, pos, plan, H, scale, inv_tau, p, seed or 0)
    return _SpatialAttentionH.apply(qkv, pos, plan, H, scale, inv_tau, p, seed or 0, attn_zero_blocks_possible(pos_extent, inv_tau))


# ----------------------------------------------------------------------------- K4-dense: MultiHeadAttention.forward as the reference exposes it
ATTN_DENSE_HEAD_DIMS = (16, 32, 64, 128)      # csrc/attn_dense.hip (narrower heads are zero-padded to the next one by the caller)


class DenseMask:
    """What the reference adds to its [B, H, Lq, Lk] score tensor before the softmax (core/attention.py:129-142,311-314), as the
    kernels of csrc/attn_dense.hip take it: ``attn_mask`` through the strides of its broadcast view (float: added; bool: -inf where
    True), ``key_padding_mask`` [B, Lk] bool, optional positions for the spatial bias.  Holds the tensors it points into."""

    __slots__ = ("bias", "bmask", "strides", "kpm", "posq", "posk", "inv_tau")

    def __init__(self, attn_mask, key_padding_mask, B: int, H: int, Lq: int, Lk: int, device, posq=None, posk=None, inv_tau: float = 0.0):
        self.bias = self.bmask = self.kpm = None
        self.strides = (0, 0, 0, 0)
        if attn_mask is not None:
            _lib.require_cuda(attn_mask)
            m = attn_mask
            if m.dtype == torch.bool:
                m = m.view(torch.uint8)
            elif m.dtype != torch.float32:
                m = m.float()              # attn_scores += attn_mask keeps the scores' fp32 (attention.py:135)
            # the reference's in-place ops broadcast the mask against [B, H, Lq, Lk]; expand() raises where they would
            view = m.expand(B, H, Lq, Lk)
            self.strides = tuple(int(v) for v in view.stride())
            if attn_mask.dtype == torch.bool:
                self.bmask = m
            else:
                self.bias = m
        if key_padding_mask is not None:
            _lib.require_cuda(key_padding_mask)
            if key_padding_mask.dtype != torch.bool:
                raise RuntimeError("masked_fill_ only supports boolean masks, but got mask with dtype %s" % key_padding_mask.dtype)
            if tuple(key_padding_mask.shape) != (B, Lk):
                raise RuntimeError(f"key_padding_mask must be [batch_size, key_len] = [{B}, {Lk}], got {tuple(key_padding_mask.shape)}")
            self.kpm = key_padding_mask.contiguous().view(torch.uint8)
        self.posq = None if posq is None else _f32c(posq)
        self.posk = None if posk is None else _f32c(posk)
        self.inv_tau = float(inv_tau)

    def args(self):
        return (_lib.ptr(self.bias), _lib.ptr(self.bmask), *self.strides, _lib.ptr(self.kpm), _lib.ptr(self.posq), _lib.ptr(self.posk), self.inv_tau)


def _dense_rows(t: torch.Tensor) -> torch.Tensor:
    """A 2-D fp32 row view the dense kernels can address (unit column stride, row stride % 4 == 0, 16-byte aligned), or a copy."""
    if t.dtype != torch.float32 or t.dim() != 2:
        raise _lib.DGDMKernelError(f"dense attention takes 2-D fp32 row matrices, got {t.dtype} {tuple(t.shape)}")
    if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
    return t


class _AttnDense(torch.autograd.Function):
    """dropout(softmax(Q K^T scale + mask)) V for a dense batch (csrc/attn_dense.hip).  q [B * Lq, H * D], k / v [B * Lk, H * D] (row
    views with unit column stride; k and v share a row stride or are made contiguous).  Returns (O [B * Lq, H * D], lse [B, H, Lq]);
    the mask gets no gradient."""

    @staticmethod
    def forward(ctx, q, k, v, B: int, Lq: int, Lk: int, H: int, D: int, scale: float, mask: DenseMask, drop_p: float, seed: int):
        lib = _lib.load()
        _lib.require_cuda(q, k, v)
        q, k, v = _dense_rows(q), _dense_rows(k), _dense_rows(v)
        if k.stride(0) != v.stride(0):
            k, v = k.contiguous(), v.contiguous()
        C = H * D
        out = torch.empty(B * Lq, C, dtype=torch.float32, device=q.device)
        lse = torch.empty(B, H, max(Lq, 1), dtype=torch.float32, device=q.device)
        _lib.check(lib.dgdm_attn_dense_fwd(q.data_ptr(), q.stride(0), k.data_ptr(), v.data_ptr(), k.stride(0), B, Lq, Lk, H, D, scale, *mask.args(),
                                           drop_p, seed, out.data_ptr(), C, lse.data_ptr(), _lib.stream_ptr(q.device)), "dgdm_attn_dense_fwd")
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.meta = (B, Lq, Lk, H, D, scale, mask, drop_p, seed)
        ctx.mark_non_differentiable(lse)
        return out, lse

    @staticmethod
    def backward(ctx, gout, _glse=None):
        lib = _lib.load()
        q, k, v, out, lse = ctx.saved_tensors
        B, Lq, Lk, H, D, scale, mask, drop_p, seed = ctx.meta
        C = H * D
        gout = _f32c(gout)
        dq = torch.empty(B * Lq, C, dtype=torch.float32, device=q.device)
        dk = torch.empty(B * Lk, C, dtype=torch.float32, device=q.device)
        dv = torch.empty_like(dk)
        delta = torch.empty_like(lse)
        _lib.check(lib.dgdm_attn_dense_bwd(q.data_ptr(), q.stride(0), k.data_ptr(), v.data_ptr(), k.stride(0), B, Lq, Lk, H, D, scale, *mask.args(),
                                           drop_p, seed, out.data_ptr(), gout.data_ptr(), C, lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), C,
                                           dk.data_ptr(), dv.data_ptr(), C, _lib.stream_ptr(q.device)), "dgdm_attn_dense_bwd")
        return dq, dk, dv, None, None, None, None, None, None, None, None, None


def attn_dense(q, k, v, B: int, Lq: int, Lk: int, H: int, scale: float, mask: Optional[DenseMask] = None, drop_p: float = 0.0,
               training: bool = False, seed: Optional[int] = None):
    """-> (O, lse, seed): see ``_AttnDense``; ``seed`` is the dropout seed the launch used (``attn_dense_weights`` takes it to return
    the weights the forward applied)."""
    D = q.size(1) // H
    if q.size(1) != H * D or D not in ATTN_DENSE_HEAD_DIMS or k.size(1) != H * D or v.size(1) != H * D:
        raise _lib.DGDMKernelError(f"dense attention kernels take head dims {ATTN_DENSE_HEAD_DIMS} (pad narrower heads with zeros), got "
                                   f"{tuple(q.shape)} / {tuple(k.shape)} / {tuple(v.shape)} at {H} heads")
    if q.size(0) != B * Lq or k.size(0) != B * Lk or v.size(0) != B * Lk:
        raise _lib.DGDMKernelError(f"dense attention: rows {q.size(0)} / {k.size(0)} / {v.size(0)} do not match B = {B}, Lq = {Lq}, Lk = {Lk}")
    if B * Lq > 0 and Lk == 0:
        raise _lib.DGDMKernelError("dense attention needs at least one key")
    p = float(drop_p) if training else 0.0
    if p > 0 and seed is None:
        seed = next_dropout_seed()
    if mask is None:
        mask = DenseMask(None, None, B, H, Lq, Lk, q.device)
    out, lse = _AttnDense.apply(q, k, v, B, Lq, Lk, H, D, scale, mask, p, seed or 0)
    return out, lse, (seed or 0)


def attn_dense_weights(q, k, lse, B: int, Lq: int, Lk: int, H: int, scale: float, mask: Optional[DenseMask] = None, drop_p: float = 0.0,
                       seed: int = 0, per_head: bool = False) -> torch.Tensor:
    """Attention weights after dropout (core/attention.py:145-146): head mean [B, Lq, Lk] or per head [B * H, Lq, Lk] (no grad)."""
    lib = _lib.load()
    with torch.no_grad():
        q, k = _dense_rows(q.detach()), _dense_rows(k.detach())
        D = q.size(1) // H
        if mask is None:
            mask = DenseMask(None, None, B, H, Lq, Lk, q.device)
        W = torch.empty(B * H if per_head else B, Lq, Lk, dtype=torch.float32, device=q.device)
        _lib.check(lib.dgdm_attn_dense_weights(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), B, Lq, Lk, H, D, scale, *mask.args(), drop_p, seed,
                                               lse.data_ptr(), int(per_head), W.data_ptr(), _lib.stream_ptr(q.device)), "dgdm_attn_dense_weights")
    return W


# ----------------------------------------------------------------------------- K5 positional encoding
class _AddPosEnc(torch.autograd.Function):
    """x + sinusoid(pos) (core/attention.py:225-259,306); d/dx = identity, pos carries no grad."""

    @staticmethod
    def forward(ctx, x, pos, plan: AttnPlan):
        return add_posenc_raw(x, pos, plan, x.size(1))

    @staticmethod
    def backward(ctx, g):
        return g, None, None


def add_posenc_raw(x: Optional[torch.Tensor], pos, plan: AttnPlan, C: int) -> torch.Tensor:
    lib = _lib.load()
    pos = _f32c(pos)
    _lib.require_cuda(pos, x)
    if x is not None:
        x = _f32c(x)
    N = pos.size(0)
    out = torch.empty(N, C, dtype=torch.float32, device=pos.device)
    ws = torch.empty(2 * max(plan.B, 1), dtype=torch.float32, device=pos.device)
    slot = new_amax_slot(pos.device)
    _lib.check(lib.dgdm_add_posenc(_lib.ptr(x), x.stride(0) if x is not None else 0, pos.data_ptr(), plan.ptr_dev.data_ptr(),
                                   plan.B, N, C, ws.data_ptr(), out.data_ptr(), out.stride(0), slot, _lib.stream_ptr(pos.device)),
               "dgdm_add_posenc")
    return tag_amax(out, slot)


def add_posenc(x, pos, plan: AttnPlan):
    return _AddPosEnc.apply(x, pos, plan)


def spatial_attention_mean_weights(qkv, pos, plan: AttnPlan, H: int, scale: float, inv_tau: float = 1.0):
    """List of head-mean attention matrices [n_g, n_g], one per graph (no grad)."""
    lib = _lib.load()
    qkv, pos = _f32c(qkv), _f32c(pos)
    C = qkv.size(1) // 3
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    sizes = [plan.ptr_host[g + 1] - plan.ptr_host[g] for g in range(plan.B)]
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + n * n)
    W = torch.empty(max(offs[-1], 1), dtype=torch.float32, device=qkv.device)
    off_dev = device_constant(offs[:-1], torch.int64, qkv.device)
    D = C // H
    if D == 128:          # csrc/attn_dense.hip, as the forward of this head width
        res = [None] * plan.B
        for g0, b, n in _dense_groups(plan):
            r0, r1 = plan.ptr_host[g0], plan.ptr_host[g0] + b * n
            m = DenseMask(None, None, b, H, n, n, qkv.device, posq=pos[r0:r1], posk=pos[r0:r1], inv_tau=inv_tau)
            _, lse, _ = attn_dense(q[r0:r1], k[r0:r1], v[r0:r1], b, n, n, H, scale, m)
            Wd = attn_dense_weights(q[r0:r1], k[r0:r1], lse, b, n, n, H, scale, m)
            for i in range(b):
                res[g0 + i] = Wd[i]
        return [w if w is not None else W[:0].view(0, 0) for w in res]
    if D != 16:           # head dims 32 / 64: csrc/attn_gen.hip (its own forward for the row log-sum-exp)
        if D not in ATTN_HEAD_DIMS:
            raise _lib.DGDMKernelError(f"spatial attention kernels take head dims {ATTN_HEAD_DIMS}, got {D}")
        out = torch.empty(qkv.size(0), C, dtype=torch.float32, device=qkv.device)
        lse = torch.empty(H, max(qkv.size(0), 1), dtype=torch.float32, device=qkv.device)
        st = _lib.stream_ptr(qkv.device)
        _lib.check(lib.dgdm_spatial_attn_gen_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), qkv.stride(0), pos.data_ptr(), plan.ptr_dev.data_ptr(),
                                                 plan.B, plan.num_q_tiles, plan.N_tot, H, D, scale, inv_tau, 0.0, 0, out.data_ptr(), C, lse.data_ptr(), st),
                   "dgdm_spatial_attn_gen_fwd")
        _lib.check(lib.dgdm_spatial_attn_gen_mean_weights(q.data_ptr(), k.data_ptr(), qkv.stride(0), pos.data_ptr(), plan.ptr_dev.data_ptr(), plan.B,
                                                          plan.num_q_tiles, plan.N_tot, H, D, scale, inv_tau, lse.data_ptr(), W.data_ptr(),
                                                          off_dev.data_ptr(), st), "dgdm_spatial_attn_gen_mean_weights")
        return [W[offs[g]:offs[g + 1]].view(sizes[g], sizes[g]) for g in range(plan.B)]
    _, lse2 = spatial_attn_fwd_raw(q, k, v, pos, plan, H, scale, inv_tau)
    _lib.check(lib.dgdm_spatial_attn_mean_weights(q.data_ptr(), k.data_ptr(), q.stride(0), pos.data_ptr(), plan.ptr_dev.data_ptr(),
                                                  plan.B, plan.num_q_tiles, plan.N_tot, H, scale, inv_tau, lse2.data_ptr(),
                                                  W.data_ptr(), off_dev.data_ptr(), _lib.stream_ptr(qkv.device)),
               "dgdm_spatial_attn_mean_weights")
    return [W[offs[g]:offs[g + 1]].view(sizes[g], sizes[g]) for g in range(plan.B)]


# ----------------------------------------------------------------------------- K6/K7 fused row kernels
ACT_NONE, ACT_GELU, ACT_RELU, ACT_SILU, ACT_ELU = 0, 1, 2, 3, 4
_ACT_IDS = {"none": ACT_NONE, "gelu": ACT_GELU, "relu": ACT_RELU, "silu": ACT_SILU, "elu": ACT_ELU}
_seed_counter = 0


def next_dropout_seed() -> int:
    """Host-side counter-based seed (no device sync): reproducible under torch.manual_seed and the
    order of calls; each dropout site of each step gets its own 32-bit stream id."""
    global _seed_counter
    _seed_counter += 1
    return (torch.initial_seed() * 0x9E3779B1 + _seed_counter * 0x85EBCA6B) & 0xFFFFFFFF


def act_id(module_or_name) -> Optional[int]:
    if module_or_name is None:
        return ACT_NONE
    if isinstance(module_or_name, str):
        return _ACT_IDS.get(module_or_name)
    import torch.nn as nn
    if isinstance(module_or_name, nn.GELU):
        return ACT_GELU if getattr(module_or_name, "approximate", "none") == "none" else None
    if isinstance(module_or_name, nn.ReLU):
        return ACT_RELU
    if isinstance(module_or_name, nn.SiLU):
        return ACT_SILU
    if isinstance(module_or_name, nn.ELU):
        return ACT_ELU if module_or_name.alpha == 1.0 else None
    if isinstance(module_or_name, nn.Identity):
        return ACT_NONE
    return None


def row_norm_supported(C: int, groups: int) -> bool:
    return C % groups == 0 and (C // groups) % 4 == 0 and (C // groups) <= 1024


class _RowNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, groups: int, eps: float, act: int, drop_p: float, seed: int):
        lib = _lib.load()
        x = _f32c(x)
        res = _f32c(res) if res is not None else None
        gamma, beta = _f32c(gamma), _f32c(beta)
        _lib.require_cuda(x, res, gamma, beta)
        N, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(N * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        slot = new_amax_slot(x.device)
        _lib.check(lib.dgdm_rownorm_fwd(x.data_ptr(), _lib.ptr(res), gamma.data_ptr(), beta.data_ptr(), N, C, groups, eps, act,
                                        drop_p, seed, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), slot, _lib.stream_ptr(x.device)),
                   "dgdm_rownorm_fwd")
        ctx.save_for_backward(x, res, gamma, beta, mean, rstd)
        ctx.meta = (groups, act, drop_p, seed)
        ctx.leaf = gamma.is_leaf and beta.is_leaf
        ctx.gb = (gamma, beta) if ctx.leaf else None      # the parameter objects themselves (see _claim_deferred)
        return tag_amax(y, slot)

    @staticmethod
    def backward(ctx, gy):
        x, res, gamma, beta, mean, rstd = ctx.saved_tensors
        groups, act, drop_p, seed = ctx.meta
        dx, dg, db = rownorm_bwd_raw(x, res, gamma, beta, mean, rstd, gy, groups, act, drop_p, seed, ctx.gb if ctx.leaf else None)
        return dx, (dx if res is not None else None), dg, db, None, None, None, None, None


def rownorm_bwd_raw(x, res, gamma, beta, mean, rstd, gy, groups: int, act: int, drop_p: float, seed: int, leaf_params=None):
    """Backward of dropout(act(norm_groups(x [+ res]) * gamma + beta)): (dx, dgamma, dbeta); dx also is the residual's gradient.
    ``leaf_params``: the (gamma, beta) PARAMETER objects when their gradients go straight to them -- inside
    ``deferred_weight_grads()`` the column sums of the row partials then join the pass's one reduction launch (as a [1, 2C]
    "weight gradient" split at C) instead of a launch of their own behind every norm."""
    lib = _lib.load()
    gy = _f32c(gy)
    N, C = x.shape
    dx = torch.empty_like(x)
    dg, db = torch.empty_like(gamma), torch.empty_like(beta)
    wsb = _lib.workspace_bytes("dgdm_rownorm_bwd_workspace_bytes", N, C, groups)
    ws = torch.empty(max(wsb, 4) // 4, dtype=torch.float32, device=x.device)
    slot = new_amax_slot(x.device)
    later = False
    if _DEFER_TN and leaf_params is not None and N > 0 and C % 4 == 0 and _claim_deferred(*leaf_params):
        slots = int(_lib.workspace_bytes("dgdm_rownorm_bwd_slots", N, C, groups))       # (memoised call, not a byte count)
        alias = lambda t: t.detach()
        later = slots > 0 and _defer_tn((ws.data_ptr(), dg.data_ptr(), db.data_ptr(), None, C, C, slots, 1, 2 * C, C),
                                        (ws, alias(dg), alias(db)), _lib.stream_ptr(x.device))
    _lib.check(lib.dgdm_rownorm_bwd(x.data_ptr(), _lib.ptr(res), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                    rstd.data_ptr(), gy.data_ptr(), N, C, groups, act, drop_p, seed, dx.data_ptr(),
                                    None if later else dg.data_ptr(), None if later else db.data_ptr(), ws.data_ptr(), wsb, slot,
                                    _lib.stream_ptr(x.device)),
               "dgdm_rownorm_bwd")
    return tag_amax(dx, slot), dg, db


def row_norm(x, weight, bias, *, res=None, groups: int = 1, eps: float = 1e-5, act: int = ACT_NONE, drop_p: float = 0.0,
             training: bool = False):
    """dropout(act(norm_groups(x [+ res]) * weight + bias)) in one HIP kernel."""
    p = float(drop_p) if training else 0.0
    return _RowNorm.apply(x, res, weight, bias, groups, eps, act, p, next_dropout_seed() if p > 0 else 0)


class _ColNorm(torch.autograd.Function):
    """nn.BatchNorm1d over the nodes of a batch + activation + dropout (csrc/colnorm.hip; models/encoders.py:95-100,211-219 with
    normalization="batch").  ``running_mean`` / ``running_var`` are updated in place in training mode, as the module does."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training: bool, momentum: float, eps: float, act: int, drop_p: float, seed: int):
        lib = _lib.load()
        x, gamma, beta = _f32c(x), _f32c(gamma), _f32c(beta)
        _lib.require_cuda(x, gamma, beta, running_mean, running_var)
        N, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        wsb = _lib.workspace_bytes("dgdm_colnorm_workspace_bytes", N, C)
        ws = torch.empty(max(wsb, 4) // 4, dtype=torch.float32, device=x.device)
        slot = new_amax_slot(x.device)
        _lib.check(lib.dgdm_colnorm_fwd(x.data_ptr(), N, C, gamma.data_ptr(), beta.data_ptr(), _lib.ptr(running_mean), _lib.ptr(running_var),
                                        int(training), momentum, eps, act, drop_p, seed, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(),
                                        wsb, slot, _lib.stream_ptr(x.device)), "dgdm_colnorm_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.meta = (bool(training), act, drop_p, seed)
        return tag_amax(y, slot)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        training, act, drop_p, seed = ctx.meta
        gy = _f32c(gy)
        N, C = x.shape
        dx = torch.empty_like(x)
        dg, db = torch.empty_like(gamma), torch.empty_like(beta)
        wsb = _lib.workspace_bytes("dgdm_colnorm_workspace_bytes", N, C)
        ws = torch.empty(max(wsb, 4) // 4, dtype=torch.float32, device=x.device)
        slot = new_amax_slot(x.device)
        _lib.check(lib.dgdm_colnorm_bwd(x.data_ptr(), gy.data_ptr(), N, C, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                        int(training), act, drop_p, seed, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, slot,
                                        _lib.stream_ptr(x.device)), "dgdm_colnorm_bwd")
        return tag_amax(dx, slot), dg, db, None, None, None, None, None, None, None, None


def batch_norm_supported(bn, x) -> bool:
    return (bn.affine and bn.track_running_stats and bn.momentum is not None and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32
            and x.size(1) % 4 == 0 and x.size(0) > 0)


def batch_norm(x, bn, act: int = ACT_NONE, drop_p: float = 0.0, training: bool = False):
    """dropout(act(bn(x))) for an ``nn.BatchNorm1d`` module ``bn`` on a 2-D node matrix; the module's running statistics and
    ``num_batches_tracked`` advance in training mode as they do in ``bn.forward``."""
    p = float(drop_p) if training else 0.0
    use_batch = bool(bn.training)
    if use_batch and x.size(0) < 2:
        raise ValueError("Expected more than 1 value per channel when training")       # torch's message for a one-row batch
    if use_batch:
        bn.num_batches_tracked.add_(1)
    return _ColNorm.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, use_batch, float(bn.momentum), float(bn.eps), act, p,
                          next_dropout_seed() if p > 0 else 0)


def _decide_arg(decide, like: torch.Tensor):
    """Kink decisions for the ReLU kernels (include/dgdm_hip.h, `decide`): uint8 [N, C] contiguous on the device, or None."""
    if decide is None:
        return None
    d = decide.to(device=like.device, dtype=torch.uint8).contiguous()
    if d.numel() != like.numel():
        raise ValueError(f"decisions have {d.numel()} elements, the activation {like.numel()}")
    return d


class _ActDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act: int, drop_p: float, seed: int, decide=None):
        lib = _lib.load()
        x = _f32c(x)
        _lib.require_cuda(x)
        y = torch.empty_like(x)
        slot = new_amax_slot(x.device)
        _lib.check(lib.dgdm_act_dropout_fwd(x.data_ptr(), x.numel(), act, drop_p, seed, y.data_ptr(), _lib.ptr(decide), slot,
                                            _lib.stream_ptr(x.device)), "dgdm_act_dropout_fwd")
        ctx.save_for_backward(x)
        ctx.meta = (act, drop_p, seed, decide)
        return tag_amax(y, slot)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        act, drop_p, seed, decide = ctx.meta
        gy = _f32c(gy)
        dx = torch.empty_like(x)
        slot = new_amax_slot(x.device)
        _lib.check(lib.dgdm_act_dropout_bwd(x.data_ptr(), gy.data_ptr(), x.numel(), act, drop_p, seed, dx.data_ptr(), _lib.ptr(decide), slot,
                                            _lib.stream_ptr(x.device)), "dgdm_act_dropout_bwd")
        return tag_amax(dx, slot), None, None, None, None


def act_dropout(x, act: int = ACT_NONE, drop_p: float = 0.0, training: bool = False, decide=None):
    """dropout(act(x)).  ``decide`` (ReLU only): kink decisions handed in by a parity test instead of the sign of x."""
    p = float(drop_p) if training else 0.0
    if act == ACT_NONE and p == 0.0:
        return x
    if x.numel() % 4:
        raise _lib.DGDMKernelError("act_dropout needs numel % 4 == 0")
    return _ActDropout.apply(x, act, p, next_dropout_seed() if p > 0 else 0, _decide_arg(decide, x))


class _QSample(torch.autograd.Function):
    """x_t = sqrt(ac[t_g]) x0 + sqrt(1 - ac[t_g]) eps, t_g = the timestep of the row's graph (DiffusionLayer.add_noise,
    core/diffusion.py:123-145, for a whole batch in one launch); d/dx0 = sqrt(ac[t_g])."""

    @staticmethod
    def forward(ctx, x0, eps, timesteps, tab_a, tab_b, plan: AttnPlan):
        x0, eps = _f32c(x0), _f32c(eps)
        ctx.save_for_backward(timesteps, tab_a)
        ctx.plan = plan
        return _qsample_raw(x0, eps, timesteps, tab_a, tab_b, plan)

    @staticmethod
    def backward(ctx, g):
        timesteps, tab_a = ctx.saved_tensors
        return _qsample_raw(_f32c(g), None, timesteps, tab_a, None, ctx.plan), None, None, None, None, None


def _qsample_raw(x, eps, timesteps, tab_a, tab_b, plan: AttnPlan):
    lib = _lib.load()
    _lib.require_cuda(x, eps, timesteps, tab_a, tab_b)
    if timesteps.dtype != torch.int64 or timesteps.numel() != plan.B:
        raise ValueError(f"timesteps must be int64 [{plan.B}], got {timesteps.dtype} {tuple(timesteps.shape)}")
    out = torch.empty_like(x)
    slot = new_amax_slot(x.device)
    _lib.check(lib.dgdm_qsample(x.data_ptr(), _lib.ptr(eps), tab_a.data_ptr(), _lib.ptr(tab_b), timesteps.contiguous().data_ptr(),
                                plan.ptr_dev.data_ptr(), plan.B, x.size(0), x.size(1), out.data_ptr(), slot, _lib.stream_ptr(x.device)),
               "dgdm_qsample")
    return tag_amax(out, slot)


def qsample(x0, eps, timesteps, tab_a, tab_b, plan: AttnPlan):
    return _QSample.apply(x0, eps.detach(), timesteps, tab_a, tab_b, plan)


class _SegmentMSE(torch.autograd.Function):
    """mean over graphs of mse(pred_g, target_g) (models/dgdm_model.py:430-433) as one fixed-order reduction."""

    @staticmethod
    def forward(ctx, pred, target, plan: AttnPlan):
        lib = _lib.load()
        pred, target = _f32c(pred), _f32c(target)
        _lib.require_cuda(pred, target)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        wsb = _lib.workspace_bytes("dgdm_segment_mse_workspace_bytes", plan.B)
        ws = torch.empty(max(wsb, 4) // 4, dtype=torch.float32, device=pred.device)
        _lib.check(lib.dgdm_segment_mse_fwd(pred.data_ptr(), target.data_ptr(), plan.ptr_dev.data_ptr(), plan.B, pred.size(0), pred.size(1),
                                            loss.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr(pred.device)), "dgdm_segment_mse_fwd")
        ctx.save_for_backward(pred, target)
        ctx.plan = plan
        return loss

    @staticmethod
    def backward(ctx, gloss):
        lib = _lib.load()
        pred, target = ctx.saved_tensors
        plan = ctx.plan
        gloss = _f32c(gloss.reshape(1))
        dpred = torch.empty_like(pred)
        slot = new_amax_slot(pred.device)
        _lib.check(lib.dgdm_segment_mse_bwd(pred.data_ptr(), target.data_ptr(), gloss.data_ptr(), plan.ptr_dev.data_ptr(), plan.B, pred.size(0),
                                            pred.size(1), dpred.data_ptr(), slot, _lib.stream_ptr(pred.device)), "dgdm_segment_mse_bwd")
        return tag_amax(dpred, slot), None, None


def segment_mse(pred, target, plan: AttnPlan):
    return _SegmentMSE.apply(pred, target.detach(), plan)


def mask_rows(x, node_map, token):
    """out[n] = token where node_map[n] >= 0 else x[n] (entity masking, models/dgdm_model.py:494-503); no gradient (inputs are data)."""
    lib = _lib.load()
    x, token = _f32c(x), _f32c(token)
    _lib.require_cuda(x, node_map, token)
    out = torch.empty_like(x)
    slot = new_amax_slot(x.device)
    _lib.check(lib.dgdm_mask_rows(x.data_ptr(), node_map.data_ptr(), token.data_ptr(), x.size(0), x.size(1), out.data_ptr(), slot,
                                  _lib.stream_ptr(x.device)), "dgdm_mask_rows")
    return tag_amax(out, slot)


def ddpm_step(x, eps, z, sqrt_one_minus_ac: float, sqrt_ac: float, sqrt_alpha: float, sqrt_var: float, last: bool, out=None):
    """One update of DiffusionLayer.sample (core/diffusion.py:255-273): x0 = (x - s*eps)/a; out = x0 or sqrt(alpha) x0 + sqrt(var) z."""
    lib = _lib.load()
    x, eps = _f32c(x), _f32c(eps)
    z = _f32c(z) if z is not None else None
    _lib.require_cuda(x, eps, z)
    out = torch.empty_like(x) if out is None else out
    _lib.check(lib.dgdm_ddpm_step(x.data_ptr(), eps.data_ptr(), _lib.ptr(z), x.numel(), sqrt_one_minus_ac, sqrt_ac, sqrt_alpha, sqrt_var,
                                  int(last), out.data_ptr(), _lib.stream_ptr(x.device)), "dgdm_ddpm_step")
    return out


# ----------------------------------------------------------------------------- segment ops / K10 pooling
SAMPLE_STEP_FUSED = True      # DiffusionLayer.sample: one launch per step (csrc/sample_step.hip) where the width is taken; False: seven


def denoise_ddpm_step_supported(C: int) -> bool:
    return (SAMPLE_STEP_FUSED and USE_WEIGHT_IMAGES and GEMM_MATH == "f16x2" and bool(_lib.load().dgdm_denoise_ddpm_step_supported(int(C))))


def denoise_ddpm_step(x, z, w0x, w1, w2, bias0, gn1, bias1, gn2, bias2, s1mac: float, sac: float, salpha: float, svar: float, last: bool):
    """One step of DiffusionLayer.sample in ONE launch (csrc/sample_step.hip): eps = denoise_net([x | t_emb]) with the time half as the
    per-step ``bias0`` [4C], then the DDPM update with ``z`` (None on the last step).  ``w0x`` = denoise_net[0].weight[:, :C], ``w1`` /
    ``w2`` the other two Linear weights (their images are the training step's), ``gn1`` / ``gn2`` the nn.GroupNorm modules.  Eval only."""
    lib = _lib.load()
    x = _f32c(x)
    _lib.require_cuda(x, z, bias0)
    N, C = x.shape
    e0, e1, e2 = WEIGHT_IMAGES.get(0, _rm_tagged(w0x)), WEIGHT_IMAGES.get(0, _rm_tagged(w1)), WEIGHT_IMAGES.get(0, _rm_tagged(w2))
    out = torch.empty_like(x)
    if z is not None:
        z = _f32c(z)
    bias0 = _f32c(bias0)
    _lib.check(lib.dgdm_denoise_ddpm_step(x.data_ptr(), x.stride(0), _lib.ptr(z), z.stride(0) if z is not None else 0, N, C,
                                          e0.img.data_ptr(), e0.tiles, e1.img.data_ptr(), e1.tiles, e2.img.data_ptr(), e2.tiles, bias0.data_ptr(),
                                          gn1.weight.data_ptr(), gn1.bias.data_ptr(), float(gn1.eps), bias1.data_ptr(), gn2.weight.data_ptr(),
                                          gn2.bias.data_ptr(), float(gn2.eps), bias2.data_ptr(), s1mac, sac, salpha, svar, int(last),
                                          out.data_ptr(), out.stride(0), _lib.stream_ptr(x.device)), "dgdm_denoise_ddpm_step")
    return out


def segment_sum_raw(x, plan: AttnPlan) -> torch.Tensor:
    lib = _lib.load()
    x = _f32c(x)
    _lib.require_cuda(x)
    C = x.size(1)
    out = torch.empty(plan.B, C, dtype=torch.float32, device=x.device)
    wsb = _lib.workspace_bytes("dgdm_segment_sum_workspace_bytes", plan.B, C)
    ws = torch.empty(max(wsb, 4) // 4, dtype=torch.float32, device=x.device)
    _lib.check(lib.dgdm_segment_sum(x.data_ptr(), plan.ptr_dev.data_ptr(), plan.B, C, out.data_ptr(), ws.data_ptr(), wsb,
                                    _lib.stream_ptr(x.device)), "dgdm_segment_sum")
    return out


class _SegmentMax(torch.autograd.Function):
    """Per-graph maximum over the rows, [N, C] -> [B, C] (GlobalMaxPool, models/dgdm_model.py:570-585): csrc/segment.hip; the
    gradient goes to the row that attained the maximum (the first on ties), as torch.max(dim=0) sends it."""

    @staticmethod
    def forward(ctx, x, plan: AttnPlan):
        lib = _lib.load()
        x = _f32c(x)
        _lib.require_cuda(x)
        C = x.size(1)
        out = torch.empty(plan.B, C, dtype=torch.float32, device=x.device)
        arg = torch.empty(plan.B, C, dtype=torch.int32, device=x.device)
        wsb = _lib.workspace_bytes("dgdm_segment_max_workspace_bytes", plan.B, C)
        ws = torch.empty(max(wsb, 4) // 4, dtype=torch.float32, device=x.device)
        _lib.check(lib.dgdm_segment_max_fwd(x.data_ptr(), x.stride(0), plan.ptr_dev.data_ptr(), plan.B, C, out.data_ptr(), arg.data_ptr(),
                                            ws.data_ptr(), wsb, _lib.stream_ptr(x.device)), "dgdm_segment_max_fwd")
        ctx.save_for_backward(arg)
        ctx.plan, ctx.n = plan, x.size(0)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, g, _garg=None):
        (arg,) = ctx.saved_tensors
        g = _f32c(g)
        dx = torch.empty(ctx.n, g.size(1), dtype=torch.float32, device=g.device)
        _lib.check(_lib.load().dgdm_segment_max_bwd(g.data_ptr(), arg.data_ptr(), ctx.plan.ptr_dev.data_ptr(), ctx.plan.B, ctx.n, g.size(1),
                                                    dx.data_ptr(), _lib.stream_ptr(g.device)), "dgdm_segment_max_bwd")
        return dx, None


def segment_max(x, plan: AttnPlan, return_arg: bool = False):
    out, arg = _SegmentMax.apply(x, plan)
    return (out, arg) if return_arg else out


class _SegmentBcastAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, src, plan: AttnPlan):
        lib = _lib.load()
        x, src = _f32c(x), _f32c(src)
        _lib.require_cuda(x, src)
        N, C = x.shape
        out = torch.empty_like(x)
        _lib.check(lib.dgdm_segment_bcast_add(x.data_ptr(), src.data_ptr(), plan.ptr_dev.data_ptr(), plan.B, N, C, out.data_ptr(),
                                              _lib.stream_ptr(x.device)), "dgdm_segment_bcast_add")
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, g):
        g = _f32c(g)
        return g, segment_sum_raw(g, ctx.plan), None


class _SegmentMean(torch.autograd.Function):
    """out[g] = mean of the rows of graph g (GlobalMeanPool, models/dgdm_model.py:552-567): the fixed-order segment sum times
    1/n_g; backward broadcasts g/n_g back over the graph's rows."""

    @staticmethod
    def forward(ctx, x, plan: AttnPlan):
        inv = device_constant([1.0 / max(plan.ptr_host[g + 1] - plan.ptr_host[g], 1) for g in range(plan.B)], torch.float32, x.device)
        ctx.plan, ctx.n = plan, x.size(0)
        ctx.save_for_backward(inv)
        return segment_sum_raw(x, plan) * inv.unsqueeze(1)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (inv,) = ctx.saved_tensors
        src = _f32c(g * inv.unsqueeze(1))
        plan, C = ctx.plan, g.size(1)
        out = torch.empty(ctx.n, C, dtype=torch.float32, device=g.device)
        _lib.check(lib.dgdm_segment_bcast_add(None, src.data_ptr(), plan.ptr_dev.data_ptr(), plan.B, ctx.n, C, out.data_ptr(),
                                              _lib.stream_ptr(g.device)), "dgdm_segment_bcast_add")
        return out, None


def segment_mean(x, plan: AttnPlan):
    return _SegmentMean.apply(x, plan)


def segment_bcast_add(x, src, plan: AttnPlan):
    """out[n] = x[n] + src[graph(n)]  (src [B, C]); backward of src is a fixed-order segment sum."""
    return _SegmentBcastAdd.apply(x, src, plan)


POOL_HEAD_DIMS = (4, 8, 16, 32, 64)


class _AttnPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kv, q_scaled, plan: AttnPlan, H: int, D: int, drop_p: float, seed: int):
        lib = _lib.load()
        kv, q_scaled = _f32c(kv), _f32c(q_scaled)
        _lib.require_cuda(kv, q_scaled)
        N, C = kv.size(0), H * D
        P = torch.empty(N, H, dtype=torch.float32, device=kv.device)
        out = torch.empty(plan.B, C, dtype=torch.float32, device=kv.device)
        max_rows = max((plan.ptr_host[g + 1] - plan.ptr_host[g] for g in range(plan.B)), default=0)
        wsb = _lib.workspace_bytes("dgdm_attn_pool_fwd_workspace_bytes", plan.B, H, D, max_rows)
        ws = torch.empty(max(wsb, 4) // 4, dtype=torch.float32, device=kv.device)
        _lib.check(lib.dgdm_attn_pool_fwd(kv.data_ptr(), kv[:, C:].data_ptr(), kv.stride(0), q_scaled.data_ptr(),
                                          plan.ptr_dev.data_ptr(), plan.B, H, D, max_rows, drop_p, seed, P.data_ptr(), out.data_ptr(),
                                          ws.data_ptr(), wsb, _lib.stream_ptr(kv.device)), "dgdm_attn_pool_fwd")
        ctx.save_for_backward(kv, q_scaled, P, out)
        ctx.meta = (plan, H, D, drop_p, seed)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        kv, q_scaled, P, out = ctx.saved_tensors
        plan, H, D, drop_p, seed = ctx.meta
        C = H * D
        gout = _f32c(gout)
        dkv = torch.empty_like(kv)
        dq_part = torch.empty(plan.B, C, dtype=torch.float32, device=kv.device)
        _lib.check(lib.dgdm_attn_pool_bwd(kv.data_ptr(), kv[:, C:].data_ptr(), kv.stride(0), q_scaled.data_ptr(),
                                          plan.ptr_dev.data_ptr(), plan.B, H, D, drop_p, seed, P.data_ptr(), out.data_ptr(),
                                          gout.data_ptr(), dkv.data_ptr(), dkv[:, C:].data_ptr(), dkv.stride(0), dq_part.data_ptr(),
                                          _lib.stream_ptr(kv.device)), "dgdm_attn_pool_bwd")
        return dkv, dq_part.sum(0), None, None, None, None, None


def attn_pool(kv, q_scaled, plan: AttnPlan, H: int, D: int, drop_p: float = 0.0, training: bool = False):
    """GlobalAttentionPool core: kv [N, 2*H*D] (K | V), q_scaled [H*D] -> [B, H*D]."""
    p = float(drop_p) if training else 0.0
    return _AttnPool.apply(kv, q_scaled, plan, H, D, p, next_dropout_seed() if p > 0 else 0)


# ----------------------------------------------------------------------------- operand maxima for the fp16 hi+lo GEMMs
class AmaxArena:
    """Device slots holding the float bits of max|x| of GEMM operands (csrc/gemm_h.hip).  A ring of uint32 slot groups, zeroed
    chunk by chunk as the bump pointer enters a chunk (one fill launch per CHUNK groups, also recorded by a graph capture at the
    same position, so a replay re-zeroes exactly the slots it refills).

    Lifetime of a slot's content: from the kernel that fills it until the ring comes round and the slot's CHUNK is zeroed again
    (SLOTS takes later; a step takes ~300).  Every chunk carries a generation counter (``chunk_gen``) bumped by each fill; a tag
    or a saved handle records the generation it was made under and is dead once the chunk has been refilled -- a long-lived
    tensor (a device-resident input reused over epochs) then falls back to a reduction launch instead of scaling by whatever
    the recycled slot holds now."""

    SLOTS, CHUNK = 1 << 12, 1 << 6          # slot groups in the ring / groups zeroed per fill launch
    GROUP_WORDS = 32 * 64                   # DGDM_AMAX_WAYS * DGDM_AMAX_STRIDE (include/dgdm_hip.h): 8 KiB per slot group

    def __init__(self, device):
        self.buf = torch.zeros(self.SLOTS * self.GROUP_WORDS, dtype=torch.int32, device=device)     # 32 MiB
        self.next = 0
        self.base = self.buf.data_ptr()
        self.end = self.base + 4 * self.SLOTS * self.GROUP_WORDS
        self.chunk_gen = [0] * (self.SLOTS // self.CHUNK)
        self.total_takes = 0
        self.recording: Optional[set] = None    # chunks entered while a stream capture records (begin_amax_recording)

    def take(self) -> int:
        """Address of a zeroed slot group."""
        i = self.next
        if i % self.CHUNK == 0:
            _lib.check(_lib.load().dgdm_fill_u32(self.base + 4 * i * self.GROUP_WORDS, self.CHUNK * self.GROUP_WORDS, 0,
                                                 _lib.stream_ptr(self.buf.device)), "dgdm_fill_u32")
            self.chunk_gen[i // self.CHUNK] += 1
        if self.recording is not None:
            self.recording.add(i // self.CHUNK)
        self.next = (i + 1) % self.SLOTS
        self.total_takes += 1
        return self.base + 4 * i * self.GROUP_WORDS

    def align(self) -> None:
        """Advance to the next chunk boundary: the next ``take`` issues (and a capture records) the fill of its chunk.  Called
        before every stream capture -- the slots a recording takes before its first boundary would otherwise never be re-zeroed by
        a replay and keep the maximum over everything earlier replays, other recordings and eager steps left there."""
        self.next = ((self.next + self.CHUNK - 1) // self.CHUNK * self.CHUNK) % self.SLOTS

    def gen_of(self, addr: int) -> Optional[int]:
        """Generation of the chunk that holds slot ``addr``; None for an address outside the ring (a weight's persistent slot)."""
        if not self.base <= addr < self.end:
            return None
        return self.chunk_gen[(addr - self.base) // (4 * self.GROUP_WORDS * self.CHUNK)]


_ARENAS: dict = {}
AMAX_FALLBACK_LOG: Optional[list] = None


def _arena(device) -> AmaxArena:
    key = device.index if device.index is not None else torch.cuda.current_device()
    a = _ARENAS.get(key)
    if a is None:
        a = _ARENAS[key] = AmaxArena(torch.device("cuda", key))
    return a


def begin_amax_recording() -> None:
    """Call right before a stream capture that takes slots: every arena moves to a chunk boundary (AmaxArena.align), so the
    recording's first ``take`` records the zero-fill of its chunk, and the chunks the recording enters are noted."""
    for a in _ARENAS.values():
        a.align()
        a.recording = set()


def end_amax_recording() -> list:
    """After the capture: [(arena, chunk indices)] for ``amax_recording_replayed``; eager takes continue in a fresh chunk."""
    rec = []
    for a in _ARENAS.values():
        if a.recording is not None:
            rec.append((a, sorted(a.recording)))
            a.recording = None
            a.align()
    return rec


def amax_recording_replayed(rec: list) -> None:
    """Call after every replay of a recording: the replay zero-filled and refilled the chunks it uses behind the host's back, so
    every tag made under their old generation (an eager tensor that happened to sit in one of them) is dead from here on."""
    for a, chunks in rec:
        for c in chunks:
            a.chunk_gen[c] += 1


def new_amax_slot(device) -> Optional[int]:
    """A zeroed slot group for a producer kernel to fill -- or None when the fp16 hi+lo GEMMs are not selected."""
    return _arena(device).take() if GEMM_MATH == "f16x2" else None


def _slot_gen(slot: int, device) -> Optional[int]:
    a = _ARENAS.get(device.index if device.index is not None else torch.cuda.current_device())
    return None if a is None else a.gen_of(slot)


def amax_handle(t: torch.Tensor):
    """(slot address, chunk generation) of a live upper bound of max|t| if a producer (or an earlier GEMM) left one, else None.
    A tag is bound to the tensor's version counter: an in-place write after tagging (the autograd engine sums a second gradient
    INTO the first one's buffer; an accumulating GEMM epilogue) invalidates it; and to the generation of the slot's chunk: once
    the ring has recycled the slot the tag is dead.  Either way the consumer falls back to a reduction launch."""
    tag = getattr(t, "_dgdm_amax", None)
    if tag is None:
        return None
    if isinstance(tag, tuple):
        slot, version, gen = tag
        if version != t._version or (gen is not None and _slot_gen(slot, t.device) != gen):
            return None
        return (slot, gen)
    return (tag, None)              # parameters: refreshed at every forward (WeightAmax), never version-bound, not in the ring


def amax_of(t: torch.Tensor) -> Optional[int]:
    """Slot address of a live upper bound of max|t| (see ``amax_handle``), else None."""
    h = amax_handle(t)
    return None if h is None else h[0]


def tag_amax(t: torch.Tensor, slot) -> torch.Tensor:
    """Bind an amax slot to ``t``.  ``slot``: the address of a slot just taken (int), or a handle saved earlier with
    ``amax_handle`` (autograd contexts keep those from forward to backward): a handle whose slot has been recycled since is dropped."""
    if slot is None:
        return t
    if isinstance(slot, tuple):
        addr, gen = slot
        if gen is not None and _slot_gen(addr, t.device) != gen:
            return t
    else:
        addr, gen = slot, _slot_gen(slot, t.device)
    t._dgdm_amax = (addr, t._version, gen)
    return t


def ensure_amax(t: torch.Tensor) -> int:
    """The tensor's amax slot; computes it with one reduction launch (dgdm_amax_bits) when no producer supplied it.  ``t``: 2-D fp32,
    unit column stride, cols % 4 == 0, 16-byte aligned (what the tile GEMMs accept anyway)."""
    slot = amax_of(t)
    if slot is None:
        if AMAX_FALLBACK_LOG is not None:       # diagnostics (tools/attic/amax_fallbacks.py): who still needs a reduction launch?
            import traceback
            fr = [f for f in traceback.extract_stack(limit=30) if "dgdm_histopath_lab_amd" in f.filename][-9:-1]
            AMAX_FALLBACK_LOG.append((tuple(t.shape), " < ".join(f"{f.name}:{f.lineno}" for f in reversed(fr))))
        slot = _arena(t.device).take()
        _lib.check(_lib.load().dgdm_amax_bits(t.data_ptr(), _ld(t), t.size(0), t.size(1), slot, _lib.stream_ptr(t.device)), "dgdm_amax_bits")
        tag_amax(t, slot)
    return slot


class WeightAmax:
    """max|w| of every parameter of a module, refreshed by ONE launch (dgdm_amax_table) at the start of a forward: weights only
    change in the optimizer step, so the values hold for the whole forward + backward.  Parameters are tagged with their slot;
    a column slice of a weight inherits the whole matrix's maximum (an upper bound is all the GEMM needs)."""

    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters() if p.dim() >= 2 and p.dtype == torch.float32 and p.is_cuda and p.is_contiguous()]
        self.ptrs = [p.data_ptr() for p in self.params]
        dev = self.params[0].device if self.params else None
        self.device = dev
        if not self.params:
            return
        import struct
        recs, CH = [], 1 << 13    # 8 loads of 16 B per thread: 64K-float records gave 109 workgroups of 64 dependent-looking steps each
        for g, p in enumerate(self.params):       # (22 us for 28 MB); large tensors are cut into records that share the tensor's group
            for off in range(0, p.numel(), CH):
                recs.append(struct.pack("<Qqq", p.data_ptr() + 4 * off, min(CH, p.numel() - off), g))
        self.count = len(recs)
        self.table = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev)
        self.slots = torch.zeros(len(self.params) * AmaxArena.GROUP_WORDS, dtype=torch.int32, device=dev)
        for i, p in enumerate(self.params):
            p._dgdm_amax = p._dgdm_wamax = self.slots.data_ptr() + 4 * i * AmaxArena.GROUP_WORDS    # _dgdm_wamax: never re-tagged

    def stale(self) -> bool:
        return any(p.data_ptr() != q for p, q in zip(self.params, self.ptrs))

    def refresh(self) -> None:
        if not self.params:
            return
        lib, st = _lib.load(), _lib.stream_ptr(self.device)
        _lib.check(lib.dgdm_fill_u32(self.slots.data_ptr(), self.slots.numel(), 0, st), "dgdm_fill_u32")
        _lib.check(lib.dgdm_amax_table(self.table.data_ptr(), self.count, self.slots.data_ptr(), st), "dgdm_amax_table")


def refresh_weight_amax(module: torch.nn.Module) -> None:
    """Call at the start of a forward (DGDMModel.forward does): (re)builds the module's weight-maximum table when its parameters
    moved, then refreshes the maxima.  No-op unless the fp16 hi+lo GEMMs are selected."""
    if GEMM_MATH != "f16x2":
        return
    wa = module.__dict__.get("_dgdm_weight_amax")
    if wa is None or wa.stale():
        wa = module.__dict__["_dgdm_weight_amax"] = WeightAmax(module)
        WEIGHT_IMAGES.dirty = True          # the parameters' amax slots moved: the image table points at the old ones
    wa.refresh()
    if USE_WEIGHT_IMAGES and wa.device is not None:
        WEIGHT_IMAGES.refresh(wa.device)


# ----------------------------------------------------------------------------- weight images (csrc/gemm_img.hip)
class _WImage:
    __slots__ = ("kind", "srcs", "img", "tiles", "cols", "k", "versions", "epoch", "used", "blocks")


class WeightImages:
    """Pre-split fp16 hi+lo images of the WEIGHT operands of y = x W^T (kind 0) and dx = dy W (kind 1).

    Weights only change in the optimizer step, so an image built at the start of a forward holds for the whole forward +
    backward.  Images whose sources are parameters (or views of parameters) are registered here and ALL rebuilt by ONE launch
    per step (``refresh``, called from ``refresh_weight_amax`` right after the weight maxima); an image of a temporary (the
    concatenated QKV weight) lives on the tensor object and costs one small launch.  A registered image is valid while the
    registry epoch and the sources' version counters are those of its last build; anything else rebuilds it on the spot, so a
    weight written in place between two calls is never served from a stale image."""

    MAX_IDLE_EPOCHS = 4      # registered images not looked up for this many refreshes are dropped (models that went away)

    def __init__(self):
        self.entries: dict = {}
        self.epoch = 0
        self.table = None          # device table of the registered images (rebuilt when the set changes)
        self.table_n = 0
        self.table_blocks = 0
        self.dirty = True

    @staticmethod
    def _key(kind, w0, w1):
        k = (kind, w0.data_ptr(), tuple(w0.shape), w0.stride(0))
        return k if w1 is None else k + (w1.data_ptr(), tuple(w1.shape), w1.stride(0))

    @staticmethod
    def _persistent(w) -> bool:
        b = w._base if w._base is not None else w
        return isinstance(b, torch.nn.Parameter)

    @staticmethod
    def _dims(kind, w0, w1):
        rows, c0 = w0.shape
        c1 = 0 if w1 is None else w1.size(1)
        return (rows, c0 + c1) if kind == 0 else (c0, rows)        # (image columns, reduction length)

    def _new(self, kind, w0, w1, keep_srcs: bool = True) -> _WImage:
        e = _WImage()
        # an image of a TEMPORARY (the concatenated QKV weight, a tensor with a grad_fn) lives in that tensor's own __dict__: holding
        # the tensor here as well would close a reference cycle that keeps the temporary -- and through its grad_fn the whole
        # autograd graph of the step, AccumulateGrad nodes included -- alive until the cyclic collector runs (the source of torch's
        # "AccumulateGrad node's stream does not match" warning when the next step runs on another stream)
        e.kind, e.srcs = kind, ((w0, w1) if keep_srcs else None)
        e.cols, e.k = self._dims(kind, w0, w1)
        lib = _lib.load()
        e.img = torch.empty(lib.dgdm_gemm_image_bytes(e.cols, e.k), dtype=torch.uint8, device=w0.device)
        e.blocks = lib.dgdm_gemm_image_blocks(e.cols, e.k)
        e.tiles = (e.cols + 31) // 32
        e.versions, e.epoch, e.used = None, -1, self.epoch
        return e

    @staticmethod
    def _param_amax(w) -> Optional[int]:
        """The persistent amax slot of the parameter ``w`` is (a view of): refreshed by WeightAmax at every forward.  A slice
        inherits the whole matrix's maximum (an upper bound is all the scale needs)."""
        if w is None:
            return None
        b = w._base if w._base is not None else w
        return getattr(b, "_dgdm_wamax", None)

    def _build_one(self, e: _WImage, w0=None, w1=None) -> None:
        if w0 is None:
            w0, w1 = e.srcs
        a0 = self._param_amax(w0) or ensure_amax(w0)
        a1 = (self._param_amax(w1) or ensure_amax(w1)) if w1 is not None else None
        _lib.check(_lib.load().dgdm_gemm_image_build(
            w0.data_ptr(), w0.stride(0), _lib.ptr(w1), w1.stride(0) if w1 is not None else 0, a0, a1, e.img.data_ptr(), w0.size(0),
            w0.size(1), 0 if w1 is None else w1.size(1), e.kind, _lib.stream_ptr(w0.device)), "dgdm_gemm_image_build")
        e.versions = (w0._version, None if w1 is None else w1._version)
        e.epoch = self.epoch

    def get(self, kind: int, w0: torch.Tensor, w1: Optional[torch.Tensor] = None) -> _WImage:
        """Image of B for ``x . [w0 | w1]^T`` (kind 0) or ``dy . w0`` (kind 1); w0 / w1: 2-D fp32, unit column stride."""
        if not (self._persistent(w0) and (w1 is None or self._persistent(w1))):
            holder = w0.__dict__.setdefault("_dgdm_images", {}) if hasattr(w0, "__dict__") else {}
            key = (kind, None if w1 is None else id(w1))
            e = holder.get(key)
            if e is None or e.versions != (w0._version, None if w1 is None else w1._version) or e.epoch != self.epoch:
                e = holder[key] = self._new(kind, w0, w1, keep_srcs=False)
                self._build_one(e, w0, w1)
            for sink in _CONSTANT_SINKS:
                sink.append(e.img)
            return e
        key = self._key(kind, w0, w1)
        e = self.entries.get(key)
        if e is None:
            e = self.entries[key] = self._new(kind, w0, w1)
            self.dirty = True
        e.used = self.epoch
        if e.epoch != self.epoch or e.versions != (w0._version, None if w1 is None else w1._version):
            e.srcs = (w0, w1)
            self._build_one(e)
        for sink in _CONSTANT_SINKS:      # a step being recorded bakes the image's address into its launches: it keeps the image alive
            sink.append(e.img)
        return e

    def _upload_table(self, device) -> None:
        import struct
        recs, block0 = [], 0
        live = {}
        for key, e in self.entries.items():
            if self.epoch - e.used > self.MAX_IDLE_EPOCHS:
                continue
            live[key] = e
        self.entries = live
        self.table_entries = []
        for e in self.entries.values():
            w0, w1 = e.srcs
            a0, a1 = self._param_amax(w0), self._param_amax(w1)
            if w0.device != device or a0 is None or (w1 is not None and a1 is None):
                continue           # no persistent maximum (a bare parameter outside a model): rebuilt on first use instead
            recs.append(struct.pack("<QQqqQQQiiiii4x", w0.data_ptr(), _lib.ptr(w1) or 0, w0.stride(0), w1.stride(0) if w1 is not None else 0,
                                    a0, a1 or 0, e.img.data_ptr(), w0.size(0), w0.size(1), 0 if w1 is None else w1.size(1), e.kind, block0))
            block0 += e.blocks
            self.table_entries.append(e)
        self.table_n, self.table_blocks = len(recs), block0
        self.table = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(device) if recs else None
        self.dirty = False

    def prepare(self, device) -> None:
        """Bring the device table up to date with the registered images (a host-to-device copy: call it OUTSIDE a stream
        capture -- training.GraphedPretrainStep does, right before it records, so that the recorded step rebuilds every image
        with the one table launch instead of one launch per weight)."""
        if self.dirty and not torch.cuda.is_current_stream_capturing():
            self._upload_table(device)

    def refresh(self, device) -> None:
        """New epoch: every registered image is rebuilt from the current weights by one launch (the weights' maxima must have
        been refreshed on the same stream just before)."""
        self.epoch += 1
        capturing = torch.cuda.is_current_stream_capturing()
        stale = self.dirty or any(self.epoch - e.used > self.MAX_IDLE_EPOCHS for e in self.entries.values())
        if stale and not capturing:
            self._upload_table(device)
        if self.table is None:
            return                      # first step: images are built one by one on first use
        _lib.check(_lib.load().dgdm_gemm_image_build_many(self.table.data_ptr(), self.table_n, self.table_blocks, _lib.stream_ptr(device)),
                   "dgdm_gemm_image_build_many")
        for sink in _CONSTANT_SINKS:      # a recorded step replays this launch: the table and every image it writes stay allocated
            sink.append(self.table)
            sink.extend(e.img for e in self.table_entries)
        for e in self.table_entries:
            w0, w1 = e.srcs
            e.versions = (w0._version, None if w1 is None else w1._version)
            e.epoch = self.epoch


WEIGHT_IMAGES = WeightImages()
USE_WEIGHT_IMAGES = True      # tools/ A/B switch: False sends the f16x2 row contractions to the register-staged kernels (gemm_h.hip)


def weights_changed() -> None:
    """Tell the image cache that weights were written behind autograd's back (a HIP-graph replay of an optimizer step bumps no
    version counter): images built before are not used again."""
    WEIGHT_IMAGES.epoch += 1


def _img_ok(a: torch.Tensor, K: int) -> bool:
    return USE_WEIGHT_IMAGES and K % 16 == 0 and K >= 16 and a.size(0) > 0


def _gemm_rows_img(a, e: _WImage, tile_begin: int, ncols: int, bias, out, accumulate: bool):
    M, K = a.shape
    if out is None:
        out = torch.empty(M, ncols, dtype=torch.float32, device=a.device)
    TIMERS.timed("gemm_img", lambda: _lib.check(_lib.load().dgdm_gemm_rows_img(
        a.data_ptr(), a.stride(0), M, K, e.img.data_ptr(), e.tiles, tile_begin, ncols, _lib.ptr(bias), out.data_ptr(), out.stride(0),
        int(accumulate), ensure_amax(a), _lib.stream_ptr(a.device)), "dgdm_gemm_rows_img"))
    return out


# ----------------------------------------------------------------------------- K3 dense contractions
GEMM_MIN_ROWS = 256   # fewer rows (one per graph / timestep): the exact-fp32 small-M kernels (csrc/smallm.hip), not a tile GEMM


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    """2-D fp32 view with unit column stride and a row stride that keeps 16-B alignment."""
    if t.dtype != torch.float32:
        raise _lib.DGDMKernelError(f"HIP kernels compute in fp32, got {t.dtype}")
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and t.stride(0) >= t.size(1):
        return t
    return t.contiguous()


def _gemm_entry(lib, name: str, math: str):
    if math not in ("fp32", "bf16x3", "f16x2"):
        raise ValueError(f"GEMM math must be 'fp32', 'bf16x3' or 'f16x2', got {math!r}")
    return getattr(lib, name if math == "fp32" else name + "_" + math)


def _rm_tagged(t: torch.Tensor) -> torch.Tensor:
    """_rowmajor that keeps the operand's amax tag when it has to copy."""
    r = _rowmajor(t)
    return r if r is t else tag_amax(r, amax_handle(t))


def gemm_nt_raw(a, w, bias=None, out=None, accumulate=False, math="fp32"):
    lib = _lib.load()
    a, w = _rm_tagged(a), _rm_tagged(w)
    M, K = a.shape
    N = w.size(0)
    if math == "f16x2" and _img_ok(a, K):
        return _gemm_rows_img(a, WEIGHT_IMAGES.get(0, w), 0, N, bias, out, accumulate)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    fn = _gemm_entry(lib, "dgdm_gemm_nt", math)
    extra = (ensure_amax(a), ensure_amax(w)) if math == "f16x2" else ()
    TIMERS.timed("gemm_nt", lambda: _lib.check(
        fn(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _lib.ptr(bias), out.data_ptr(), out.stride(0), M, N, K,
           int(accumulate), *extra, _lib.stream_ptr(a.device)), "dgdm_gemm_nt"))
    return out


def gemm_nt_split_raw(a, w0, w1, bias=None, math="bf16x3"):
    """a [M, K0 + K1] . [w0 | w1]^T (+ bias) without materialising the concatenated weight (16-bit-pipe kernels only)."""
    lib = _lib.load()
    a0, w00, w10 = a, w0, w1
    a, w0, w1 = _rowmajor(a), _rowmajor(w0), _rowmajor(w1)
    for new, old in ((a, a0), (w0, w00), (w1, w10)):
        if new is not old:
            tag_amax(new, amax_handle(old))
    M, K = a.shape
    N, K0 = w0.shape
    if w1.size(0) != N or K0 + w1.size(1) != K:
        raise ValueError(f"weights {tuple(w0.shape)} | {tuple(w1.shape)} do not match the operand {tuple(a.shape)}")
    if math == "f16x2" and _img_ok(a, K):
        return _gemm_rows_img(a, WEIGHT_IMAGES.get(0, w0, w1), 0, N, bias, None, False)
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    fn = lib.dgdm_gemm_nt_split_f16x2 if math == "f16x2" else lib.dgdm_gemm_nt_split_bf16x3
    extra = (ensure_amax(a), ensure_amax(w0), ensure_amax(w1)) if math == "f16x2" else ()
    TIMERS.timed("gemm_nt", lambda: _lib.check(
        fn(a.data_ptr(), a.stride(0), w0.data_ptr(), w0.stride(0), K0, w1.data_ptr(), w1.stride(0), _lib.ptr(bias),
           out.data_ptr(), out.stride(0), M, N, K, 0, *extra, _lib.stream_ptr(a.device)), "dgdm_gemm_nt_split"))
    return out


def gemm_nn_raw(a, w, out=None, accumulate=False, math="fp32"):
    """a [M,N] . w [N,K] -> [M,K]"""
    lib = _lib.load()
    a, w = _rm_tagged(a), _rm_tagged(w)
    M, N = a.shape
    K = w.size(1)
    if math == "f16x2" and _img_ok(a, N):
        return _gemm_rows_img(a, WEIGHT_IMAGES.get(1, w), 0, K, None, out, accumulate)
    if out is None:
        out = torch.empty(M, K, dtype=torch.float32, device=a.device)
    fn = _gemm_entry(lib, "dgdm_gemm_nn", math)
    extra = (ensure_amax(a), ensure_amax(w)) if math == "f16x2" else ()
    TIMERS.timed("gemm_nn", lambda: _lib.check(
        fn(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, N, K, int(accumulate), *extra,
           _lib.stream_ptr(a.device)), "dgdm_gemm_nn"))
    return out


# ---- deferred reduction of the split-M weight-gradient GEMMs ---------------------------------------------------------------
# Inside ``with deferred_weight_grads():`` the dW GEMMs of a backward pass only write their chunk partials; ONE launch per 24 GEMMs
# reduces them when the autograd engine finishes the pass (queue_callback) -- ~50 tiny reduction launches per step otherwise.  The
# returned dW / db tensors are filled by that launch: only valid for loops that start the backward with ``.grad is None`` (the
# engine then just stores the tensor) -- GraphedPretrainStep, DGDMTrainer.fit and bench.py do; anything that reads a weight
# gradient before the pass ends must call ``flush_deferred_tn()`` first (parallel.FlatGradAllReducer's early bucket does).
_DEFER_TN = False
_PENDING_TN: list = []
_DEFER_CLAIMED: set = set()      # ids of the parameters that already have a deferred producer in the running pass


class deferred_weight_grads:
    def __enter__(self):
        global _DEFER_TN
        self.prev, _DEFER_TN = _DEFER_TN, True
        _DEFER_CLAIMED.clear()
        return self

    def __exit__(self, *exc):
        global _DEFER_TN
        _DEFER_TN = self.prev
        flush_deferred_tn()
        return False


def _claim_deferred(*params) -> bool:
    """May the gradients of these parameters be returned UNFILLED and written at the end of the running pass?  Only if the engine's
    AccumulateGrad merely stores the tensor it is handed: every parameter is a leaf whose ``.grad`` is None (an existing gradient
    would be added to in place, right away) and has no other producer in this pass (the engine would sum the two tensors when the
    second arrives).  A second producer flushes what is pending -- the first one's tensor is then filled before the engine adds --
    and reduces immediately itself (tied weights, a module called twice per step, gradient accumulation)."""
    if not _DEFER_TN:
        return False
    ps = [p for p in params if p is not None]
    if any(id(p) in _DEFER_CLAIMED for p in ps):
        flush_deferred_tn()
        return False
    if any((not p.is_leaf) or p.grad is not None for p in ps):
        return False
    _DEFER_CLAIMED.update(id(p) for p in ps)
    return True


def flush_deferred_tn(on_stream: Optional[int] = None):
    """Run the held-back dW GEMMs (many-problem launches) and reduce every pending GEMM's chunk partials, all in fixed order.
    ``on_stream``: launch everything on this stream instead of the streams the problems were queued on (the caller orders it behind
    their producers and in front of their consumers); the problems' keep-alive references are then RETURNED -- the caller holds them
    until the join has been enqueued (their memory belongs to the producers' stream)."""
    if not _PENDING_TN:
        return None
    lib = _lib.load()
    pend = list(_PENDING_TN)
    _PENDING_TN.clear()
    if on_stream is not None:
        pend = [(desc, keep, on_stream, partial) for desc, keep, _s, partial in pend]
    if not _DEFER_TN:
        _DEFER_CLAIMED.clear()
    # on the stream each GEMM's partials were launched on (the engine's end-of-pass callback may run under another current stream
    # than the backward nodes did, e.g. when the step runs on a side stream)
    by_stream: dict = {}
    small: dict = {}
    for desc, keep, stream, partial in pend:
        by_stream.setdefault(stream, []).append(desc)
        if partial is not None:
            small.setdefault(stream, []).append((-(partial[8] // max(1, desc[6])), len(small.get(stream, ())), partial))
    for stream, parts in small.items():       # the partial GEMMs that were held back: up to TN_PARTIAL_MAX problems per launch
        # longest workgroups first (rows per chunk = M / slots): the backward meets the largest layers LAST, and a launch that
        # starts its longest workgroups last ends in a tail of a few of them
        parts = [t[2] for t in sorted(parts)] if TN_GROUP_SORT else [t[2] for t in parts]
        for i in range(0, len(parts), _lib.TN_PARTIAL_MAX):
            part = parts[i:i + _lib.TN_PARTIAL_MAX]
            arr = (_lib.TnPartial * len(part))()
            for j, d in enumerate(part):
                arr[j] = _lib.TnPartial(*d)
            _lib.check(lib.dgdm_gemm_tn_partial_many_f16x2(arr, len(part), stream), "dgdm_gemm_tn_partial_many_f16x2")
    for stream, descs in by_stream.items():
        for i in range(0, len(descs), _lib.TN_REDUCE_MAX):
            part = descs[i:i + _lib.TN_REDUCE_MAX]
            arr = (_lib.TnReduce * len(part))()
            for j, desc in enumerate(part):
                arr[j] = _lib.TnReduce(*desc)
            _lib.check(lib.dgdm_gemm_tn_reduce_many(arr, len(part), stream), "dgdm_gemm_tn_reduce_many")
    return [keep for _d, keep, _s, _p in pend] if on_stream is not None else None


def _defer_tn(desc, keep, stream, partial=None) -> bool:
    """Queue a reduction for the end of the running backward pass; False when no pass is running.  ``partial``: the arguments of the
    partial GEMM itself (a _lib.TnPartial tuple) when that launch is held back too and joins the pass's many-problem launch."""
    if not _PENDING_TN:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(flush_deferred_tn)
        except RuntimeError:
            return False
    _PENDING_TN.append((desc, keep, stream, partial))
    return True


# Inside ``deferred_weight_grads()`` the fp16 hi+lo dW GEMMs themselves are held back and run in the many-problem launches at the
# end of the pass (24 problems per launch, each cut into fewer and longer row chunks than a launch of its own would use: the
# gradients equal the immediate ones up to fp32 summation order, and are repeatable bit for bit).  False: one launch per GEMM.
TN_GROUPED = True
TN_GROUP_SORT = True


def gemm_tn_raw(dy, x, with_bias: bool, math="fp32", split: Optional[int] = None, out: Optional[torch.Tensor] = None,
                may_defer: bool = False):
    """dW [N,K] = dy[M,N]^T x[M,K]; db [N] = colsum(dy) (fixed-order split-M reduction).
    ``split=K0``: dW is delivered as two contiguous matrices (dW[:, :K0], dW[:, K0:]) -- returns ((dW0, dW1), db).
    ``out``: write dW there (a [N, K] view with unit column stride, e.g. a column block of a wider gradient matrix).
    ``may_defer``: the caller guarantees that dW / db go STRAIGHT to leaf parameters (the engine only stores them), so inside
    ``deferred_weight_grads()`` their reduction may wait for the end of the pass; a gradient that another backward node reads
    (padding / slicing of a concatenated weight) must not be deferred."""
    lib = _lib.load()
    dy, x = _rm_tagged(dy), _rm_tagged(x)
    M, N = dy.shape
    K = x.size(1)
    db = torch.empty(N, dtype=torch.float32, device=x.device) if with_bias else None
    extra = (ensure_amax(dy), ensure_amax(x)) if math == "f16x2" else ()
    defer = _DEFER_TN and may_defer and math in ("bf16x3", "f16x2") and M > 0
    grouped = defer and math == "f16x2" and TN_GROUPED
    if grouped:     # part of the pass's many-problem launch: its own (coarser) chunking, see dgdm_gemm_tn_partial_many_f16x2
        slots = _lib.workspace_bytes("dgdm_gemm_tn_chunks_grouped", M, N, K)
        wsb = slots * (N * K + (N if with_bias else 0)) * 4
    else:
        wsb = _lib.workspace_bytes("dgdm_gemm_tn_workspace_bytes" if math == "fp32" else f"dgdm_gemm_tn_{math}_workspace_bytes", M, N, K, int(with_bias))
    ws = torch.empty(max(wsb, 4) // 4, dtype=torch.float32, device=x.device)
    if defer:
        if split is None:
            dW = torch.empty(N, K, dtype=torch.float32, device=x.device) if out is None else out
            d0, d1, k0 = dW, None, K
        else:
            if not 0 < split < K:
                raise ValueError(f"split must lie inside (0, {K}), got {split}")
            d0 = torch.empty(N, split, dtype=torch.float32, device=x.device)
            d1 = torch.empty(N, K - split, dtype=torch.float32, device=x.device)
            k0 = split
        if not grouped:
            slots = lib.dgdm_gemm_tn_chunks(M, N, K)
        desc = (ws.data_ptr(), d0.data_ptr(), _lib.ptr(d1), _lib.ptr(db), d0.stride(0), d1.stride(0) if d1 is not None else 0, slots, N, K, k0)
        # keep-alive ALIASES of the outputs (same storage, other tensor objects): holding d0 itself would raise its reference count and
        # make the engine's AccumulateGrad CLONE it -- a copy of memory this launch has not filled yet -- instead of adopting it
        alias = lambda t: None if t is None else t.detach()
        keep = (ws, alias(d0), alias(d1), alias(db), dy, x)
        if grouped:
            partial = (dy.data_ptr(), x.data_ptr(), ws.data_ptr(), extra[0], extra[1], dy.stride(0), x.stride(0), wsb, M, N, K, int(with_bias))
            if _defer_tn(desc, keep, _lib.stream_ptr(x.device), partial):
                return (d0 if split is None else (d0, d1)), db
            # no backward pass is running: the immediate launch below, with the workspace of its own chunking
            wsb = _lib.workspace_bytes(f"dgdm_gemm_tn_{math}_workspace_bytes", M, N, K, int(with_bias))
            ws = torch.empty(max(wsb, 4) // 4, dtype=torch.float32, device=x.device)
        elif _defer_tn(desc, keep, _lib.stream_ptr(x.device)):
            fn = getattr(lib, "dgdm_gemm_tn_partial_" + math)
            TIMERS.timed("gemm_tn", lambda: _lib.check(
                fn(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), int(with_bias), M, N, K, ws.data_ptr(), wsb, *extra,
                   _lib.stream_ptr(x.device)), "dgdm_gemm_tn_partial"))
            return (d0 if split is None else (d0, d1)), db
        if split is not None:      # no backward pass is running: reduce right away into the tensors just made
            fn = _gemm_entry(lib, "dgdm_gemm_tn_split", math)
            _lib.check(fn(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), d0.data_ptr(), d0.stride(0), split, d1.data_ptr(), d1.stride(0),
                          _lib.ptr(db), M, N, K, ws.data_ptr(), wsb, *extra, _lib.stream_ptr(x.device)), "dgdm_gemm_tn_split")
            return (d0, d1), db
    if split is None:
        dW = torch.empty(N, K, dtype=torch.float32, device=x.device) if out is None else out
        assert dW.shape == (N, K) and dW.stride(1) == 1
        fn = _gemm_entry(lib, "dgdm_gemm_tn", math)
        TIMERS.timed("gemm_tn", lambda: _lib.check(
            fn(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dW.data_ptr(), dW.stride(0), _lib.ptr(db), M, N, K,
               ws.data_ptr(), wsb, *extra, _lib.stream_ptr(x.device)), "dgdm_gemm_tn"))
        return dW, db
    if not 0 < split < K:
        raise ValueError(f"split must lie inside (0, {K}), got {split}")
    dW0 = torch.empty(N, split, dtype=torch.float32, device=x.device)
    dW1 = torch.empty(N, K - split, dtype=torch.float32, device=x.device)
    fn = _gemm_entry(lib, "dgdm_gemm_tn_split", math)
    TIMERS.timed("gemm_tn", lambda: _lib.check(
        fn(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dW0.data_ptr(), dW0.stride(0), split, dW1.data_ptr(), dW1.stride(0),
           _lib.ptr(db), M, N, K, ws.data_ptr(), wsb, *extra, _lib.stream_ptr(x.device)), "dgdm_gemm_tn_split"))
    return (dW0, dW1), db


# Arithmetic of the dense contractions (``configure(gemm=...)``; both are this library's kernels):
#   "bf16x3" (shipped) csrc/gemm3.hip: exact three-way bf16 split of every fp32 operand, six bf16 MFMAs per product,
#            fp32 accumulate -- fp32-level accuracy at 2-2.5x the fp32 matrix rate;
#   "f16x2"  csrc/gemm_h.hip: fp16 hi + lo operands (22 significand bits), three fp16 MFMAs per product, each operand scaled by a
#            power of two derived from its absolute maximum (kept by the producing kernels / one reduction launch otherwise);
#   "fp32"   csrc/gemm.hip: fp32 MFMA for all three contractions.
GEMM_MATH = "f16x2"


def configure(attention: Optional[str] = None, gemm: Optional[str] = None) -> dict:
    """Select the arithmetic of the attention kernels ("fp16x2" | "fp32") and of the tile GEMMs ("bf16x3" | "fp32") for the
    calls that follow; returns the previous setting (pass it back as ``configure(**prev)`` to restore)."""
    global ATTN_PRECISION, GEMM_MATH
    prev = dict(attention=ATTN_PRECISION, gemm=GEMM_MATH)
    if attention is not None:
        if attention not in ("fp16x2", "fp32"):
            raise ValueError(f"attention precision must be 'fp16x2' or 'fp32', got {attention!r}")
        ATTN_PRECISION = attention
    if gemm is not None:
        if gemm not in ("bf16x3", "f16x2", "fp32"):
            raise ValueError(f"GEMM math must be 'bf16x3', 'f16x2' or 'fp32', got {gemm!r}")
        GEMM_MATH = gemm
    return prev


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.leaf = w.is_leaf and (b is None or b.is_leaf)     # gradients go straight to parameters: their reduction may be deferred
        ctx.wb = (w, b) if ctx.leaf else None                 # ... if the engine only stores them (_claim_deferred looks at these)
        x, w = _rm_tagged(x), _rm_tagged(w)
        ctx.save_for_backward(x, w)
        ctx.has_bias, ctx.math = b is not None, GEMM_MATH
        y = gemm_nt_raw(x, w, b, math=GEMM_MATH)
        ctx.amax = (amax_handle(x), amax_handle(w))      # slots the forward GEMM used (or filled): the backward GEMMs reuse them while live
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        tag_amax(x, ctx.amax[0]); tag_amax(w, ctx.amax[1])
        gy = _rm_tagged(gy)
        dx = gemm_nn_raw(gy, w, math=ctx.math) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW, db = gemm_tn_raw(gy, x, ctx.has_bias, math=ctx.math, may_defer=ctx.leaf and _claim_deferred(*ctx.wb))
        return dx, dW, db


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.size(0) > 1 else max(t.size(1), t.stride(0))


def _rows(t: torch.Tensor) -> torch.Tensor:
    """2-D fp32 view with unit column stride (any row stride): what the small-M kernels address."""
    if t.dtype != torch.float32:
        raise _lib.DGDMKernelError(f"HIP kernels compute in fp32, got {t.dtype}")
    return t if (t.dim() == 2 and t.stride(1) == 1 and (t.size(0) <= 1 or t.stride(0) >= t.size(1))) else t.contiguous()


def linear_small_fwd_raw(x, w, b, act: int = ACT_NONE, want_pre: bool = False):
    lib = _lib.load()
    x, w = _rows(x), _rows(w)
    _lib.require_cuda(x, w, b)
    M, K = x.shape
    N = w.size(0)
    if w.size(1) != K:
        raise ValueError(f"weight {tuple(w.shape)} does not match the input {tuple(x.shape)}")
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    pre = torch.empty(M, N, dtype=torch.float32, device=x.device) if want_pre else None
    b = _f32c(b) if b is not None else None
    _lib.check(lib.dgdm_linear_small_fwd(x.data_ptr(), _ld(x), w.data_ptr(), _ld(w), _lib.ptr(b), M, N, K, act, y.data_ptr(), N,
                                         _lib.ptr(pre), N, _lib.stream_ptr(x.device)), "dgdm_linear_small_fwd")
    return y, pre


class _LinearSmall(torch.autograd.Function):
    """act(x w^T + b) for few rows (csrc/smallm.hip): exact fp32, fixed-order reductions.  ``w`` may be a column block of a larger
    weight (row stride > K); its gradient is then returned as a dense [N, K] block and autograd assembles the full matrix."""

    @staticmethod
    def forward(ctx, x, w, b, act: int):
        x, w = _rows(x), _rows(w)
        y, pre = linear_small_fwd_raw(x, w, b, act, want_pre=act != ACT_NONE)
        ctx.save_for_backward(x, w, pre)
        ctx.act, ctx.has_bias = act, b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, w, pre = ctx.saved_tensors
        gy = _rows(gy)
        M, K = x.shape
        N = w.size(0)
        dev = x.device
        dx = torch.empty(M, K, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dw = torch.empty(N, K, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        db = torch.empty(N, dtype=torch.float32, device=dev) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        _lib.check(lib.dgdm_linear_small_bwd(gy.data_ptr(), _ld(gy), _lib.ptr(pre), N, ctx.act, x.data_ptr(), _ld(x), w.data_ptr(), _ld(w),
                                             M, N, K, _lib.ptr(dx), K, _lib.ptr(dw), K, _lib.ptr(db), _lib.stream_ptr(dev)),
                   "dgdm_linear_small_bwd")
        return dx, dw, db, None


SMALL_MAX_K = 2048


def linear_small(x, weight, bias=None, act: int = ACT_NONE):
    return _LinearSmall.apply(x, weight, bias, act)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x @ weight^T + bias on this library's kernels: the tile GEMMs (``GEMM_MATH``) for >= GEMM_MIN_ROWS rows with K and N
    multiples of 4, the exact-fp32 small-M kernels otherwise.  There is no library / CPU fallback."""
    if x.dim() != 2 or x.dtype != torch.float32:
        raise _lib.DGDMKernelError(f"linear expects a 2-D fp32 matrix, got {x.dtype} {tuple(x.shape)}")
    _lib.require_cuda(x, weight, bias)
    if x.size(0) >= GEMM_MIN_ROWS and x.size(1) % 4 == 0 and weight.size(0) % 4 == 0:
        return _Linear.apply(x, weight, bias)
    if x.size(1) > SMALL_MAX_K:
        raise _lib.DGDMKernelError(f"no kernel for a [{x.size(0)}, {x.size(1)}] x [{weight.size(0)}, {weight.size(1)}]^T contraction "
                                   "(tile GEMMs need K, N % 4 == 0 and >= 256 rows; the small-M kernels K <= 2048)")
    return _LinearSmall.apply(x, weight, bias, ACT_NONE)


class _LinearAddInto(torch.autograd.Function):
    """acc + x w^T + b, accumulated by the GEMM's epilogue INTO ``acc`` (FeatureEncoder: encoder output + residual projection,
    models/encoders.py:119-122) -- no separate element-wise add, no third [N, C] buffer."""

    @staticmethod
    def forward(ctx, acc, x, w, b):
        ctx.leaf = w.is_leaf and (b is None or b.is_leaf)
        ctx.wb = (w, b) if ctx.leaf else None
        x, w = _rm_tagged(x), _rm_tagged(w)
        ctx.save_for_backward(x, w)
        ctx.has_bias, ctx.math = b is not None, GEMM_MATH
        gemm_nt_raw(x, w, b, out=acc, accumulate=True, math=GEMM_MATH)
        ctx.amax = (amax_handle(x), amax_handle(w))
        ctx.mark_dirty(acc)                     # bumps acc's version: a maximum tagged before no longer applies (amax_of checks)
        return acc

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        tag_amax(x, ctx.amax[0]); tag_amax(w, ctx.amax[1])
        gy = _rm_tagged(gy)
        dx = gemm_nn_raw(gy, w, math=ctx.math) if ctx.needs_input_grad[1] else None
        dW = db = None
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            dW, db = gemm_tn_raw(gy, x, ctx.has_bias, math=ctx.math, may_defer=ctx.leaf and _claim_deferred(*ctx.wb))
        return gy, dx, dW, db


def linear_add_into(acc: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``acc += x @ weight^T + bias`` in place (returns ``acc``); falls back to a separate add for shapes outside the tile GEMMs."""
    if (x.dim() == 2 and x.size(0) >= GEMM_MIN_ROWS and x.size(1) % 4 == 0 and weight.size(0) % 4 == 0 and acc.is_contiguous()
            and acc.dtype == torch.float32 and not acc.is_leaf):
        return _LinearAddInto.apply(acc, x, weight, bias)
    return acc + linear(x, weight, bias)


class _DenoiseFirstLayer(torch.autograd.Function):
    """First Linear of the denoiser on [x_t | t_emb] (core/diffusion.py:165-170) without the concat:
        h[n] = x_t[n] W[:, :C]^T + (te[g(n)] W[:, C:]^T + b)
    as ONE autograd node, so that the gradient of W is written once, in place, as a whole matrix (the column block of the node half
    by the split-M GEMM, the block of the time half by the small-M kernel) instead of two slice gradients that autograd pads and adds."""

    @staticmethod
    def forward(ctx, x, te, w, b, plan: AttnPlan, gamma=None, beta=None, norm=None):
        """``norm`` = (groups, eps, act, drop_p, seed) with ``gamma`` / ``beta``: the GroupNorm + SiLU + dropout behind the layer
        (core/diffusion.py:96-98) runs as the GEMM's epilogue, the per-graph time bias enters there as a per-segment residual --
        ONE launch instead of GEMM + broadcast add + row norm."""
        lib = _lib.load()
        x, te = _rm_tagged(x), _rows(te)
        C = x.size(1)
        wx, wt = tag_amax(w[:, :C], amax_handle(w)), w[:, C:]                       # a column block is bounded by the whole matrix's maximum
        pg, _ = linear_small_fwd_raw(te, wt, b)                                # [B, N_out]
        ctx.plan, ctx.math, ctx.has_bias, ctx.norm = plan, GEMM_MATH, b is not None, norm
        ctx.leaf = w.is_leaf and (gamma is None or (gamma.is_leaf and beta.is_leaf))
        ctx.wb = (w,) if ctx.leaf else None
        ctx.gb = (gamma, beta) if (ctx.leaf and gamma is not None) else None
        if norm is not None:
            groups, eps, act, drop_p, seed = norm
            out, ssum, mean, rstd = gemm_img_norm_raw(x, WEIGHT_IMAGES.get(0, wx), w.size(0), None, pg, gamma, beta, groups, eps, act, drop_p,
                                                      seed, res_plan=plan)
            ctx.amax = (amax_handle(x), amax_handle(wx))
            ctx.save_for_backward(x, te, w, ssum, mean, rstd, gamma, beta)
            return out
        h = gemm_nt_raw(x, wx, None, math=GEMM_MATH) if x.size(0) >= GEMM_MIN_ROWS else linear_small_fwd_raw(x, wx, None)[0]
        ctx.amax = (amax_handle(x), amax_handle(wx))
        out = torch.empty_like(h)
        _lib.check(lib.dgdm_segment_bcast_add(h.data_ptr(), pg.data_ptr(), plan.ptr_dev.data_ptr(), plan.B, h.size(0), h.size(1), out.data_ptr(),
                                              _lib.stream_ptr(x.device)), "dgdm_segment_bcast_add")
        ctx.save_for_backward(x, te, w)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        dgam = dbet = None
        if ctx.norm is not None:
            x, te, w, ssum, mean, rstd, gamma, beta = ctx.saved_tensors
            groups, eps, act, drop_p, seed = ctx.norm
            g, dgam, dbet = rownorm_bwd_raw(ssum, None, gamma, beta, mean, rstd, g, groups, act, drop_p, seed, ctx.gb)
        else:
            x, te, w = ctx.saved_tensors
        plan, C = ctx.plan, x.size(1)
        g = _rm_tagged(g)
        tag_amax(x, ctx.amax[0])
        N_out, K = w.shape
        dev = x.device
        big = x.size(0) >= GEMM_MIN_ROWS
        dx = None
        if ctx.needs_input_grad[0]:
            dx = gemm_nn_raw(g, tag_amax(w[:, :C], ctx.amax[1]), math=ctx.math) if big else _small_dx(g, w[:, :C])
        gpg = segment_sum_raw(g, plan)                                         # [B, N_out]: gradient of the per-graph bias
        dw = torch.empty(N_out, K, dtype=torch.float32, device=dev)
        if big:
            gemm_tn_raw(g, x, False, math=ctx.math, out=dw[:, :C], may_defer=ctx.leaf and _claim_deferred(*ctx.wb))
        else:
            _lib.check(lib.dgdm_linear_small_bwd(g.data_ptr(), _ld(g), None, 0, ACT_NONE, x.data_ptr(), _ld(x), None, 0, x.size(0), N_out, C,
                                                 None, 0, dw.data_ptr(), K, None, _lib.stream_ptr(dev)), "dgdm_linear_small_bwd")
        dte = torch.empty_like(te) if ctx.needs_input_grad[1] else None
        db = torch.empty(N_out, dtype=torch.float32, device=dev) if ctx.has_bias else None
        wt = w[:, C:]
        _lib.check(lib.dgdm_linear_small_bwd(gpg.data_ptr(), _ld(gpg), None, 0, ACT_NONE, te.data_ptr(), _ld(te), wt.data_ptr(), _ld(wt), te.size(0),
                                             N_out, K - C, _lib.ptr(dte), K - C, dw[:, C:].data_ptr(), K, _lib.ptr(db), _lib.stream_ptr(dev)),
                   "dgdm_linear_small_bwd")
        return dx, dte, dw, db, None, dgam, dbet, None


def _small_dx(g, w):
    lib = _lib.load()
    g, w = _rows(g), _rows(w)
    dx = torch.empty(g.size(0), w.size(1), dtype=torch.float32, device=g.device)
    _lib.check(lib.dgdm_linear_small_bwd(g.data_ptr(), _ld(g), None, 0, ACT_NONE, None, 0, w.data_ptr(), _ld(w), g.size(0), w.size(0), w.size(1),
                                         dx.data_ptr(), w.size(1), None, 0, None, _lib.stream_ptr(g.device)), "dgdm_linear_small_bwd")
    return dx


def denoise_first_layer(x, te, weight, bias, plan: AttnPlan, norm=None, drop_p: float = 0.0, training: bool = False):
    """First denoiser Linear on [x_t | t_emb]; with ``norm`` (the nn.GroupNorm behind it) also that norm + SiLU + dropout
    (core/diffusion.py:94-98) -- as the GEMM's epilogue where the shape is taken, by the row-norm kernel otherwise."""
    if weight.size(0) % 4 or x.size(1) % 4 or (weight.size(1) - x.size(1)) > SMALL_MAX_K or not weight.is_contiguous():
        raise _lib.DGDMKernelError("denoiser widths must be multiples of 4 (and the time embedding at most 2048 wide)")
    if norm is None:
        return _DenoiseFirstLayer.apply(x, te, weight, bias, plan)
    p = float(drop_p) if training else 0.0
    if (x.size(0) >= GEMM_MIN_ROWS and epilogues_available(x.size(1)) and gemm_img_norm_supported(weight.size(0), norm.num_groups)):
        return _DenoiseFirstLayer.apply(x, te, weight, bias, plan, norm.weight, norm.bias,
                                        (norm.num_groups, norm.eps, ACT_SILU, p, next_dropout_seed() if p > 0 else 0))
    h = _DenoiseFirstLayer.apply(x, te, weight, bias, plan)
    return row_norm(h, norm.weight, norm.bias, groups=norm.num_groups, eps=norm.eps, act=ACT_SILU, drop_p=p, training=training)


def lin(module, x: torch.Tensor) -> torch.Tensor:
    """Apply an nn.Linear through `linear`."""
    return linear(x, module.weight, module.bias)


class _GraphConvLinear(torch.autograd.Function):
    """GraphConvolution in one autograd node:  out = [A_hat x | EA_hat] . [W | W_e]^T + b  (graph_layers.py:89-110).

    forward : one SpMM that also lays EA_hat next to its result (dgdm_spmm_concat), one cat of the two weights, one GEMM;
    backward: dW and dW_e leave the split-M GEMM as two contiguous matrices (no slicing copies), the input gradient is
              contracted over W alone (EA_hat depends only on the inputs) and scattered back by the transposed SpMM."""

    @staticmethod
    def forward(ctx, x, ea_hat, gs: GraphStructure, w, we, b, skip: bool = False):
        """``skip=True`` also returns ``x`` itself as a second output: a caller that feeds x to a residual connection as well
        uses that alias, so both gradients of x arrive here and are added while the scattered one is written (dgdm_spmm_add)."""
        lib = _lib.load()
        x_in = x
        x = x if (x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0) else _f32c(x)
        ea_hat = _rowmajor(ea_hat)
        n, cin, ed = gs.num_nodes, x.size(1), ea_hat.size(1)
        buf = torch.empty(n, cin + ed, dtype=torch.float32, device=x.device)
        slot = new_amax_slot(x.device)
        TIMERS.timed(f"spmm_c{cin}", lambda: _lib.check(
            lib.dgdm_spmm_concat(gs.rowptr.data_ptr(), gs.col.data_ptr(), gs.w.data_ptr(), x.data_ptr(), x.stride(0), x.size(0),
                                 ea_hat.data_ptr(), ea_hat.stride(0), ed, buf.data_ptr(), buf.stride(0), n, cin, slot,
                                 _lr(gs.long_rows()), _lib.stream_ptr(x.device)), "dgdm_spmm_concat"))
        tag_amax(buf, slot)
        ctx.gs, ctx.cin, ctx.has_bias, ctx.skip, ctx.math = gs, cin, b is not None, skip, GEMM_MATH
        ctx.amax = (None, None)
        ctx.leaf = w.is_leaf and we.is_leaf and (b is None or b.is_leaf)
        ctx.wb = (w, we, b) if ctx.leaf else None
        if GEMM_MATH in ("bf16x3", "f16x2") and cin % 4 == 0:
            w = _rm_tagged(w)
            ctx.save_for_backward(buf, w)          # the kernel reads the two weights side by side: no concatenated copy
            y = gemm_nt_split_raw(buf, w, we, b, math=GEMM_MATH)
            ctx.amax = (amax_handle(buf), amax_handle(w))
        else:
            wcat = torch.cat([w, we], dim=1)
            ctx.save_for_backward(buf, wcat)
            y = gemm_nt_raw(buf, wcat, b, math="fp32")
        return (y, x_in.view_as(x_in)) if skip else y

    @staticmethod
    def backward(ctx, gy, gskip=None):
        if gy is None:      # only the alias of x was used downstream
            return gskip, None, None, None, None, None, None
        buf, wsaved = ctx.saved_tensors
        gs, cin = ctx.gs, ctx.cin
        gy = _rm_tagged(gy)
        tag_amax(buf, ctx.amax[0]); tag_amax(wsaved, ctx.amax[1])
        math = ctx.math
        dx = None
        if ctx.needs_input_grad[0]:
            # node_lin.weight itself, or a view into the concatenated copy (bounded by the copy's maximum)
            w_only = wsaved if wsaved.size(1) == cin else tag_amax(wsaved[:, :cin], amax_handle(wsaved))
            dagg = gemm_nn_raw(gy, w_only, math=math)
            dx = spmm_raw(gs.rowptr_t, gs.col_t, gs.w_t, dagg, gs.num_nodes, addend=gskip, long_rows=gs.long_rows(True))
        dw = dwe = db = None
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[4] or (ctx.has_bias and ctx.needs_input_grad[5]):
            (dw, dwe), db = gemm_tn_raw(gy, buf, ctx.has_bias, math=math, split=cin, may_defer=ctx.leaf and _claim_deferred(*ctx.wb))
        return dx, None, None, dw, dwe, db, None


def graph_conv_linear(x, ea_hat, gs: GraphStructure, w, we, b, skip: bool = False):
    return _GraphConvLinear.apply(x, ea_hat, gs, w, we, b, skip)


# ----------------------------------------------------------------------------- K3 + K6 in one launch: GEMMs with fused epilogues
# Round 5, measured (tools/microbench_epilogues.py -> profiles/r05_epilogue_microbench.txt; tools/ab_ws.sh -> profiles/r05_fusion_ab.txt):
# the fused epilogues are correct (tests/test_hip_gemm_img.py, test_hip_graph_ops.py) and SLOWER than the launches they replace
# wherever the GEMM has fewer waves than the chip has SIMDs: a wave finishes a 32 x 128 tile = 64 outputs per lane alone
# (GELU + dropout: ~28 vector instructions per output, 4 cycles each = 3.5 us of one SIMD), while the streaming kernel spreads the
# same instructions over all 1024 SIMDs.  At 5 000 rows: GEMM 7.3 us, + activation kernel 10.5, fused 16.9; at 40 000 rows a draw
# (22.3 / 22.3); LayerNorm epilogue 29.4 against 25.9; whole step 13.07 against 13.00 ms.  So the default keeps every activation /
# norm behind a GEMM in a kernel of its own; True selects the fused path (same-box A/B, the parity tests run both).
# (Round 5 also had an "auto" policy -- activation backwards as epilogues from 20 000 rows on; a same-box A/B gave 12.63 against 12.63 ms
# per step, a draw to four digits, profiles/r05_fuse_auto_ab.txt: removed in round 6.  True stays as the A/B switch, like the
# arithmetic variants of ops.configure.)
FUSE_EPILOGUES = False


def epilogues_available(*widths: int) -> bool:
    """The fused-epilogue GEMMs are the weight-image kernels: fp16 hi+lo arithmetic, every reduction length a multiple of 16."""
    return FUSE_EPILOGUES is True and USE_WEIGHT_IMAGES and GEMM_MATH == "f16x2" and all(w % 16 == 0 and w >= 16 for w in widths)


def _act_raw(pre, act: int, drop_p: float, seed: int):
    """dropout(act(pre)) by the streaming kernel (what a fused forward epilogue replaces); the result carries its operand maximum."""
    y = torch.empty_like(pre)
    slot = new_amax_slot(pre.device)
    _lib.check(_lib.load().dgdm_act_dropout_fwd(pre.data_ptr(), pre.numel(), act, drop_p, seed, y.data_ptr(), None, slot,
                                                _lib.stream_ptr(pre.device)), "dgdm_act_dropout_fwd")
    return tag_amax(y, slot)


def _act_bwd_raw(pre, g, act: int, drop_p: float, seed: int):
    """g * act'(pre) * mask in place of g (the streaming backward kernel)."""
    slot = new_amax_slot(pre.device)
    _lib.check(_lib.load().dgdm_act_dropout_bwd(pre.data_ptr(), g.data_ptr(), pre.numel(), act, drop_p, seed, g.data_ptr(), None, slot,
                                                _lib.stream_ptr(pre.device)), "dgdm_act_dropout_bwd")
    g._dgdm_amax = None
    return tag_amax(g, slot)


def gemm_img_act_raw(a, e: "_WImage", ncols: int, bias, act: int, drop_p: float, seed: int, want_pre: bool = True):
    """(Y, pre): pre = a . B + bias, Y = dropout(act(pre)) from ONE launch (dgdm_gemm_rows_img_act); Y carries its operand maximum."""
    M, K = a.shape
    y = torch.empty(M, ncols, dtype=torch.float32, device=a.device)
    pre = torch.empty_like(y) if want_pre else None
    slot = new_amax_slot(a.device)
    TIMERS.timed("gemm_img_act", lambda: _lib.check(_lib.load().dgdm_gemm_rows_img_act(
        a.data_ptr(), a.stride(0), M, K, e.img.data_ptr(), e.tiles, 0, ncols, _lib.ptr(bias), _lib.ptr(pre), ncols, y.data_ptr(), ncols,
        act, drop_p, seed, ensure_amax(a), slot, _lib.stream_ptr(a.device)), "dgdm_gemm_rows_img_act"))
    return tag_amax(y, slot), pre


def gemm_img_act_bwd_raw(a, e: "_WImage", ncols: int, pre, act: int, drop_p: float, seed: int):
    """G = (a . B) * act'(pre) * mask (dgdm_gemm_rows_img_act_bwd): the activation's backward as the epilogue of the GEMM that
    forms its incoming gradient."""
    M, K = a.shape
    g = torch.empty(M, ncols, dtype=torch.float32, device=a.device)
    slot = new_amax_slot(a.device)
    TIMERS.timed("gemm_img_act_bwd", lambda: _lib.check(_lib.load().dgdm_gemm_rows_img_act_bwd(
        a.data_ptr(), a.stride(0), M, K, e.img.data_ptr(), e.tiles, 0, ncols, pre.data_ptr(), pre.stride(0), g.data_ptr(), ncols,
        act, drop_p, seed, ensure_amax(a), slot, _lib.stream_ptr(a.device)), "dgdm_gemm_rows_img_act_bwd"))
    return tag_amax(g, slot)


def gemm_img_norm_supported(ncols: int, groups: int) -> bool:
    return bool(_lib.load().dgdm_gemm_rows_img_norm_supported(ncols, groups))


def gemm_img_norm_raw(a, e: "_WImage", ncols: int, bias, res, gamma, beta, groups: int, eps: float, act: int = ACT_NONE,
                      drop_p: float = 0.0, seed: int = 0, want_sum: bool = True, res_plan: Optional[AttnPlan] = None,
                      pre_drop_p: float = 0.0, pre_seed: int = 0):
    """(Y, S, mean, rstd): S = dropout_pre(a . B + bias) [+ res], Y = dropout(act(norm_groups(S) * gamma + beta)) from ONE launch
    (dgdm_gemm_rows_img_norm).  S is what ``rownorm_bwd_raw`` takes as the norm's input (with res=None).  ``res_plan``: ``res`` has
    one row per graph of the plan (broadcast over the graph's rows)."""
    M, K = a.shape
    y = torch.empty(M, ncols, dtype=torch.float32, device=a.device)
    ssum = torch.empty_like(y) if want_sum else None
    mean = torch.empty(M * groups, dtype=torch.float32, device=a.device)
    rstd = torch.empty_like(mean)
    slot = new_amax_slot(a.device)
    rp, rs = (None, 0) if res_plan is None else (res_plan.ptr_dev.data_ptr(), res_plan.B)
    TIMERS.timed("gemm_img_norm", lambda: _lib.check(_lib.load().dgdm_gemm_rows_img_norm(
        a.data_ptr(), a.stride(0), M, K, e.img.data_ptr(), e.tiles, 0, ncols, _lib.ptr(bias), pre_drop_p, pre_seed, _lib.ptr(res),
        0 if res is None else _ld(res), rp, rs, gamma.data_ptr(), beta.data_ptr(), groups, eps, _lib.ptr(ssum), ncols, y.data_ptr(), ncols,
        mean.data_ptr(), rstd.data_ptr(), act, drop_p, seed, ensure_amax(a), slot, _lib.stream_ptr(a.device)), "dgdm_gemm_rows_img_norm"))
    return tag_amax(y, slot), ssum, mean, rstd


class _LinearNorm(torch.autograd.Function):
    """dropout(act(norm_groups(dropout_pre(x W^T + b) [+ res]) * gamma + beta)) with the norm as the GEMM's epilogue: ONE launch for a
    Linear followed by a LayerNorm / GroupNorm site (models/encoders.py:262-269 dim_proj -> LayerNorm -> GELU -> dropout;
    core/diffusion.py:98-102 Linear -> GroupNorm(8) -> SiLU -> dropout; core/attention.py:176-181,325 LN(x + dropout(out_proj(o)))).
    Backward: the row-norm backward kernel on the stored pre-norm sum, then the Linear's two GEMMs."""

    @staticmethod
    def forward(ctx, x, w, b, res, gamma, beta, groups, eps, act, drop_p, seed, pre_drop_p, pre_seed):
        x, w = _rm_tagged(x), _rm_tagged(w)
        n_out = w.size(0)
        y, ssum, mean, rstd = gemm_img_norm_raw(x, WEIGHT_IMAGES.get(0, w), n_out, b, None if res is None else _rowmajor(res), gamma, beta,
                                                groups, eps, act, drop_p, seed, pre_drop_p=pre_drop_p, pre_seed=pre_seed)
        ctx.save_for_backward(x, w, ssum, mean, rstd, gamma, beta)
        ctx.meta = (groups, act, drop_p, seed, pre_drop_p, pre_seed, b is not None, res is not None)
        ctx.amax = (amax_handle(x), amax_handle(w))
        params = (w, b, gamma, beta)
        ctx.leaf = all(p is None or p.is_leaf for p in params)
        ctx.params = params if ctx.leaf else None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, ssum, mean, rstd, gamma, beta = ctx.saved_tensors
        groups, act, drop_p, seed, pre_drop_p, pre_seed, has_bias, has_res = ctx.meta
        tag_amax(x, ctx.amax[0]); tag_amax(w, ctx.amax[1])
        P = ctx.params
        ds, dgam, dbet = rownorm_bwd_raw(ssum, None, gamma, beta, mean, rstd, gy, groups, act, drop_p, seed, (P[2], P[3]) if ctx.leaf else None)
        do = ds
        if pre_drop_p > 0:       # the projection's gradient passes the residual dropout's mask; the residual's does not
            do = torch.empty_like(ds)
            slot = new_amax_slot(ds.device)
            _lib.check(_lib.load().dgdm_act_dropout_bwd(ds.data_ptr(), ds.data_ptr(), ds.numel(), ACT_NONE, pre_drop_p, pre_seed, do.data_ptr(),
                                                        None, slot, _lib.stream_ptr(ds.device)), "dgdm_act_dropout_bwd")
            tag_amax(do, slot)
        dx = gemm_nn_raw(do, w, math="f16x2") if ctx.needs_input_grad[0] else None
        dw, db = gemm_tn_raw(do, x, has_bias, math="f16x2", may_defer=ctx.leaf and _claim_deferred(P[0], P[1]))
        return dx, dw, db, (ds if has_res else None), dgam, dbet, None, None, None, None, None, None, None


def linear_norm(x, weight, bias, norm_weight, norm_bias, *, groups: int = 1, eps: float = 1e-5, res=None, act: int = ACT_NONE,
                drop_p: float = 0.0, pre_drop_p: float = 0.0, training: bool = False):
    """``row_norm(dropout_pre(linear(x, weight, bias)), ..., res=res)``; one launch (the norm as the GEMM's epilogue) where the shape
    is taken, the separate kernels otherwise.  Dropout seeds are drawn in the order the separate kernels draw them."""
    p, pp = (float(drop_p), float(pre_drop_p)) if training else (0.0, 0.0)
    if (x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and x.size(0) >= GEMM_MIN_ROWS and epilogues_available(x.size(1))
            and weight.size(0) % 4 == 0 and gemm_img_norm_supported(weight.size(0), groups)):
        pre_seed = next_dropout_seed() if pp > 0 else 0
        seed = next_dropout_seed() if p > 0 else 0
        return _LinearNorm.apply(x, weight, bias, res, norm_weight, norm_bias, groups, eps, act, p, seed, pp, pre_seed)
    h = linear(x, weight, bias)
    if pp > 0:
        h = act_dropout(h, ACT_NONE, pp, True)
    return row_norm(h, norm_weight, norm_bias, res=res, groups=groups, eps=eps, act=act, drop_p=p, training=training)


def _spmm_concat_raw(x, ea_hat, gs: GraphStructure):
    """[A_hat x | EA_hat] with its operand maximum (dgdm_spmm_concat)."""
    n, cin, ed = gs.num_nodes, x.size(1), ea_hat.size(1)
    buf = torch.empty(n, cin + ed, dtype=torch.float32, device=x.device)
    slot = new_amax_slot(x.device)
    TIMERS.timed(f"spmm_c{cin}", lambda: _lib.check(
        _lib.load().dgdm_spmm_concat(gs.rowptr.data_ptr(), gs.col.data_ptr(), gs.w.data_ptr(), x.data_ptr(), x.stride(0), x.size(0),
                                     ea_hat.data_ptr(), ea_hat.stride(0), ed, buf.data_ptr(), buf.stride(0), n, cin, slot,
                                     _lr(gs.long_rows()), _lib.stream_ptr(x.device)), "dgdm_spmm_concat"))
    return tag_amax(buf, slot)


def _spmm_t_amax_raw(g, gs: GraphStructure):
    """A_hat^T g with its operand maximum (the result feeds a GEMM): dgdm_spmm_concat on the transposed index, no tail."""
    n, c = gs.num_nodes, g.size(1)
    out = torch.empty(n, c, dtype=torch.float32, device=g.device)
    slot = new_amax_slot(g.device)
    TIMERS.timed(f"spmm_c{c}", lambda: _lib.check(
        _lib.load().dgdm_spmm_concat(gs.rowptr_t.data_ptr(), gs.col_t.data_ptr(), gs.w_t.data_ptr(), g.data_ptr(), g.stride(0), g.size(0),
                                     None, 0, 0, out.data_ptr(), out.stride(0), n, c, slot, _lr(gs.long_rows(True)),
                                     _lib.stream_ptr(g.device)), "dgdm_spmm_concat"))
    return tag_amax(out, slot)


class _GraphLayer(torch.autograd.Function):
    """DynamicGraphLayer (core/graph_layers.py:207-247 of the reference) as ONE autograd node:

        norm1( output_proj( drop(GELU( conv2( drop(GELU( conv1(x) )) ) )) ) + x )

    forward, 5 launches (8 before): [A x | EA] -> GEMM + bias + GELU + dropout -> [A h1 | EA] -> GEMM + bias + GELU + dropout ->
    GEMM + bias + residual + LayerNorm (a GEMM and a row-norm launch when the row does not fit one wave's columns);
    backward, 6 launches (8): LayerNorm backward -> GEMM (. W_o) * GELU'(pre2) * mask -> A^T -> GEMM (. W_2) * GELU'(pre1) * mask
    -> GEMM (. W_1) -> A^T + residual gradient; the second convolution's input gradient is formed as (A^T dpre2) . W_2 instead of
    A^T (dpre2 . W_2) -- the same sum, associated so that the activation's backward is a GEMM epilogue.  The three weight-gradient
    GEMMs and the norm's dgamma / dbeta join the pass's deferred reduction launches exactly as the separate nodes did."""

    @staticmethod
    def forward(ctx, x, ea_hat, gs, w1, we1, b1, w2, we2, b2, wo, bo, gamma, beta, eps, drop_p, seeds):
        x = x if (x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0) else _f32c(x)
        ea_hat = _rowmajor(ea_hat)
        w1, w2, wo = _rm_tagged(w1), _rm_tagged(w2), _rm_tagged(wo)
        hid, node = w1.size(0), wo.size(0)
        full = True
        z1 = _spmm_concat_raw(x, ea_hat, gs)
        if full:
            h1, pre1 = gemm_img_act_raw(z1, WEIGHT_IMAGES.get(0, w1, we1), hid, b1, ACT_GELU, drop_p, seeds[0])
        else:
            pre1 = _gemm_rows_img(z1, WEIGHT_IMAGES.get(0, w1, we1), 0, hid, b1, None, False)
            h1 = _act_raw(pre1, ACT_GELU, drop_p, seeds[0])
        z2 = _spmm_concat_raw(h1, ea_hat, gs)
        if full:
            h2, pre2 = gemm_img_act_raw(z2, WEIGHT_IMAGES.get(0, w2, we2), hid, b2, ACT_GELU, drop_p, seeds[1])
        else:
            pre2 = _gemm_rows_img(z2, WEIGHT_IMAGES.get(0, w2, we2), 0, hid, b2, None, False)
            h2 = _act_raw(pre2, ACT_GELU, drop_p, seeds[1])
        fused_norm = full and gemm_img_norm_supported(node, 1)
        if fused_norm:
            y, ssum, mean, rstd = gemm_img_norm_raw(h2, WEIGHT_IMAGES.get(0, wo), node, bo, x, gamma, beta, 1, eps)
            res = None
        else:
            ssum = _gemm_rows_img(h2, WEIGHT_IMAGES.get(0, wo), 0, node, bo, None, False)
            y = torch.empty_like(ssum)
            mean = torch.empty(ssum.size(0), dtype=torch.float32, device=x.device)
            rstd = torch.empty_like(mean)
            slot = new_amax_slot(x.device)
            _lib.check(_lib.load().dgdm_rownorm_fwd(ssum.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ssum.size(0), node, 1,
                                                    eps, ACT_NONE, 0.0, 0, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), slot,
                                                    _lib.stream_ptr(x.device)), "dgdm_rownorm_fwd")
            tag_amax(y, slot)
            res = x
        ctx.save_for_backward(z1, pre1, z2, pre2, h2, ssum, res, mean, rstd, w1, w2, wo, gamma, beta)
        ctx.gs, ctx.meta = gs, (x.size(1), hid, node, drop_p, tuple(seeds))
        ctx.amax = tuple(amax_handle(t) for t in (z1, z2, h2, w1, w2, wo))
        params = (w1, we1, b1, w2, we2, b2, wo, bo, gamma, beta)
        ctx.leaf = all(p.is_leaf for p in params)
        ctx.params = params if ctx.leaf else None
        return y

    @staticmethod
    def backward(ctx, gy):
        z1, pre1, z2, pre2, h2, ssum, res, mean, rstd, w1, w2, wo, gamma, beta = ctx.saved_tensors
        gs = ctx.gs
        cin, hid, node, drop_p, seeds = ctx.meta
        for t, h in zip((z1, z2, h2, w1, w2, wo), ctx.amax):
            tag_amax(t, h)
        P = ctx.params
        claim = lambda *idx: ctx.leaf and _claim_deferred(*(P[i] for i in idx))
        ds, dgam, dbet = rownorm_bwd_raw(ssum, res, gamma, beta, mean, rstd, gy, 1, ACT_NONE, 0.0, 0, (P[8], P[9]) if ctx.leaf else None)
        dwo, dbo = gemm_tn_raw(ds, h2, True, math="f16x2", may_defer=claim(6, 7))
        fuse_bwd = True
        if fuse_bwd:
            dpre2 = gemm_img_act_bwd_raw(ds, WEIGHT_IMAGES.get(1, wo), hid, pre2, ACT_GELU, drop_p, seeds[1])
        else:
            dpre2 = _act_bwd_raw(pre2, _gemm_rows_img(ds, WEIGHT_IMAGES.get(1, wo), 0, hid, None, None, False), ACT_GELU, drop_p, seeds[1])
        (dw2, dwe2), db2 = gemm_tn_raw(dpre2, z2, True, math="f16x2", split=hid, may_defer=claim(3, 4, 5))
        g2 = _spmm_t_amax_raw(dpre2, gs)
        if fuse_bwd:
            dpre1 = gemm_img_act_bwd_raw(g2, WEIGHT_IMAGES.get(1, w2), hid, pre1, ACT_GELU, drop_p, seeds[0])
        else:
            dpre1 = _act_bwd_raw(pre1, _gemm_rows_img(g2, WEIGHT_IMAGES.get(1, w2), 0, hid, None, None, False), ACT_GELU, drop_p, seeds[0])
        (dw1, dwe1), db1 = gemm_tn_raw(dpre1, z1, True, math="f16x2", split=cin, may_defer=claim(0, 1, 2))
        dx = None
        if ctx.needs_input_grad[0]:
            dagg = _gemm_rows_img(dpre1, WEIGHT_IMAGES.get(1, w1), 0, cin, None, None, False)
            dx = spmm_raw(gs.rowptr_t, gs.col_t, gs.w_t, dagg, gs.num_nodes, addend=ds, long_rows=gs.long_rows(True))
        return dx, None, None, dw1, dwe1, db1, dw2, dwe2, db2, dwo, dbo, dgam, dbet, None, None, None


def graph_layer_supported(x, ea_hat, node_dim: int, hidden: int, edge_dim: int) -> bool:
    return (ea_hat is not None and x.dim() == 2 and x.dtype == torch.float32 and x.is_cuda and x.size(0) >= GEMM_MIN_ROWS
            and x.size(1) == node_dim and ea_hat.size(1) == edge_dim
            and epilogues_available(node_dim + edge_dim, hidden + edge_dim, hidden, node_dim))


def graph_layer(x, ea_hat, gs: GraphStructure, conv1, conv2, output_proj, norm, drop_p: float, training: bool):
    """The whole DynamicGraphLayer through `_GraphLayer` (conv1 / conv2: GraphConvolution modules, norm: nn.LayerNorm)."""
    p = float(drop_p) if training else 0.0
    seeds = (next_dropout_seed(), next_dropout_seed()) if p > 0 else (0, 0)
    return _GraphLayer.apply(x, ea_hat, gs, conv1.node_lin.weight, conv1.edge_lin.weight, conv1.bias, conv2.node_lin.weight,
                             conv2.edge_lin.weight, conv2.bias, output_proj.weight, output_proj.bias, norm.weight, norm.bias,
                             norm.eps, p, seeds)


# ----------------------------------------------------------------------------- K9 top-k pooling / unpooling
def pool_supported(C: int, C2: int) -> bool:
    """Shapes the K9 kernels are built for (C = node width, C2 = width of the score MLP's hidden layer)."""
    return C % 4 == 0 and C2 % 4 == 0 and C2 <= 256 and 256 % (C2 // 4) == 0


POOL_NONLINEARITY = {"tanh": 0, "sigmoid": 1, "none": 2}


class _PoolScore(torch.autograd.Function):
    """s = f(w2 . relu(h) + b2), f = tanh / sigmoid / identity; h [N, C2] is the first score layer's output (a GEMM through `lin`)."""

    @staticmethod
    def forward(ctx, h, w2, b2, decide=None, nl: int = 0):
        lib = _lib.load()
        h = _rowmajor(h)
        w2c, b2c = _f32c(w2.reshape(-1)), _f32c(b2.reshape(-1))
        N, C2 = h.shape
        s = torch.empty(N, dtype=torch.float32, device=h.device)
        _lib.check(lib.dgdm_pool_score_fwd(h.data_ptr(), h.stride(0), w2c.data_ptr(), b2c.data_ptr(), N, C2, s.data_ptr(),
                                           _lib.ptr(decide), nl, _lib.stream_ptr(h.device)), "dgdm_pool_score_fwd")
        ctx.save_for_backward(h, w2c, s)
        ctx.w2_shape, ctx.b2_shape, ctx.decide, ctx.nl = w2.shape, b2.shape, decide, nl
        return s

    @staticmethod
    def backward(ctx, ds):
        lib = _lib.load()
        h, w2c, s = ctx.saved_tensors
        N, C2 = h.shape
        ds = _f32c(ds)
        dh = torch.empty_like(h)
        dw2 = torch.empty(C2, dtype=torch.float32, device=h.device)
        db2 = torch.empty(1, dtype=torch.float32, device=h.device)
        wsb = _lib.workspace_bytes("dgdm_pool_score_bwd_workspace_bytes", N, C2)
        ws = torch.empty(max(wsb, 4) // 4, dtype=torch.float32, device=h.device)
        slot = new_amax_slot(h.device)
        _lib.check(lib.dgdm_pool_score_bwd(h.data_ptr(), h.stride(0), w2c.data_ptr(), s.data_ptr(), ds.data_ptr(), N, C2, dh.data_ptr(),
                                           dh.stride(0), dw2.data_ptr(), db2.data_ptr(), _lib.ptr(ctx.decide), ctx.nl, ws.data_ptr(), wsb,
                                           slot, _lib.stream_ptr(h.device)), "dgdm_pool_score_bwd")
        return tag_amax(dh, slot), dw2.view(ctx.w2_shape), db2.view(ctx.b2_shape), None, None


class _VecSoftmax(torch.autograd.Function):
    """softmax over ALL entries of a vector (AdaptiveGraphPooling(nonlinearity='softmax'), graph_layers.py:280-281)."""

    @staticmethod
    def forward(ctx, z):
        z = _f32c(z)
        _lib.require_cuda(z)
        s = torch.empty_like(z)
        _lib.check(_lib.load().dgdm_vec_softmax_fwd(z.data_ptr(), z.numel(), s.data_ptr(), _lib.stream_ptr(z.device)), "dgdm_vec_softmax_fwd")
        ctx.save_for_backward(s)
        return s

    @staticmethod
    def backward(ctx, ds):
        (s,) = ctx.saved_tensors
        ds = _f32c(ds)
        dz = torch.empty_like(s)
        _lib.check(_lib.load().dgdm_vec_softmax_bwd(s.data_ptr(), ds.data_ptr(), s.numel(), dz.data_ptr(), _lib.stream_ptr(s.device)),
                   "dgdm_vec_softmax_bwd")
        return dz


def pool_score(h, w2, b2, decide=None, nonlinearity: str = "tanh"):
    """Node scores of AdaptiveGraphPooling (graph_layers.py:276-296): tanh / sigmoid of the score MLP, or its softmax over all nodes."""
    if nonlinearity == "softmax":
        return _VecSoftmax.apply(_PoolScore.apply(h, w2, b2, _decide_arg(decide, h), POOL_NONLINEARITY["none"]))
    return _PoolScore.apply(h, w2, b2, _decide_arg(decide, h), POOL_NONLINEARITY[nonlinearity])


def count_ge(s: torch.Tensor, threshold: float) -> int:
    """#{i : s[i] >= threshold} -- ONE host sync (min_score pooling keeps a data-dependent number of nodes, graph_layers.py:302-303)."""
    s = _f32c(s.detach())
    out = torch.empty(1, dtype=torch.int32, device=s.device)
    _lib.check(_lib.load().dgdm_count_ge(s.data_ptr(), s.numel(), float(threshold), out.data_ptr(), _lib.stream_ptr(s.device)), "dgdm_count_ge")
    return int(out.item())


def topk_perm(s: torch.Tensor, k: int):
    """Exact top-k of the scores: (perm int64 [k] ascending node ids, node_map int32 [N] new id or -1).
    No host synchronisation; ties at the k-th value keep the lowest ids."""
    lib = _lib.load()
    s = _f32c(s.detach())
    N = s.numel()
    perm = torch.empty(k, dtype=torch.int64, device=s.device)
    node_map = torch.empty(N, dtype=torch.int32, device=s.device)
    wsb = _lib.workspace_bytes("dgdm_topk_perm_workspace_bytes", N)
    ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=s.device)
    _lib.check(lib.dgdm_topk_perm(s.data_ptr(), N, k, perm.data_ptr(), node_map.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr(s.device)),
               "dgdm_topk_perm")
    return perm, node_map


class _PoolGather(torch.autograd.Function):
    """out[j] = x[perm[j]] * s[perm[j]] * mult"""

    @staticmethod
    def forward(ctx, x, s, perm, node_map, mult):
        lib = _lib.load()
        x, s = _rowmajor(x), _f32c(s)
        k, C = perm.numel(), x.size(1)
        out = torch.empty(k, C, dtype=torch.float32, device=x.device)
        _lib.check(lib.dgdm_pool_gather_fwd(x.data_ptr(), x.stride(0), s.data_ptr(), perm.data_ptr(), k, C, float(mult), out.data_ptr(),
                                            out.stride(0), _lib.stream_ptr(x.device)), "dgdm_pool_gather_fwd")
        ctx.save_for_backward(x, s, node_map)
        ctx.mult = float(mult)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, s, node_map = ctx.saved_tensors
        g = _rowmajor(g)
        N, C = x.shape
        dx = torch.empty(N, C, dtype=torch.float32, device=x.device)
        ds = torch.empty(N, dtype=torch.float32, device=x.device)
        _lib.check(lib.dgdm_pool_gather_bwd(g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), s.data_ptr(), node_map.data_ptr(), N, C,
                                            ctx.mult, dx.data_ptr(), dx.stride(0), ds.data_ptr(), _lib.stream_ptr(x.device)),
                   "dgdm_pool_gather_bwd")
        return dx, ds, None, None, None


def pool_gather(x, s, perm, node_map, mult: float = 1.0):
    return _PoolGather.apply(x, s, perm, node_map, mult)


def edge_relabel(edge_index: torch.Tensor, node_map: torch.Tensor) -> torch.Tensor:
    """[2, E] int64 -> [2, E] int64 in the pooled numbering; edges with a dropped end become (-1, -1)."""
    lib = _lib.load()
    ei = edge_index.contiguous()
    out = torch.empty_like(ei)
    _lib.check(lib.dgdm_edge_relabel(ei.data_ptr(), ei.size(1), node_map.data_ptr(), node_map.numel(), out.data_ptr(),
                                     _lib.stream_ptr(ei.device)), "dgdm_edge_relabel")
    return out


class _UnpoolAddRelu(torch.autograd.Function):
    """relu(skip + unpool(xc)): unpool = zero-fill + index write of the reference (graph_layers.py:441-444)."""

    @staticmethod
    def forward(ctx, xc, skip, node_map, decide=None):
        lib = _lib.load()
        _lib.require_cuda(xc, skip, node_map)          # safe on its own: no earlier layer has to have raised first
        xc, skip = _rowmajor(xc), _rowmajor(skip)
        N, C = skip.shape
        if C % 4 or xc.size(1) != C or node_map.numel() != N:
            raise _lib.DGDMKernelError(f"unpool_add_relu: widths must match and be multiples of 4 (got {tuple(xc.shape)} onto {tuple(skip.shape)})")
        out = torch.empty(N, C, dtype=torch.float32, device=skip.device)
        _lib.check(lib.dgdm_unpool_add_relu_fwd(xc.data_ptr(), xc.stride(0), skip.data_ptr(), skip.stride(0), node_map.data_ptr(), N, C,
                                                out.data_ptr(), out.stride(0), _lib.ptr(decide), _lib.stream_ptr(skip.device)),
                   "dgdm_unpool_add_relu_fwd")
        ctx.save_for_backward(out, node_map)
        ctx.k, ctx.decide = xc.size(0), decide
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        out, node_map = ctx.saved_tensors
        g = _rowmajor(g)
        N, C = out.shape
        dskip = torch.empty(N, C, dtype=torch.float32, device=out.device)
        dxc = torch.empty(ctx.k, C, dtype=torch.float32, device=out.device)
        _lib.check(lib.dgdm_unpool_add_relu_bwd(g.data_ptr(), g.stride(0), out.data_ptr(), out.stride(0), node_map.data_ptr(), N, C,
                                                dskip.data_ptr(), dskip.stride(0), dxc.data_ptr(), dxc.stride(0), _lib.ptr(ctx.decide),
                                                _lib.stream_ptr(out.device)), "dgdm_unpool_add_relu_bwd")
        return dxc, dskip, None, None


def unpool_add_relu(xc, skip, node_map, decide=None):
    return _UnpoolAddRelu.apply(xc, skip, node_map, _decide_arg(decide, skip))

"""Graph containers for the DGDM hot path.

``GraphData`` / ``GraphBatch`` reproduce the batch layout the reference's data pipeline hands
to ``DGDMModel.forward`` (torch_geometric ``Data`` / ``Batch.from_data_list``, used at
data/datamodule.py:9,173 and tests/test_basic.py:238-255): node tensors concatenated, every
graph's ``edge_index`` offset by the number of nodes before it, a sorted ``batch`` vector and
``ptr`` offsets.  The model duck-types its input, so a real PyG ``Batch`` works as well.

``GraphStructure`` is the device-side index set one (batched) edge list needs: CSR by
destination (forward aggregation), CSR by source (backward), GCN weights -- built once per
batch by the K1 kernels and reused by every graph convolution and its backward.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import _lib


class GraphData:
    """One tissue graph: x [N,F], edge_index [2,E] int64 (row 0 = source, row 1 = destination),
    optional edge_attr [E,32], pos [N,2], y."""

    _fields = ("x", "edge_index", "edge_attr", "pos", "y", "batch", "ptr")

    def __init__(self, x=None, edge_index=None, edge_attr=None, pos=None, y=None, **extra):
        self.x, self.edge_index, self.edge_attr, self.pos, self.y = x, edge_index, edge_attr, pos, y
        self.batch = None
        self.ptr = None
        for k, v in extra.items():
            setattr(self, k, v)

    @property
    def num_nodes(self) -> int:
        return 0 if self.x is None else self.x.size(0)

    def host_max_degree(self) -> Optional[int]:
        """``max_degree`` if set, else the largest in- / out-degree of ``edge_index`` while it still lives on the HOST (no device sync);
        else None.  GraphStructure sizes its long-row tables by it."""
        d = getattr(self, "max_degree", None)
        if d is None and isinstance(self.edge_index, torch.Tensor) and not self.edge_index.is_cuda:
            if self.edge_index.numel() == 0:
                return 0
            n = max(self.num_nodes, int(self.edge_index.max()) + 1)
            d = int(max(torch.bincount(self.edge_index[0], minlength=n).max(), torch.bincount(self.edge_index[1], minlength=n).max()))
        return d

    def host_pos_extent(self) -> Optional[float]:
        """``pos_extent`` if set, else the largest coordinate range of ``pos`` when it still lives on the HOST (a loader calls this
        before the upload: no device sync), rounded up to a power of two so that batches of one data set share one value; else None.
        An upper bound of the coordinate range inside any graph: the spatial attention uses it to tell, without looking at the
        device, whether any of its block pairs can be exactly zero (ops.attn_zero_blocks_possible)."""
        e = getattr(self, "pos_extent", None)
        if e is None and isinstance(self.pos, torch.Tensor) and not self.pos.is_cuda and self.pos.numel() > 0:
            e = float((self.pos.max(dim=0).values - self.pos.min(dim=0).values).max())
        if e is None:
            return None
        import math
        return float(2.0 ** math.ceil(math.log2(e))) if e > 0 else 0.0

    @property
    def num_edges(self) -> int:
        return 0 if self.edge_index is None else self.edge_index.size(1)

    @property
    def num_graphs(self) -> int:
        return 1 if self.ptr is None else len(self.ptr) - 1

    def _apply(self, fn):
        out = self.__class__.__new__(self.__class__)
        for k, v in self.__dict__.items():
            setattr(out, k, fn(v) if isinstance(v, torch.Tensor) else v)
        return out

    def clone(self):
        return self._apply(lambda t: t.clone())

    def to(self, device, non_blocking: bool = False):
        return self._apply(lambda t: t.to(device, non_blocking=non_blocking))

    def pin_memory(self):
        return self._apply(lambda t: t.pin_memory())


class GraphBatch(GraphData):
    """Block-diagonal batch of graphs (PyG collation rule)."""

    @classmethod
    def from_data_list(cls, graphs: Sequence[GraphData]) -> "GraphBatch":
        if len(graphs) == 0:
            raise ValueError("cannot batch an empty list of graphs")
        xs, eis, eas, poss, ys, bvec, ptr = [], [], [], [], [], [], [0]
        for g, d in enumerate(graphs):
            n = d.x.size(0)
            xs.append(d.x)
            eis.append(d.edge_index + ptr[-1])
            if d.edge_attr is not None:
                eas.append(d.edge_attr)
            if d.pos is not None:
                poss.append(d.pos)
            if getattr(d, "y", None) is not None:
                ys.append(d.y.reshape(1, -1) if d.y.dim() <= 1 else d.y)
            bvec.append(torch.full((n,), g, dtype=torch.long, device=d.x.device))
            ptr.append(ptr[-1] + n)
        for name, lst in (("edge_attr", eas), ("pos", poss)):
            if lst and len(lst) != len(graphs):
                raise ValueError(f"either every graph or no graph of a batch may carry {name}")
        out = cls(x=torch.cat(xs), edge_index=torch.cat(eis, dim=1),
                  edge_attr=torch.cat(eas) if eas else None, pos=torch.cat(poss) if poss else None,
                  y=torch.cat(ys) if ys else None)
        out.batch = torch.cat(bvec)
        out.ptr = ptr  # host-side python ints: the per-graph node offsets
        ext = [d.host_pos_extent() if isinstance(d, GraphData) else getattr(d, "pos_extent", None) for d in graphs] if poss else []
        out.pos_extent = max(ext) if ext and all(e is not None for e in ext) else None
        deg = [d.host_max_degree() if isinstance(d, GraphData) else getattr(d, "max_degree", None) for d in graphs]
        out.max_degree = max(deg) if all(v is not None for v in deg) else None
        return out


def graph_ptr(data, num_nodes: int) -> List[int]:
    """Per-graph node offsets [B+1] as host ints.  Uses ``data.ptr`` when the batch carries it
    (no device sync); otherwise derives it from the sorted ``batch`` vector (one sync, once per
    batch -- the reference syncs B times per stage, dgdm_model.py:342,410,604).  ``batch=None``
    means a single graph (repair R4)."""
    ptr = getattr(data, "ptr", None)
    if ptr is not None:
        return [int(v) for v in (ptr.tolist() if isinstance(ptr, torch.Tensor) else ptr)]
    batch = getattr(data, "batch", None)
    if batch is None:
        return [0, num_nodes]
    counts = torch.bincount(batch)
    return [0] + torch.cumsum(counts, 0).tolist()


class GraphStructure:
    """Device CSR/CSC + GCN weights of one edge list over ``num_nodes`` nodes (self loops added)."""

    def assert_ok(self) -> None:
        """Reads the builder's status word (ONE host sync -- not on the training path): raises if a scatter slot or a row
        extent fell outside its bounds, i.e. the counters the scatter trusts were inconsistent."""
        if self.status is not None and int(self.status.item()) != 0:
            raise _lib.DGDMKernelError(f"CSR build reported inconsistent counters (status {int(self.status.item())}): the affected entries "
                                       "were skipped")

    __slots__ = ("num_nodes", "num_edges", "num_entries", "rowptr", "col", "eid", "w",
                 "rowptr_t", "col_t", "eid_t", "w_t", "dinv", "status", "long_tables", "long_partial", "_long")

    LONG_PARTIAL_LD = 1024      # floats per partial slot: the widest row dgdm_spmm* takes

    def long_rows(self, transposed: bool = False):
        """The ``long_rows`` argument of dgdm_spmm* for the by-destination (forward) or by-source (backward) CSR: rows longer than
        DGDM_SPMM_LONG_ROW entries (hubs) are cut into segments that run on lane groups of their own (include/dgdm_hip.h,
        "Long rows").  None when the builder left no table (the "single" pipeline, more than 2^20 nodes)."""
        return None if self._long is None else self._long[1 if transposed else 0]

    def extended(self, num_nodes: int, ea_hat: Optional[torch.Tensor] = None):
        """The index set of the same edge list over ``num_nodes`` >= self.num_nodes nodes (self loops on): what
        ``GraphStructure(edge_index, num_nodes)`` builds, derived by ONE copying launch (dgdm_csr_extend) -- the nodes beyond
        ``self.num_nodes`` have their self loop only.  That is the decoder of the reference's graph U-Net (D10,
        core/graph_layers.py:420,453: level j convolves its n_j nodes with the edge list of level j + 1).  Returns
        ``(structure, ea_hat padded with zero rows or None)``.  The long-row tables (and their scratch) are shared with ``self``:
        no appended row is long, and the two structures are used one after the other on one stream."""
        n_old, n_new = self.num_nodes, int(num_nodes)
        if n_new < n_old or n_old <= 0 or self.num_entries != self.num_edges + n_old:
            raise ValueError("extended() needs a structure built with self loops and a node count that does not shrink")
        lib = _lib.load()
        dev = self.rowptr.device
        g = GraphStructure.__new__(GraphStructure)
        n_ent = self.num_edges + n_new
        g.num_nodes, g.num_edges, g.num_entries = n_new, self.num_edges, n_ent
        i32 = dict(dtype=torch.int32, device=dev)
        g.rowptr, g.rowptr_t = torch.empty(n_new + 1, **i32), torch.empty(n_new + 1, **i32)
        g.col, g.col_t = torch.empty(n_ent, **i32), torch.empty(n_ent, **i32)
        g.eid, g.eid_t = torch.empty(n_ent, **i32), torch.empty(n_ent, **i32)
        g.dinv = torch.empty(n_new, dtype=torch.float32, device=dev)
        g.w, g.w_t = torch.empty(n_ent, dtype=torch.float32, device=dev), torch.empty(n_ent, dtype=torch.float32, device=dev)
        g.status = None
        g.long_tables, g.long_partial, g._long = self.long_tables, self.long_partial, self._long
        ea_out = None
        if ea_hat is not None:
            if ea_hat.dtype != torch.float32 or not ea_hat.is_contiguous() or ea_hat.size(0) != n_old:
                raise ValueError("ea_hat must be a contiguous float32 [num_nodes, edge_dim] tensor")
            ea_out = torch.empty(n_new, ea_hat.size(1), dtype=torch.float32, device=dev)
        _lib.check(lib.dgdm_csr_extend(self.rowptr.data_ptr(), self.col.data_ptr(), self.eid.data_ptr(), self.w.data_ptr(),
                                       self.rowptr_t.data_ptr(), self.col_t.data_ptr(), self.eid_t.data_ptr(), self.w_t.data_ptr(),
                                       self.dinv.data_ptr(), _lib.ptr(ea_hat), 0 if ea_hat is None else ea_hat.size(1),
                                       self.num_edges, n_old, n_new, n_ent, g.rowptr.data_ptr(), g.col.data_ptr(), g.eid.data_ptr(),
                                       g.w.data_ptr(), g.rowptr_t.data_ptr(), g.col_t.data_ptr(), g.eid_t.data_ptr(), g.w_t.data_ptr(),
                                       g.dinv.data_ptr(), _lib.ptr(ea_out), _lib.stream_ptr(dev)), "dgdm_csr_extend")
        return g, ea_out

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, add_loops: bool = True, pipeline: str = "pair", normalize: bool = True,
                 max_degree: Optional[int] = None):
        """``pipeline``: "pair" (dgdm_csr_build_pair) or "single" (one entry point per array set; same results).
        ``normalize=False``: the entry weights are 1 instead of deg^-1/2 deg^-1/2 (GraphConvolution(normalize=False): a plain sum
        over the incoming edges, core/graph_layers.py:76-86).
        ``max_degree``: a HOST-side upper bound of the largest in- / out-degree of the edge list (GraphData.host_max_degree: a loader
        takes it before the upload), None when unknown.  With it the long-row tables and their scratch (23 MB per index set at the
        headline batch, 0.6 GB at 6 M entries) are only allocated when a row CAN be longer than DGDM_SPMM_LONG_ROW entries; without
        it they are sized for the worst case (whether a graph has hubs is otherwise known on the device only)."""
        self._build(edge_index, num_nodes, add_loops, pipeline, max_degree)
        if not normalize and self.num_entries > 0:
            lib, st = _lib.load(), _lib.stream_ptr(self.w.device)
            for w in (self.w, self.w_t):
                _lib.check(lib.dgdm_fill_u32(w.data_ptr(), w.numel(), 0x3F800000, st), "dgdm_fill_u32")       # 1.0f

    LONG_ROW = 128      # DGDM_SPMM_LONG_ROW (include/dgdm_hip.h)

    def _build(self, edge_index: torch.Tensor, num_nodes: int, add_loops: bool, pipeline: str, max_degree: Optional[int] = None) -> None:
        _lib.require_cuda(edge_index)
        lib = _lib.load()
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError(f"edge_index must be int64 [2,E], got {edge_index.dtype} {tuple(edge_index.shape)}")
        ei = edge_index.contiguous()
        dev = ei.device
        E, N = ei.size(1), int(num_nodes)
        n_ent = E + (N if add_loops else 0)
        self.num_nodes, self.num_edges, self.num_entries = N, E, n_ent
        i32 = dict(dtype=torch.int32, device=dev)
        st = _lib.stream_ptr(dev)
        self.rowptr, self.rowptr_t = torch.empty(N + 1, **i32), torch.empty(N + 1, **i32)
        self.col, self.col_t = torch.empty(n_ent, **i32), torch.empty(n_ent, **i32)
        self.eid, self.eid_t = torch.empty(n_ent, **i32), torch.empty(n_ent, **i32)
        self.dinv = torch.empty(N, dtype=torch.float32, device=dev)
        self.w = torch.empty(n_ent, dtype=torch.float32, device=dev)
        self.w_t = torch.empty(n_ent, dtype=torch.float32, device=dev)
        self.status = None
        self.long_tables = self.long_partial = self._long = None
        if pipeline == "pair":       # both orientations, dinv and the weights in five launches
            ws_bytes = _lib.workspace_bytes("dgdm_csr_build_pair_workspace_bytes", E, N, int(add_loops))
            ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
            if N > 0:    # the builder's overflow flag (include/dgdm_hip.h): kept as a 4-byte view, read only by assert_ok()
                off = _lib.workspace_bytes("dgdm_csr_build_pair_status_offset", E, N, int(add_loops))
                self.status = ws[off:off + 4].view(torch.int32)
            lt0 = lt1 = None
            item_cap = 0
            may_be_long = max_degree is None or int(max_degree) + int(add_loops) > self.LONG_ROW
            if 0 < N <= (1 << 20) and n_ent > 0 and may_be_long:     # long-row tables (hubs): both orientations share one scratch for partial sums
                words = lib.dgdm_spmm_long_table_words(n_ent)
                item_cap, slot_cap = lib.dgdm_spmm_long_item_cap(n_ent), lib.dgdm_spmm_long_slot_cap(n_ent)
                self.long_tables = torch.empty(2, words, **i32)
                self.long_partial = torch.empty(slot_cap, self.LONG_PARTIAL_LD, dtype=torch.float32, device=dev)
                lt0, lt1 = self.long_tables[0].data_ptr(), self.long_tables[1].data_ptr()
                self._long = tuple(_lib.LongRows(t, self.long_partial.data_ptr(), self.LONG_PARTIAL_LD, item_cap, slot_cap) for t in (lt0, lt1))
            _lib.check(lib.dgdm_csr_build_pair(ei.data_ptr(), E, N, int(add_loops), self.rowptr.data_ptr(), self.col.data_ptr(),
                                               self.eid.data_ptr(), self.w.data_ptr(), self.rowptr_t.data_ptr(), self.col_t.data_ptr(),
                                               self.eid_t.data_ptr(), self.w_t.data_ptr(), self.dinv.data_ptr(), ws.data_ptr(),
                                               ws_bytes, lt0, lt1, item_cap, st), "dgdm_csr_build_pair")
            return
        if pipeline != "single":
            raise ValueError(f"unknown CSR pipeline {pipeline!r}")
        ws_bytes = lib.dgdm_csr_build_workspace_bytes(E, N, int(add_loops))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        for by_src, (rowptr, col, eid) in enumerate(((self.rowptr, self.col, self.eid), (self.rowptr_t, self.col_t, self.eid_t))):
            _lib.check(lib.dgdm_csr_build(ei.data_ptr(), E, N, int(add_loops), by_src, rowptr.data_ptr(), col.data_ptr(),
                                          eid.data_ptr(), ws.data_ptr(), ws_bytes, st), "dgdm_csr_build")
        _lib.check(lib.dgdm_gcn_dinv(self.rowptr.data_ptr(), N, self.dinv.data_ptr(), st), "dgdm_gcn_dinv")
        _lib.check(lib.dgdm_csr_edge_weights(self.rowptr.data_ptr(), self.col.data_ptr(), self.dinv.data_ptr(), N,
                                             self.w.data_ptr(), st), "dgdm_csr_edge_weights")
        _lib.check(lib.dgdm_csr_edge_weights(self.rowptr_t.data_ptr(), self.col_t.data_ptr(), self.dinv.data_ptr(), N,
                                             self.w_t.data_ptr(), st), "dgdm_csr_edge_weights")

"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the integer (index) half of the DGDM hot path.

Nothing in the shipped package may import this module: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only as the
checker ("Oracle").

What it restates (reference = /root/reference/dgdm_histopath, read as text):

* ``core/graph_layers.py:76-84``  -- ``GraphConvolution.forward`` index preparation:
  self-loop edges ``(i, i)`` are appended *after* the E input edges (PyG
  ``add_self_loops``: concatenation, no de-duplication of loops/multi-edges that are
  already there), ``deg = degree(col)`` is the in-degree (destination count) including the
  appended loop, ``norm_e = deg^-1/2[src_e] * deg^-1/2[dst_e]`` with ``inf -> 0``.
* the aggregation order of ``MessagePassing.propagate(aggr='add')`` (third-party,
  torch-geometric >= 2.3, ``requirements.txt:4``): a scatter-add over ``edge_index[1]``.
  The build fixes the summation order to *ascending edge id inside each destination row*,
  i.e. a stable sort of the edge list by destination -- that is what this file defines and
  what the HIP CSR builder must reproduce bit for bit (``rowptr``/``col``/``eid``).
* ``core/graph_layers.py:298-329`` -- ``AdaptiveGraphPooling`` index work: top-k mask,
  ``perm = mask.nonzero()`` (ascending node id), ``node_map``, edge filter + relabel.

Parity pin: these are pure integer definitions; they are pinned by the hand-computed
known-answer cases in ``tests/test_oracle_csr.py`` and, through the float oracle that
consumes them, by the golden vectors captured from the reference classes
(``tests/golden``, ``oracle/capture_golden.py``).
"""
from __future__ import annotations

import numpy as np


def append_self_loops(edge_index: np.ndarray, num_nodes: int) -> np.ndarray:
    """``add_self_loops``: [2,E] -> [2,E+N]; loop edge for node i gets edge id E+i."""
    ei = np.asarray(edge_index, dtype=np.int64).reshape(2, -1)
    loops = np.arange(num_nodes, dtype=np.int64)
    return np.concatenate([ei, np.stack([loops, loops])], axis=1)


def csr_by_key(keys: np.ndarray, vals: np.ndarray, num_nodes: int):
    """Stable bucket of edges by ``keys``; returns (rowptr[N+1], col, eid) as int32.

    ``col[p]`` is ``vals`` of the p-th edge in (key, edge id) order, ``eid[p]`` its id.
    """
    keys = np.asarray(keys, dtype=np.int64)
    order = np.argsort(keys, kind="stable")
    counts = np.bincount(keys, minlength=num_nodes).astype(np.int64)
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return (rowptr.astype(np.int32), np.asarray(vals, dtype=np.int64)[order].astype(np.int32),
            order.astype(np.int32))


def gcn_csr(edge_index: np.ndarray, num_nodes: int, add_loops: bool = True) -> dict:
    """Everything a GraphConvolution needs from the edge list.

    Returns a dict with
      rowptr, col, eid      CSR by destination (col = source ids): forward aggregation
      rowptr_t, col_t, eid_t CSR by source (col_t = destination ids): backward aggregation
      dinv [N] f32          deg^-1/2 (0 where deg == 0)
      norm [E'] f32         per-entry weight in CSR-by-destination order
      norm_t [E'] f32       same weights in CSR-by-source order
      src, dst [E'] i64     the (loop-extended) COO list in edge-id order
      norm_coo [E'] f32     weights in edge-id order
    """
    ei = np.asarray(edge_index, dtype=np.int64).reshape(2, -1)
    if add_loops:
        ei = append_self_loops(ei, num_nodes)
    src, dst = ei[0], ei[1]
    rowptr, col, eid = csr_by_key(dst, src, num_nodes)
    rowptr_t, col_t, eid_t = csr_by_key(src, dst, num_nodes)
    deg = np.diff(rowptr.astype(np.int64)).astype(np.float32)
    with np.errstate(divide="ignore"):
        dinv = np.where(deg > 0, np.float32(1.0) / np.sqrt(deg, dtype=np.float32), np.float32(0.0))
    dinv = dinv.astype(np.float32)
    norm_coo = (dinv[src] * dinv[dst]).astype(np.float32)
    return dict(rowptr=rowptr, col=col, eid=eid, rowptr_t=rowptr_t, col_t=col_t, eid_t=eid_t,
                dinv=dinv, norm=norm_coo[eid], norm_t=norm_coo[eid_t], src=src, dst=dst,
                norm_coo=norm_coo, num_nodes=num_nodes, num_input_edges=int(edge_index.shape[1]))


def topk_pool_indices(scores: np.ndarray, edge_index: np.ndarray, ratio: float = 0.5, perm=None) -> dict:
    """AdaptiveGraphPooling index work (graph_layers.py:306-324).

    k = max(1, int(ratio*N)); keep the k largest scores (ties: lowest node id first -- the
    reference's ``torch.topk`` tie order is implementation-defined, fixtures avoid ties);
    perm ascending; edges with a dropped endpoint are removed, survivors keep their order
    and are renumbered.
    """
    scores = np.asarray(scores)
    n = scores.shape[0]
    k = max(1, int(ratio * n))
    keep = np.zeros(n, dtype=bool)
    if perm is not None:      # kept node ids handed in (dgdm_oracle.DECISIONS): the index work below is the same
        keep[np.asarray(perm, dtype=np.int64)] = True
        assert keep.sum() == k, "injected perm must keep exactly k nodes"
    else:
        order = np.lexsort((np.arange(n), -scores.astype(np.float64)))  # score desc, id asc
        keep[order[:k]] = True
    perm = np.nonzero(keep)[0].astype(np.int64)
    node_map = np.full(n, -1, dtype=np.int64)
    node_map[perm] = np.arange(perm.shape[0], dtype=np.int64)
    ei = np.asarray(edge_index, dtype=np.int64).reshape(2, -1)
    emask = (node_map[ei[0]] >= 0) & (node_map[ei[1]] >= 0)
    kept = np.nonzero(emask)[0].astype(np.int64)
    return dict(perm=perm, node_map=node_map, edge_keep=kept, edge_index=node_map[ei[:, emask]])

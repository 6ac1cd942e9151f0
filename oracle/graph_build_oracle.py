"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's tissue-graph edge construction
(dgdm_histopath/preprocessing/tissue_graph_builder.py:269-365 and the edge part of
_to_pytorch_geometric :367-414), in numpy float64 like the reference (its coordinates and
similarities are float64 numpy arrays; scikit-learn >= 1.0, un-pinned in requirements.txt, supplies
NearestNeighbors / cosine_similarity).  Pinned against the reference's own functions executed in the
dev container: tests/golden/g8_graph_build_*.npz written by oracle/capture_graph_golden.py.

Algorithm (reference line):
  spatial      kNN over coords, K = min(spatial_k + 1, N) neighbours sorted by distance, column 0
               skipped as "self" (:286-295); weight = exp(-10 d); edge i -> nbr if weight >= thr,
               features [d, weight] (:297-309)
  morphological kNN over features (Euclidean, K = min(morphological_k + 1, N)), column 0 skipped;
               cosine similarity of the pair; edge if similarity >= thr, weight = similarity,
               features [similarity] (:318-343)
  dedup        key = sorted (source, target); a later edge replaces an earlier one only if its weight
               is strictly larger; output in first-occurrence order of the keys (:346-357)
  emit         every kept edge as two consecutive directed columns (src,tgt), (tgt,src) with the
               same attributes and a type code spatial=0 / morphological=1 (:384-402)
Repair R6: the reference np.stack()s 2-wide spatial and 1-wide morphological
attribute rows, which raises whenever both kinds survive; the model needs 32 columns.  The rows are
zero-padded to `edge_dim` (32): spatial [d, w, 0...], morphological [similarity, 0...].
kNN ties (equal distance) are implementation-defined in scikit-learn; here: lower index first.
"""
from __future__ import annotations

import numpy as np


def knn(X: np.ndarray, k_plus_1: int):
    """(distances [N,K], indices [N,K]) of the K nearest rows (self included), ascending by (distance, index)."""
    X = np.asarray(X, dtype=np.float64)
    n = X.shape[0]
    K = min(k_plus_1, n)
    sq = (X * X).sum(1)
    d2 = np.maximum(sq[:, None] + sq[None, :] - 2.0 * (X @ X.T), 0.0)
    np.fill_diagonal(d2, 0.0)
    # exact recomputation for the candidates near the cut would be overkill for fixtures: direct differences
    if X.shape[1] <= 8:
        diff = X[:, None, :] - X[None, :, :]
        d2 = (diff * diff).sum(-1)
    order = np.lexsort((np.broadcast_to(np.arange(n), (n, n)), d2), axis=1)[:, :K]
    return np.sqrt(np.take_along_axis(d2, order, 1)), order


def create_edges(features: np.ndarray, coords: np.ndarray, spatial_k: int = 8, morphological_k: int = 16, edge_threshold: float = 0.7):
    """Kept undirected edges in output order: dict(src, tgt, type, weight, feat [U,2])."""
    features = np.asarray(features, dtype=np.float64)
    coords = np.asarray(coords, dtype=np.float64)
    n = coords.shape[0]
    cand = []   # (src, tgt, type, weight, f0, f1)
    dist, idx = knn(coords, spatial_k + 1)
    for i in range(n):
        for j in range(1, idx.shape[1]):
            d = dist[i, j]
            w = np.exp(-d * 10)
            if w >= edge_threshold:
                cand.append((i, int(idx[i, j]), 0, w, d, w))
    norm = np.sqrt((features * features).sum(1))
    _, fidx = knn(features, morphological_k + 1)
    for i in range(n):
        for j in range(1, fidx.shape[1]):
            t = int(fidx[i, j])
            sim = float(features[i] @ features[t]) / (norm[i] * norm[t]) if norm[i] > 0 and norm[t] > 0 else 0.0
            if sim >= edge_threshold:
                cand.append((i, t, 1, sim, sim, 0.0))
    kept = {}
    for e in cand:
        key = (min(e[0], e[1]), max(e[0], e[1]))
        if key not in kept or e[3] > kept[key][3]:
            kept[key] = e              # dict keeps the key's first insertion position
    out = list(kept.values())
    arr = lambda c, dt: np.array([e[c] for e in out], dtype=dt).reshape(-1)
    return dict(src=arr(0, np.int64), tgt=arr(1, np.int64), type=arr(2, np.int64), weight=arr(3, np.float64),
                feat=np.array([[e[4], e[5]] for e in out], dtype=np.float64).reshape(-1, 2))


def to_edge_arrays(edges: dict, edge_dim: int = 32):
    """edge_index [2, 2U] int64, edge_attr [2U, edge_dim] float32 (R6 padding), edge_type [2U] int64."""
    u = edges["src"].shape[0]
    ei = np.empty((2, 2 * u), dtype=np.int64)
    ei[0, 0::2], ei[1, 0::2] = edges["src"], edges["tgt"]
    ei[0, 1::2], ei[1, 1::2] = edges["tgt"], edges["src"]
    ea = np.zeros((2 * u, edge_dim), dtype=np.float32)
    ea[0::2, :2] = edges["feat"]; ea[1::2, :2] = edges["feat"]
    et = np.repeat(edges["type"], 2)
    return ei, ea, et

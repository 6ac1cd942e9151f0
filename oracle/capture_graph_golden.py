"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/g8_graph_build_*.npz by EXECUTING the reference's own
TissueGraphBuilder._create_edges (with scikit-learn's NearestNeighbors / cosine_similarity) in the
dev container.

    python -m oracle.capture_graph_golden

The reference module is imported from /root/reference at run time; nothing of it is stored.  Two of
its imports are absent from this image (ordinary ModuleNotFoundError): torch_geometric (stand-in:
oracle/pyg_standin.py) and cv2, needed only by the sibling slide_processor module from which the
builder takes two plain dataclasses -- a stand-in module with those two records
(slide_processor.py:33-52: PatchInfo, SlideData) is registered instead.  The builder object is made
without running its constructor (which downloads a vision backbone); only the edge parameters are set.
"""
from __future__ import annotations

import importlib
import os
import sys
import types
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

REF_ROOT = "/root/reference"
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_builder_module():
    from . import pyg_standin
    pyg_standin.install()
    for name in ("dgdm_histopath", "dgdm_histopath.preprocessing"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF_ROOT, *name.split("."))]
            sys.modules[name] = m
    sp = types.ModuleType("dgdm_histopath.preprocessing.slide_processor")

    @dataclass
    class PatchInfo:
        x: int
        y: int
        level: int
        magnification: float
        patch_id: str
        tissue_percentage: float
        features: Optional[np.ndarray] = None

    @dataclass
    class SlideData:
        slide_id: str
        patches: List[PatchInfo]
        metadata: Dict
        thumbnail: Optional[np.ndarray] = None
        tissue_mask: Optional[np.ndarray] = None

    sp.PatchInfo, sp.SlideData = PatchInfo, SlideData
    sys.modules["dgdm_histopath.preprocessing.slide_processor"] = sp
    return importlib.import_module("dgdm_histopath.preprocessing.tissue_graph_builder"), sp


def make_case(seed: int, n: int, f: int, clusters: int, spread: float, box: float):
    """Clustered unit-ish features (so cosine similarities pass 0.7 inside a cluster) and coordinates
    dense enough that exp(-10 d) >= 0.7 (d <= 0.0357) happens for some of the 8 nearest neighbours."""
    g = np.random.default_rng(seed)
    centers = g.normal(size=(clusters, f))
    lab = g.integers(0, clusters, size=n)
    feats = (centers[lab] + spread * g.normal(size=(n, f))).astype(np.float32)
    coords = (g.random(size=(n, 2)) * box).astype(np.float64)
    return feats, coords


CASES = {  # name: (seed, n, f, clusters, spread, box, spatial_k, morph_k, thr)
    "small": (1, 40, 16, 3, 0.35, 0.15, 8, 16, 0.7),
    "mid": (2, 300, 64, 5, 0.6, 0.45, 8, 16, 0.7),
    "tiny_k_gt_n": (3, 6, 8, 2, 0.3, 0.05, 8, 16, 0.7),
    "loose": (4, 120, 32, 4, 0.8, 0.3, 4, 6, 0.5),
}


def run_reference(mod, sp, feats, coords, spatial_k, morph_k, thr):
    b = object.__new__(mod.TissueGraphBuilder)
    b.spatial_k, b.morphological_k, b.edge_threshold = spatial_k, morph_k, thr
    nodes = []
    for i in range(feats.shape[0]):
        p = sp.PatchInfo(x=0, y=0, level=0, magnification=20.0, patch_id=f"p{i:05d}", tissue_percentage=1.0, features=feats[i])
        nodes.append(mod.GraphNode(node_id=p.patch_id, patch_info=p, features=feats[i], spatial_coords=(float(coords[i, 0]), float(coords[i, 1]))))
    edges = b._create_edges(nodes)
    ident = {nd.node_id: i for i, nd in enumerate(nodes)}
    src = np.array([ident[e.source_id] for e in edges], dtype=np.int64)
    tgt = np.array([ident[e.target_id] for e in edges], dtype=np.int64)
    typ = np.array([0 if e.edge_type == "spatial" else 1 for e in edges], dtype=np.int64)
    w = np.array([e.weight for e in edges], dtype=np.float64)
    feat = np.zeros((len(edges), 2))
    for r, e in enumerate(edges):
        feat[r, :len(e.features)] = e.features
    return dict(src=src, tgt=tgt, type=typ, weight=w, feat=feat)


def main():
    mod, sp = load_builder_module()
    os.makedirs(OUT_DIR, exist_ok=True)
    for name, (seed, n, f, c, spread, box, sk, mk, thr) in CASES.items():
        feats, coords = make_case(seed, n, f, c, spread, box)
        ref = run_reference(mod, sp, feats, coords, sk, mk, thr)
        np.savez_compressed(os.path.join(OUT_DIR, f"g8_graph_build_{name}.npz"), features=feats, coords=coords,
                            params=np.array([sk, mk, thr], dtype=np.float64), **{f"ref_{k}": v for k, v in ref.items()})
        print(name, "nodes", n, "kept edges", ref["src"].shape[0], "spatial", int((ref["type"] == 0).sum()), "morph", int((ref["type"] == 1).sum()))


if __name__ == "__main__":
    main()

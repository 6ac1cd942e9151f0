#!/usr/bin/env python3
"""Throughput of GPU tissue-graph construction (K11) at the north-star graph size, with the CPU oracle
(numpy restatement of the reference's scikit-learn path) timed beside it.

    python tools/bench_graph_build.py [--nodes 10000] [--feat 768] [--iters 20] [--no-cpu]
"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10000); ap.add_argument("--feat", type=int, default=768)
    ap.add_argument("--iters", type=int, default=20); ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    from dgdm_histopath_lab_amd.graph_build import TissueGraphBuilder
    g = np.random.default_rng(0)
    centers = g.normal(size=(12, a.feat))
    feats = (centers[g.integers(0, 12, a.nodes)] + 0.7 * g.normal(size=(a.nodes, a.feat))).astype(np.float32)
    coords = (g.random((a.nodes, 2)) * (a.nodes / 1200.0) ** 0.5 * 0.9).astype(np.float32)
    x, c = torch.from_numpy(feats).cuda(), torch.from_numpy(coords).cuda()
    b = TissueGraphBuilder()
    for _ in range(3):
        e = b.build_edges(x, c)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.iters):
        e = b.build_edges(x, c)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
    flop = 2.0 * a.nodes * a.nodes * a.feat
    res = {"metric": "graphs/s (kNN edge construction, spatial_k=8, morphological_k=16, threshold 0.7)", "value": round(1 / dt, 2),
           "ms_per_graph": round(dt * 1e3, 3), "nodes": a.nodes, "feat": a.feat, "directed_edges": int(e["edge_index"].size(1)),
           "gram_gemm_flop": flop, "gram_tflops_if_all_time": round(flop / dt / 1e12, 1)}
    if not a.no_cpu:
        from oracle.graph_build_oracle import create_edges
        t0 = time.perf_counter(); create_edges(feats, coords.astype(np.float64)); cpu = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(1 / cpu, 4), "unit": "graphs/s", "kind": "port", "cores": os.cpu_count(),
                               "sample": f"1 graph of {a.nodes} nodes, numpy float64 oracle, {cpu:.1f} s"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()

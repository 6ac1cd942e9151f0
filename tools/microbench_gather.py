"""North-star microbenchmark: message-passing gather (K2) at 10k nodes x 768 features, 50k edges
(+10k self loops).  Algorithmic bytes per launch:
(E+N)*C*4 gathered + N*C*4 written + (E+N)*8 (col+w) + (N+1)*4 = 215.6 MB at C=768."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import GraphStructure, ops  # noqa: E402
from dgdm_histopath_lab_amd.synthetic import synthetic_batch  # noqa: E402


def algorithmic_bytes(n, e, c):
    ent = e + n
    return ent * c * 4 + n * c * 4 + ent * 8 + (n + 1) * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10000)
    ap.add_argument("--edges", type=int, default=50000)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--widths", type=int, nargs="*", default=[768, 512, 256, 128, 32])
    ap.add_argument("--skew", action="store_true",
                    help="degree-skew cases at the north-star size (VERDICT r2 item 9): the synthetic kNN-like graph, a Poisson multigraph "
                         "(torch.randint endpoints, core/graph_layers.py:92 sees whatever the builder emits), and a graph with ONE hub of "
                         "5000 in-neighbours -- one wave walks a destination row serially (csrc/spmm.hip), so a hub row is the worst case")
    ap.add_argument("--cold", action="store_true",
                    help="every launch behind a rewrite of a 1 GiB buffer (L2 and the 256 MiB Infinity Cache evicted): single launches, "
                         "median; what the rocprofv3 kernel-trace / PMC passes of tools/profile_gather.sh call the cold case")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.cold:
        cold_case(dev, a.nodes, a.edges, a.widths[0], a.iters)
        return
    if a.skew:
        skew_cases(dev, a.nodes, a.edges, a.iters)
        return
    b = synthetic_batch(0, a.batch, a.nodes, a.edges, 8)
    n, e = b.x.size(0), b.edge_index.size(1)
    gs = GraphStructure(b.edge_index.to(dev), n)
    for c in a.widths:
        x = torch.randn(n, c, device=dev)
        y = torch.empty(n, c, device=dev)
        for _ in range(10):
            ops.spmm_raw(gs.rowptr, gs.col, gs.w, x, n, out=y)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0.record()
        for _ in range(a.iters):
            ops.spmm_raw(gs.rowptr, gs.col, gs.w, x, n, out=y)
        t1.record(); torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 1e3 / a.iters
        by = algorithmic_bytes(n, e, c)
        print(json.dumps(dict(kernel="dgdm_spmm", nodes=n, edges=e, C=c, us=round(us, 2), algorithmic_MB=round(by / 1e6, 1),
                              GBps=round(by / us / 1e3, 1), frac_of_8TBps=round(by / us / 1e3 / 8000, 3))))
    # CSR build cost (both orientations + weights)
    ei = b.edge_index.to(dev)
    for _ in range(3):
        GraphStructure(ei, n)
    torch.cuda.synchronize(); t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
    for _ in range(20):
        GraphStructure(ei, n)
    t1.record(); torch.cuda.synchronize()
    print(json.dumps(dict(kernel="graph_structure_build", nodes=n, edges=e, us=round(t0.elapsed_time(t1) * 1e3 / 20, 1))))


def cold_case(dev, n, e, c, iters):
    b = synthetic_batch(0, 1, n, e, 8)
    gs = GraphStructure(b.edge_index.to(dev), n)
    x = torch.randn(n, c, device=dev)
    y = torch.empty(n, c, device=dev)
    evict = torch.zeros(256 * 1024 * 1024, dtype=torch.float32, device=dev)       # 1 GiB
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        evict.add_(1.0)
        t0.record()
        ops.spmm_raw(gs.rowptr, gs.col, gs.w, x, n, out=y)
        t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) * 1e3)
    ts.sort()
    us = ts[len(ts) // 2]
    by = algorithmic_bytes(n, e, c)
    print(json.dumps(dict(kernel="dgdm_spmm", case="cold: 1 GiB rewritten before every launch", nodes=n, edges=e, C=c, launches=iters,
                          us_median=round(us, 2), us_min=round(ts[0], 2), algorithmic_MB=round(by / 1e6, 1), GBps=round(by / us / 1e3, 1),
                          frac_of_8TBps=round(by / us / 1e3 / 8000, 3))))


def skew_cases(dev, n, e, iters, c=768):
    g = torch.Generator().manual_seed(5)
    cases = {"synthetic (uniform undirected pairs, both directions)": synthetic_batch(0, 1, n, e, 8).edge_index}
    cases["poisson multigraph (independent uniform endpoints, duplicates and loops kept)"] = torch.randint(0, n, (2, e), generator=g)
    hub = torch.randint(0, n, (2, e), generator=g)
    hub[1, :5000] = 17                                   # 5000 edges INTO node 17 (its row of the by-destination CSR)
    hub[0, :5000] = torch.randperm(n, generator=g)[:5000]
    cases["one hub: 5000 in-neighbours of one node among 10000 (rest uniform)"] = hub
    both = hub.clone()
    both[0, 5000:10000] = 17                             # and 5000 edges OUT of it (the transposed CSR of the backward)
    cases["hub with 5000 in- and 5000 out-edges, by-source orientation (the backward's gather)"] = both
    for name, ei in cases.items():
        gs = GraphStructure(ei.to(dev), n)
        x = torch.randn(n, c, device=dev)
        y = torch.empty(n, c, device=dev)
        tr = name.startswith("hub with")
        rp, cl, w = (gs.rowptr_t, gs.col_t, gs.w_t) if tr else (gs.rowptr, gs.col, gs.w)
        deg = (rp[1:] - rp[:-1])
        by = algorithmic_bytes(n, ei.size(1), c)
        for label, lr in (("one wave per row", None), ("long rows split (default)", gs.long_rows(tr))):
            for _ in range(10):
                ops.spmm_raw(rp, cl, w, x, n, out=y, long_rows=lr)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); t0.record()
            for _ in range(iters):
                ops.spmm_raw(rp, cl, w, x, n, out=y, long_rows=lr)
            t1.record(); torch.cuda.synchronize()
            us = t0.elapsed_time(t1) * 1e3 / iters
            print(json.dumps(dict(kernel="dgdm_spmm", case=name, mode=label, nodes=n, edges=int(ei.size(1)), C=c, max_row=int(deg.max()),
                                  mean_row=round(float(deg.float().mean()), 2), us=round(us, 2), algorithmic_MB=round(by / 1e6, 1),
                                  GBps=round(by / us / 1e3, 1), frac_of_8TBps=round(by / us / 1e3 / 8000, 3))))
        ei_d = ei.to(dev)
        for _ in range(3):
            GraphStructure(ei_d, n)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0.record()
        for _ in range(10):
            GraphStructure(ei_d, n)
        t1.record(); torch.cuda.synchronize()
        print(json.dumps(dict(kernel="graph_structure_build (K1)", case=name, us=round(t0.elapsed_time(t1) * 1e3 / 10, 1))))


if __name__ == "__main__":
    main()

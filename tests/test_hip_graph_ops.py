"""GPU parity: K1 (CSR build, bit-exact) and K2 (SpMM gather-reduce) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import csr_oracle

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _rand_edges(n, e, seed, hub=False):
    rng = np.random.default_rng(seed)
    ei = rng.integers(0, max(n, 1), size=(2, e)).astype(np.int64)
    if hub and e > 10:
        ei[1, : e // 2] = 3 % n          # one very long destination row
        ei[0, e // 4: e // 2] = 5 % n    # and a long source row
    return ei


@pytest.mark.parametrize("n,e,hub", [(1, 0, False), (7, 0, False), (16, 40, False), (1000, 5000, False), (3000, 20000, True),
                                     (10000, 50000, False), (70001, 300007, False), (1100003, 2000001, False)])
@pytest.mark.parametrize("pipeline", ["pair", "single"])
def test_csr_build_bit_exact(n, e, hub, pipeline):
    from dgdm_histopath_lab_amd import GraphStructure
    ei = _rand_edges(n, e, n + e, hub)
    gs = GraphStructure(torch.from_numpy(ei).to(_dev()), n, pipeline=pipeline)
    o = csr_oracle.gcn_csr(ei, n)
    gs.assert_ok()          # bounds guard of the scatter kernels: status word 0 (pair pipeline; the other has none)
    for k in ("rowptr", "col", "eid", "rowptr_t", "col_t", "eid_t"):
        got = getattr(gs, k).cpu().numpy()
        assert got.dtype == np.int32 and np.array_equal(got, o[k]), k
    np.testing.assert_allclose(gs.dinv.cpu().numpy(), o["dinv"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(gs.w.cpu().numpy(), o["norm"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(gs.w_t.cpu().numpy(), o["norm_t"], rtol=1e-6, atol=0)


def test_csr_build_no_loops_and_determinism():
    from dgdm_histopath_lab_amd import GraphStructure
    ei = _rand_edges(500, 4000, 1)
    t = torch.from_numpy(ei).to(_dev())
    a = GraphStructure(t, 500, add_loops=False)
    o = csr_oracle.gcn_csr(ei, 500, add_loops=False)
    assert np.array_equal(a.col.cpu().numpy(), o["col"]) and np.array_equal(a.eid_t.cpu().numpy(), o["eid_t"])
    b = GraphStructure(t, 500, add_loops=False)
    assert torch.equal(a.col, b.col) and torch.equal(a.eid, b.eid)  # atomics only order-free counts


@pytest.mark.parametrize("n_old,n_new,e,hub", [(1000, 1700, 6000, False), (3000, 6000, 20000, True), (10000, 20000, 100000, False),
                                               (500, 500, 2000, False)])
def test_csr_extend_is_bit_for_bit_the_build_over_more_nodes(n_old, n_new, e, hub):
    """dgdm_csr_extend (GraphStructure.extended): the index set of an edge list with ids < n_old over n_new nodes equals what the
    builder produces for n_new nodes -- rowptr, the entries in use of col / eid / w in both orientations, dinv -- and so do the
    aggregated edge attributes and a gather through either structure, hub rows (segmented long-row path) and edges marked -1
    included.  The U-Net's decoder (D10, core/graph_layers.py:420,453) takes its three index sets this way."""
    from dgdm_histopath_lab_amd import GraphStructure, ops
    ei = _rand_edges(n_old, e, 11 + n_old, hub)
    ei[0, ::9] = -1                      # dropped edges of the sync-free pooling
    t = torch.from_numpy(ei).to(_dev())
    base, full = GraphStructure(t, n_old), GraphStructure(t, n_new)
    ea = torch.randn(e, 32, generator=torch.Generator().manual_seed(3)).to(_dev())
    ea_base = ops.aggregate_edge_attr(ea, base)
    ext, ea_ext = base.extended(n_new, ea_base)
    m = int(full.rowptr[-1])
    assert m == int(ext.rowptr[-1]) == int(ext.rowptr_t[-1]) and (ext.num_nodes, ext.num_edges, ext.num_entries) == (n_new, e, e + n_new)
    assert torch.equal(ext.rowptr, full.rowptr) and torch.equal(ext.rowptr_t, full.rowptr_t) and torch.equal(ext.dinv, full.dinv)
    for k in ("col", "eid", "w", "col_t", "eid_t", "w_t"):
        assert torch.equal(getattr(ext, k)[:m], getattr(full, k)[:m]), k
    assert torch.equal(ea_ext, ops.aggregate_edge_attr(ea, full))
    if hub:
        assert int((base.rowptr[1:] - base.rowptr[:-1]).max()) > 128 and ext.long_rows() is not None
    x = torch.randn(n_new, 64, generator=torch.Generator().manual_seed(5)).to(_dev())
    for tr in (False, True):
        rp, cl, w = (ext.rowptr_t, ext.col_t, ext.w_t) if tr else (ext.rowptr, ext.col, ext.w)
        rf, cf, wf = (full.rowptr_t, full.col_t, full.w_t) if tr else (full.rowptr, full.col, full.w)
        a = ops.spmm_raw(rp, cl, w, x, n_new, long_rows=ext.long_rows(tr))
        b = ops.spmm_raw(rf, cf, wf, x, n_new, long_rows=full.long_rows(tr))
        assert torch.equal(a, b), tr
        assert torch.equal(a[n_old:], x[n_old:])          # a node beyond the edge list's range sees its self loop only
    with pytest.raises(ValueError):
        base.extended(n_old - 1)


def test_csr_pipelines_agree_with_dropped_edges():
    """Edges with an endpoint outside [0, N) (how the sync-free pooling marks dropped edges) are skipped by both pipelines;
    every defined array entry and all weights agree bit for bit."""
    from dgdm_histopath_lab_amd import GraphStructure
    ei = _rand_edges(4000, 30000, 7)
    ei[0, ::7] = -1
    ei[1, 3::11] = 4000 + 5
    t = torch.from_numpy(ei).to(_dev())
    for loops in (True, False):
        a, b = GraphStructure(t, 4000, add_loops=loops), GraphStructure(t, 4000, add_loops=loops, pipeline="single")
        m = int(a.rowptr[-1])
        assert m == int(b.rowptr[-1]) == int(a.rowptr_t[-1]) and m < a.num_entries
        assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.rowptr_t, b.rowptr_t) and torch.equal(a.dinv, b.dinv)
        for k in ("col", "eid", "w", "col_t", "eid_t", "w_t"):
            assert torch.equal(getattr(a, k)[:m], getattr(b, k)[:m]), k


@pytest.mark.parametrize("c", [4, 32, 36, 128, 256, 512, 768, 1024])
def test_spmm_forward_backward_vs_oracle(c):
    from dgdm_histopath_lab_amd import GraphStructure, ops
    n, e = 2000, 8000
    ei = _rand_edges(n, e, c, hub=True)
    gs = GraphStructure(torch.from_numpy(ei).to(_dev()), n)
    o = csr_oracle.gcn_csr(ei, n)
    g = torch.Generator().manual_seed(c)
    x = torch.randn(n, c, generator=g)
    gy = torch.randn(n, c, generator=g)
    # float64 reference: the hub rows (4000 / 2000 entries, half of them duplicates of one edge) are summed in segments by the
    # kernel, and an fp32 reference adding 2000 equal terms one by one carries a rounding bias of ~1e-4 of its own
    src, dst, w = torch.from_numpy(o["src"]), torch.from_numpy(o["dst"]), torch.from_numpy(o["norm_coo"]).double()
    xr = x.double().clone().requires_grad_(True)
    yr = torch.zeros(n, c, dtype=torch.float64).index_add_(0, dst, w[:, None] * xr[src])
    yr.backward(gy.double())
    xd = x.to(_dev()).requires_grad_(True)
    y = ops.aggregate(xd, gs)
    y.backward(gy.to(_dev()))
    assert_close(y, yr, 1e-5, "Y")
    assert_close(xd.grad, xr.grad, 1e-5, "dX")
    y2 = ops.aggregate(xd.detach(), gs)
    assert torch.equal(y2, y.detach())  # bitwise reproducible (no float atomics)


@pytest.mark.parametrize("c", [32, 128, 512, 768])
def test_spmm_long_rows_are_split_and_reduced_in_fixed_order(c):
    """Hub rows: rows of 128 entries (not split), 129 (three segments), 1000 and 5000, by destination AND by
    source; every entry point (plain + bias + accumulate, + addend, concat + operand maximum) against the float64 gather, bitwise
    repeatable, and the short rows bit-identical to the run without the long-row table."""
    from dgdm_histopath_lab_amd import GraphStructure, ops
    dev = _dev()
    n, e = 9000, 30000
    rng = np.random.default_rng(c)
    ei = rng.integers(100, n, size=(2, e)).astype(np.int64)          # the background edges stay clear of the special nodes below
    at = 0
    for node, deg in ((11, 127), (12, 128), (13, 999), (14, 4999)):        # + the self loop: 128, 129, 1000, 5000 entries
        ei[1, at:at + deg] = node; ei[0, at:at + deg] = rng.choice(n, size=deg, replace=False); at += deg
    for node, deg in ((21, 128), (22, 3000)):
        ei[0, at:at + deg] = node; ei[1, at:at + deg] = rng.choice(n, size=deg, replace=False); at += deg
    gs = GraphStructure(torch.from_numpy(ei).to(dev), n)
    gs.assert_ok()
    t0 = gs.long_tables.cpu()
    assert int(t0[0, 0]) == 3 and int(t0[1, 0]) == 2          # rows longer than 128 entries: 3 by destination, 2 by source
    for k in ("rowptr", "col", "eid", "rowptr_t", "col_t", "eid_t"):     # long rows are ordered by k_rank_long: still bit-exact
        assert np.array_equal(getattr(gs, k).cpu().numpy(), csr_oracle.gcn_csr(ei, n)[k]), k
    o = csr_oracle.gcn_csr(ei, n)
    g = torch.Generator().manual_seed(c)
    x, bias, add = torch.randn(n, c, generator=g), torch.randn(c, generator=g), torch.randn(n, c, generator=g)
    src, dst, w = torch.from_numpy(o["src"]), torch.from_numpy(o["dst"]), torch.from_numpy(o["norm_coo"]).double()
    fwd = torch.zeros(n, c, dtype=torch.float64).index_add_(0, dst, w[:, None] * x.double()[src])
    bwd = torch.zeros(n, c, dtype=torch.float64).index_add_(0, src, w[:, None] * x.double()[dst])
    xd = x.to(dev)
    y = ops.spmm_raw(gs.rowptr, gs.col, gs.w, xd, n, bias=bias.to(dev), long_rows=gs.long_rows())
    assert_close(y, fwd + bias.double(), 2e-6, "forward + bias")
    ops.spmm_raw(gs.rowptr, gs.col, gs.w, xd, n, out=y, accumulate=True, long_rows=gs.long_rows())
    assert_close(y, 2 * fwd + bias.double(), 2e-6, "accumulate")
    yt = ops.spmm_raw(gs.rowptr_t, gs.col_t, gs.w_t, xd, n, addend=add.to(dev), long_rows=gs.long_rows(True))
    assert_close(yt, bwd + add.double(), 2e-6, "transposed + addend")
    plain = ops.spmm_raw(gs.rowptr, gs.col, gs.w, xd, n)                              # one wave per row, hubs walked serially
    split = ops.spmm_raw(gs.rowptr, gs.col, gs.w, xd, n, long_rows=gs.long_rows())
    deg = (gs.rowptr[1:] - gs.rowptr[:-1]).cpu()
    short = deg <= 128
    assert int((~short).sum()) == 3
    assert torch.equal(plain.cpu()[short], split.cpu()[short])                        # short rows: same wave, same order
    # the serial walk of 5000 fp32 terms is the less accurate of the two orders (the split result held 2e-6 against float64 above)
    assert_close(split.cpu()[~short], plain.cpu()[~short].double(), 2e-5, "long rows, two summation orders")
    for _ in range(3):                                                                # arrival order varies, the result does not
        assert torch.equal(ops.spmm_raw(gs.rowptr, gs.col, gs.w, xd, n, long_rows=gs.long_rows()), split)
    # the graph convolution's forward (concat + operand maximum) and backward (addend) go through the same tables
    if c % 32 == 0:
        prev = ops.configure(gemm="f16x2")
        try:
            ea = torch.randn(n, 32, generator=g).to(dev)
            wgt, wge = (torch.randn(64, c, generator=g) / c ** 0.5).to(dev), (torch.randn(64, 32, generator=g) / 6).to(dev)
            xg = xd.clone().requires_grad_(True)
            out = ops.graph_conv_linear(xg, ea, gs, wgt, wge, None)
            ref = torch.cat([fwd, ea.cpu().double()], 1) @ torch.cat([wgt, wge], 1).cpu().double().t()
            assert_close(out, ref, 2e-5, "graph convolution over hub rows")
            out.backward(torch.ones_like(out))
            gref = torch.zeros(n, c, dtype=torch.float64).index_add_(0, src, w[:, None] * (torch.ones(n, 64, dtype=torch.float64) @ wgt.cpu().double())[dst])
            assert_close(xg.grad, gref, 2e-5, "its input gradient")
        finally:
            ops.configure(**prev)


def test_spmm_strided_bias_accumulate_and_edge_attr():
    from dgdm_histopath_lab_amd import GraphStructure, ops
    n, e, c = 777, 3000, 64
    ei = _rand_edges(n, e, 5)
    dev = _dev()
    gs = GraphStructure(torch.from_numpy(ei).to(dev), n)
    o = csr_oracle.gcn_csr(ei, n)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, c, generator=g); bias = torch.randn(c, generator=g); ea = torch.randn(e, 32, generator=g)
    src, dst, w = torch.from_numpy(o["src"]), torch.from_numpy(o["dst"]), torch.from_numpy(o["norm_coo"])
    ref = torch.zeros(n, c).index_add_(0, dst, w[:, None] * x[src])
    buf = torch.full((n, c + 32), 7.0, device=dev)          # write into a column-strided view
    ops.spmm_raw(gs.rowptr, gs.col, gs.w, x.to(dev), n, out=buf[:, :c], bias=bias.to(dev))
    assert_close(buf[:, :c], ref + bias, 1e-5, "strided+bias")
    assert (buf[:, c:] == 7.0).all()
    ops.spmm_raw(gs.rowptr, gs.col, gs.w, x.to(dev), n, out=buf[:, :c], accumulate=True)
    assert_close(buf[:, :c], 2 * ref + bias, 1e-5, "accumulate")
    ea_ext = torch.cat([ea, torch.zeros(n, 32)])            # R1: loops carry zero rows
    ref_ea = torch.zeros(n, 32).index_add_(0, dst, w[:, None] * ea_ext)
    assert_close(ops.aggregate_edge_attr(ea.to(dev), gs), ref_ea, 1e-5, "edge attr aggregate")
    assert ops.aggregate_edge_attr(None, gs) is None


def test_ops_fail_loudly_on_cpu_tensors():
    from dgdm_histopath_lab_amd import DGDMKernelError, GraphStructure
    with pytest.raises(DGDMKernelError):
        GraphStructure(torch.zeros(2, 3, dtype=torch.long), 4)


@pytest.mark.parametrize("n,e,cin,cout", [(3000, 15000, 128, 128), (10000, 50000, 512, 256), (300, 900, 36, 64)])
def test_graph_conv_linear_is_the_two_step_path(n, e, cin, cout):
    """ops.graph_conv_linear (SpMM with the edge aggregate laid beside it, one GEMM, split weight gradients) against
    aggregate_concat + linear on a concatenated weight: same values, same gradients."""
    from dgdm_histopath_lab_amd import GraphStructure, ops
    dev = _dev()
    ei = _rand_edges(n, e, 11)
    gs = GraphStructure(torch.from_numpy(ei).to(dev), n)
    g = torch.Generator().manual_seed(5)
    ea_hat = ops.aggregate_edge_attr(torch.randn(e, 32, generator=g).to(dev), gs)
    gy = torch.randn(n, cout, generator=g).to(dev)

    def leaves():
        gg = torch.Generator().manual_seed(9)
        return [t.to(dev).requires_grad_(True) for t in (torch.randn(n, cin, generator=gg), torch.randn(cout, cin, generator=gg) / cin ** 0.5,
                                                         torch.randn(cout, 32, generator=gg) / 6, torch.randn(cout, generator=gg))]
    x1, w1, we1, b1 = leaves()
    y1 = ops.graph_conv_linear(x1, ea_hat, gs, w1, we1, b1)
    y1.backward(gy)
    x2, w2, we2, b2 = leaves()
    y2 = ops.linear(ops.aggregate_concat(x2, ea_hat, gs), torch.cat([w2, we2], dim=1), b2)
    y2.backward(gy)
    assert torch.equal(y1, y2)
    assert w1.grad.is_contiguous() and we1.grad.is_contiguous()
    assert torch.equal(w1.grad, w2.grad) and torch.equal(we1.grad, we2.grad) and torch.equal(b1.grad, b2.grad)
    # the input gradient is contracted over W alone instead of [W | W_e]: other K, other summation tree
    scale = float(x2.grad.abs().max())
    assert torch.allclose(x1.grad, x2.grad, rtol=1e-4, atol=1e-5 * scale)


def test_graph_conv_skip_alias_adds_the_residual_gradient_in_the_backward_kernel():
    """graph_conv_linear(skip=True): the alias of x carries the residual's gradient into the convolution's backward, where
    dgdm_spmm_add sums it with the scattered gradient -- same numbers as letting autograd add the two."""
    from dgdm_histopath_lab_amd import GraphStructure, ops
    dev = _dev()
    n, e, cin, cout = 3000, 15000, 128, 128
    ei = _rand_edges(n, e, 21)
    gs = GraphStructure(torch.from_numpy(ei).to(dev), n)
    g = torch.Generator().manual_seed(3)
    ea_hat = ops.aggregate_edge_attr(torch.randn(e, 32, generator=g).to(dev), gs)
    gy, gres = torch.randn(n, cout, generator=g).to(dev), torch.randn(n, cin, generator=g).to(dev)

    def leaves():
        gg = torch.Generator().manual_seed(9)
        return [t.to(dev).requires_grad_(True) for t in (torch.randn(n, cin, generator=gg), torch.randn(cout, cin, generator=gg) / cin ** 0.5,
                                                         torch.randn(cout, 32, generator=gg) / 6, torch.randn(cout, generator=gg))]
    x1, w1, we1, b1 = leaves()
    y1, xs = ops.graph_conv_linear(x1, ea_hat, gs, w1, we1, b1, skip=True)
    assert xs.data_ptr() == x1.data_ptr()
    ((y1 * gy).sum() + (xs * gres).sum()).backward()
    x2, w2, we2, b2 = leaves()
    y2 = ops.graph_conv_linear(x2, ea_hat, gs, w2, we2, b2)
    ((y2 * gy).sum() + (x2 * gres).sum()).backward()
    assert torch.equal(y1, y2) and torch.equal(w1.grad, w2.grad) and torch.equal(b1.grad, b2.grad)
    assert torch.allclose(x1.grad, x2.grad, rtol=1e-6, atol=1e-6)      # (a + b) vs a then + b: one rounding apart
    x3 = leaves()[0]
    _, xs3 = ops.graph_conv_linear(x3, ea_hat, gs, w2.detach(), we2.detach(), b2.detach(), skip=True)
    (xs3 * gres).sum().backward()                                      # only the alias used
    assert torch.equal(x3.grad, gres)


@pytest.mark.parametrize("node,hid,n,e", [(128, 128, 3000, 12000), (512, 256, 1500, 6000), (256, 128, 700, 3000), (64, 32, 300, 900)])
@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("mode", [True])
def test_graph_layer_as_one_node_is_the_layer_of_separate_kernels(node, hid, n, e, training, mode, monkeypatch):
    """DynamicGraphLayer through ops._GraphLayer (activations and LayerNorm as GEMM epilogues, the second convolution's input gradient
    associated as (A^T dpre) . W) against the same module on the separate kernels (ops.FUSE_EPILOGUES = False): output and every
    gradient, eval mode and TRAINING mode -- both paths draw the same two dropout seeds in the same order and the epilogues' mask is
    the streaming kernels' function of (seed, element index), so the training-mode results agree to rounding as well; and against a
    float64 composition of the reference's formula in eval mode (core/graph_layers.py:207-247).  (Round 5's "auto" policy -- a draw in
    the same-box A/B -- left the tree in round 6; True stays as the A/B switch.)"""
    from dgdm_histopath_lab_amd import ops
    from dgdm_histopath_lab_amd.core.graph_layers import DynamicGraphLayer, GraphContext
    DEV = "cuda:0"
    torch.manual_seed(node + hid + n)
    layer = DynamicGraphLayer(node, 32, hid, num_heads=8).to(DEV).train(training)
    with torch.no_grad():
        for p in layer.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    g = torch.Generator().manual_seed(n)
    x0 = torch.randn(n, node, generator=g).to(DEV)
    ei = torch.randint(0, n, (2, e), generator=g).to(DEV)
    ea = torch.randn(e, 32, generator=g).to(DEV)
    gy = torch.randn(n, node, generator=g).to(DEV)
    ctx = GraphContext(ei, n, ea)
    names = [k for k, p in layer.named_parameters() if not k.startswith(("node_to_qkv", "edge_to_key", "norm2"))]

    was = ops.FUSE_EPILOGUES

    def run(fused):
        ops.FUSE_EPILOGUES = fused
        ops._seed_counter = 1000
        layer.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        try:
            y = layer(x, ctx)
            y.backward(gy)
        finally:
            ops.FUSE_EPILOGUES = was
        return y.detach(), x.grad, {k: dict(layer.named_parameters())[k].grad.clone() for k in names}, type(y.grad_fn).__name__

    yf, dxf, gf, nf = run(mode)
    yu, dxu, gu, nu = run(False)
    assert "GraphLayer" in nf and "GraphLayer" not in nu
    assert_close(yf, yu, 2e-5, "output")
    assert_close(dxf, dxu, 2e-5, "dx")
    for k in names:
        assert_close(gf[k], gu[k], 5e-5, k)
    if training:
        assert float((yf == 0).float().mean()) < 1e-3       # LayerNorm output: no dropout behind it
        return
    # float64 composition of the reference formula
    P = {k: v.detach().double().cpu() for k, v in layer.named_parameters()}
    xd = x0.double().cpu().requires_grad_(True)
    src, dst = torch.cat([ei[0].cpu(), torch.arange(n)]), torch.cat([ei[1].cpu(), torch.arange(n)])
    ead = torch.cat([ea.double().cpu(), torch.zeros(n, 32, dtype=torch.float64)])
    deg = torch.zeros(n, dtype=torch.float64).index_add_(0, dst, torch.ones(e + n, dtype=torch.float64))
    norm = deg[src].rsqrt() * deg[dst].rsqrt()

    def conv(h, pre):
        msg = (h @ P[pre + ".node_lin.weight"].t())[src] + ead @ P[pre + ".edge_lin.weight"].t()
        return torch.zeros(n, P[pre + ".bias"].numel(), dtype=torch.float64).index_add_(0, dst, msg * norm[:, None]) + P[pre + ".bias"]
    gelu = torch.nn.functional.gelu
    h = gelu(conv(gelu(conv(xd, "graph_conv1")), "graph_conv2"))
    out = torch.nn.functional.layer_norm(h @ P["output_proj.weight"].t() + P["output_proj.bias"] + xd, (node,), P["norm1.weight"], P["norm1.bias"])
    out.backward(gy.double().cpu())
    assert_close(yf, out.detach(), 1e-4, "output vs float64")
    assert_close(dxf, xd.grad, 1e-4, "dx vs float64")


@pytest.mark.gpu
def test_long_row_tables_are_only_allocated_when_the_degree_bound_allows_a_long_row():
    """graph.py (VERDICT r3-r5 note): with a host-side degree bound the index sets of ordinary tissue graphs carry no long-row tables and
    no 23 MB scratch; the result is the same bits, and a WRONG bound (a hub the loader did not know of) costs time, not correctness."""
    from dgdm_histopath_lab_amd import ops
    from dgdm_histopath_lab_amd.graph import GraphStructure
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(3)
    n, e = 3000, 15000
    ei = torch.randint(0, n, (2, e), generator=g)
    ei[1, :400] = 7                                       # one hub of 400 incoming edges
    x = torch.randn(n, 64, generator=g).to(DEV)
    full = GraphStructure(ei.to(DEV), n)
    hinted = GraphStructure(ei.to(DEV), n, max_degree=int(max(torch.bincount(ei[0]).max(), torch.bincount(ei[1]).max())))
    wrong = GraphStructure(ei.to(DEV), n, max_degree=9)
    assert full.long_tables is not None and hinted.long_tables is not None and wrong.long_tables is None and wrong.long_partial is None
    ys = [ops.spmm_raw(s.rowptr, s.col, s.w, x, n, long_rows=s.long_rows()) for s in (full, hinted, wrong)]
    for s in (full, hinted, wrong):
        s.assert_ok()
        assert torch.equal(s.rowptr, full.rowptr) and torch.equal(s.col, full.col) and torch.equal(s.eid, full.eid) and torch.equal(s.w, full.w)
    assert torch.equal(ys[0], ys[1])
    torch.testing.assert_close(ys[2], ys[0], rtol=1e-5, atol=1e-5)        # the hub's row summed by one wave instead of segments: other order
    ei2 = torch.randint(0, n, (2, e), generator=g)
    small = GraphStructure(ei2.to(DEV), n, max_degree=int(max(torch.bincount(ei2[0]).max(), torch.bincount(ei2[1]).max())))
    ref = GraphStructure(ei2.to(DEV), n)
    assert small.long_tables is None and ref.long_tables is not None
    assert torch.equal(ops.spmm_raw(small.rowptr, small.col, small.w, x, n, long_rows=small.long_rows()),
                       ops.spmm_raw(ref.rowptr, ref.col, ref.w, x, n, long_rows=ref.long_rows()))

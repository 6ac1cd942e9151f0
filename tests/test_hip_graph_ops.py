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
                                     (10000, 50000, False), (70001, 300007, False)])
def test_csr_build_bit_exact(n, e, hub):
    from dgdm_histopath_lab_amd import GraphStructure
    ei = _rand_edges(n, e, n + e, hub)
    gs = GraphStructure(torch.from_numpy(ei).to(_dev()), n)
    o = csr_oracle.gcn_csr(ei, n)
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
    src, dst, w = torch.from_numpy(o["src"]), torch.from_numpy(o["dst"]), torch.from_numpy(o["norm_coo"])
    xr = x.clone().requires_grad_(True)
    yr = torch.zeros(n, c).index_add_(0, dst, w[:, None] * xr[src])
    yr.backward(gy)
    xd = x.to(_dev()).requires_grad_(True)
    y = ops.aggregate(xd, gs)
    y.backward(gy.to(_dev()))
    assert_close(y, yr, 1e-5, "Y")
    assert_close(xd.grad, xr.grad, 1e-5, "dX")
    y2 = ops.aggregate(xd.detach(), gs)
    assert torch.equal(y2, y.detach())  # bitwise reproducible (no float atomics)


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

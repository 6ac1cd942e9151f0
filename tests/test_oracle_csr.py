"""Integer oracle: hand-computed known answers (G8) for CSR build, GCN norm and top-k pooling."""
import numpy as np

from oracle import csr_oracle as C


def test_csr_known_answer_path_graph():
    # 4-node path 0-1-2-3, both directions, plus a duplicate edge 1->2 and an existing loop 3->3
    ei = np.array([[0, 1, 1, 2, 2, 3, 1, 3],
                   [1, 0, 2, 1, 3, 2, 2, 3]])
    g = C.gcn_csr(ei, 4)
    # destination rows (edge ids, loops get 8+i): d0: e1, L0 | d1: e0,e3,L1 | d2: e2,e5,e6,L2 | d3: e4,e7,L3
    assert g["rowptr"].tolist() == [0, 2, 5, 9, 12]
    assert g["eid"].tolist() == [1, 8, 0, 3, 9, 2, 5, 6, 10, 4, 7, 11]
    assert g["col"].tolist() == [1, 0, 0, 2, 1, 1, 3, 1, 2, 2, 3, 3]
    # by source: s0: e0,L0 | s1: e1,e2,e6,L1 | s2: e3,e4,L2 | s3: e5,e7,L3
    assert g["rowptr_t"].tolist() == [0, 2, 6, 9, 12]
    assert g["eid_t"].tolist() == [0, 8, 1, 2, 6, 9, 3, 4, 10, 5, 7, 11]
    assert g["col_t"].tolist() == [1, 0, 0, 2, 2, 1, 1, 3, 2, 2, 3, 3]
    deg = np.array([2, 3, 4, 3], dtype=np.float32)  # in-degree incl. the appended loop
    np.testing.assert_allclose(g["dinv"], deg ** -0.5, rtol=1e-6)
    # norm uses the in-degree for BOTH ends (graph_layers.py:80-84)
    e = 6  # edge 1->2
    p = g["eid"].tolist().index(e)
    np.testing.assert_allclose(g["norm"][p], (3 ** -0.5) * (4 ** -0.5), rtol=1e-6)
    assert g["norm"].dtype == np.float32 and g["rowptr"].dtype == np.int32


def test_csr_isolated_and_empty():
    g = C.gcn_csr(np.zeros((2, 0), dtype=np.int64), 3)
    assert g["rowptr"].tolist() == [0, 1, 2, 3] and g["col"].tolist() == [0, 1, 2]
    np.testing.assert_allclose(g["norm"], 1.0)
    g = C.gcn_csr(np.array([[0], [2]]), 3, add_loops=False)
    assert g["rowptr"].tolist() == [0, 0, 0, 1]
    assert g["dinv"].tolist() == [0.0, 0.0, 1.0]  # deg 0 -> inf -> 0 (graph_layers.py:83)
    assert g["norm"].tolist() == [0.0]            # dinv[src=0] = 0


def test_csr_stable_order_random():
    rng = np.random.default_rng(0)
    n, e = 50, 400
    ei = rng.integers(0, n, size=(2, e))
    g = C.gcn_csr(ei, n)
    for d in range(n):
        seg = g["eid"][g["rowptr"][d]:g["rowptr"][d + 1]]
        assert (np.diff(seg) > 0).all()
        assert (g["dst"][seg] == d).all() and (g["src"][seg] == g["col"][g["rowptr"][d]:g["rowptr"][d + 1]]).all()


def test_topk_pool_known_answer():
    scores = np.array([0.1, 0.9, -0.3, 0.5, 0.7, 0.2], dtype=np.float32)
    ei = np.array([[0, 1, 3, 4, 4, 5], [1, 3, 4, 1, 2, 4]])
    r = C.topk_pool_indices(scores, ei, 0.5)  # k = 3 -> nodes 1, 4, 3 -> perm ascending [1, 3, 4]
    assert r["perm"].tolist() == [1, 3, 4]
    assert r["edge_keep"].tolist() == [1, 2, 3]
    assert r["edge_index"].tolist() == [[0, 1, 2], [1, 2, 0]]
    r1 = C.topk_pool_indices(np.array([0.3], dtype=np.float32), np.zeros((2, 0), dtype=np.int64), 0.5)
    assert r1["perm"].tolist() == [0]  # k = max(1, 0)

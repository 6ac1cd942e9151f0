"""The graph-construction oracle against the reference's own TissueGraphBuilder._create_edges outputs
(tests/golden/g8_graph_build_*.npz, written by oracle/capture_graph_golden.py by running the reference
with scikit-learn in the dev container)."""
import glob
import os

import numpy as np
import pytest

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "g8_graph_build_*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[15:-4] for p in GOLD])
def test_oracle_reproduces_reference_edges(path):
    from oracle.graph_build_oracle import create_edges, to_edge_arrays
    z = np.load(path)
    sk, mk, thr = int(z["params"][0]), int(z["params"][1]), float(z["params"][2])
    out = create_edges(z["features"], z["coords"], sk, mk, thr)
    assert np.array_equal(out["src"], z["ref_src"]) and np.array_equal(out["tgt"], z["ref_tgt"])      # index work: exact, in order
    assert np.array_equal(out["type"], z["ref_type"])
    np.testing.assert_allclose(out["weight"], z["ref_weight"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(out["feat"], z["ref_feat"], rtol=1e-6, atol=1e-7)
    ei, ea, et = to_edge_arrays(out)
    u = out["src"].shape[0]
    assert ei.shape == (2, 2 * u) and ea.shape == (2 * u, 32) and et.shape == (2 * u,)
    assert np.array_equal(ei[:, 0::2], ei[::-1, 1::2])                      # both directions, consecutive (builder :384-386)
    keys = {(min(a, b), max(a, b)) for a, b in zip(out["src"], out["tgt"])}
    assert len(keys) == u                                                   # deduplicated


def test_golden_set_is_present():
    assert len(GOLD) >= 4

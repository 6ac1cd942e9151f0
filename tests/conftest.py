import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def weights(g, prefix="w."):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def T(a):
    return torch.from_numpy(np.asarray(a))


def assert_close(a, b, tol=1e-3, name=""):
    """SURVEY.md 8(d) parity metric: max|a-b| <= tol*max(1,max|b|) and rel-L2 <= tol."""
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{name}: shape {a.shape} vs {b.shape}"
    assert np.isfinite(a).all(), f"{name}: non-finite values"
    mx = np.abs(a - b).max() if a.size else 0.0
    scale = max(1.0, np.abs(b).max() if b.size else 0.0)
    nb = np.linalg.norm(b)
    rel = np.linalg.norm(a - b) / nb if nb > 0 else np.linalg.norm(a - b)
    assert mx <= tol * scale, f"{name}: max abs err {mx:.3e} > {tol}*{scale:.3e}"
    assert rel <= tol, f"{name}: rel-L2 {rel:.3e} > {tol}"
    return mx, rel


@pytest.fixture(scope="session")
def golden():
    return load_golden

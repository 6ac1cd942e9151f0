import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


BENCH_REHEARSAL = {}      # filled by pytest_configure on a GPU run: {"proc": Popen, "out": path, "err": path}


def _start_bench_rehearsal():
    """The N > 1 forms of bench.py (plain `python bench.py --gpus 2`, the same with `--mixed`, and the launcher form the driver
    uses), run ONE AFTER THE OTHER by the child process tests/_bench_rehearsal.py as fresh processes, two ranks on the one GPU of
    the box over gloo, started here -- before anything in this process has touched the GPU: a process that has initialised HIP
    must not fork+exec on this pool.  tests/test_parallel.py::test_bench_two_rank_rehearsal_* wait for it and check the lines; the
    ranks run beside the first tests of the session (3 processes on the card at any time, limit 6)."""
    import subprocess
    import tempfile
    d = tempfile.mkdtemp(prefix="dgdm_bench_rehearsal_")
    log = open(os.path.join(d, "driver.log"), "w")
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_bench_rehearsal.py"), d], stdout=log, stderr=log, cwd=ROOT)
    BENCH_REHEARSAL.update(proc=proc, dir=d)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    expr = (config.getoption("-m") or "").strip()
    # torch.cuda.device_count() does not initialise the GPU on this image (torch.cuda.is_available() does)
    if expr == "gpu" and os.environ.get("DGDM_NO_BENCH_REHEARSAL") != "1" and torch.cuda.device_count() > 0:
        _start_bench_rehearsal()


def pytest_unconfigure(config):
    proc = BENCH_REHEARSAL.get("proc")
    if proc is not None and proc.poll() is None:      # the waiting test was deselected or never reached: let the running case finish
        try:                                           # (its ranks are grandchildren: killing the driver would orphan them on the card)
            proc.wait(900)
        except Exception:
            proc.kill()


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


# ---------------------------------------------------------------------------------------------
# Discrete decisions of the graph U-Net (ReLU kinks, top-k selections).
#
# Gradients of a ReLU / top-k network are only comparable between two implementations when both
# took the same side of every kink.  A pre-activation within rounding of zero may fall on either
# side (2 of 512 000 at 2 x 2 000 nodes with exact fp32 kernels), so the parity tests
# run the checker first, hand ITS decisions to the HIP model (`decisions=`, GraphUNet.forward) and
# then hold every decision the HIP model would have taken differently to the rounding margin: a
# flipped element must have |own pre-activation| <= DECISION_MARGIN * max(1, max|pre|), which is
# what "the two activations agree to the tolerance and differ in sign" implies.  A backward bug or
# a wrong activation cannot hide behind this: it shows up as a flip outside the margin, as a
# gradient mismatch, or both.
DECISION_MARGIN = 2e-4
FLIPS_PER_MARGIN = 5.0     # a unit-scale activation has ~0.8 * margin * max|pre| of its mass within the margin of zero; at most that can flip


def decisions_from_trace(trace):
    """Oracle trace (post-ReLU tensors `relu.*`, `perm*`) -> the `decisions` dict of DGDMModel.forward."""
    dec = {}
    for k, v in trace.items():
        if k.startswith("relu."):
            dec[k] = (v.detach() > 0)
        elif k.startswith("perm") and k[4:].isdigit():
            dec[k] = v.detach().clone()
    return dec


def decisions_from_golden(g):
    """`dec.*` arrays of a g7_model_* fixture (recorded from the reference's own run by oracle/capture_golden.py)."""
    return {k[4:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("dec.") and not k.startswith("dec.margin.")}


def check_decision_margins(own_trace, decisions, margin=DECISION_MARGIN):
    max_fraction = FLIPS_PER_MARGIN * margin
    """Every decision the HIP run would have taken differently lies within the rounding margin.  Returns (#flips, #decisions)."""
    flips = total = 0
    for k, mask in decisions.items():
        if k.startswith("relu."):
            pre = own_trace["pre." + k[5:]].detach().cpu().double()
            mask = mask.cpu().reshape(pre.shape)
            diff = (pre > 0) != mask
            n = int(diff.sum())
            total += pre.numel()
            if n:
                worst = float(pre[diff].abs().max())
                bound = margin * max(1.0, float(pre.abs().max()))
                assert worst <= bound, f"{k}: a ReLU decision differs from the reference at |pre-activation| = {worst:.3e} > {bound:.3e}"
                assert n <= max(2, max_fraction * pre.numel()), f"{k}: {n} of {pre.numel()} ReLU decisions differ"
            flips += n
        else:  # perm{i}: the kept node set may differ only by scores tied with the k-th within the margin
            i = k[4:]
            own, s = own_trace["own_perm" + i].cpu(), own_trace["score" + i].detach().cpu().double()
            ref = mask.cpu()
            total += s.numel()
            if not torch.equal(own, ref):
                kth = float(s[own].min())
                sym = torch.tensor(sorted(set(own.tolist()) ^ set(ref.tolist())), dtype=torch.long)
                worst = float((s[sym] - kth).abs().max())
                assert worst <= margin, f"{k}: top-k selection differs by nodes whose score is {worst:.3e} from the k-th"
                assert sym.numel() <= max(2, max_fraction * s.numel()), f"{k}: {sym.numel()} nodes differ"
                flips += sym.numel()
    return flips, total

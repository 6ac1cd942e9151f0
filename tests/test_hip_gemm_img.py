"""GPU parity of the weight-image GEMMs (csrc/gemm_img.hip: y = x W^T + b and dx = dy W with the weight pre-split into an fp16
hi+lo image) against float64 torch, and of the image cache's validity rules."""
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("m,k,n", [(1, 16, 4), (31, 48, 12), (130, 32, 100), (1000, 768, 512), (4001, 544, 512), (2000, 160, 128),
                                   (777, 128, 384), (5000, 288, 256), (129, 1024, 260), (40000, 128, 128), (333, 80, 129)])
def test_rows_img_matches_float64(m, k, n):
    from dgdm_histopath_lab_amd import ops
    g = torch.Generator().manual_seed(m + k + n)
    x = torch.randn(m, k, generator=g); w = torch.randn(n, k, generator=g) / k ** 0.5; b = torch.randn(n, generator=g)
    gy = torch.randn(m, n, generator=g)
    xd, wd, bd, gd = x.to(DEV), w.to(DEV), b.to(DEV), gy.to(DEV)
    assert ops.USE_WEIGHT_IMAGES
    y = ops.gemm_nt_raw(xd, wd, bd, math="f16x2")
    assert_close(y, x.double() @ w.double().t() + b.double(), 1e-5, "nt")
    y2 = ops.gemm_nt_raw(xd, wd, None, out=y.clone(), accumulate=True, math="f16x2")
    assert_close(y2, 2 * (x.double() @ w.double().t()) + b.double(), 1e-5, "nt accumulate")
    if n % 4 == 0:
        dx = ops.gemm_nn_raw(gd, wd, math="f16x2")
        assert_close(dx, gy.double() @ w.double(), 1e-5, "nn")
    # the image path is the one that ran: the weight carries an image now
    assert "_dgdm_images" in wd.__dict__ and any(kind == 0 for kind, _ in wd.__dict__["_dgdm_images"])


def test_rows_img_is_the_register_staged_kernel_up_to_summation_order():
    """Same operands, same split, same three products per term: the two f16x2 kernels differ only in the order of the fp32 sums."""
    from dgdm_histopath_lab_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3000, 544, generator=g).to(DEV); w = (torch.randn(256, 544, generator=g) / 23).to(DEV)
    ref = x.double() @ w.double().t()
    y_img = ops.gemm_nt_raw(x, w, None, math="f16x2")
    ops.USE_WEIGHT_IMAGES = False
    try:
        y_reg = ops.gemm_nt_raw(x, w, None, math="f16x2")
    finally:
        ops.USE_WEIGHT_IMAGES = True
    e_img = float((y_img.double() - ref).abs().max()); e_reg = float((y_reg.double() - ref).abs().max())
    assert e_img < 2 * e_reg + 1e-6 and float((y_img - y_reg).abs().max()) < 4e-6 * float(ref.abs().max())


def test_column_range_of_an_image_and_two_matrix_weight():
    from dgdm_histopath_lab_amd import ops
    g = torch.Generator().manual_seed(11)
    m, k0, k1, n = 2500, 128, 32, 320
    a = torch.randn(m, k0 + k1, generator=g).to(DEV)
    w0, w1 = torch.randn(n, k0, generator=g).to(DEV) / 12, torch.randn(n, k1, generator=g).to(DEV) / 12
    b = torch.randn(n, generator=g).to(DEV)
    ref = a.double() @ torch.cat([w0, w1], 1).double().t() + b.double()
    assert_close(ops.gemm_nt_split_raw(a, w0, w1, b, math="f16x2"), ref, 1e-5, "two-matrix weight")
    e = ops.WEIGHT_IMAGES.get(0, w0, w1)
    # columns [64, 64 + 200) of the product from tiles [2, ...) of the same image
    part = ops._gemm_rows_img(a, e, 2, 200, b[64:264].contiguous(), None, False)
    assert_close(part, ref[:, 64:264], 1e-5, "tile range")
    # dx = dy . W through the transposed image, W a column slice of a wider matrix
    wide = torch.randn(n, 800, generator=g).to(DEV) / 20
    gy = torch.randn(m, n, generator=g).to(DEV)
    assert_close(ops.gemm_nn_raw(gy, wide[:, 128:128 + 544], math="f16x2"), gy.double() @ wide[:, 128:128 + 544].double(), 1e-5, "nn slice")


def test_image_follows_the_weight():
    """An image is never served stale: an in-place write bumps the version counter (rebuilt on the next call); writes behind
    autograd's back (``.data``, a graph replay) are announced by ``weights_changed`` / the refresh at the start of a forward."""
    from dgdm_histopath_lab_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1000, 64, generator=g).to(DEV)
    lin = torch.nn.Linear(64, 96).to(DEV)
    w = lin.weight
    with torch.no_grad():
        y0 = ops.gemm_nt_raw(x, w, None, math="f16x2")
        assert_close(y0, x.double() @ w.double().t(), 1e-5, "first")
        w.mul_(3.0)                                           # in place: version counter moves
        assert_close(ops.gemm_nt_raw(x, w, None, math="f16x2"), x.double() @ w.double().t(), 1e-5, "after in-place")
        w.data.mul_(0.5)                                      # no version bump
        ops.weights_changed()
        assert_close(ops.gemm_nt_raw(x, w, None, math="f16x2"), x.double() @ w.double().t(), 1e-5, "after weights_changed")
        # registered parameters: rebuilt by the one launch of refresh_weight_amax
        for _ in range(3):
            w.data.mul_(1.7)
            ops.refresh_weight_amax(lin)
            assert_close(ops.gemm_nt_raw(x, w, None, math="f16x2"), x.double() @ w.double().t(), 1e-5, "after refresh")
            assert_close(ops.gemm_nn_raw(torch.ones(1000, 96, device=DEV), w, math="f16x2"),
                         torch.ones(1000, 96, dtype=torch.float64) @ w.double().cpu(), 1e-5, "nn after refresh")
        assert ops.WEIGHT_IMAGES.table is not None and ops.WEIGHT_IMAGES.table_n >= 2


def test_asm_issued_loads_are_retired_before_use_staging_canary(monkeypatch):
    """the image GEMMs and the dW kernel issue their activation loads as inline asm and retire them with
    one hand-placed s_waitcnt per stage; the compile-time zero-spill check (tests/test_abi.py) cannot see a consumer that the
    scheduler moved above the wait.  The canary build of the same kernels (lib/canary, -DDGDM_STAGE_CANARY) poisons every staging
    register with NaN when its load is issued: on cache-cold operands (a 1 GiB buffer rewritten before each call, so the loads take
    their full HBM latency) a premature read multiplies NaNs into the result.  Results must be finite AND bit-identical to the
    shipped build's."""
    import os
    from dgdm_histopath_lab_amd import _build, _lib, ops
    assert os.path.exists(_build.CANARY_PATH), "lib/canary/libdgdm_hip.so is built by __graft_entry__.build()"
    canary = _lib.open_library(_build.CANARY_PATH)
    prev = ops.configure(gemm="f16x2")
    try:
        g = torch.Generator().manual_seed(12)
        evict = torch.zeros(256 * 1024 * 1024, dtype=torch.float32, device=DEV)     # 1 GiB: L2 and the Infinity Cache turn over
        cases = [(40000, 768, 512), (40000, 512, 128), (20000, 160, 128), (5000, 128, 128), (4099, 144, 36)]
        for m, k, n in cases:
            x = torch.randn(m, k, generator=g).to(DEV)
            w = (torch.randn(n, k, generator=g) / k ** 0.5).to(DEV)
            b = torch.randn(n, generator=g).to(DEV)
            gy = torch.randn(m, n, generator=g).to(DEV)

            def run():
                ops.weights_changed()
                evict.add_(1.0)
                y = ops.gemm_nt_raw(x, w, b, math="f16x2")                # k_gemm_img8 / k_gemm_img<4,1> (weight image as B)
                evict.add_(1.0)
                dx = ops.gemm_nn_raw(gy, w, math="f16x2")                 # the dx form of the same kernels
                evict.add_(1.0)
                dw, db = ops.gemm_tn_raw(gy, x, True, math="f16x2")       # k_gemmh_tn32
                torch.cuda.synchronize()
                return y, dx, dw, db
            want = run()
            monkeypatch.setattr(_lib, "_lib", canary)
            got = run()
            monkeypatch.undo()
            for name, a, c in zip(("y", "dx", "dW", "db"), want, got):
                assert torch.isfinite(c).all(), f"{name} at {(m, k, n)}: a staging register was read before its load landed"
                assert torch.equal(a, c), f"{name} at {(m, k, n)}"
            ref = x.double() @ w.double().t() + b.double()
            assert_close(got[0], ref, 1e-5, "y vs float64")
        del evict
    finally:
        ops.configure(**prev)

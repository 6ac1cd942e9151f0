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
        # (150000, 160, 128): more workgroups than resident slots in the narrow kernel (a second round of workgroups on warm CUs)
        cases = [(40000, 768, 512), (40000, 512, 128), (20000, 160, 128), (5000, 128, 128), (4099, 144, 36), (150000, 160, 128)]
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
                extra = ()
                if n % 32 == 0 and n <= 256:                              # the fused epilogues of the same kernels (round 5)
                    e = ops.WEIGHT_IMAGES.get(0, w)
                    evict.add_(1.0)
                    ya, pre = ops.gemm_img_act_raw(x, e, n, b, 1, 0.1, 77)
                    evict.add_(1.0)
                    yn, ssum, mean, rstd = ops.gemm_img_norm_raw(x, e, n, b, gy, b, b, 1, 1e-5, 1, 0.1, 78)
                    evict.add_(1.0)
                    gb = ops.gemm_img_act_bwd_raw(gy, ops.WEIGHT_IMAGES.get(1, w), k, x, 1, 0.1, 79) if k % 32 == 0 and k <= 256 else ya
                    extra = (ya, pre, yn, ssum, mean, rstd, gb)
                torch.cuda.synchronize()
                return (y, dx, dw, db) + extra
            want = run()
            monkeypatch.setattr(_lib, "_lib", canary)
            got = run()
            monkeypatch.undo()
            for name, a, c in zip(("y", "dx", "dW", "db", "act y", "act pre", "norm y", "norm sum", "norm mean", "norm rstd", "act-bwd g"), want, got):
                assert torch.isfinite(c).all(), f"{name} at {(m, k, n)}: a staging register was read before its load landed"
                assert torch.equal(a, c), f"{name} at {(m, k, n)}"
            ref = x.double() @ w.double().t() + b.double()
            assert_close(got[0], ref, 1e-5, "y vs float64")
        del evict
    finally:
        ops.configure(**prev)


# ---------------------------------------------------------------------------------------------------------------------------------
# fused epilogues (round 5): the layer behind a Linear finished in the GEMM's registers

def _amax_bits(t, slot):
    """Maximum over the ways of an amax slot group (device address `slot`), as the integer the kernels keep (float bits)."""
    from dgdm_histopath_lab_amd.ops import _arena
    ar = _arena(t.device)
    words = ar.buf.view(torch.int32)
    i0 = (slot - ar.buf.data_ptr()) // 4
    return int(words[i0:i0 + 32 * 64:64].max().item())       # DGDM_AMAX_WAYS x DGDM_AMAX_STRIDE (include/dgdm_hip.h)


def _operands(m, k, n, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(m, k, generator=g).to(DEV)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(DEV)
    b = torch.randn(n, generator=g).to(DEV)
    return x, w, b


@pytest.mark.parametrize("m,k,n", [(1000, 160, 128), (4100, 544, 512), (300, 288, 256), (257, 64, 36), (40000, 160, 128), (2049, 128, 384)])
@pytest.mark.parametrize("act,p", [(1, 0.1), (1, 0.0), (3, 0.25), (2, 0.1), (0, 0.5), (4, 0.1)])
def test_act_epilogue_is_the_gemm_followed_by_the_activation_kernel(m, k, n, act, p):
    """dgdm_gemm_rows_img_act (core/graph_layers.py:233-239, dropout(GELU(conv(x))) as the GEMM's epilogue) against the two launches
    it replaces: the stored pre-activation equals the plain GEMM's output to fp32 rounding, the output is dgdm_act_dropout_fwd of
    that pre-activation bit for bit (same activation code, same (seed, element index) mask), and against float64."""
    from dgdm_histopath_lab_amd import _lib, ops
    x, w, b = _operands(m, k, n, m + k + n + act)
    seed = 0x1234567 + act
    e = ops.WEIGHT_IMAGES.get(0, w)
    y, pre = ops.gemm_img_act_raw(x, e, n, b, act, p, seed)
    plain = ops.gemm_nt_raw(x, w, b, math="f16x2")
    ref = x.double() @ w.double().t() + b.double()
    assert_close(pre, ref, 1e-5, "pre vs float64")
    assert float((pre - plain).abs().max()) <= 2e-6 * float(ref.abs().max())
    want = torch.empty_like(pre)
    _lib.check(_lib.load().dgdm_act_dropout_fwd(pre.data_ptr(), pre.numel(), act, p, seed, want.data_ptr(), None, None,
                                                _lib.stream_ptr(pre.device)), "act")
    assert torch.equal(y, want)
    if p > 0:
        kept = float((y != 0).float().mean()) / max(float((torch.nn.functional.gelu(pre) != 0).float().mean()) if act == 1 else 1.0, 1e-9)
        if act in (0, 1, 3):
            assert abs(kept - (1 - p)) < 0.02, kept
    # the operand maximum the epilogue kept is max|y| exactly (integer maximum of float bits)
    slot = ops.amax_of(y)
    assert slot is not None and _amax_bits(y, slot) == int(y.abs().max().view(torch.int32).item())


@pytest.mark.parametrize("m,k,n", [(1000, 128, 128), (4100, 512, 512), (300, 256, 256), (40000, 128, 128), (2049, 384, 128)])
@pytest.mark.parametrize("act,p", [(1, 0.1), (1, 0.0), (3, 0.25), (2, 0.1), (4, 0.1)])
def test_act_backward_epilogue_is_the_gemm_followed_by_the_activation_backward(m, k, n, act, p):
    """dgdm_gemm_rows_img_act_bwd: G = (dY . W) * act'(pre) * mask from one launch, against dgdm_gemm_rows_img followed by
    dgdm_act_dropout_bwd (same derivative code, same mask)."""
    from dgdm_histopath_lab_amd import _lib, ops
    g = torch.Generator().manual_seed(m + k + n)
    dy = torch.randn(m, k, generator=g).to(DEV)
    w = (torch.randn(k, n, generator=g) / k ** 0.5).to(DEV)          # dx = dy . w
    pre = torch.randn(m, n, generator=g).to(DEV)
    seed = 0x7654321
    e = ops.WEIGHT_IMAGES.get(1, w)
    got = ops.gemm_img_act_bwd_raw(dy, e, n, pre, act, p, seed)
    plain = ops.gemm_nn_raw(dy, w, math="f16x2")
    want = torch.empty_like(plain)
    _lib.check(_lib.load().dgdm_act_dropout_bwd(pre.data_ptr(), plain.data_ptr(), plain.numel(), act, p, seed, want.data_ptr(), None, None,
                                                _lib.stream_ptr(pre.device)), "act bwd")
    assert torch.equal(got == 0, want == 0)
    assert float((got - want).abs().max()) <= 4e-6 * float(want.abs().max())
    assert ops.amax_of(got) is not None


@pytest.mark.parametrize("m,k,n,groups", [(1000, 128, 128, 1), (4100, 128, 256, 1), (300, 384, 512, 8), (2049, 512, 256, 8), (40000, 128, 128, 1),
                                          (777, 64, 64, 2), (513, 256, 256, 2), (1500, 64, 32, 1)])
@pytest.mark.parametrize("with_res,act,p", [(True, 0, 0.0), (False, 3, 0.1), (True, 1, 0.25)])
def test_norm_epilogue_is_the_gemm_followed_by_the_row_norm_kernel(m, k, n, groups, with_res, act, p):
    """dgdm_gemm_rows_img_norm (LayerNorm(out_proj(h) + x), core/graph_layers.py:241-245; GroupNorm(8) + SiLU + dropout behind the
    denoiser's Linears, core/diffusion.py:94-102) against dgdm_gemm_rows_img + dgdm_rownorm_fwd and against float64."""
    from dgdm_histopath_lab_amd import _lib, ops
    assert ops.gemm_img_norm_supported(n, groups)
    x, w, b = _operands(m, k, n, m + k + n + groups)
    g = torch.Generator().manual_seed(5)
    res = torch.randn(m, n, generator=g).to(DEV) if with_res else None
    gamma, beta = (1 + 0.3 * torch.randn(n, generator=g)).to(DEV), (0.2 * torch.randn(n, generator=g)).to(DEV)
    seed = 0x13572468
    e = ops.WEIGHT_IMAGES.get(0, w)
    y, ssum, mean, rstd = ops.gemm_img_norm_raw(x, e, n, b, res, gamma, beta, groups, 1e-5, act, p, seed)
    s64 = x.double() @ w.double().t() + b.double() + (res.double() if with_res else 0)
    assert_close(ssum, s64, 1e-5, "sum")
    v = s64.view(m * groups, n // groups)
    mu, var = v.mean(1), v.var(1, unbiased=False)
    assert_close(mean, mu, 1e-5, "mean")
    assert_close(rstd, (var + 1e-5).rsqrt(), 1e-4, "rstd")
    plain = ops.gemm_nt_raw(x, w, b, math="f16x2")
    want, m2, r2 = torch.empty_like(plain), torch.empty_like(mean), torch.empty_like(rstd)
    _lib.check(_lib.load().dgdm_rownorm_fwd(plain.data_ptr(), _lib.ptr(res), gamma.data_ptr(), beta.data_ptr(), m, n, groups, 1e-5, act, p,
                                            seed, want.data_ptr(), m2.data_ptr(), r2.data_ptr(), None, _lib.stream_ptr(plain.device)), "rownorm")
    assert torch.equal(y == 0, want == 0) or p == 0            # the same dropout mask
    assert float((y - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    assert ops.amax_of(y) is not None


def test_norm_epilogue_shapes_that_are_not_taken():
    from dgdm_histopath_lab_amd import ops
    assert not ops.gemm_img_norm_supported(512, 1)        # a 512-wide row spans two workgroups
    assert not ops.gemm_img_norm_supported(96, 1) and not ops.gemm_img_norm_supported(192, 2)   # 96 is not a power-of-two number of tiles
    assert ops.gemm_img_norm_supported(512, 8) and ops.gemm_img_norm_supported(256, 1) and ops.gemm_img_norm_supported(128, 4)

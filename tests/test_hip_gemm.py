"""GPU parity of the GEMMs (K3: fp32 MFMA, and the exact bf16x3 split on the 16-bit pipe) against float64 torch."""
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


MATHS = ["fp32", "bf16x3", "f16x2"]


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("m,k,n", [(1, 4, 4), (130, 36, 12), (1000, 768, 512), (4001, 544, 512), (2000, 160, 128), (777, 128, 384),
                                   (5000, 288, 256), (300, 40, 64), (129, 1024, 260), (40000, 128, 128)])
def test_gemm_nt_nn_tn(m, k, n, math):
    from dgdm_histopath_lab_amd import ops
    g = torch.Generator().manual_seed(m + k + n)
    x = torch.randn(m, k, generator=g); w = torch.randn(n, k, generator=g) / k ** 0.5; b = torch.randn(n, generator=g)
    gy = torch.randn(m, n, generator=g)
    xd, wd, bd, gd = x.to(DEV), w.to(DEV), b.to(DEV), gy.to(DEV)
    y = ops.gemm_nt_raw(xd, wd, bd, math=math)
    assert_close(y, x.double() @ w.double().t() + b.double(), 1e-5, "nt")
    y2 = ops.gemm_nt_raw(xd, wd, None, out=y.clone(), accumulate=True, math=math)
    assert_close(y2, 2 * (x.double() @ w.double().t()) + b.double(), 1e-5, "nt accumulate")
    dx = ops.gemm_nn_raw(gd, wd, math=math)
    assert_close(dx, gy.double() @ w.double(), 1e-5, "nn")
    dW, db = ops.gemm_tn_raw(gd, xd, True, math=math)
    assert_close(dW, gy.double().t() @ x.double(), 1e-5, "tn dW")
    assert_close(db, gy.double().sum(0), 1e-5, "tn db")
    dW2, db2 = ops.gemm_tn_raw(gd, xd, True, math=math)
    assert torch.equal(dW2, dW) and torch.equal(db2, db)    # fixed reduction order: bitwise reproducible
    dW3, none = ops.gemm_tn_raw(gd, xd, False, math=math)
    assert none is None
    assert_close(dW3, gy.double().t() @ x.double(), 1e-5, "tn dW (no bias)")


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("m,k,k0,n", [(4001, 544, 512, 512), (40000, 160, 128, 128), (300, 36, 4, 64)])
def test_gemm_tn_split_output_is_the_unsplit_result(m, k, k0, n, math):
    """dgdm_gemm_tn_split: the same sums, delivered as two contiguous matrices (bit for bit)."""
    from dgdm_histopath_lab_amd import ops
    g = torch.Generator().manual_seed(m + k)
    x, gy = torch.randn(m, k, generator=g).to(DEV), torch.randn(m, n, generator=g).to(DEV)
    for with_bias in (True, False):
        dW, db = ops.gemm_tn_raw(gy, x, with_bias, math=math)
        (d0, d1), db2 = ops.gemm_tn_raw(gy, x, with_bias, math=math, split=k0)
        assert d0.is_contiguous() and d1.is_contiguous() and d0.shape == (n, k0) and d1.shape == (n, k - k0)
        assert torch.equal(d0, dW[:, :k0]) and torch.equal(d1, dW[:, k0:])
        assert (db is None and db2 is None) or torch.equal(db, db2)
    with pytest.raises(ValueError):
        ops.gemm_tn_raw(gy, x, False, math=math, split=k)


@pytest.mark.parametrize("m,k0,k1,n", [(4001, 512, 32, 512), (40000, 128, 32, 128), (300, 36, 32, 64), (1000, 4, 8, 12)])
def test_gemm_nt_split_weight_is_the_concatenated_product(m, k0, k1, n):
    """dgdm_gemm_nt_split_bf16x3 reads [W0 | W1] in place: bit for bit the GEMM on the concatenated weight."""
    from dgdm_histopath_lab_amd import ops
    g = torch.Generator().manual_seed(m + k0)
    a = torch.randn(m, k0 + k1, generator=g).to(DEV)
    w0, w1, b = torch.randn(n, k0, generator=g).to(DEV), torch.randn(n, k1, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    want = ops.gemm_nt_raw(a, torch.cat([w0, w1], dim=1), b, math="bf16x3")
    assert torch.equal(ops.gemm_nt_split_raw(a, w0, w1, b), want)
    wide = torch.randn(n, k0 + 8, generator=g).to(DEV)              # W0 as a column slice of a wider matrix
    want2 = ops.gemm_nt_raw(a, torch.cat([wide[:, :k0], w1], dim=1), None, math="bf16x3")
    assert torch.equal(ops.gemm_nt_split_raw(a, wide[:, :k0], w1), want2)
    # fp16 hi+lo kernel: the two weights share the scale of the larger maximum = the scale of the concatenated matrix
    assert torch.equal(ops.gemm_nt_split_raw(a, w0, w1, b, math="f16x2"), ops.gemm_nt_raw(a, torch.cat([w0, w1], dim=1), b, math="f16x2"))
    with pytest.raises(ValueError):
        ops.gemm_nt_split_raw(a, w0, w1[:, :4])


def test_bf16x3_matches_fp32_mfma_accuracy():
    """The split GEMMs' (bf16 x3 and fp16 hi+lo) error against fp64 is of the size of the fp32-MFMA kernel's own (accumulation
    rounding), also for operands spanning many binades and for gradient-sized (1e-6) values."""
    from dgdm_histopath_lab_amd import ops
    g = torch.Generator().manual_seed(5)
    m, k, n = 4096, 768, 512
    for scale_x, scale_g in [(1.0, 1.0), (1e3, 1e-6), (1e-4, 1e4)]:
        x = (torch.randn(m, k, generator=g) * torch.exp2(torch.randint(-12, 12, (m, k), generator=g).float()) * scale_x).to(DEV)
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(DEV)
        gy = (torch.randn(m, n, generator=g) * scale_g).to(DEV)
        ref_y = x.double() @ w.double().t()
        ref_dx = gy.double() @ w.double()
        ref_dw = gy.double().t() @ x.double()
        err = {}
        for math in MATHS:
            y = ops.gemm_nt_raw(x, w, None, math=math); dx = ops.gemm_nn_raw(gy, w, math=math); dw, _ = ops.gemm_tn_raw(gy, x, False, math=math)
            err[math] = [float((a.double() - r).abs().max() / r.abs().max()) for a, r in ((y, ref_y), (dx, ref_dx), (dw, ref_dw))]
        for math in ("bf16x3", "f16x2"):
            for e3, e32 in zip(err[math], err["fp32"]):
                assert e3 < 2e-6 and e3 < 4 * e32 + 2e-7, (math, scale_x, scale_g, err)


def test_gemm_strided_operands_and_autograd():
    from dgdm_histopath_lab_amd import ops
    g = torch.Generator().manual_seed(0)
    big = torch.randn(3000, 544 + 64, generator=g)           # A = column slice (row stride 608)
    wbig = torch.randn(256, 800, generator=g) / 20           # W = column slice of a wider parameter
    b = torch.randn(256, generator=g)
    A = big[:, :544]; W = wbig[:, 128:128 + 544]
    Ad = big.to(DEV)[:, :544].requires_grad_(True)
    wfull = wbig.to(DEV).requires_grad_(True); bd = b.to(DEV).requires_grad_(True)
    y = ops.linear(Ad, wfull[:, 128:128 + 544], bd)
    gy = torch.randn(3000, 256, generator=g)
    y.backward(gy.to(DEV))
    Ar, Wr, br = A.double().requires_grad_(True), W.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch.nn.functional.linear(Ar, Wr, br); yr.backward(gy.double())
    assert_close(y, yr, 1e-5, "y"); assert_close(Ad.grad, Ar.grad, 1e-5, "dA")
    assert_close(wfull.grad[:, 128:128 + 544], Wr.grad, 1e-5, "dW slice"); assert_close(bd.grad, br.grad, 1e-5, "db")
    assert wfull.grad[:, :128].abs().max() == 0
    # shapes outside the kernel's domain fall back to the library GEMM but stay correct
    small = torch.randn(7, 10, device=DEV); ws = torch.randn(3, 10, device=DEV)
    assert_close(ops.linear(small, ws), small.double().cpu() @ ws.double().cpu().t(), 1e-5, "fallback")


def test_f16x2_range_handling():
    """What fp16 cannot hold without the per-operand power-of-two scale: gradients of 1e-30, inputs of 1e30, an all-zero operand,
    an operand whose maximum is a single huge outlier, and an amax tag that is only an UPPER BOUND (a slice of a tagged matrix)."""
    from dgdm_histopath_lab_amd import ops
    g = torch.Generator().manual_seed(9)
    m, k, n = 1024, 256, 128
    w = (torch.randn(n, k, generator=g) / 16).to(DEV)
    for sx in (1e-30, 1.0, 1e30):
        x = (torch.randn(m, k, generator=g) * sx).to(DEV)
        y = ops.gemm_nt_raw(x, w, None, math="f16x2")
        ref = x.double() @ w.double().t()
        assert torch.isfinite(y).all() and float((y.double() - ref).abs().max() / ref.abs().max()) < 2e-6, sx
    z = ops.gemm_nt_raw(torch.zeros(m, k, device=DEV), w, None, math="f16x2")
    assert torch.equal(z, torch.zeros_like(z))
    x = torch.randn(m, k, generator=g).to(DEV)
    x[5, 7] = 3e4                                  # one outlier sets the scale: everything else sits 15 binades lower
    y = ops.gemm_nt_raw(x, w, None, math="f16x2")
    ref = x.double() @ w.double().t()
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    small = x[:, :128]
    big = torch.randn(m, 512, generator=g).to(DEV) * 100
    big[:, :128] = small
    ops.ensure_amax(big)                           # tag = max over the WIDE matrix (100x larger than the slice's own)
    sl = ops.tag_amax(big[:, :128], ops.amax_of(big))
    y2 = ops.gemm_nt_raw(sl, w[:, :128].contiguous(), None, math="f16x2")
    ref2 = small.double() @ w[:, :128].double().t()
    assert float((y2.double() - ref2).abs().max() / ref2.abs().max()) < 2e-6


def test_producer_kernels_keep_the_operand_maximum():
    """The kernels that produce GEMM operands (row norm fwd/bwd, activation+dropout fwd/bwd, the graph convolution's
    aggregate-and-concat) leave max|out| in the tensor's amax slot group: bit-identical to a reduction over the finished tensor."""
    from dgdm_histopath_lab_amd import GraphStructure, ops
    prev = ops.configure(gemm="f16x2")
    try:
        g = torch.Generator().manual_seed(3)

        def slot_max(t):
            s = ops.amax_of(t)
            assert s is not None
            a = ops._arena(t.device)
            off = (s - a.base) // 4
            return a.buf[off: off + a.GROUP_WORDS].max().view(torch.float32)

        x = torch.randn(5000, 512, generator=g).to(DEV).requires_grad_(True)
        w, b = torch.randn(512, generator=g).to(DEV).requires_grad_(True), torch.randn(512, generator=g).to(DEV).requires_grad_(True)
        y = ops.row_norm(x, w, b, act=ops.ACT_GELU, drop_p=0.1, training=True)
        assert torch.equal(slot_max(y), y.detach().abs().max())
        a = ops.act_dropout(y, ops.ACT_GELU, 0.1, True)
        assert torch.equal(slot_max(a), a.detach().abs().max())
        grads = {}
        y.register_hook(lambda t: grads.setdefault("dy", t))
        x.register_hook(lambda t: grads.setdefault("dx", t))
        (a * torch.randn(5000, 512, generator=g).to(DEV)).sum().backward()
        assert torch.equal(slot_max(grads["dy"]), grads["dy"].abs().max())      # act_dropout_bwd
        assert torch.equal(slot_max(grads["dx"]), grads["dx"].abs().max())      # rownorm_bwd
        n = 3000
        gs = GraphStructure(torch.randint(0, n, (2, 12000), generator=g).to(DEV), n)
        ea = torch.randn(n, 32, generator=g).to(DEV) * 3
        conv_w, conv_we = torch.randn(256, 128, generator=g).to(DEV) / 12, torch.randn(256, 32, generator=g).to(DEV) / 6
        xin = torch.randn(n, 128, generator=g).to(DEV)
        seen = {}
        orig = ops.gemm_nt_split_raw
        ops.gemm_nt_split_raw = lambda buf, *a_, **k_: (seen.setdefault("buf", buf), orig(buf, *a_, **k_))[1]
        try:
            ops.graph_conv_linear(xin, ea, gs, conv_w, conv_we, None)
        finally:
            ops.gemm_nt_split_raw = orig
        assert torch.equal(slot_max(seen["buf"]), seen["buf"].abs().max())      # spmm_concat: aggregate AND the copied edge block
    finally:
        ops.configure(**prev)


def test_many_problem_dw_launch_against_the_single_launches():
    """Inside ops.deferred_weight_grads() the fp16 hi+lo dW GEMMs are held back and run as ONE launch per 24 problems at the end of
    the pass (k_gemmh_tn32_many, its own row chunking): the gradients equal those of the single launches up to the fp32 summation
    order and are repeatable bit for bit.  30 layers (two launches), ragged widths, row counts from one stage to many chunks,
    with and without bias."""
    from dgdm_histopath_lab_amd import ops
    prev = ops.configure(gemm="f16x2")
    try:
        g = torch.Generator().manual_seed(21)
        widths = [128, 12, 20, 132, 64, 256, 36, 128, 100, 8, 128, 160, 32, 128, 516, 128, 4, 44, 128, 128, 96, 192, 128, 64, 320, 128, 24,
                  128, 72, 128, 128]
        for m in (257, 5000):
            x = torch.randn(m, widths[0], generator=g).to(DEV)
            ws = [(torch.randn(o, i, generator=g) / i ** 0.5).to(DEV) for i, o in zip(widths[:-1], widths[1:])]
            bs = [torch.randn(w.size(0), generator=g).to(DEV) if k % 3 else None for k, w in enumerate(ws)]

            def run(deferred):
                W = [w.clone().requires_grad_(True) for w in ws]
                Bv = [None if b is None else b.clone().requires_grad_(True) for b in bs]
                h = x
                for w, b in zip(W, Bv):
                    h = ops.linear(h, w, b)
                loss = h.square().mean()
                if deferred:
                    with ops.deferred_weight_grads():
                        loss.backward()
                    assert not ops._PENDING_TN
                else:
                    loss.backward()
                return [p.grad for p in W + [b for b in Bv if b is not None]]
            a, b, b2 = run(False), run(True), run(True)
            for u, v in zip(a, b):
                assert float((u - v).abs().max()) <= 4e-6 * float(u.abs().max()) + 1e-12
            assert all(torch.equal(u, v) for u, v in zip(b, b2))
            assert all(torch.isfinite(v).all() for v in b) and sum(float(v.abs().max()) > 0 for v in b) > len(b) // 2
    finally:
        ops.configure(**prev)


def test_posenc_maximum_with_a_partial_last_wave():
    """dgdm_add_posenc with N * C / 4 not a multiple of 64 (an odd node count at C = 128): the last wave is half in range.  The maximum
    in the slot must still be exactly max|out| -- the kernel once ran its amax commit (which holds a workgroup barrier) on both sides
    of the range check, the barrier count went wrong and thread 0 committed whatever the LDS held (after a weight-image GEMM: fp16
    halfs with the sign bit set, which an unsigned maximum prefers to every legitimate value; the next GEMM then scaled its operand
    into fp16 infinity).  Run behind a GEMM that leaves such halfs in the LDS, many times."""
    from dgdm_histopath_lab_amd import ops
    prev = ops.configure(gemm="f16x2")
    try:
        g = torch.Generator().manual_seed(8)
        xg, wg = torch.randn(4096, 128, generator=g).to(DEV), -torch.rand(128, 128, generator=g).to(DEV)
        for n in (8495, 10603, 1001, 63):
            plan = ops.AttnPlan([0, n], DEV)
            x, pos = torch.randn(n, 128, generator=g).to(DEV), torch.rand(n, 2, generator=g).to(DEV)
            for _ in range(12):
                ops.gemm_nt_raw(xg, wg, None, math="f16x2")
                y = ops.add_posenc_raw(x, pos, plan, 128)
                s = ops.amax_of(y)
                a = ops._arena(y.device)
                off = (s - a.base) // 4
                ways = a.buf[off: off + a.GROUP_WORDS]
                assert int(ways.min()) >= 0, "a way holds the bit pattern of a negative float"
                assert torch.equal(ways.max().view(torch.float32), y.abs().max())
    finally:
        ops.configure(**prev)


@pytest.mark.parametrize("math", ["bf16x3", "f16x2"])
def test_deferred_weight_gradient_reduction_is_bitwise_the_immediate_one(math):
    """Inside ops.deferred_weight_grads() the dW GEMMs of a backward pass leave their chunk partials and ONE launch reduces them
    when the pass ends (csrc/dw_reduce.hip), for few and for many chunks, for the split form (two parameters behind one
    contraction) and for an output that is a column block.  bf16x3: same sums in the same order -- bit-identical to the immediate
    reduction.  f16x2: the GEMMs themselves run in the pass's many-problem launches with a coarser row chunking -- equal up to the
    fp32 summation order, and repeatable bit for bit."""
    from dgdm_histopath_lab_amd import GraphStructure, ops
    prev = ops.configure(gemm=math)
    try:
        g = torch.Generator().manual_seed(8)
        x = torch.randn(40000, 128, generator=g).to(DEV)
        ws = [(torch.randn(o, i, generator=g) / i ** 0.5).to(DEV) for o, i in ((512, 128), (128, 512), (256, 128), (128, 256))]
        bs = [torch.randn(w.size(0), generator=g).to(DEV) for w in ws]
        n = x.size(0)
        gs = GraphStructure(torch.randint(0, n, (2, 100000), generator=g).to(DEV), n)
        ea = torch.randn(n, 32, generator=g).to(DEV)
        cw, cwe = (torch.randn(128, 128, generator=g) / 11).to(DEV), (torch.randn(128, 32, generator=g) / 6).to(DEV)

        def run(deferred):
            params = [t.clone().requires_grad_(True) for t in ws + bs + [cw, cwe]]
            W, Bv, (pw, pwe) = params[:4], params[4:8], params[8:]
            h = ops.linear(ops.linear(x, W[0], Bv[0]), W[1], Bv[1])
            h = ops.graph_conv_linear(h, ea, gs, pw, pwe, None)
            h = ops.linear(ops.linear(h, W[2], Bv[2]), W[3], Bv[3])
            loss = h.square().mean()
            if deferred:
                with ops.deferred_weight_grads():
                    loss.backward()
                assert not ops._PENDING_TN
            else:
                loss.backward()
            return [p.grad for p in params]
        a, b = run(False), run(True)
        if math == "bf16x3":
            assert all(torch.equal(u, v) for u, v in zip(a, b))
        else:
            for u, v in zip(a, b):
                assert float((u - v).abs().max()) <= 4e-6 * float(u.abs().max()) + 1e-12
            assert all(torch.equal(u, v) for u, v in zip(b, run(True)))
        assert all(torch.isfinite(v).all() and v.abs().max() > 0 for v in b)
    finally:
        ops.configure(**prev)


def test_deferred_weight_gradients_with_shared_weights_and_existing_grads():
    """a deferred dW tensor is filled at the END of the pass, which is only right if the engine merely stores
    it.  A weight used by two nodes (tied weights / a module called twice) makes the engine ADD the two gradients when the second
    arrives, and a backward that starts with ``.grad`` set adds in place right away -- both must give the immediate path's values."""
    from dgdm_histopath_lab_amd import ops
    prev = ops.configure(gemm="f16x2")
    try:
        g = torch.Generator().manual_seed(31)
        x = torch.randn(3000, 128, generator=g).to(DEV)
        w0 = (torch.randn(128, 128, generator=g) / 11).to(DEV)
        b0 = torch.randn(128, generator=g).to(DEV)
        w1 = (torch.randn(64, 128, generator=g) / 11).to(DEV)
        gam, bet = torch.rand(128, generator=g).to(DEV) + 0.5, torch.randn(128, generator=g).to(DEV)

        def run(deferred, accumulate):
            W, B, W1, G, Be = (t.clone().requires_grad_(True) for t in (w0, b0, w1, gam, bet))
            reps = 2 if accumulate else 1
            for _ in range(reps):       # second round: .grad is already set -> AccumulateGrad adds in place
                h = ops.linear(x, W, B)                               # the SAME weight twice in one graph ...
                h = ops.row_norm(h, G, Be, act=ops.ACT_GELU)
                h = ops.linear(h, W, B)
                h = ops.row_norm(h, G, Be, act=ops.ACT_GELU)          # ... and the same norm parameters twice
                loss = ops.linear(h, W1, None).square().mean()
                if deferred:
                    with ops.deferred_weight_grads():
                        loss.backward()
                    assert not ops._PENDING_TN
                else:
                    loss.backward()
            return [p.grad.clone() for p in (W, B, W1, G, Be)]
        for accumulate in (False, True):
            a, b = run(False, accumulate), run(True, accumulate)
            for name, u, v in zip(("W", "b", "W1", "gamma", "beta"), a, b):
                assert torch.isfinite(v).all() and float(u.abs().max()) > 0
                err = float((u - v).abs().max()) / float(u.abs().max())
                assert err <= 4e-6, (accumulate, name, err)
    finally:
        ops.configure(**prev)


def test_amax_tag_dies_when_the_ring_recycles_its_slot():
    """the operand-maximum ring (4096 slot groups, ~300 taken per step) wraps after ~13 steps; a tag on a
    long-lived tensor (a device-resident input reused over epochs) then points at a slot that was zeroed and handed to another
    tensor.  Tags carry the generation of their slot's chunk: after the wrap the GEMM must take a fresh maximum instead of scaling by
    whatever the recycled slot holds (here: the bits of 1e-6, which would scale the operand by 2^30 into fp16 infinity)."""
    from dgdm_histopath_lab_amd import ops
    prev = ops.configure(gemm="f16x2")
    try:
        g = torch.Generator().manual_seed(17)
        x = (torch.randn(2048, 256, generator=g) * 3.0).to(DEV)
        w = (torch.randn(128, 256, generator=g) / 16).to(DEV)
        y0 = ops.gemm_nt_raw(x, w, None, math="f16x2")
        sx, sw = ops.amax_of(x), ops.amax_of(w)
        assert sx is not None and sw is not None
        y1 = ops.gemm_nt_raw(x, w, None, math="f16x2")
        assert ops.amax_of(x) == sx and torch.equal(y0, y1)          # live tag: reused, no new reduction
        a = ops._arena(x.device)
        other = torch.full((512, 128), 1e-6, device=DEV)
        takes0 = a.total_takes
        for _ in range(a.SLOTS + 2 * a.CHUNK):                        # more takes than the ring has slots, by tensors with a tiny maximum
            other._dgdm_amax = None
            ops.ensure_amax(other)
        assert a.total_takes - takes0 > a.SLOTS
        assert ops.amax_of(x) is None and ops.amax_of(w) is None       # the old tags are dead ...
        y2 = ops.gemm_nt_raw(x, w, None, math="f16x2")                 # ... and the GEMM re-derives the maxima
        assert torch.isfinite(y2).all() and torch.equal(y0, y2)
        assert ops.amax_of(x) not in (None,)
        # a handle saved by an autograd context across the wrap is dropped as well
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        y = ops.linear(xr, wr, None)
        for _ in range(a.SLOTS + 2 * a.CHUNK):
            other._dgdm_amax = None
            ops.ensure_amax(other)
        y.square().mean().backward()
        ref = (2 * (x.double() @ w.double().t()) / y.numel())
        assert float((xr.grad.double() - ref @ w.double()).abs().max()) <= 1e-5 * float((ref @ w.double()).abs().max())
        assert float((wr.grad.double() - ref.t() @ x.double()).abs().max()) <= 1e-5 * float((ref.t() @ x.double()).abs().max())
    finally:
        ops.configure(**prev)

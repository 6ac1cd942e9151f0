"""GPU parity of DiffusionLayer.sample (K8, reference core/diffusion.py:214-275) and of the small-M dense kernels under it.

The reference's own `sample` was executed by oracle/capture_golden.py (core/diffusion.py runs as-is): `g5_diffusion["sample"]`
(32-wide, 6 steps, all draws stored) and `g5b_sample_base` (Base widths, 300 rows, 10 and 50 steps, draws from a stored seed).
The HIP `sample()` runs eagerly and as a replayed HIP graph against those vectors."""
import pytest
import torch
import torch.nn.functional as F

from conftest import T, assert_close, load_golden, weights
from oracle import dgdm_oracle as O
from test_oracle_golden import sample_draws

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3


def _layer(C, Hd, Tn, state):
    from dgdm_histopath_lab_amd.core.diffusion import DiffusionLayer
    dl = DiffusionLayer(C, Hd, num_timesteps=Tn)
    dl.load_state_dict(state, strict=True)
    return dl.to(DEV).eval()


@pytest.mark.parametrize("graphed", [False, True])
def test_sample_matches_reference_golden_small(graphed):
    g = load_golden("g5_diffusion")
    dl = _layer(32, 64, int(g["T"]), weights(g))
    n = g["x_init"].shape[0]
    for _ in range(2 if graphed else 1):      # second call = pure replay
        s = dl.sample((n, 32), DEV, num_inference_steps=int(g["steps"]), x_init=T(g["x_init"]), step_noise=list(T(g["step_noise"])),
                      graphed=graphed)
        assert_close(s, g["sample"], TOL, "sample")


@pytest.mark.parametrize("graphed", [False, True])
@pytest.mark.parametrize("steps", [10, 50])
def test_sample_matches_reference_golden_base_widths(steps, graphed):
    g = load_golden("g5b_sample_base")
    P = O.init_params(O.OracleConfig(), seed=int(g["init_seed"]), perturb=float(g["init_perturb"]))
    n, C = int(g["n"]), int(g["C"])
    dl = _layer(C, 2 * C, int(g["T"]), {k[len("diffusion_layer."):]: v for k, v in P.items() if k.startswith("diffusion_layer.")})
    x_init, noises = sample_draws(n, C, steps, int(g["draw_seed_base"]) + steps)
    for _ in range(2 if graphed else 1):
        s = dl.sample((n, C), DEV, num_inference_steps=steps, x_init=x_init, step_noise=noises, graphed=graphed)
        assert_close(s, g[f"sample{steps}"], TOL, f"sample{steps}")


def test_sample_draws_its_own_noise_and_replays_with_fresh_draws():
    """Without injected draws: eager and graphed calls run, are finite, and two replays of the recorded loop differ (the normal
    draws inside the graph come from torch's graph-aware generator)."""
    cfg = O.OracleConfig()
    P = O.init_params(cfg, seed=2, perturb=0.05)
    dl = _layer(128, 256, 10, {k[len("diffusion_layer."):]: v for k, v in P.items() if k.startswith("diffusion_layer.")})
    torch.manual_seed(0)
    a = dl.sample((2000, 128), DEV, num_inference_steps=10)
    b = dl.sample((2000, 128), DEV, num_inference_steps=10, graphed=True)
    c = dl.sample((2000, 128), DEV, num_inference_steps=10, graphed=True)
    for t in (a, b, c):
        assert t.shape == (2000, 128) and torch.isfinite(t).all()
    assert not torch.equal(b, c)
    torch.manual_seed(3); d = dl.sample((2000, 128), DEV, num_inference_steps=10, graphed=True)
    torch.manual_seed(3); e = dl.sample((2000, 128), DEV, num_inference_steps=10, graphed=True)
    assert torch.equal(d, e)                   # same generator state, same recorded loop: bit-identical


def test_sample_at_the_headline_size_matches_oracle():
    """10 000 rows x 128 (one BASELINE configs[1] graph), T = 10, 10 steps: HIP path (tile GEMMs) against the float64 oracle."""
    cfg = O.OracleConfig()
    P = O.init_params(cfg, seed=9, perturb=0.05)
    dl = _layer(128, 256, 10, {k[len("diffusion_layer."):]: v for k, v in P.items() if k.startswith("diffusion_layer.")})
    x_init, noises = sample_draws(10000, 128, 10, 77)
    P64 = {k: v.double() for k, v in P.items()}
    sched = {k: v.double() for k, v in O.diffusion_schedule(10, "cosine").items()}
    with torch.no_grad():
        ref = O.ddpm_sample(P64, sched, 10, x_init.double(), [z.double() for z in noises], 10)
    s = dl.sample((10000, 128), DEV, num_inference_steps=10, x_init=x_init, step_noise=noises, graphed=True)
    assert_close(s, ref, TOL, "sample 10k")


@pytest.mark.parametrize("M,N,K,act", [(1, 128, 128, 0), (4, 256, 128, 3), (4, 512, 256, 0), (10, 512, 256, 0), (7, 3, 64, 0),
                                        (255, 96, 130, 1), (33, 130, 2048, 0), (2, 5, 7, 2)])
def test_small_m_linear_forward_backward(M, N, K, act):
    """dgdm_linear_small_fwd/bwd against float64 (incl. a weight that is a column block of a wider matrix, as the time half of
    the denoiser's first Linear is)."""
    from dgdm_histopath_lab_amd import ops
    g = torch.Generator().manual_seed(M * 1000 + N + K)
    x = torch.randn(M, K, generator=g)
    wfull = torch.randn(N, K + 24, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    gy = torch.randn(M, N, generator=g)
    acts = {0: lambda z: z, 1: F.gelu, 2: F.relu, 3: F.silu}
    xr, wr, br = x.double().requires_grad_(True), wfull.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = acts[act](F.linear(xr, wr[:, 24:], br))
    yr.backward(gy.double())
    xd, wd, bd = x.to(DEV).requires_grad_(True), wfull.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = ops.linear_small(xd, wd[:, 24:], bd, act)
    y.backward(gy.to(DEV))
    assert_close(y, yr, 1e-5, "y")
    assert_close(xd.grad, xr.grad, 1e-5, "dx")
    assert_close(wd.grad, wr.grad, 1e-5, "dw")
    assert_close(bd.grad, br.grad, 1e-5, "db")
    # the generic entry point routes few-row inputs here (no library GEMM on the path)
    y2 = ops.linear(x.to(DEV), wfull.to(DEV)[:, 24:].contiguous(), b.to(DEV))
    assert_close(y2, F.linear(x.double(), wfull.double()[:, 24:], b.double()), 1e-5, "ops.linear")


def test_graphed_sample_follows_weight_updates():
    """the recorded sampling loop must rebuild the fp16 weight images it reads.  Sample graphed, change the
    weights (in place, as an optimizer step or load_state_dict does), sample graphed again: equal to the eager path on the new
    weights -- and different from the first result."""
    cfg = O.OracleConfig()
    P = O.init_params(cfg, seed=4, perturb=0.05)
    dl = _layer(128, 256, 10, {k[len("diffusion_layer."):]: v for k, v in P.items() if k.startswith("diffusion_layer.")})
    x_init, noises = sample_draws(2000, 128, 10, 123)
    kw = dict(num_inference_steps=10, x_init=x_init, step_noise=noises)
    first = dl.sample((2000, 128), DEV, graphed=True, **kw)
    again = dl.sample((2000, 128), DEV, graphed=True, **kw)
    assert torch.equal(first, again)
    P2 = O.init_params(cfg, seed=5, perturb=0.05)
    dl.load_state_dict({k[len("diffusion_layer."):]: v for k, v in P2.items() if k.startswith("diffusion_layer.")}, strict=True)
    new_g = dl.sample((2000, 128), DEV, graphed=True, **kw)
    new_e = dl.sample((2000, 128), DEV, graphed=False, **kw)
    assert_close(new_g, new_e.double(), 1e-5, "graphed vs eager after a weight update")
    assert float((new_g - first).abs().max()) > 1e-2
    P64 = {k: v.double() for k, v in P2.items()}
    sched = {k: v.double() for k, v in O.diffusion_schedule(10, "cosine").items()}
    with torch.no_grad():
        ref = O.ddpm_sample(P64, sched, 10, x_init.double(), [z.double() for z in noises], 10)
    assert_close(new_g, ref, TOL, "graphed sample on the new weights vs oracle")


@pytest.mark.parametrize("C,n", [(128, 1003), (128, 31), (256, 330), (128, 20011)])
def test_fused_sample_step_matches_float64_and_the_seven_launch_path(C, n, monkeypatch):
    """Round 6: one step of DiffusionLayer.sample as ONE launch (csrc/sample_step.hip; reference core/diffusion.py:147-172,245-273) at
    node_dim 128 (Base, up to more workgroups than the chip holds at once) and 256 (Large), ragged last row tile, large |x| (the early steps divide by sqrt(ac) ~ 0.02: per-workgroup
    power-of-two scale), first / middle / last step -- against a float64 composition of the reference's formulas, and against the
    seven-launch path of rounds 2-5 (`ops.SAMPLE_STEP_FUSED = False`) over a whole 10-step loop."""
    from dgdm_histopath_lab_amd import ops
    from dgdm_histopath_lab_amd.core.diffusion import DiffusionLayer
    torch.manual_seed(C + n)
    dl = DiffusionLayer(C, 2 * C, num_timesteps=10).to(DEV).eval()
    with torch.no_grad():
        for p in dl.parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn_like(p))
    g = torch.Generator().manual_seed(7)
    sch = dl.scheduler
    dn = dl.denoise_net
    P = {k: v.detach().double().cpu() for k, v in dl.state_dict().items()}
    for t, scale, last in ((9, 40.0, False), (4, 1.0, False), (0, 1.0, True)):
        x = (torch.randn(n, C, generator=g) * scale)
        z = torch.randn(n, C, generator=g)
        bias0 = dl.time_bias(torch.tensor([t], device=DEV))[0]
        s1mac, sac = float(torch.sqrt(1 - sch.alphas_cumprod)[t]), float(torch.sqrt(sch.alphas_cumprod)[t])
        salpha, svar = float(torch.sqrt(sch.alphas)[t]), float(torch.sqrt(sch.posterior_variance)[t])
        assert ops.denoise_ddpm_step_supported(C)
        got = ops.denoise_ddpm_step(x.to(DEV), None if last else z.to(DEV), dn[0].weight[:, :C], dn[4].weight, dn[8].weight, bias0, dn[1], dn[4].bias,
                                    dn[5], dn[8].bias, s1mac, sac, salpha, svar, last)
        # float64: predict_noise (diffusion.py:147-172) + the update (:255-273)
        te = O.timestep_embedding(torch.tensor([t])).double()
        te = F.linear(F.silu(F.linear(te, P["time_embed.0.weight"], P["time_embed.0.bias"])), P["time_embed.2.weight"], P["time_embed.2.bias"])
        h = F.linear(torch.cat([x.double(), te.expand(n, -1)], 1), P["denoise_net.0.weight"], P["denoise_net.0.bias"])
        h = F.silu(F.group_norm(h, 8, P["denoise_net.1.weight"], P["denoise_net.1.bias"], 1e-5))
        h = F.linear(h, P["denoise_net.4.weight"], P["denoise_net.4.bias"])
        h = F.silu(F.group_norm(h, 8, P["denoise_net.5.weight"], P["denoise_net.5.bias"], 1e-5))
        eps = F.linear(h, P["denoise_net.8.weight"], P["denoise_net.8.bias"])
        x0 = (x.double() - s1mac * eps) / sac
        ref = x0 if last else salpha * x0 + svar * z.double()
        assert_close(got, ref, 1e-4, f"fused step t={t} scale={scale}")
    # the whole loop, fused against the seven-launch path, same draws
    x_init = torch.randn(n, C, generator=g).to(DEV)
    noises = [torch.randn(n, C, generator=g).to(DEV) for _ in range(9)]
    a = dl.sample((n, C), DEV, num_inference_steps=10, x_init=x_init, step_noise=noises)
    monkeypatch.setattr(ops, "SAMPLE_STEP_FUSED", False)
    b = dl.sample((n, C), DEV, num_inference_steps=10, x_init=x_init, step_noise=noises)
    assert_close(a, b, 1e-4, "10-step loop: one launch per step against seven")
    # training mode with dropout keeps the seven-launch path (the fused kernel is eval only)
    monkeypatch.setattr(ops, "SAMPLE_STEP_FUSED", True)
    calls = []
    real = ops.denoise_ddpm_step
    monkeypatch.setattr(ops, "denoise_ddpm_step", lambda *a_, **k_: (calls.append(1), real(*a_, **k_))[1])
    dl.train()
    dl.sample((n, C), DEV, num_inference_steps=3)
    assert not calls
    dl.eval()
    dl.sample((n, C), DEV, num_inference_steps=3)
    assert len(calls) == 3


def test_fused_sample_step_is_affine_in_the_draw_at_full_size():
    """Size-independent property at BASELINE's rows (4 x 10 000, more workgroups than the chip holds at once; nothing the oracle could
    finish in seconds): the step is x' = f(x) + sqrt(var) z, so two launches on the same x differ by exactly sqrt(var) (z1 - z2) up to the
    rounding of the last add, row by row, and the last step ignores z.  reference: core/diffusion.py:255-273."""
    from dgdm_histopath_lab_amd import ops
    from dgdm_histopath_lab_amd.core.diffusion import DiffusionLayer
    torch.manual_seed(5)
    C, n = 128, 40000
    dl = DiffusionLayer(C, 2 * C, num_timesteps=10).to(DEV).eval()
    dn = dl.denoise_net
    x = torch.randn(n, C, device=DEV) * 3
    z1, z2 = torch.randn(n, C, device=DEV), torch.randn(n, C, device=DEV)
    bias0 = dl.time_bias(torch.tensor([5], device=DEV))[0]
    args = (dn[0].weight[:, :C], dn[4].weight, dn[8].weight, bias0, dn[1], dn[4].bias, dn[5], dn[8].bias, 0.6, 0.8, 0.95, 0.3)
    a = ops.denoise_ddpm_step(x, z1, *args, False)
    b = ops.denoise_ddpm_step(x, z2, *args, False)
    c = ops.denoise_ddpm_step(x, torch.zeros_like(z1), *args, False)
    assert torch.isfinite(a).all() and torch.equal(a, ops.denoise_ddpm_step(x, z1, *args, False))      # repeatable bit for bit
    scale = float(c.abs().max()) + 0.3 * float(z1.abs().max())
    assert float(((a - b) - 0.3 * (z1 - z2)).abs().max()) <= 4e-7 * scale
    assert float(((a - c) - 0.3 * z1).abs().max()) <= 4e-7 * scale
    last = ops.denoise_ddpm_step(x, None, *args, True)
    assert_close(last * 0.95, c, 1e-6, "last step = x0 prediction; the others = sqrt(alpha) x0 + sqrt(var) z")
    # rows are independent: a permutation of the rows permutes the output (up to the workgroups' power-of-two scales of x: another
    # grouping of rows, another scale, another place where the lo halves end)
    perm = torch.randperm(n, device=DEV)
    assert_close(ops.denoise_ddpm_step(x[perm].contiguous(), z1[perm].contiguous(), *args, False), a[perm], 1e-6, "rows permuted")

"""CPU check of the branch-free GELU of csrc/rowmath.hpp (half_erfc_abs / gelu_f / gelu_df): the same fp32 operation sequence
emulated in numpy (every FMA rounded once), against float64 and against torch's fp32 GELU, over [-12, 12].  Also refits the
polynomial (scipy) to show where the coefficients come from:  python tools/check_fast_gelu.py [--fit]"""
import sys

import numpy as np
import torch
from scipy import special

C = [1.6279072761535645, 0.918442964553833, 0.14830751717090607, -0.02772335335612297, -8.649988012621179e-05, 0.00227622059173882,
     -0.0008489217725582421, 0.00015313828771468252, -1.1622888450801838e-05]
f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(np.float32)


def half_erfc_abs(z):
    with np.errstate(invalid="ignore", over="ignore"):
        ta = (np.abs(z) * f32(0.70710678118654752)).astype(np.float32)
        t = np.minimum(ta, f32(4.0)).astype(np.float32)
        r = np.full_like(t, f32(C[8]))
        for k in range(7, -1, -1):
            r = fma(r, t, f32(C[k]))
        h = np.exp2(-fma(t, r, f32(1.0)).astype(np.float64)).astype(np.float32)
    return np.where(ta < f32(4.0), h, f32(0.0)).astype(np.float32)       # exactly 0 past the clamp (round 6)


def gelu(z):
    h = half_erfc_abs(z)
    with np.errstate(invalid="ignore", over="ignore"):
        return np.where(z >= 0, fma(-z, h, z), (z * h).astype(np.float32)).astype(np.float32)


def dgelu(z):
    h = half_erfc_abs(z)
    Phi = np.where(z >= 0, (f32(1) - h).astype(np.float32), h)
    with np.errstate(invalid="ignore", over="ignore"):
        ph = np.exp2(((z * z).astype(np.float32) * f32(-0.72134752044448170)).astype(np.float64)).astype(np.float32)
        return fma((z * f32(0.3989422804014327)).astype(np.float32), ph, Phi)


def fit(deg=8, T=4.0):
    n = 6000
    t = np.maximum((np.cos(np.pi * (np.arange(n) + 0.5) / n) + 1) / 2 * T, 1e-12)
    y = -np.log2(special.erfc(t)) / t
    w = t * special.erfc(t)                      # d erf = ln2 * t * erfc(t) * dq: weight the fit by the absolute error of erf
    A = np.vander(t, deg + 1, increasing=True)
    return np.linalg.lstsq(A * w[:, None], y * w, rcond=None)[0]


if __name__ == "__main__":
    if "--fit" in sys.argv:
        print("refit:", [float(f32(v)) for v in fit()])
    z = np.linspace(-12, 12, 4000001).astype(np.float32)
    z64 = z.astype(np.float64)
    g_ref = 0.5 * z64 * (1 + special.erf(z64 / np.sqrt(2)))
    d_ref = 0.5 * (1 + special.erf(z64 / np.sqrt(2))) + z64 * np.exp(-0.5 * z64 * z64) / np.sqrt(2 * np.pi)
    zt = torch.from_numpy(z).requires_grad_(True)
    gt = torch.nn.functional.gelu(zt)
    gt.sum().backward()
    big = np.abs(g_ref) > 1e-3
    print(f"gelu : max abs err {np.abs(gelu(z) - g_ref).max():.2e} (torch fp32 {np.abs(gt.detach().numpy() - g_ref).max():.2e}); "
          f"max rel err where |gelu| > 1e-3: {(np.abs(gelu(z) - g_ref) / np.abs(g_ref))[big].max():.2e} "
          f"(torch fp32 {(np.abs(gt.detach().numpy() - g_ref) / np.abs(g_ref))[big].max():.2e})")
    print(f"gelu': max abs err {np.abs(dgelu(z) - d_ref).max():.2e} (torch fp32 {np.abs(zt.grad.numpy() - d_ref).max():.2e})")
    # the ends: large |z| and the infinities, beside torch's fp32 GELU
    ends = np.array([np.inf, -np.inf, 1e4, -1e4, 130.0, -130.0, 12.0, -12.0, 5.7, -5.7, 5.6, -5.6], dtype=np.float32)
    et = torch.from_numpy(ends).requires_grad_(True)
    eg = torch.nn.functional.gelu(et)
    eg.sum().backward()
    print("   z          gelu (kernel arithmetic / torch fp32)        gelu' (kernel arithmetic / torch fp32)")
    for zi, a, b, c, d in zip(ends, gelu(ends), eg.detach().numpy(), dgelu(ends), et.grad.numpy()):
        print(f"  {zi:>9.4g}   {a:>14.7g} / {b:<14.7g}   {c:>14.7g} / {d:<14.7g}")
    zz = np.concatenate([np.linspace(-1e4, -12, 200001), np.linspace(12, 1e4, 200001)]).astype(np.float32)
    z64 = zz.astype(np.float64)
    print(f"|z| in [12, 1e4]: gelu max abs err {np.abs(gelu(zz) - 0.5 * z64 * (1 + special.erf(z64 / np.sqrt(2)))).max():.2e}, "
          f"gelu' max abs err {np.abs(dgelu(zz) - (0.5 * (1 + special.erf(z64 / np.sqrt(2))))).max():.2e}")

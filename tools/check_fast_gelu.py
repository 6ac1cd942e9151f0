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
    t = np.minimum(np.abs(z) * f32(0.70710678118654752), f32(4.0)).astype(np.float32)
    r = np.full_like(t, f32(C[8]))
    for k in range(7, -1, -1):
        r = fma(r, t, f32(C[k]))
    return np.exp2(-fma(t, r, f32(1.0)).astype(np.float64)).astype(np.float32)


def gelu(z):
    return (np.maximum(z, f32(0)) - np.abs((z * half_erfc_abs(z)).astype(np.float32))).astype(np.float32)


def dgelu(z):
    h = half_erfc_abs(z)
    Phi = np.where(z >= 0, (f32(1) - h).astype(np.float32), h)
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

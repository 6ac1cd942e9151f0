import torch, os
print("torch", torch.__version__, "prec", torch.get_float32_matmul_precision(), "allow_tf32", torch.backends.cuda.matmul.allow_tf32)
try: print("preferred blas", torch.backends.cuda.preferred_blas_library())
except Exception as e: print(e)
print({k: v for k, v in os.environ.items() if "TF32" in k or "BLAS" in k or "TUNABLE" in k})
g = torch.Generator().manual_seed(0)
def err(a, b, f):
    r = f(a.double(), b.double())
    o = f(a.cuda(), b.cuda()).cpu().double()
    return ((o - r).norm() / r.norm()).item()
for (m, k, n) in [(4000, 768, 512), (4000, 512, 512), (512, 4000, 32), (512, 4000, 544), (4000, 544, 512), (4000, 128, 384), (40000, 128, 128)]:
    a = torch.randn(m, k, generator=g); b = torch.randn(k, n, generator=g)
    print((m, k, n), "mm rel err %.2e" % err(a, b, lambda x, y: x @ y),
          " linear %.2e" % err(a, b.t().contiguous(), lambda x, w: torch.nn.functional.linear(x, w)))
torch.backends.cuda.preferred_blas_library("cublas")
print("after preferring rocblas:")
for (m, k, n) in [(4000, 768, 512), (512, 4000, 32), (4000, 128, 384)]:
    a = torch.randn(m, k, generator=g); b = torch.randn(k, n, generator=g)
    print((m, k, n), "mm rel err %.2e" % err(a, b, lambda x, y: x @ y))

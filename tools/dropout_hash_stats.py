"""numpy model of the attention-dropout hash of csrc/attn_common.hpp (cell mixing + per-element 24-bit multiply, signed
top-16-bit compare): drop rate and mask correlations along keys, queries, diagonals, inside a 2x2 cell, across heads,
seeds and graphs.  CPU only:  python tools/dropout_hash_stats.py"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)
U = np.uint64


def mul24(a, b):
    return ((a & U(0xFFFFFF)) * (U(b) & U(0xFFFFFF))) & M32


def fmix32(x):
    x = x & M32; x ^= x >> U(16); x = (x * U(0x85EBCA6B)) & M32; x ^= x >> U(13); x = (x * U(0xC2B2AE35)) & M32; x ^= x >> U(16)
    return x


def keep_mask(seed, n0, head, nq, nk, p):
    thresh = max(1, int(p * 65536))
    hs = fmix32(U(seed) ^ ((U(n0) * U(0xC2B2AE35)) & M32) ^ ((U(head + 1) * U(0x27D4EB2F)) & M32))
    g = int(fmix32(hs ^ U(0x9E3779B9)))
    m = np.array([[(g ^ 0x3C6D2B), ((g >> 4) ^ 0x6A09E7)], [((g >> 8) ^ 0x52DCE5), ((g >> 3) ^ 0x2545F5)]], dtype=np.uint64)
    m = (m & U(0xFFFFFF)) | U(1)
    q = np.arange(nq, dtype=np.uint64)[:, None]; k = np.arange(nk, dtype=np.uint64)[None, :]
    x = hs ^ mul24(q >> U(1), 0x79B1A5) ^ mul24(k >> U(1), 0x5BCA6B)
    x ^= x >> U(16)
    w = ((x & U(0xFFFFFF)) * m[(q & U(1)).astype(int), (k & U(1)).astype(int)]) & M32
    s = (w >> U(16)).astype(np.int64)
    s = np.where(s >= 32768, s - 65536, s)
    return s >= thresh - 32768


def corr(a, b):
    return float(np.corrcoef(a.astype(np.float64).ravel(), b.astype(np.float64).ravel())[0, 1])


if __name__ == "__main__":
    for p in (0.1, 0.25, 0.5):
        K = keep_mask(1234, 0, 0, 4096, 4096, p)
        print(f"p={p}: drop rate {1 - K.mean():.6f} (target {max(1, int(p * 65536)) / 65536:.6f})")
        print("  adjacent keys %.1e  adjacent queries %.1e  diagonal %.1e  anti-diagonal %.1e  k+2 %.1e  q+2 %.1e" % (
            corr(K[:, :-1], K[:, 1:]), corr(K[:-1], K[1:]), corr(K[:-1, :-1], K[1:, 1:]), corr(K[:-1, 1:], K[1:, :-1]),
            corr(K[:, :-2], K[:, 2:]), corr(K[:-2], K[2:])))
        cell = [K[0::2, 0::2], K[0::2, 1::2], K[1::2, 0::2], K[1::2, 1::2]]
        print("  cell mates", ["%.1e" % corr(cell[i], cell[j]) for i in range(4) for j in range(i + 1, 4)])
        print("  heads %.1e  seeds %.1e  graphs %.1e" % (corr(K, keep_mask(1234, 0, 1, 4096, 4096, p)), corr(K, keep_mask(1235, 0, 0, 4096, 4096, p)),
                                                          corr(K, keep_mask(1234, 10000, 0, 4096, 4096, p))))
        print("  row-rate std %.5f, column-rate std %.5f (binomial %.5f)" % ((1 - K.mean(1)).std(), (1 - K.mean(0)).std(), (p * (1 - p) / 4096) ** 0.5))

"""One-off GPU diagnostics (not a test): which CPU model reproduces v_mfma_f64_16x16x4_f64 bit for bit,
and a first look at kernel timings.  Run on the GPU box: python tests/gpu_probe.py"""
import importlib
import os
import sys
import time
from fractions import Fraction

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import oracle as O  # noqa: E402

mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0)
n, txt = mpf.device_report()
print(txt)
rng = np.random.default_rng(7)


def wide(shape):
    return np.asfortranarray(rng.standard_normal(shape) * np.exp2(rng.integers(-20, 20, shape)))


for K in (4, 8, 256):
    m = nn = 48
    A, B, C = wide((m, K)), wide((K, nn)), wide((m, nn))
    dC = ctx.from_numpy_f(C)
    ctx.dgemm_minus(dC, ctx.from_numpy_f(A), ctx.from_numpy_f(B))
    ctx.synchronize()
    G = ctx.to_numpy_f(dC)
    asc = C.copy(order="F"); O.dgemm_minus(asc, A, B)
    desc = C.copy(order="F"); O.dgemm_minus(desc, np.asfortranarray(A[:, ::-1]), np.asfortranarray(B[::-1, :]))
    # unfused mul/sub chain
    unf = C.copy()
    for k in range(K):
        unf = unf - np.outer(A[:, k], B[k, :])
    # exact with one rounding (few elements)
    ex_match = 0
    for (i, j) in [(0, 0), (5, 7), (17, 33), (47, 47), (13, 2), (31, 16)]:
        acc = Fraction(C[i, j])
        for k in range(K):
            acc -= Fraction(A[i, k]) * Fraction(B[k, j])
        ex_match += int(float(acc) == G[i, j])
    print(f"K={K}: match asc-fma {np.mean(G == asc):.4f}  desc-fma {np.mean(G == desc):.4f}  "
          f"unfused {np.mean(G == unf):.4f}  exact-single-rounding {ex_match}/6  maxrel {np.max(np.abs(G-asc)/np.abs(asc)):.2e}")

# timing of the big pieces at a moderate size
N, nb = 8192, 256
Ah = O.matgen_skip(N)
dA = ctx.from_numpy_f(Ah)
for rep in range(2):
    W = dA.clone()
    torch.cuda.synchronize(); t = time.time()
    ipiv, info = ctx.factor(W, nb, sync_timing=(rep == 1))
    torch.cuda.synchronize(); dt = time.time() - t
    s = ctx.stats()
    print(f"factor N={N} nb={nb}: wall {dt*1e3:.1f} ms dev {s.ms_total:.1f} ms -> {2/3*N**3/s.ms_total/1e6:.1f} GFLOP/s; "
          f"hpanel {s.ms_hpanel:.1f} laswp {s.ms_laswp:.1f} dpanel {s.ms_dpanel:.1f} trsm {s.ms_trsm:.1f} gemm {s.ms_gemm:.1f} info {info}")

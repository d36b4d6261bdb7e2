"""fp64 update kernel alone on a sub-block of a matrix with leading dimension ld: does a power-of-two ld cost anything?
Emulates the row-major schedule's launch (C' = sub-block, A' = U rows with the same ld, B' = the packed L image, ld 256).
Usage: python tools/gemm_ld_probe.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0)
N, k = 32768, 256
for pad in (0, 16, 32, 64, 144, 528):
    ld = N + pad
    buf = torch.empty(ld * N, dtype=torch.float64, device="cuda")
    buf.normal_()
    M = buf.view(N, ld).t()                       # column-major N x N, leading dimension ld
    Lt = torch.randn(N, k, dtype=torch.float64, device="cuda").t()   # column-major k x N: ld = k
    line = f"ld = N + {pad:3d}:"
    for n in (30464, 18304, 10112, 6016):
        o = N - n
        C = M[o:, o:]
        A = M[o:, o - k:o]                         # m x k with leading dimension ld
        B = Lt[:, o:]                              # k x n with leading dimension k
        for _ in range(2): ctx.dgemm_minus(C, A, B)
        ctx.synchronize()
        reps = 6
        t0 = time.perf_counter()
        for _ in range(reps): ctx.dgemm_minus(C, A, B)
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        line += f"  n={n}: {ms:6.3f} ms {2.0 * n * n * k / ms / 1e9:5.1f} TF"
    print(line, flush=True)
    del buf, M

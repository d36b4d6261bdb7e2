"""Kernel-level timing probe (run on the GPU box): on-box peaks + each hot-path kernel alone."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
dev = ctx.device

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

if "peaks" in sys.argv or len(sys.argv) == 1:
    print("f64 MFMA issue-rate TFLOP/s:", round(ctx.microbench(0), 2))
    print("f16 MFMA issue-rate TFLOP/s:", round(ctx.microbench(1), 1))
    print("HBM stream copy TB/s (r+w):", round(ctx.microbench(2), 2))

if "gemm" in sys.argv or len(sys.argv) == 1:
    for (m, n, k) in [(16384, 16384, 256), (32512, 32512, 256), (8192, 8192, 256), (4096, 4096, 256), (16384, 16384, 128)]:
        ld = 32768
        big = torch.rand((ld, ld), dtype=torch.float64, device=dev).t()
        A = big[256:256 + m, 0:k]; B = big[0:k, 256:256 + n]; Cm = big[256:256 + m, 256:256 + n]
        ms = timeit(lambda: ctx.dgemm_minus(Cm, A, B), 3)
        print(f"dgemm_minus m={m} n={n} k={k}: {ms:.3f} ms  {2.0*m*n*k/ms/1e9:.1f} TFLOP/s")
        del big

if "panel" in sys.argv or len(sys.argv) == 1:
    ld = 32768
    big = (torch.randint(0, 100, (1024, ld), device=dev, dtype=torch.int32).to(torch.float64) / 10.0).t()  # ld x 1024 col-major
    for rows in (256, 2048, 8192, 32768):
        P = big[:rows, :256]
        ms = timeit(lambda: ctx.hgetf2_pivots(P), 3)
        print(f"hgetf2_pivots rows={rows} cols=256: {ms:.3f} ms  {ms*1e3/256:.2f} us/col  {rows*256*8/ms/1e6:.1f} GB/s")
    for rows in (256, 8192, 32768):
        W = big[:rows, 256:512].clone()  # note: clone keeps strides? make explicit
        Wc = ctx.colmajor(rows, 256); Wc.copy_(big[:rows, 256:512]); Wc[:256, :256] += 50 * torch.eye(256, device=dev, dtype=torch.float64)
        ms = timeit(lambda: ctx.dgetf2_npv(Wc), 1)
        print(f"dgetf2_npv rows={rows} cols=256: {ms:.3f} ms")
    Lm = ctx.colmajor(256, 256); Lm.copy_(torch.rand((256, 256), dtype=torch.float64, device=dev) * 0.01)
    for n in (256, 8192, 32512):
        Bm = ctx.colmajor(256, n); Bm.copy_(torch.rand((256, n), dtype=torch.float64, device=dev))
        ms = timeit(lambda: ctx.dtrsm_llnu(Lm, Bm), 3)
        print(f"dtrsm m=256 n={n}: {ms:.3f} ms")
    n = 32768
    Afull = torch.rand((4096, n), dtype=torch.float64, device=dev).t()  # n x 4096 col-major
    piv = (torch.randint(256, n, (256,), device=dev, dtype=torch.int32) + 1)
    ms = timeit(lambda: ctx.laswp(Afull, 0, 256, piv), 3)
    print(f"laswp n={n} ncols=4096 cols=256: {ms:.3f} ms -> per 32768 cols {ms*8:.3f} ms")

if "hgemm" in sys.argv or len(sys.argv) == 1:
    for (m, n, k) in [(16384, 16384, 256), (32512, 32512, 256), (8192, 8192, 256)]:
        ld = 32768
        big = torch.rand((ld, ld), dtype=torch.float64, device=dev).t()
        A = big[256:256 + m, 0:k]; B = big[0:k, 256:256 + n]; Cm = big[256:256 + m, 256:256 + n]
        ms = timeit(lambda: ctx.hgemm_minus(Cm, A, B), 3)
        print(f"hgemm_minus (incl. operand conversion) m={m} n={n} k={k}: {ms:.3f} ms  {2.0*m*n*k/ms/1e9:.1f} TFLOP/s  {16.0*m*n/ms/1e9:.2f} TB/s algorithmic")
        del big

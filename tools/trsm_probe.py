"""TRSM launch time against the width of the right-hand side (column-major B, m = 256): python tools/trsm_probe.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0)
torch.manual_seed(1)
m = 256
L = (torch.randn(m, m, dtype=torch.float64, device="cuda") * 0.05).t().contiguous().t()
for n in (256, 1024, 4096, 8192, 8448, 16384, 32768):
    B = torch.randn(n, m, dtype=torch.float64, device="cuda").t()      # column-major m x n
    for _ in range(3): ctx.dtrsm_llnu(L, B)
    ctx.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps): ctx.dtrsm_llnu(L, B)
    ctx.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    print(f"n={n:6d}: {us:7.1f} us per launch (back to back), {m * m * n / us / 1e6:6.2f} TFLOP/s", flush=True)

"""fp64 GEMM (dgemm_minus) rate against K, alone on the chip."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
dev = ctx.device
for m in (8192, 16384, 28672):
    Cm = torch.randn(m, m, dtype=torch.float64, device=dev).t()
    for k in (256, 512, 1024):
        A = torch.randn(k, m, dtype=torch.float64, device=dev).t()
        B = torch.randn(m, k, dtype=torch.float64, device=dev).t()
        ctx.dgemm_minus(Cm, A, B); ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): ctx.dgemm_minus(Cm, A, B)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"dgemm m=n={m} k={k}: {ms:.3f} ms  {2*m*m*k/ms/1e9:.1f} TF", flush=True)
        del A, B
    del Cm

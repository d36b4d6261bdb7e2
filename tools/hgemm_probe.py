"""Standalone timing of the fp16 / fp16x3 trailing-update kernel (operand conversion included)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
dev = ctx.device
import sys
k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for m in (16384, 28672):
    A = torch.randn(k, m, dtype=torch.float64, device=dev).t()      # m x k column-major
    B = torch.randn(m, k, dtype=torch.float64, device=dev).t()      # k x m
    Cm = torch.randn(m, m, dtype=torch.float64, device=dev).t()
    for split in (False, True):
        ctx.hgemm_minus(Cm, A, B, split=split); ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ctx.hgemm_minus(Cm, A, B, split=split)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"m=n={m} k={k} split={split}: {ms:.3f} ms  {2*m*m*k/ms/1e9:.1f} TF  C traffic {16*m*m/ms/1e6:.0f} GB/s", flush=True)
    del A, B, Cm

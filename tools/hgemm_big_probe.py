"""Standalone timing of the fp16 / fp16x3 big-K update on the fp32 working copy (operand conversion excluded from the GEMM
time: the images are converted once, the kernel is timed through the step operator with conversion, and alone by
difference against a K-independent conversion run)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
big = int(os.environ.get("BIG", "1"))
ctx = mpf.MPFContext(0, options={"hgemm_big": big})
dev = ctx.device
ks = [int(a) for a in sys.argv[1:]] or [512, 1024]
for m in (16384, 28672):
    Cm = torch.randn(m, m, dtype=torch.float32, device=dev).t()
    for k in ks:
        A = torch.randn(k, m, dtype=torch.float64, device=dev).t()      # m x k column-major
        B = torch.randn(m, k, dtype=torch.float64, device=dev).t()      # k x m
        for split in (False, True):
            ctx.hgemm_minus_f32(Cm, A, B, split=split); ctx.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ctx.hgemm_minus_f32(Cm, A, B, split=split)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"big={big} m=n={m} k={k} split={split}: {ms:.3f} ms incl. image conversion  {2*m*m*k/ms/1e9:.1f} TF  C traffic {8*m*m/ms/1e6:.0f} GB/s", flush=True)
        del A, B
    del Cm

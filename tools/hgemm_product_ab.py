"""The big fp16 update through the PRODUCT library's step operator (operand conversion + kernel), for A/B runs of library builds:
MPF_LIB=<path to a libmpf_amd.so> python tools/hgemm_product_ab.py [K ...].  Prints ms per call (median of 7) at m = n = 28672."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0)
m = 28672
Cm = torch.randn(m, m, dtype=torch.float32, device=ctx.device).t()
for k in ([int(a) for a in sys.argv[1:]] or [1024, 2048]):
    A = torch.randn(k, m, dtype=torch.float64, device=ctx.device).t(); B = torch.randn(m, k, dtype=torch.float64, device=ctx.device).t()
    ts = []
    for rep in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.hgemm_minus_f32(Cm, A, B); e1.record(); torch.cuda.synchronize()
        if rep >= 2: ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{os.environ.get('MPF_LIB', 'lib/libmpf_amd.so')}: K={k}: {ts[len(ts) // 2]:.3f} ms per call incl. operand conversion ({2.0 * m * m * k / ts[len(ts) // 2] / 1e9:.0f} TFLOP/s)", flush=True)

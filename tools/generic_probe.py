"""The generic (never-waiting) schedule, measured: ms per factorization and per panel column, for the shapes the tuned
schedules do not cover (r > 256) and for pivot_path = 1 at a tuned shape.  Usage: python tools/generic_probe.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0)
for n, nb, pp in ((8192, 256, 0), (8192, 256, 1), (8192, 512, 0), (16384, 512, 0)):
    A = ctx.matgen(n)
    W = A.clone()
    ctx.factor(W, nb, pivot_path=pp)
    W.copy_(A)
    torch.cuda.synchronize()
    t = time.perf_counter()
    ctx.factor(W, nb, pivot_path=pp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) * 1e3
    st = ctx.stats()
    print(f"N={n} nb={nb} pivot_path={pp} (generic pivots used: {st.pivot_path}): {dt:.1f} ms = {2*n**3/3/dt/1e9:.2f} TFLOP/s; pivot kernels {st.ms_hpanel:.1f} ms "
          f"= {st.ms_hpanel * 1e3 / n:.1f} us per column; laswp {st.ms_laswp:.1f} dpanel {st.ms_dpanel:.1f} trsm {st.ms_trsm:.1f} gemm {st.ms_gemm:.1f}", flush=True)

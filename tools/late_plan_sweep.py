"""mpf_factor_host's late-upload plan by first-part share and number of late segments: call time (best of 2 after a warm-up).
usage: late_plan_sweep.py N [first_pct,... [parts,...]]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(sys.argv[1])
firsts = [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "25,35,50").split(",")]
parts = [int(a) for a in (sys.argv[3] if len(sys.argv) > 3 else "2,3").split(",")]
ctx = mpf.MPFContext(0)
Ah0 = ctx.to_numpy_f(ctx.matgen(n))
torch.cuda.empty_cache()
ctx.set_option("host_late_min_n", 0)
for p in parts:
    for f in firsts:
        ctx.set_option("host_late_parts", p); ctx.set_option("host_first_pct", f)
        ts = []
        for rep in range(3):
            Ah = Ah0.copy(order="F"); ip = np.arange(1, n + 1, dtype=np.int32)
            t0 = time.perf_counter(); ctx.factor_host(Ah, 256, ip); ts.append((time.perf_counter() - t0) * 1e3)
        s = ctx.stats()
        print(f"N={n} parts={p} first={f}%: call {min(ts[1:]):.1f} ms (up {s.ms_h2d:.1f} factor {s.ms_total:.1f} home {s.ms_d2h:.1f}; segments {s.host_late_segments})", flush=True)

"""First call of a process through mpf_factor_host (what benchmark.cpp's single-matrix runs see): wall clock of mpf_create + the first
call and of the second call.  usage: first_call_probe.py [N]   env: MPF_HOST_SINK, MPF_HOST_LATE_PARTS as usual"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
g = mpf.MPFContext(0)
Ah0 = g.to_numpy_f(g.matgen(n))
g.close(); del g
torch.cuda.empty_cache(); torch.cuda.synchronize()
t0 = time.perf_counter()
ctx = mpf.MPFContext(0)
t1 = time.perf_counter()
for rep in range(2):
    Ah = Ah0.copy(order="F"); ip = np.arange(1, n + 1, dtype=np.int32)
    t2 = time.perf_counter()
    ctx.factor_host(Ah, 256, ip)
    t3 = time.perf_counter()
    s = ctx.stats()
    print(f"call {rep + 1}: {(t3 - t2) * 1e3:7.1f} ms (create {(t1 - t0) * 1e3:.1f} ms before the first) | up {s.ms_h2d:.1f} factor {s.ms_total:.1f} home {s.ms_d2h:.1f} | rows streamed {s.host_rows_streamed} late {s.host_late_segments}", flush=True)

"""mpf_factor_host at N (default 32768), nb = 256, by transfer mode: the call's wall clock and the factorization's phase timers.
usage: host_path_probe.py [N]   env REPS=2; modes: plain (one copy up, one copy home), sink (block rows home while it factors), late (+ late column segments)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nb = 256
reps = int(os.environ.get("REPS", "2"))
ctx = mpf.MPFContext(0)
A0 = ctx.matgen(n) if hasattr(ctx, "matgen") else None
if A0 is None:
    A0 = torch.randn(n, n, dtype=torch.float64, device=ctx.device).t()
Ah0 = ctx.to_numpy_f(A0)
del A0
torch.cuda.empty_cache()
for rep in range(reps):
    for name, sink, parts in (("plain", 0, 0), ("sink", 1, 0), ("late", 1, int(os.environ.get("PARTS", "3")))):
        ctx.set_option("host_sink", sink); ctx.set_option("host_late_parts", parts)
        Ah = Ah0.copy(order="F")
        ip = np.arange(1, n + 1, dtype=np.int32)
        t0 = time.perf_counter()
        ctx.factor_host(Ah, nb, ip)
        t = (time.perf_counter() - t0) * 1e3
        s = ctx.stats()
        print(f"{name:5s}: call {t:7.1f} ms | up {s.ms_h2d:6.1f} factor {s.ms_total:6.1f} home {s.ms_d2h:6.1f} | gemm {s.ms_gemm:6.1f} ({s.gemm_launches} launches) chain {s.ms_hpanel:6.1f} "
              f"dpanel {s.ms_dpanel:5.1f} trsm {s.ms_trsm:5.1f} laswp {s.ms_laswp:5.1f} cvt {s.ms_cvt:5.1f} | rows streamed {s.host_rows_streamed} late segments {s.host_late_segments}", flush=True)
        del Ah

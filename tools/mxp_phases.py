import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
dev = ctx.device
n, nb = 32768, 256
g = torch.Generator(device=dev); g.manual_seed(1)
A = (torch.randint(0, 100, (n, n), generator=g, device=dev, dtype=torch.int32).to(torch.float64) / 10.0).t()
idx = torch.arange(n, device=dev); A[idx, idx] += A.sum(dim=1)
for mode, name in ((mpf.TRAIL_FP16, "fp16"), (mpf.TRAIL_FP64, "fp64")):
    for sync in (True, False):
        W = A.clone()
        ctx.factor(W, nb, trailing=mode, sync_timing=sync)
        s = ctx.stats()
        print(f"{name} sync_timing={sync}: total {s.ms_total:.1f} ms | hpanel {s.ms_hpanel:.1f} laswp {s.ms_laswp:.1f} dpanel {s.ms_dpanel:.1f} trsm {s.ms_trsm:.1f} gemm {s.ms_gemm:.1f} (launches {s.gemm_launches})")
        del W

"""fp64 panel alone on the chip: single-launch vs multi-launch variant (MPF_DPANEL_SINGLE=0/1 in the environment)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
dev = ctx.device
ld = 32768
big = (torch.randint(0, 100, (256, ld), device=dev, dtype=torch.int32).to(torch.float64) / 10.0).t()
big[torch.arange(256), torch.arange(256)] += 500.0
for rows in (32768, 16384, 8192, 2048, 256):
    P = big[:rows, :256]
    W = P.clone()
    def cp(): W.copy_(P)
    def both():
        W.copy_(P); ctx.dgetf2_npv(W)
    ts = []
    for f in (cp, both):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    print(f"rows={rows:6d}: dgetf2_npv {1e3*(ts[1]-ts[0]):8.1f} us (copy {1e3*ts[0]:.1f} us)", flush=True)

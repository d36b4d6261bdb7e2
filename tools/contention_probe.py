"""How much do the chain kernels slow down under a concurrent HBM-streaming or MFMA-bound kernel?"""
import importlib, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)          # runs on torch's current stream
dev = ctx.device
ld = 32768
big = (torch.randint(0, 100, (256, ld), device=dev, dtype=torch.int32).to(torch.float64) / 10.0).t()
side = torch.cuda.Stream()
src = torch.empty(1 << 28, dtype=torch.float64, device=dev)   # 2 GiB
dst = torch.empty_like(src)
m = 16384
Ag = torch.randn(256, m, dtype=torch.float64, device=dev).t()
Bg = torch.randn(m, 256, dtype=torch.float64, device=dev).t()
Cg = torch.randn(m, m, dtype=torch.float64, device=dev).t()

def time_chain(label, load):
    for rows in (32768, 8192):
        P = big[:rows, :256]
        W = P.clone()
        def fn():
            ctx.hgetf2_pivots(P)
        def fn2():
            W.copy_(P); ctx.dgetf2_npv(W)
        for name, f in (("hgetf2", fn), ("dgetf2_npv(+copy)", fn2)):
            f(); torch.cuda.synchronize()
            if load is not None:
                with torch.cuda.stream(side):
                    for _ in range(40): load()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / 5
            torch.cuda.synchronize()
            print(f"{label:28s} rows={rows:6d} {name:18s} {ms*1e3:8.1f} us", flush=True)

time_chain("alone", None)
time_chain("under HBM copy (torch)", lambda: dst.copy_(src))
ctx_side = mpf.MPFContext(0, probe=True, stream=side)
time_chain("under hgemm 16384^2", lambda: ctx_side.hgemm_minus(Cg, Ag, Bg))
time_chain("under dgemm 16384^2", lambda: ctx_side.dgemm_minus(Cg, Ag, Bg))

"""Pivot kernel, full slab (a CU per workgroup) against the column-window form (two per CU): alone and beside an update."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
dev = ctx.device
side = torch.cuda.Stream()
ctx_side = mpf.MPFContext(0, probe=True, stream=side)
ld = 32768
big = (torch.randint(0, 100, (256, ld), device=dev, dtype=torch.int32).to(torch.float64) / 10.0).t()
m = 16384
Ag = torch.randn(256, m, dtype=torch.float64, device=dev).t()
Bg = torch.randn(m, 256, dtype=torch.float64, device=dev).t()
Cg = torch.randn(m, m, dtype=torch.float64, device=dev).t()
Ah = torch.randn(1024, m, dtype=torch.float64, device=dev).t()
Bh = torch.randn(m, 1024, dtype=torch.float64, device=dev).t()
loads = {"alone": None, "under dgemm": lambda: ctx_side.dgemm_minus(Cg, Ag, Bg), "under hgemm K=1024": lambda: ctx_side.hgemm_minus(Cg, Ah, Bh)}
for label, load in loads.items():
    for rows in (32768, 16384, 4096, 256):
        for cols in (256, 128):
            P = big[:rows, :cols]
            out = []
            for window in (0, 1):
                ctx.set_option("hp_window", window)
                ctx.hgetf2_pivots(P); torch.cuda.synchronize()
                if load is not None:
                    with torch.cuda.stream(side):
                        for _ in range(30): load()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): ctx.hgetf2_pivots(P)
                e1.record(); e1.synchronize()
                out.append(e0.elapsed_time(e1) / 5 * 1e3 / cols)
                torch.cuda.synchronize()
            print(f"{label:20s} rows={rows:6d} cols={cols:4d}: slab {out[0]:6.2f} us/col   window {out[1]:6.2f} us/col", flush=True)

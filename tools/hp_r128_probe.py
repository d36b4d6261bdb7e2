"""Pivot kernel alone: 256-row slabs (product) against the probe build's 128-row slabs, and the single-XCD form, by panel shape.
us per column from 5 launches on a random panel (fp64 source).  usage: hp_r128_probe.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
dev = ctx.device
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ld = 32768
big = torch.randn(256, ld, device=dev, dtype=torch.float64).t()
for rows in (256, 1024, 4096, 8192, 16384, 32768):
    for cols in (128, 256):
        P = big[:rows, :cols]
        res = []
        for name, opts in (("R=256 across XCDs", {"hp_r256_upto": 1 << 30, "hp_local_xcd": 0}), ("R=256 one XCD", {"hp_r256_upto": 1 << 30, "hp_local_xcd": 2}),
                           ("R=128", {"hp_r256_upto": 0, "hp_local_xcd": 0})):
            for k, v in opts.items(): ctx.set_option(k, v)
            try:
                ms = timeit(lambda: ctx.hgetf2_pivots(P), 5)
                res.append(f"{name}: {ms * 1e3 / cols:.2f}")
            except Exception as e:
                res.append(f"{name}: ERR {str(e)[:40]}")
        print(f"rows={rows:6d} cols={cols}: us per column  " + "   ".join(res), flush=True)

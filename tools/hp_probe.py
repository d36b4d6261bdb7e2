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
big = (torch.randint(0, 100, (256, ld), device=dev, dtype=torch.int32).to(torch.float64) / 10.0).t()
for rows in (256, 4096, 32768):
    for cols in (16, 64, 128, 256):
        P = big[:rows, :cols]
        ms = timeit(lambda: ctx.hgetf2_pivots(P), 5)
        print(f"rows={rows} cols={cols}: {ms*1e3:.1f} us total  {ms*1e3/cols:.2f} us/col")

"""Where a column step of the fp16 pivot kernel spends its cycles (diagnostic build, MPF_HP_STAMP=1): segment sums of
wave 0 of workgroup 0 over a 256-column panel, for several panel heights."""
import importlib, os, sys
os.environ["MPF_HP_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
names = ["cand update+publish", "own deferred update", "sweep+row fetch", "barrier(3)", "critical part", "barrier(1)+readback"]
big = (torch.randint(0, 100, (256, 32768), device=ctx.device, dtype=torch.int32).to(torch.float64) / 10.0).t()
for rows in (256, 2048, 8192, 32768):
    P = big[:rows, :256]
    for rep in range(2):
        ctx.hgetf2_pivots(P)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ctx.hgetf2_pivots(P); e1.record(); torch.cuda.synchronize()
    segs = [ctx.microbench(70 + i) for i in range(6)]
    tot = sum(segs)
    print(f"rows={rows}: {e0.elapsed_time(e1)*1e3/256:.2f} us/col (stamped build); cycles/col by segment:",
          ", ".join(f"{n}={s/256:.0f}" for n, s in zip(names, segs)), f"total={tot/256:.0f}")
    pw = int(ctx.microbench(76)); have = ctx.microbench(77)
    if rows > 256:
        G = (rows + 255) // 256
        print(f"          polls per column {(pw >> 32)/256:.2f}; keys present at the first poll {have/256/G*100:.0f} % of {G}; first poll's best is the winner in {(pw & 0xFFFFFFFF)/256*100:.0f} % of the columns")

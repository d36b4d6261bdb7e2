"""What the HIP-event pairs around every timed region cost: option event_timers 2 / 1 / 0.  python tools/timers_probe.py [N]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for mode in (0, 1, 2):
    for lvl in (2, 1, 0):
        ctx = mpf.MPFContext(0, options={"event_timers": lvl})
        A = ctx.matgen(n)
        if mode == 1:
            idx = torch.arange(n, device=ctx.device)
            A[idx, idx] += A.sum(dim=1)
        W = A.clone()
        best = 1e9
        for rep in range(4):
            W.copy_(A)
            ctx.factor(W, 256, trailing=mode)
            best = min(best, ctx.stats().ms_total)
        print(f"mode {mode} event_timers={lvl}: best {best:.1f} ms", flush=True)
        del A, W
        ctx.close()

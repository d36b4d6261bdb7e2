"""The Python-driven multi-GPU schedule (dist.factor_lookahead) with world = 1, against the C++ single-GPU schedule:
shows what the per-panel host driving and the broadcast-shaped data path cost before any communication."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
dist = importlib.import_module("mixed-precision_lu_factorization_amd.dist")
n, nb = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 256
dev = torch.device("cuda", 0)
ctx = mpf.MPFContext(0, probe=True)
side = torch.cuda.Stream(device=dev, priority=-1)
ctx_side = mpf.MPFContext(0, probe=True, stream=side)
layout = dist.BlockCyclic(n, nb, 0, 1)
A0 = dist.colmajor_empty(n, n, dev)
for b in layout.my_blocks:
    w = layout.width(b)
    A0[:, layout.local_col(b):layout.local_col(b) + w] = dist.synth_block(n, w, b, dev)
W = dist.colmajor_empty(n, n, dev)
for name, fn in (("dist.factor_lookahead (world 1)", lambda: dist.factor_lookahead(ctx, ctx_side, W, layout)),
                 ("dist.factor (world 1, no look-ahead)", lambda: dist.factor(ctx, W, layout)),
                 ("C++ mpf_factor_dev", lambda: ctx.factor(W, nb))):
    for rep in range(2):
        W.copy_(A0); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {dt*1e3:.1f} ms ({2*n**3/3/dt/1e12:.1f} TF)", flush=True)

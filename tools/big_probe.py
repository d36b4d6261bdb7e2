"""N = 65536 on one GPU (34 GB matrix): factor + refinement residual, all three trailing modes."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = mpf.MPFContext(0, probe=True)
dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(7)
A = torch.empty((n, n), dtype=torch.float64, device=dev).t()
for c0 in range(0, n, 8192):   # generator distribution, built in slabs to bound temporaries
    A[:, c0:c0 + 8192] = (torch.randint(0, 100, (8192, n), generator=g, device=dev, dtype=torch.int32).to(torch.float64) / 10.0).t()
xs = torch.ones(n, dtype=torch.float64, device=dev)
b = A @ xs
W = torch.empty((n, n), dtype=torch.float64, device=dev).t()
for mode, name in ((mpf.TRAIL_FP64, "fp64"), (mpf.TRAIL_FP16X3, "fp16x3")):
    W.copy_(A)
    ipiv, info = ctx.factor(W, 256, trailing=mode)
    st = ctx.stats()
    x, ir = ctx.solve_ir(A, W, ipiv, b, max_iter=10, tol=1e-12)
    print(f"N={n} {name}: factor {st.ms_total:.0f} ms ({2*n**3/3/st.ms_total/1e9:.1f} TF) info={info} timeouts={st.hpanel_timeouts} "
          f"IR iters={ir.iterations} converged={ir.converged} rel_residual={ir.rel_residual:.2e} max|x-1|={float((x-xs).abs().max()):.2e}", flush=True)

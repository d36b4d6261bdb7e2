import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
dev = ctx.device
for n, nb in ((1024, 32), (4096, 128), (8192, 128), (16384, 256), (32768, 256)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    A = (torch.randint(0, 100, (n, n), generator=g, device=dev, dtype=torch.int32).to(torch.float64) / 10.0).t()
    xs = torch.ones(n, dtype=torch.float64, device=dev)
    b = A @ xs
    for mode, name in ((mpf.TRAIL_FP16, "fp16"), (mpf.TRAIL_FP16X3, "fp16x3"), (mpf.TRAIL_FP64, "fp64")):
        W = A.clone()
        ipiv, info = ctx.factor(W, nb, trailing=mode)
        W.copy_(A)
        ipiv, info = ctx.factor(W, nb, trailing=mode)
        ms = ctx.stats().ms_total
        x, st = ctx.solve_ir(A, W, ipiv, b, max_iter=30, tol=1e-12)
        h = [f"{v:.1e}" for v in list(st.history)[:st.iterations + 1]]
        print(f"generator matrix N={n} nb={nb} {name}: factor {ms:.1f} ms ({2*n**3/3/ms/1e9:.1f} TF) ir {st.ms_total:.1f} ms converged={st.converged} iters={st.iterations} hist={h[:12]}")
        del W
    del A

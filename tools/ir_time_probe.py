"""Time of one refinement sweep (residual + two triangular solves) at N = 32768 on fp64 factors."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
for n in ([int(a) for a in sys.argv[1:]] or [8192, 32768]):
    A = ctx.matgen(n)
    W = A.clone()
    ipiv, info = ctx.factor(W, 256)
    xs = torch.ones(n, dtype=torch.float64, device=ctx.device)
    b = A @ xs
    for rep in range(2):
        x, st = ctx.solve_ir(A, W, ipiv, b, max_iter=8, tol=1e-30)   # unreachable tolerance: runs until it stalls / max_iter
    sweeps = st.iterations + 1
    x2, st2 = ctx.solve_ir(A, W, ipiv, b, max_iter=8, tol=1e-30)
    print(f"N={n}: {st.ms_total:.2f} ms for {sweeps} solves + {sweeps} residuals = {st.ms_total / sweeps:.2f} ms per sweep; "
          f"residual history {[f'{v:.1e}' for v in list(st.history)[:sweeps]]}; bitwise reproducible x: {bool(torch.equal(x, x2))}")

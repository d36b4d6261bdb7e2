"""Repeated factorizations / solves in all modes on two contexts: memory must stay flat, results identical run to run."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
dev = torch.device("cuda", 0)
ctxs = [mpf.MPFContext(0, probe=True), mpf.MPFContext(0, probe=True, stream=torch.cuda.Stream())]
n = 4096
g = torch.Generator(device=dev); g.manual_seed(5)
A = (torch.randint(0, 100, (n, n), generator=g, device=dev, dtype=torch.int32).to(torch.float64) / 10.0).t()
xs = torch.ones(n, dtype=torch.float64, device=dev); b = A @ xs
ref = {}
torch.cuda.synchronize()
free0 = None
for it in range(60):
    ctx = ctxs[it % 2]
    mode = it % 3
    nb = (256, 128, 96)[(it // 3) % 3]
    W = A.clone()
    with torch.cuda.stream(ctx.stream) if ctx.stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
        ipiv, info = ctx.factor(W, nb, trailing=mode)
        x, st = ctx.solve_ir(A, W, ipiv, b, max_iter=25, tol=1e-12)
    ctx.synchronize()
    key = (mode, nb)
    sig = (float(W.double().sum()), int(ipiv.sum()), st.iterations)
    if key in ref: assert ref[key] == sig, (key, ref[key], sig)
    ref[key] = sig
    assert info == 0 and (st.converged == 1 or mode == 1), (key, st.iterations, st.rel_residual)
    if it == 20: free0 = torch.cuda.mem_get_info()[0]
free1 = torch.cuda.mem_get_info()[0]
print("deterministic across", len(ref), "configurations; free HBM after warm-up", free0, "at end", free1, "delta", free0 - free1)
for c in ctxs: c.close()
print("ok")

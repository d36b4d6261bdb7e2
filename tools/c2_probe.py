"""C2's factorization alone (N = 8192, nb = 128, fp16 mode, diagonally dominant input), five times: for rocprofv3 --kernel-trace --stats."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0)
n, nb = int(os.environ.get("N", "8192")), int(os.environ.get("NB", "128"))
A = ctx.matgen(n); idx = torch.arange(n, device=ctx.device); A[idx, idx] += A.sum(dim=1)
W = A.clone()
for rep in range(6):
    W.copy_(A)
    ipiv, info = ctx.factor(W, nb, trailing=mpf.TRAIL_FP16)
    ctx.synchronize()
print("ms", ctx.stats().ms_total)

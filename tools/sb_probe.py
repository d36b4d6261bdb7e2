"""fp16-mode factorization at N = 32768 by super-panel width (option superpanel_fp16): time and the big-K update's rate in the schedule."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
mode_name = sys.argv[2] if len(sys.argv) > 2 else "fp16"
ctx = mpf.MPFContext(0)
A = ctx.matgen(n); idx = torch.arange(n, device=ctx.device)
if mode_name == 'fp16': A[idx, idx] += A.sum(dim=1)
MODE = mpf.TRAIL_FP16 if mode_name == 'fp16' else mpf.TRAIL_FP16X3
W = torch.empty((n, n), dtype=torch.float64, device=ctx.device).t()
b = A @ torch.ones(n, dtype=torch.float64, device=ctx.device)
for sb in (4, 5, 6, 8):
    ctx.set_option("superpanel_fp16", sb)
    for rep in range(3):
        W.copy_(A)
        ipiv, info = ctx.factor(W, 256, trailing=MODE)
    s = ctx.stats()
    x, ir = ctx.solve_ir(A, W, ipiv, b, max_iter=5, tol=1e-12)
    print(f"N={n} {mode_name} sb={sb}: {s.ms_total:.2f} ms; big-K updates: {s.gemm_big_launches} launches, {s.ms_gemm_big:.2f} ms, {s.gemm_big_flops / max(s.ms_gemm_big, 1e-9) / 1e9:.1f} TFLOP/s "
          f"({s.gemm_big_flops / max(s.ms_gemm_big, 1e-9) / 1e9 / 2500:.4f} of spec); IR {ir.iterations} it, {ir.rel_residual:.1e}", flush=True)

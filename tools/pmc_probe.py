"""Runs the kernels whose HBM traffic we want from PMC counters, once each, at known algorithmic byte counts.
Use under:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python tools/pmc_probe.py
(and a second pass with WRITE_SIZE)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
dev = ctx.device
# calibration 1: 16 B/lane stream copy, 2 GiB read + 2 GiB write
print("stream copy TB/s", ctx.microbench(2))
# the GEMMs: C tile 16384 x 16384 inside an ld = 32768 matrix, K = 256
ld, m, n, k = 32768, 16384, 16384, 256
big = torch.rand((ld // 2 + 512, ld), dtype=torch.float64, device=dev).t()   # ld x (ld/2+512), column-major
A = big[256:256 + m, 0:k]; B = big[0:k, 256:256 + n]; Cm = big[256:256 + m, 256:256 + n]
ctx.dgemm_minus(Cm, A, B); ctx.synchronize()
print("dgemm algorithmic bytes: C r+w", 2 * m * n * 8, "A+B", (m + n) * k * 8)
ctx.hgemm_minus(Cm, A, B); ctx.synchronize()
print("hgemm algorithmic bytes: C r+w", 2 * m * n * 8)
# calibration 2: 8 B/lane coalesced read-only stream (residual GEMV): reads n2*n2*8 bytes
n2 = 16384
Asq = big[:n2, :n2]
x = torch.ones(n2, dtype=torch.float64, device=dev); b = torch.ones(n2, dtype=torch.float64, device=dev)
W = Asq.clone(); ip = torch.arange(1, n2 + 1, dtype=torch.int32, device=dev)
xx, st = ctx.solve_ir(Asq, W, ip, b, max_iter=0, tol=0.0)   # one residual pass (plus two cheap solves)
print("residual algorithmic bytes", n2 * n2 * 8)

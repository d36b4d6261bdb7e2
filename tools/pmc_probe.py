"""Runs the kernels whose HBM traffic / MFMA utilisation we want from PMC counters, once each, at known algorithmic byte and
flop counts (printed as PMCINFO lines for tools/pmc_summarize.py).  Use under
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_probe.py
(separate passes: WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE).
The update kernels run from the PRODUCT library; the probe library only supplies the calibration loops."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0)
pctx = mpf.MPFContext(0, probe=True)
dev = ctx.device
info = {}
# calibration: 16 B/lane stream copy, 2 GiB read + 2 GiB write; register-only MFMA loops
info["stream_copy_TBps"] = pctx.microbench(2)
info["f64_mfma_loop_tflops"] = pctx.microbench(0)
info["f16_mfma_loop_tflops"] = pctx.microbench(1)
# fp64 update: C tile 16384 x 16384 inside an ld = 32768 matrix, K = 256 and 1024
ld, m, n = 32768, 16384, 16384
big = torch.rand((ld // 2 + 1280, ld), dtype=torch.float64, device=dev).t()   # ld x (ld/2 + 1280), column-major
for k in (256, 1024):
    A = big[1280:1280 + m, 0:k]; B = big[0:k, 1280:1280 + n]; Cm = big[1280:1280 + m, 1280:1280 + n]
    ctx.dgemm_minus(Cm, A, B); ctx.synchronize()
    info[f"dgemm_k{k}"] = {"m": m, "n": n, "k": k, "flops": 2.0 * m * n * k, "read_bytes": m * n * 8 + (m + n) * k * 8, "write_bytes": m * n * 8}
# fp16 update on the fp32 working copy (what the two-level schedule runs), plain and split operands, K = 512 and 1024
m2 = 28672
C32 = torch.rand((m2, m2), dtype=torch.float32, device=dev).t()
Abuf = torch.rand((1024, m2), dtype=torch.float64, device=dev).t()      # m2 x 1024, column-major
Bbuf = torch.rand((m2, 1024), dtype=torch.float64, device=dev).t()      # 1024 x m2
for k in (512, 1024):
    A = Abuf[:, 0:k]; B = Bbuf[0:k, :]
    assert A.shape == (m2, k) and B.shape == (k, m2)
    for split in (False, True):
        ctx.hgemm_minus_f32(C32, A, B, split=split); ctx.synchronize()
        opb = 4 if split else 2
        info[f"hgemm_big_k{k}_{'split' if split else 'plain'}"] = {"m": m2, "n": m2, "k": k, "flops": 2.0 * m2 * m2 * k,
                                                                  "read_bytes": m2 * m2 * 4 + 2 * m2 * k * opb, "write_bytes": m2 * m2 * 4}
# calibration 2: 8 B/lane coalesced read-only stream (residual GEMV): reads n2*n2*8 bytes
n2 = 16384
Asq = big[:n2, :n2]
b = torch.ones(n2, dtype=torch.float64, device=dev)
W = Asq.clone(); ip = torch.arange(1, n2 + 1, dtype=torch.int32, device=dev)
xx, st = ctx.solve_ir(Asq, W, ip, b, max_iter=0, tol=0.0)   # one residual pass (plus two cheap solves)
info["residual"] = {"read_bytes": n2 * n2 * 8}
print("PMCINFO " + json.dumps(info))

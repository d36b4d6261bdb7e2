"""Where a dgemm tile spends its cycles (diagnostic build MPF_GEMM_STAMP=1): block 0 / wave 0 segment sums."""
import importlib, os, sys
os.environ["MPF_GEMM_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0)
dev = ctx.device
names = ["prologue (C + first stage)", "MFMA phases", "staging wait + LDS writes", "barriers", "epilogue (C stores)"]
print("variant MPF_GEMM_MF =", os.environ.get("MPF_GEMM_MF", "0"))
for (m, n, k) in [(128, 128, 256), (16384, 16384, 256), (32512 // 128 * 128, 32512 // 128 * 128, 256), (16384, 16384, 1024)]:
    ld = 36864
    big = torch.rand((ld, ld), dtype=torch.float64, device=dev).t()
    A = big[2048:2048 + m, 0:k]; B = big[0:k, 2048:2048 + n]; Cm = big[2048:2048 + m, 2048:2048 + n]
    for rep in range(3):
        ctx.dgemm_minus(Cm, A, B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ctx.dgemm_minus(Cm, A, B); e1.record(); torch.cuda.synchronize()
    segs = [ctx.microbench(70 + i) for i in range(5)]
    rt = ctx.microbench(75)
    ms = e0.elapsed_time(e1)
    clk = sum(segs) / (rt * 10.0) if rt > 0 else 0.0   # cycles per ns of the stamped tile: the shader clock it ran at
    print(f"m={m} n={n} k={k}: {ms:.3f} ms {2.0*m*n*k/ms/1e9:.1f} TFLOP/s; block 0 cycles:", ", ".join(f"{a}={b:.0f}" for a, b in zip(names, segs)),
          f"total={sum(segs):.0f} clock={clk:.2f} GHz; per stage: mfma={segs[1]/(k/16):.0f} stage-wait={segs[2]/(k/16):.0f} barrier={segs[3]/(k/16):.0f}")
    del big

"""f64 matrix-instruction issue patterns (mpf_microbench 100..): cycles per MFMA seen by one wave."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
pats = ["16x16x4 distinct A/B x16 acc (2048 flop)", "4x4x4_4b x16 acc (512 flop)", "16x16x4 GEMM-style operand reuse x16", "16x16x4 ONE accumulator (latency)", "4x4x4_4b x8 acc"]
cfgs = ["one wave alone", "1 wave/SIMD all CUs", "2 waves/SIMD all CUs", "4 waves/SIMD all CUs"]
for p, pn in enumerate(pats):
    for c, cn in enumerate(cfgs):
        cyc = ctx.microbench(100 + 10 * p + c)
        flop = 512 if p in (1, 4) else 2048
        waves = [1, 1, 2, 4][c]
        print(f"{pn:45s} {cn:22s}: {cyc:7.1f} cycles/MFMA per wave -> {flop * waves / cyc:6.1f} flop/clk/SIMD")

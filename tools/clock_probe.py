import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
print("f64 MFMA TFLOP/s (2 WG/CU, 8 acc):", round(ctx.microbench(0), 2))
print("cycles per v_mfma_f64_16x16x4_f64 as seen by one wave (2 waves/SIMD share the pipe):", round(ctx.microbench(3), 1))
print("sustained shader clock during the loop, GHz:", round(ctx.microbench(4), 3))
for w in (1, 2, 3):
    for v, na in ((0, 4), (1, 8), (2, 16)):
        print(f"waves/SIMD={w} accumulators={na}: {ctx.microbench(10*w+v):.1f} TFLOP/s (whole launch, HIP events)")
for v, name in ((0, "back to back"), (1, "one LDS-read operand between"), (2, "two LDS-read operands between"), (3, "s_nop 7 between")):
    print(f"f64 MFMA, 2 waves/SIMD, {name}: {ctx.microbench(50+v):.1f} TFLOP/s")

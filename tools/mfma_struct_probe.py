"""fp16 MFMA structure probes (libmpf_probe.so, microbench 90..95): cycles per 32x32x16-equivalent MFMA per SIMD, the clock the
chip holds, TFLOP/s -- for one / two waves per SIMD, with and without a barrier every 16 / 8 MFMAs, and the 16x16x32 shape."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
names = {90: "32x32x16, 1 wave/SIMD, no barrier", 91: "32x32x16, 2 waves/SIMD, no barrier", 92: "32x32x16, 2 waves, barrier / 16 MFMAs",
         93: "32x32x16, 2 waves, barrier / 8 MFMAs", 94: "16x16x32, 2 waves, barrier / 16 equiv.", 95: "16x16x32, 2 waves, no barrier"}
for rd in range(2):
    for v in sorted(names):
        cyc, clk, tf = ctx.microbench(v), ctx.microbench(100 + v), ctx.microbench(200 + v)
        if rd: print(f"{names[v]:42s}: {cyc:6.2f} cycles per MFMA-equivalent per SIMD, clock {clk:.3f} GHz, {tf:7.1f} TFLOP/s", flush=True)

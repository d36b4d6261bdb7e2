"""Stream copy (2 GiB -> 2 GiB) in a few forms: loads in flight per lane x blocks per CU, non-temporal or not (libmpf_probe.so,
microbench 400 + 10 u + b): which form reads the box's HBM rate (MI355X_MICROARCH.md: 6.29 TB/s for a float4 copy)?"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
for u, un in enumerate(("4 nt", "8 nt", "16 nt", "8 plain")):
    for b, bn in enumerate((8, 16, 32, 64)):
        print(f"{un:8s} loads in flight, {bn:2d} blocks per CU: {ctx.microbench(400 + 10 * u + b):.2f} TB/s", flush=True)
print(f"which = 2 (bench.py): {ctx.microbench(2):.2f} TB/s")

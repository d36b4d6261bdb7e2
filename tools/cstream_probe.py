"""The big fp16 update's C stream alone (round 5): through registers (load, subtract, store) against returnless fp32 atomic
adds, with and without the next tile's MFMAs behind it (csrc/microbench.hip, cstream_probe_kernel).  One line per form on stderr."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
for rd in range(2):
    for mode in (2, 0, 1):
        for sp in (0, 1, 2, 3):
            if mode == 2 and sp == 0: continue
            ctx.microbench(600 + 10 * mode + sp)

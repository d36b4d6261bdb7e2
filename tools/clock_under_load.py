"""Sustained shader clock while the library's update kernels run (one wave on the side stream compares s_memtime with the
constant 100 MHz clock over 20 ms)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
for which, name in ((302, "idle"), (300, "fp64 update m=n=16384 k=256 x24"), (301, "fp16 update x24"), (4, "register-only f64 MFMA loop"), (300, "fp64 update again"), (302, "idle again")):
    print(f"{name}: {ctx.microbench(which):.3f} GHz", flush=True)

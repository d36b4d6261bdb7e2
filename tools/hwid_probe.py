"""Dispatch map of a launch shaped like the fp64 update (512 threads, 73.7 KB LDS): block -> XCC / SE / CU / slot."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
ctx.microbench(310)

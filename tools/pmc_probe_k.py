"""hgemm at K = 256 / 1024 (plain and split), one launch each, for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
dev = ctx.device
print("stream copy TB/s", ctx.microbench(2))
m = 16384
Cm = torch.rand((m, m), dtype=torch.float64, device=dev).t()
for k in (256, 1024):
    A = torch.rand((k, m), dtype=torch.float64, device=dev).t()
    B = torch.rand((m, k), dtype=torch.float64, device=dev).t()
    for split in (False, True):
        ctx.hgemm_minus(Cm, A, B, split=split); ctx.synchronize()
        print("hgemm k", k, "split", split, "C r+w bytes", 2 * m * m * 8, "image bytes", 2 * m * k * 2 * (2 if split else 1))

"""The big fp16 update alone (plain and split, K = 1024, fp32 copy) for wave-level PMC passes:
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -- python3 tools/pmc_hgemm_only.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0)
m2 = 16384
C32 = torch.rand((m2, m2), dtype=torch.float32, device=ctx.device).t()
A = torch.rand((1024, m2), dtype=torch.float64, device=ctx.device).t()
B = torch.rand((m2, 1024), dtype=torch.float64, device=ctx.device).t()
for split in (False, True):
    ctx.hgemm_minus_f32(C32, A, B, split=split); ctx.synchronize()
print("done")

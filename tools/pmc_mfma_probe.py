"""One launch each of the MFMA-carrying kernels (and the register-only MFMA loops as calibration) for
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
dev = ctx.device
print("f64 mfma loop TF", ctx.microbench(0)); print("f16 mfma loop TF", ctx.microbench(1))
m = 16384
Cm = torch.rand((m, m), dtype=torch.float64, device=dev).t()
for k in (256, 1024):
    A = torch.rand((k, m), dtype=torch.float64, device=dev).t()
    B = torch.rand((m, k), dtype=torch.float64, device=dev).t()
    ctx.dgemm_minus(Cm, A, B); ctx.synchronize()
    for split in (False, True):
        ctx.hgemm_minus(Cm, A, B, split=split); ctx.synchronize()
    print("done k", k)

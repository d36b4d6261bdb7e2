"""One whole factorization per mode for per-kernel PMC passes (HBM traffic of the NON-GEMM kernels -- pivot
kernel, fp64 panel, interchanges, transposes, conversions -- in the shapes the schedule really launches):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 tools/pmc_factor_probe.py      (and again with WRITE_SIZE)
chain_pipeline = 0: counter collection runs one dispatch at a time, and a gate kernel that waits for a pivot kernel on another
stream would wait for its bounded two seconds.  Prints PMCFACTOR {json} with what was run (N, nb, modes, order)."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(os.environ.get("N", "32768")); nb = int(os.environ.get("NB", "256"))
ctx = mpf.MPFContext(0, options={"chain_pipeline": 0})
A0 = ctx.matgen(n)
W = A0.clone()
modes = [m for m in os.environ.get("MODES", "fp64,fp16").split(",") if m]
tm = {"fp64": mpf.TRAIL_FP64, "fp16": mpf.TRAIL_FP16, "fp16x3": mpf.TRAIL_FP16X3}
for m in modes:
    if m != "fp64":       # the IR-friendly input of the fp16 modes
        W.copy_(A0); idx = torch.arange(n, device=W.device); W[idx, idx] += A0.sum(dim=1)
    else:
        W.copy_(A0)
    ip, info = ctx.factor(W, nb, trailing=tm[m])
    ctx.synchronize()
    st = ctx.stats()
    print(f"{m}: info {info} ms {st.ms_total:.1f}", flush=True)
print("PMCFACTOR " + json.dumps({"n": n, "nb": nb, "modes": modes, "options": {"chain_pipeline": 0}}))

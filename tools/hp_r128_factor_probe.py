"""Whole factorizations through the PROBE library with the pivot kernel on 128-row slabs (hp_r256_upto = 0) against 256-row slabs:
usage: hp_r128_factor_probe.py   env MODES=fp16,fp64 SIZES=8192,16384,32768 NB=256"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
sizes = [int(a) for a in os.environ.get("SIZES", "8192,16384,32768").split(",")]
modes = os.environ.get("MODES", "fp16,fp64").split(",")
nb = int(os.environ.get("NB", "256"))
M = {"fp64": mpf.TRAIL_FP64, "fp16": mpf.TRAIL_FP16, "fp16x3": mpf.TRAIL_FP16X3}
ctxs = {"R=256": mpf.MPFContext(0, probe=True, options={"hp_r256_upto": 1 << 30}), "R=128": mpf.MPFContext(0, probe=True, options={"hp_r256_upto": 0})}
for n in sizes:
    A = ctxs["R=256"].matgen(n)
    Ad = A.clone(); idx = torch.arange(n, device=A.device); Ad[idx, idx] += A.sum(dim=1)
    W = torch.empty((n, n), dtype=torch.float64, device=A.device).t()
    for m in modes:
        src = Ad if m == "fp16" else A
        out = []
        for rnd in range(2):
            for name, ctx in ctxs.items():
                try:
                    for rep in range(2):
                        W.copy_(src); ipiv, info = ctx.factor(W, nb, trailing=M[m])
                    st = ctx.stats()
                    out.append(f"{name} {st.ms_total:.2f} ms (timeouts {st.hpanel_timeouts})")
                except Exception as e:
                    out.append(f"{name} ERR {str(e)[:60]}")
        print(f"N={n} nb={nb} {m}: " + " | ".join(out), flush=True)

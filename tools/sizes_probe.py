"""Factorization time by size and trailing mode on one GPU (device events, second call at each size); any error code is printed,
never hidden.  usage: sizes_probe.py [N ...]   default: 4096 .. 65536 in steps of 4096.  env MODES=fp64,fp16,fp16x3 NB=256"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0)
sizes = [int(a) for a in sys.argv[1:]] or list(range(4096, 65536 + 1, 4096))
modes = os.environ.get("MODES", "fp64,fp16,fp16x3").split(",")
nb = int(os.environ.get("NB", "256"))
M = {"fp64": mpf.TRAIL_FP64, "fp16": mpf.TRAIL_FP16, "fp16x3": mpf.TRAIL_FP16X3}
for n in sizes:
    A = ctx.matgen(n)
    Ad = A.clone()
    idx = torch.arange(n, device=ctx.device)
    Ad[idx, idx] += A.sum(dim=1)
    W = torch.empty((n, n), dtype=torch.float64, device=ctx.device).t()
    for m in modes:
        src = Ad if m == "fp16" else A
        try:
            for rep in range(2):
                W.copy_(src)
                ipiv, info = ctx.factor(W, nb, trailing=M[m])
            st = ctx.stats()
            print(f"N={n} {m} gen={'diagdom' if m == 'fp16' else 'ref'}: {st.ms_total:.2f} ms  {2 / 3 * n ** 3 / st.ms_total / 1e9:.1f} TFLOP/s  info={info} timeouts={st.hpanel_timeouts}", flush=True)
        except Exception as e:
            print(f"N={n} {m}: ERR {str(e)[:160]}", flush=True)
    del A, Ad, W
    torch.cuda.empty_cache()

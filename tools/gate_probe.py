"""The pipelined chain's gate as hipStreamWaitValue64 (option gate_wait_value) against the kernel that spins on the progress word:
fp16-mode and fp64-mode factorization time, bits compared.  usage: gate_probe.py [N ...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
for n in ([int(a) for a in sys.argv[1:]] or [8192, 32768]):
    ctxs = {0: mpf.MPFContext(0, options={"gate_wait_value": 0}), 1: mpf.MPFContext(0, options={"gate_wait_value": 1})}
    A = ctxs[0].matgen(n); Ad = A.clone(); idx = torch.arange(n, device=A.device); Ad[idx, idx] += A.sum(dim=1)
    W = torch.empty((n, n), dtype=torch.float64, device=A.device).t()
    for mode, name, src in ((mpf.TRAIL_FP16, "fp16", Ad), (mpf.TRAIL_FP64, "fp64", A)):
        res = {}
        for rd in range(3):
            for g, c in ctxs.items():
                W.copy_(src); ip, info = c.factor(W, 256, trailing=mode)
                st = c.stats()
                res.setdefault(g, []).append(st.ms_total)
                if rd == 2: res[("bits", g)] = (ip.clone(), W.clone() if n <= 16384 else None)
        same = bool(torch.equal(res[("bits", 0)][0], res[("bits", 1)][0])) and (res[("bits", 0)][1] is None or bool(torch.equal(res[("bits", 0)][1], res[("bits", 1)][1])))
        print(f"N={n} {name}: spinning gate {min(res[0][1:]):.2f} ms, stream wait value {min(res[1][1:]):.2f} ms, same bits {same}, timeouts {ctxs[1].stats().hpanel_timeouts}", flush=True)
    for c in ctxs.values(): c.close()

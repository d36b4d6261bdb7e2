"""One warm fp16-mode factorization under rocprofv3 --kernel-trace (default options: chain pipelined); tools/trace_chain_gaps.py
turns the trace into the kernels that run between two pivot kernels.  Usage (GPU box):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace16 -- python3 tools/trace_fp16_factor.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(os.environ.get("N", "32768")); nb = 256
ctx = mpf.MPFContext(0, options={"event_timers": 0})
A0 = ctx.matgen(n)
idx = torch.arange(n, device=A0.device); A0[idx, idx] += A0.sum(dim=1)
W = A0.clone()
for rep in range(2):
    W.copy_(A0)
    ctx.synchronize()
    ip, info = ctx.factor(W, nb, trailing=mpf.TRAIL_FP16)
    ctx.synchronize()
    print(f"rep {rep}: info {info} ms {ctx.stats().ms_total:.1f}", flush=True)

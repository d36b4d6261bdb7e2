"""fp16 / fp16x3 factorization time at N = 32768 for each form of the big-K update (option hgemm_big_tile), interleaved in one
process: does a faster kernel alone make a faster factorization?  usage: tile_mode_probe.py [tiles e.g. 0,3,4] [modes e.g. 1,2]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(os.environ.get("N", "32768"))
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "0,3,4").split(",")]
modes = [int(t) for t in (sys.argv[2] if len(sys.argv) > 2 else "1").split(",")]
ctx = mpf.MPFContext(0)
A = ctx.matgen(n)
idx = torch.arange(n, device=A.device)
Ad = A.clone(); Ad[idx, idx] += A.sum(dim=1)
W = torch.empty((n, n), dtype=torch.float64, device=A.device).t()
res = {}
for rd in range(4):
    for mode in modes:
        for t in tiles:
            ctx.set_option("hgemm_big_tile", t)
            W.copy_(Ad if mode == 1 else A)
            ctx.factor(W, 256, trailing=mode)
            st = ctx.stats()
            if rd: res.setdefault((mode, t), []).append((st.ms_total, st.ms_gemm_big, st.gemm_big_flops, st.gemm_big_launches))
for (mode, t), v in sorted(res.items()):
    v.sort()
    ms, gb, fl, nl = v[len(v) // 2]
    print(f"mode {mode} hgemm_big_tile {t}: factor median {ms:.1f} ms (min {v[0][0]:.1f}); big-K launches {nl}: {gb:.1f} ms = {fl / (gb * 1e-3) / 1e12 if gb > 0 else 0:.0f} TFLOP/s in the schedule", flush=True)

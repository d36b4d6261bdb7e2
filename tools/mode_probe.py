"""Time one factorization per trailing mode at N (default 32768), nb 256, matrix resident in HBM.
Usage: python tools/mode_probe.py [N] [modes, e.g. 012] [superpanel list] [gen]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
modes = sys.argv[2] if len(sys.argv) > 2 else "012"
ctx = mpf.MPFContext(0)
dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(1)
if len(sys.argv) > 4 and sys.argv[4] == "gen":      # the reference generator's own matrix (real pivoting: interchanges cost)
    A = ctx.matgen(n)
else:
    A = (torch.randint(0, 100, (n, n), generator=g, device=dev, dtype=torch.int32).to(torch.float64) / 10.0).t()
    idx = torch.arange(n, device=dev)
    A[idx, idx] += A.sum(dim=1)
W = torch.empty((n, n), dtype=torch.float64, device=dev).t()
sbs = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
for mode in modes:
  mode = int(mode)
  for sb in sbs:
    best = 1e9
    for rep in range(3):
        W.copy_(A)
        ctx.factor(W, 256, trailing=mode, superpanel=sb)
        st = ctx.stats()
        best = min(best, st.ms_total)
    print(f"N={n} mode={mode} superpanel={st.superpanel} gemm {st.gemm_flops / (st.ms_gemm * 1e-3) / 1e12 if st.ms_gemm > 0 else 0:.0f} TF: {best:.1f} ms ({2*n**3/3/best/1e9:.1f} TF)  hgetf2 {st.ms_hpanel:.1f} laswp+dpanel {st.ms_dpanel:.1f} gemm {st.ms_gemm:.1f} trsm {st.ms_trsm:.1f} laswp {st.ms_laswp:.1f} | big-K {st.gemm_big_launches} launches {st.ms_gemm_big:.1f} ms = {st.gemm_big_flops / (st.ms_gemm_big * 1e-3) / 1e12 if st.ms_gemm_big > 0 else 0:.0f} TF, {st.gemm_big_bytes / (st.ms_gemm_big * 1e-3) / 1e12 if st.ms_gemm_big > 0 else 0:.2f} TB/s; cvt {st.ms_cvt:.1f} block-row {st.ms_blockrow:.1f} timeouts {st.hpanel_timeouts}", flush=True)

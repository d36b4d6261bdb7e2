"""fp64 step against the two-lane threshold (option fp64_two_lanes): python tools/two_lane_probe.py [N]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for thr, pct, below in ((0, 50, 18432), (8192, 50, 18432), (8192, 40, 18432), (8192, 30, 18432), (8192, 45, 18432), (4096, 50, 15360), (4096, 40, 15360), (4096, 40, 12288)):
    ctx = mpf.MPFContext(0, options={"fp64_two_lanes": thr, "fp64_lane_a_pct": pct, "chain_pipeline_below": below})
    A = ctx.matgen(n)
    W = A.clone()
    best = 1e9
    for rep in range(3):
        W.copy_(A)
        ctx.factor(W, 256, trailing=0)
        best = min(best, ctx.stats().ms_total)
    st = ctx.stats()
    print(f"fp64_two_lanes={thr:6d} lane A {pct}% chain_pipeline_below={below}: best {best:.1f} ms  (last: hgetf2 {st.ms_hpanel:.1f} gemm {st.ms_gemm:.1f} trsm {st.ms_trsm:.1f} laswp {st.ms_laswp:.1f} cvt {st.ms_cvt:.1f})", flush=True)
    del W, A
    ctx.close()

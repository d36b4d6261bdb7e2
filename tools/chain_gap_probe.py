"""Per panel of a factorization (MPF_TIMELINE=1 prints every timed region): pivot kernel duration, what is left of the fp64 panel
when it ends, and the gap until the next pivot kernel starts.  Usage: python tools/chain_gap_probe.py [N] [mode 0|1|2]"""
import importlib, os, subprocess, sys
if os.environ.get("MPF_TIMELINE") != "1":
    env = dict(os.environ, MPF_TIMELINE="1")
    out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    rows = [l.split() for l in out.stderr.splitlines() if l.startswith("TL ")]
    print(out.stdout[-400:])
    runs, cur = [], []
    for r in rows:
        if cur and float(r[2]) == 0.0 and float(cur[-1][2]) > 0: runs.append(cur); cur = []
        cur.append(r)
    runs.append(cur)
    rows = runs[-1]
    ids = sorted({int(r[1]) for r in rows})
    hp_id, dp_id = ids[0], ids[2]          # ms_hpanel, ms_laswp, ms_dpanel, ... are consecutive doubles of mpf_stats
    hp = sorted([(float(r[2]), float(r[3])) for r in rows if int(r[1]) == hp_id])
    dp = sorted([(float(r[2]), float(r[3])) for r in rows if int(r[1]) == dp_id])
    print(f"{len(hp)} pivot kernels, {len(dp)} fp64 panels; region ids {ids}")
    print("panel  pivot_start  pivot_ms  panel_left_ms  gap_to_next_pivot_ms")
    tot_p = tot_left = tot_gap = 0.0
    for i in range(len(hp) - 1):
        left = max(0.0, dp[i][1] - hp[i][1]) if i < len(dp) else 0.0
        gap = hp[i + 1][0] - max(hp[i][1], dp[i][1] if i < len(dp) else 0.0)
        tot_p += hp[i][1] - hp[i][0]; tot_left += left; tot_gap += gap
        if i % 8 == 0 or i > len(hp) - 4 or os.environ.get("GAP_ALL"):
            print(f"{i:4d}  {hp[i][0]:9.2f}  {hp[i][1]-hp[i][0]:7.3f}  {left:7.3f}  {gap:7.3f}")
    print(f"sums: pivot kernels {tot_p:.1f} ms, fp64 panel left after them {tot_left:.1f} ms, gaps to the next pivot kernel {tot_gap:.1f} ms")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = mpf.MPFContext(0, probe=True)
A = ctx.matgen(n)
W = A.clone()
for rep in range(2):
    W.copy_(A)
    ctx.factor(W, 256, trailing=mode)
st = ctx.stats()
print(f"N={n} mode={mode}: {st.ms_total:.1f} ms hgetf2 {st.ms_hpanel:.1f} dpanel {st.ms_dpanel:.1f} gemm {st.ms_gemm:.1f} trsm {st.ms_trsm:.1f} laswp {st.ms_laswp:.1f}")

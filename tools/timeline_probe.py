"""Per-panel timeline of the look-ahead schedule (MPF_TIMELINE=1 makes the library print every timed region).
Usage: python tools/timeline_probe.py [N] [mode]  -> table: panel, chain (hgetf2, laswp+dpanel), strip, rest, period"""
import importlib, os, subprocess, sys
if os.environ.get("MPF_TIMELINE") != "1":
    env = dict(os.environ, MPF_TIMELINE="1")
    out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    rows = [l.split() for l in out.stderr.splitlines() if l.startswith("TL ")]
    print(out.stdout[-400:])
    # keep the LAST factorization's lines: a new one starts where the start time goes back to 0
    runs, cur = [], []
    for r in rows:
        t0 = float(r[2])
        if cur and t0 == 0.0 and float(cur[-1][2]) > 0: runs.append(cur); cur = []
        cur.append(r)
    runs.append(cur)
    rows = runs[-1]
    names = {}
    ids = sorted({int(r[1]) for r in rows})
    # offsets (in doubles) inside mpf_stats: ms_hpanel, ms_laswp, ms_dpanel, ms_trsm, ms_gemm are consecutive
    label = dict(zip(ids, ["hgetf2", "laswp", "dpanel", "trsm", "gemm"])) if len(ids) == 5 else {i: str(i) for i in ids}
    ev = [(label[int(r[1])], float(r[2]), float(r[3])) for r in rows]
    hp = [e for e in ev if e[0] == "hgetf2"]
    dp = [e for e in ev if e[0] == "dpanel"]
    gm = [e for e in ev if e[0] == "gemm"]
    print("panel  hgetf2[start dur]  dpanel[dur]  chain_end | gemm strip[start dur] rest[start dur end]")
    # gemm pairs come as strip, rest per panel
    gi = 0
    for i in range(1, len(hp)):
        h = hp[i]; d = dp[i - 1] if i - 1 < len(dp) else None
        strip = gm[2 * (i - 1)] if 2 * (i - 1) < len(gm) else None
        rest = gm[2 * (i - 1) + 1] if 2 * (i - 1) + 1 < len(gm) else None
        if i % 8 == 0 or i > len(hp) - 4:
            print(f"{i:4d}  {h[1]:8.2f} {h[2]-h[1]:6.3f}  {d[2]-d[1] if d else 0:6.3f}  {d[2] if d else 0:8.2f} | "
                  f"{strip[1] if strip else 0:8.2f} {strip[2]-strip[1] if strip else 0:6.3f}  {rest[1] if rest else 0:8.2f} {rest[2]-rest[1] if rest else 0:6.3f} {rest[2] if rest else 0:8.2f}")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = mpf.MPFContext(0, probe=True)
A = ctx.matgen(n)
W = A.clone()
for rep in range(2):
    W.copy_(A)
    ctx.factor(W, 256, trailing=mode)
st = ctx.stats()
print(f"N={n} mode={mode}: {st.ms_total:.1f} ms hgetf2 {st.ms_hpanel:.1f} dpanel {st.ms_dpanel:.1f} gemm {st.ms_gemm:.1f} trsm {st.ms_trsm:.1f} laswp {st.ms_laswp:.1f}")

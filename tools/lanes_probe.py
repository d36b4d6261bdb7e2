"""Timeline of the two-level schedule's lanes (MPF_TIMELINE=1 makes the library print every timed region as TL <timer> <t0> <t1>).
Usage: python tools/lanes_probe.py [N] [mode] [sb] [gen]  -> per-panel gap between consecutive pivot kernels and what ran in it."""
import importlib, os, subprocess, sys
NAMES = {0: "hgetf2", 1: "laswp", 2: "dpanel", 3: "trsm", 4: "gemm", 15: "cvt"}
if os.environ.get("MPF_TIMELINE") != "1":
    env = dict(os.environ, MPF_TIMELINE="1")
    out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    rows = [l.split() for l in out.stderr.splitlines() if l.startswith("TL ")]
    print(out.stdout[-600:])
    runs, cur = [], []
    for r in rows:                      # a new factorization starts where the start time goes back to 0
        if cur and float(r[2]) == 0.0 and float(cur[-1][2]) > 0: runs.append(cur); cur = []
        cur.append(r)
    runs.append(cur)
    ev = sorted(((NAMES.get(int(r[1]), r[1]), float(r[2]), float(r[3])) for r in runs[-1]), key=lambda e: e[1])
    hp = [e for e in ev if e[0] == "hgetf2"]
    tot_piv = sum(e[2] - e[1] for e in hp)
    gaps = [hp[i + 1][1] - hp[i][2] for i in range(len(hp) - 1)]
    print(f"pivot kernels: {len(hp)}, sum of durations {tot_piv:.1f} ms, sum of gaps between them {sum(gaps):.1f} ms, end of last {hp[-1][2]:.1f} ms, "
          f"last event ends {max(e[2] for e in ev):.1f} ms")
    for lo in range(0, len(hp) - 1, 16):
        g = gaps[lo:lo + 16]
        print(f"panels {lo:3d}..{lo + len(g) - 1:3d}: pivot avg {sum(h[2] - h[1] for h in hp[lo:lo + 16]) / len(hp[lo:lo + 16]):.3f} ms, gap avg {sum(g) / len(g):.3f} max {max(g):.3f}")
    for p in (9, 10, 11, 12, 60, 61):
        if p + 1 >= len(hp): continue
        a, b = hp[p][2], hp[p + 1][1]
        print(f"--- between pivot kernel {p} (ends {a:.3f}) and {p + 1} (starts {b:.3f}; runs {hp[p+1][2]-hp[p+1][1]:.3f}):")
        for e in ev:
            if e[2] > a - 0.3 and e[1] < hp[p + 1][2] and e[0] != "hgetf2":
                print(f"      {e[0]:7s} {e[1]:9.3f} -> {e[2]:9.3f}  ({e[2] - e[1]:.3f})")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = mpf.MPFContext(0)
A = ctx.matgen(n)
if not (len(sys.argv) > 4 and sys.argv[4] == "gen"):
    idx = torch.arange(n, device=ctx.device)
    A[idx, idx] += A.sum(dim=1)
W = A.clone()
for rep in range(2):
    W.copy_(A)
    ctx.factor(W, 256, trailing=mode, superpanel=sb)
st = ctx.stats()
print(f"N={n} mode={mode} sb={st.superpanel}: {st.ms_total:.1f} ms hgetf2 {st.ms_hpanel:.1f} dpanel {st.ms_dpanel:.1f} gemm {st.ms_gemm:.1f} (big {st.ms_gemm_big:.1f}) trsm {st.ms_trsm:.1f} "
      f"(block-row {st.ms_blockrow:.1f}) laswp {st.ms_laswp:.1f} cvt {st.ms_cvt:.1f}")

"""Where the fp64 step's time goes: main-stream busy / idle time per group of panels (option timeline = every timed region on stderr).
Usage: python tools/fp64_timeline.py [N]"""
import importlib, os, subprocess, sys
NAMES = {0: "hgetf2", 1: "laswp", 2: "dpanel", 3: "trsm", 4: "gemm", 15: "cvt"}
if os.environ.get("MPF_TIMELINE") != "1":
    env = dict(os.environ, MPF_TIMELINE="1")
    out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    rows = [l.split() for l in out.stderr.splitlines() if l.startswith("TL ")]
    print(out.stdout[-600:])
    runs, cur = [], []
    for r in rows:
        if cur and float(r[2]) == 0.0 and float(cur[-1][2]) > 0: runs.append(cur); cur = []
        cur.append(r)
    runs.append(cur)
    ev = sorted(((NAMES.get(int(r[1]), r[1]), float(r[2]), float(r[3])) for r in runs[-1]), key=lambda e: e[1])
    hp = [e for e in ev if e[0] == "hgetf2"]
    main = [e for e in ev if e[0] in ("laswp", "trsm", "gemm")]
    cvt = [e for e in ev if e[0] == "cvt"]
    end = max(e[2] for e in ev)
    print(f"{len(hp)} pivot kernels; last event ends {end:.1f} ms")
    # group by pivot kernel start times: panel p's period = [start of pivot kernel p, start of pivot kernel p + 1)
    edges = [h[1] for h in hp] + [end]
    G = 16
    print("panels      period  pivots  gemm    trsm   laswp  cvt(any stream)  main-stream idle (gaps > 3 us, cvt inside them subtracted)")
    for lo in range(0, len(hp), G):
        a, b = edges[lo], edges[min(lo + G, len(hp))]
        n = min(lo + G, len(hp)) - lo
        def inside(lst, kind=None):
            return sum(min(e[2], b) - max(e[1], a) for e in lst if (kind is None or e[0] == kind) and e[2] > a and e[1] < b)
        m = [e for e in main if e[2] > a and e[1] < b]
        idle = 0.0
        for x, y in zip(m, m[1:]):
            g = y[1] - x[2]
            if g > 0.003:
                g -= sum(min(c[2], y[1]) - max(c[1], x[2]) for c in cvt if c[2] > x[2] and c[1] < y[1])
                idle += max(g, 0.0)
        print(f"{lo:3d}..{lo + n - 1:3d}  {(b - a) / n:7.3f} {inside(hp) / n:7.3f} {inside(main, 'gemm') / n:7.3f} {inside(main, 'trsm') / n:6.3f} "
              f"{inside(main, 'laswp') / n:6.3f} {inside(cvt) / n:6.3f}          {idle / n:7.3f}    (ms per panel)")
    tail = [e for e in ev if e[1] >= hp[-1][2]]
    print("after the last pivot kernel:", ", ".join(f"{e[0]} {e[2] - e[1]:.2f}" for e in tail if e[2] - e[1] > 0.2))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = mpf.MPFContext(0)
A = ctx.matgen(n)
W = A.clone()
for rep in range(2):
    W.copy_(A)
    ctx.factor(W, 256, trailing=0)
st = ctx.stats()
print(f"N={n} fp64: {st.ms_total:.1f} ms hgetf2 {st.ms_hpanel:.1f} dpanel {st.ms_dpanel:.1f} gemm {st.ms_gemm:.1f} trsm {st.ms_trsm:.1f} laswp {st.ms_laswp:.1f} cvt {st.ms_cvt:.1f}")

"""Random shapes and plans through mpf_factor_host (block-row sink + late column segments forced on at every size) against the device
entry point, bit for bit.  usage: host_path_fuzz.py [cases] [seed]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ctx = mpf.MPFContext(0)
ctx.set_option("host_sink_min_n", 0); ctx.set_option("host_late_min_n", 0); ctx.set_option("fp64_rowmajor_min_n", 0)
bad = 0
for i in range(cases):
    nb = int(rng.choice([32, 48, 64, 96, 128, 200, 256]))
    npan = int(rng.integers(33, 70))
    n = nb * npan - int(rng.integers(0, nb))          # ragged last panel most of the time
    parts = int(rng.integers(0, 5)); first = int(rng.integers(10, 90)); qpct = int(rng.integers(20, 300)); sink = int(rng.integers(0, 4) != 0)
    A = np.asfortranarray(rng.standard_normal((n, n)))
    if rng.integers(0, 3) == 0: A[np.arange(n), np.arange(n)] += 50.0      # sometimes hardly any pivoting
    dA = ctx.from_numpy_f(A)
    ipiv_d, info = ctx.factor(dA, nb); ctx.synchronize()
    LU_d, ip_d = ctx.to_numpy_f(dA), ipiv_d.cpu().numpy()
    for k, v in (("host_sink", sink), ("host_late_parts", parts), ("host_first_pct", first), ("host_late_q_pct", qpct)): ctx.set_option(k, v)
    Ah = A.copy(order="F"); ip, _ = ctx.factor_host(Ah, nb)
    st = ctx.stats()
    ok = np.array_equal(ip, ip_d) and np.array_equal(Ah.view(np.uint64), LU_d.view(np.uint64))
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} n={n} nb={nb} sink={sink} parts={parts} first={first}% q={qpct}%: rows streamed {st.host_rows_streamed} late segments {st.host_late_segments}"
          + ("" if ok else f"  differing elements {int((Ah.view(np.uint64) != LU_d.view(np.uint64)).sum())}"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)

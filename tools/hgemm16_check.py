"""Where does hgemm16_big_kernel differ from fp64 arithmetic on the rounded operands?  Prints count, size and pattern of the wrong
elements of one update on the fp32 copy.  usage: hgemm16_check.py [m n k]   env TILE=4|5 (one tile per workgroup / persistent)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, options={"hgemm_big_tile": int(os.environ.get("TILE", "4"))})
m, n, k = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (1024, 1024, 256)
rng = np.random.default_rng(1)
A = np.asfortranarray(rng.standard_normal((m, k))); B = np.asfortranarray(rng.standard_normal((k, n)))
Ah = A.astype(np.float32).astype(np.float16).astype(np.float64); Bh = B.astype(np.float32).astype(np.float16).astype(np.float64)   # (through fp32, as the kernels round)
P = Ah @ Bh
S = np.abs(Ah) @ np.abs(Bh)
for extra in (0, 3):
    for rep in range(int(os.environ.get('REPS', '3'))):
        Cm = rng.standard_normal((m + extra, n)).astype(np.float32)
        d = torch.empty((n, m + extra), dtype=torch.float32, device=ctx.device).t()
        d.copy_(torch.from_numpy(Cm))
        ctx.hgemm_minus_f32(d[:m, :], ctx.from_numpy_f(A), ctx.from_numpy_f(B)); ctx.synchronize()
        got = d.cpu().numpy().astype(np.float64)
        want = Cm.astype(np.float64); want[:m] -= P
        err = np.abs(got - want)
        tol = np.zeros_like(err); tol[:m] = 2.0 * (k / 32 + 2) * 2.0 ** -24 * (S + np.abs(want[:m])) + 1e-7
        bad = err > tol
        print(f"ld = m + {extra} rep {rep}: wrong {bad.sum()} of {bad.size}, max err {err.max():.3e} (max tol {tol.max():.3e}); rows below m touched: {not np.array_equal(got[m:], Cm[m:].astype(np.float64))}", flush=True)
        if bad.any():
            r, c = np.nonzero(bad)
            print("  rows mod 256:", sorted(set((r % 256).tolist()))[:40], " rows // 256:", sorted(set((r // 256).tolist())))
            print("  cols mod 256:", sorted(set((c % 256).tolist()))[:40], " cols // 256:", sorted(set((c // 256).tolist())))
            print("  first few:", [(int(a), int(b), float(f"{err[a, b]:.3e}")) for a, b in zip(r[:8], c[:8])])
            big = err > 0.05
            if big.any():
                r, c = np.nonzero(big)
                print(f"  LARGE errors: {big.sum()}; rows {r.min()}..{r.max()} cols {c.min()}..{c.max()}; rows mod 4: {sorted(set((r % 4).tolist()))}; cols: {sorted(set(c.tolist()))[:20]}; rows: {sorted(set(r.tolist()))[:24]}")
                a, b2 = int(r[0]), int(c[0])
                print(f"  at ({a},{b2}): got {got[a, b2]:.6f} want {want[a, b2]:.6f} C {Cm[a, b2]:.6f} P {P[a, b2] if a < m else 0:.6f}")

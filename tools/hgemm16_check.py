"""Where does hgemm16_big_kernel differ from fp64 arithmetic?  Prints the pattern of wrong rows / columns of one update on the
fp32 copy for an aligned and an unaligned leading dimension.  usage: hgemm16_check.py [m n k]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0)
m, n, k = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (1024, 1024, 256)
rng = np.random.default_rng(1)
A = np.asfortranarray(rng.standard_normal((m, k))); B = np.asfortranarray(rng.standard_normal((k, n)))
Ah = A.astype(np.float16).astype(np.float64); Bh = B.astype(np.float16).astype(np.float64)
P = Ah @ Bh
for extra in (0, 3, 4):
    Cm = rng.standard_normal((m + extra, n)).astype(np.float32)
    d = torch.empty((n, m + extra), dtype=torch.float32, device=ctx.device).t()
    d.copy_(torch.from_numpy(Cm))
    ctx.hgemm_minus_f32(d[:m, :], ctx.from_numpy_f(A), ctx.from_numpy_f(B)); ctx.synchronize()
    got = d.cpu().numpy().astype(np.float64)
    want = Cm.astype(np.float64); want[:m] -= P
    bad = np.abs(got - want) > 1e-3 * (1 + np.abs(want))
    print(f"ld = m + {extra}: wrong elements {bad.sum()} of {bad.size}; rows below m touched: {not np.array_equal(got[m:], Cm[m:].astype(np.float64))}")
    if bad.any():
        r, c = np.nonzero(bad)
        print("  wrong rows mod 64:", sorted(set((r % 64).tolist()))[:70])
        print("  wrong rows // 64 :", sorted(set((r // 64).tolist()))[:40])
        print("  wrong cols mod 64:", sorted(set((c % 64).tolist()))[:70])
        print("  first few:", list(zip(r[:6].tolist(), c[:6].tolist())))
        # is a wrong element some OTHER element's value?
        i, j = int(r[0]), int(c[0])
        near = [(di, dj) for di in range(-64, 65) for dj in range(-64, 65) if 0 <= i + di < m and 0 <= j + dj < n and abs(got[i, j] - want[i + di, j + dj]) < 1e-4]
        print("  got[%d,%d] equals want at offsets:" % (i, j), near[:8])

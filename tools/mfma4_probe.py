"""v_mfma_f64_4x4x4_4b_f64 on gfx950: throughput (HIP events), operand / result lane layout, broadcast and NEG controls,
and which CPU arithmetic reproduces it bit for bit."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
L = ctx.L
L.mpf_debug_mfma4.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]

names = ["16x16x4 distinct x16", "4x4x4_4b x16", "16x16x4 reuse x16", "16x16x4 one acc", "4x4x4_4b x8"]
for p in (0, 1, 3, 4):
    for w in (1, 2, 3, 4):
        print(f"{names[p]:24s} {w} WG/CU (waves/SIMD): {ctx.microbench(200 + 10 * p + w):7.1f} TFLOP/s by events", flush=True)

def run(a, b, c, variant):
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64); c = np.ascontiguousarray(c, dtype=np.float64)
    out = np.zeros(64)
    rc = L.mpf_debug_mfma4(ctx.h, a.ctypes.data, b.ctypes.data, c.ctypes.data, variant, out.ctypes.data)
    assert rc == 0, rc
    return out

# layout: A = one-hot per lane, B = distinct powers-free integers -> which (a-lane, b-lane) pairs feed which output lane
z = np.zeros(64)
pairs = {}
for la in range(64):
    a = z.copy(); a[la] = 1.0
    b = np.arange(64) + 1.0   # b lane value identifies the B lane
    d = run(a, b, z, 0)
    for lo in np.nonzero(d)[0]:
        pairs.setdefault(int(lo), []).append((la, int(round(d[lo])) - 1))
print("variant 0 (no broadcast): output lane <- list of (A lane, B lane) products")
for lo in (0, 1, 4, 5, 15, 16, 17, 21, 63):
    print("  D lane", lo, "<-", pairs.get(lo))
# full decode into (block, i, j, k) form
ok = True
for lo in range(64):
    blk, r = divmod(lo, 16)
    got = sorted(pairs.get(lo, []))
    print_first = lo < 2
# broadcast variants
for v in (1, 2, 3, 4, 8, 9):
    pairs_v = {}
    for la in range(64):
        a = z.copy(); a[la] = 1.0
        d = run(a, np.arange(64) + 1.0, z, v)
        for lo in np.nonzero(d)[0]:
            pairs_v.setdefault(int(lo), []).append((la, int(round(d[lo])) - 1))
    print(f"variant {v}: D lane 0 <- {pairs_v.get(0)}; D lane 17 <- {pairs_v.get(17)}; D lane 63 <- {pairs_v.get(63)}")
# NEG controls (BLGP bits)
rng = np.random.default_rng(0)
a, b, c = rng.standard_normal(64), rng.standard_normal(64), rng.standard_normal(64)
d0 = run(a, b, c, 0)
for v, nm in ((5, "BLGP=1"), (6, "BLGP=2"), (7, "BLGP=4")):
    d = run(a, b, c, v)
    print(nm, "== mfma(-a,b,c):", np.array_equal(d, run(-a, b, c, 0)), " == mfma(a,-b,c):", np.array_equal(d, run(a, -b, c, 0)),
          " == mfma(a,b,-c):", np.array_equal(d, run(a, b, -c, 0)), " == plain:", np.array_equal(d, d0))
# arithmetic: fma chain k ascending?
def wide(n):
    return rng.standard_normal(n) * np.exp2(rng.integers(-20, 20, n))
from fractions import Fraction
def fma(x, y, z):
    return float(Fraction(x) * Fraction(y) + Fraction(z))   # exact, rounded once
match_asc = match_desc = total = 0
for trial in range(50):
    a, b, c = wide(64), wide(64), wide(64)
    d = run(a, b, c, 0)
    for lo in range(64):
        pr = sorted(pairs[lo])          # assume the B lane order within an output follows k
        asc = c[lo]
        for (la, lb) in pr:
            asc = fma(a[la], b[lb], asc)
        desc = c[lo]
        for (la, lb) in reversed(pr):
            desc = fma(a[la], b[lb], desc)
        match_asc += d[lo] == asc; match_desc += d[lo] == desc; total += 1
print(f"fma chain in ascending A-lane order matches {match_asc}/{total}; descending {match_desc}/{total} ")
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "mfma4_pairs.npy"), np.array([pairs[l] for l in range(64)], dtype=object), allow_pickle=True)

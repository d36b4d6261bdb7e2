"""One-way latency of a flag between two workgroups on different XCDs, per write method (the pivot kernel's hand-off is made of
such trips): mpf_microbench 500 + method; 0 write-through store, 1 atomic exchange, 2 release store, 3 atomic add, 4 atomic max."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
names = ["write-through store (sc1)", "atomic exchange", "release store (wbl2 + store)", "atomic add", "atomic max"]
for rep in range(2):
    for m, n in enumerate(names):
        print(f"{n:32s}: {ctx.microbench(500 + m):7.0f} ns one way", flush=True)

# round 5: the same trip with the accesses spelled out, between XCDs (workgroups 0 and 1) and inside one (0 and 8)
lms = ["sc1 load", "sc0 load", "sc0 sc1 load"]; sms = ["sc1 store", "plain store", "sc0 sc1 store"]
for same in (0, 1):
    for lm in range(3):
        for sm in range(3):
            v = ctx.microbench(510 + 30 * same + 10 * lm + sm)
            print(f"{'same XCD' if same else 'two XCDs'}: {lms[lm]:13s} / {sms[sm]:14s}: " + (f"{v:7.0f} ns one way" if v >= 0 else f"NOT VISIBLE (gave up after {int(-v)} rounds)"), flush=True)

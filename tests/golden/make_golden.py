"""Regenerates tests/golden/mpf_golden.json from the CPU oracle (oracle/mpf_oracle.c).

The reference ships no golden vectors and its CUDA path cannot be built here, so these vectors pin the
oracle (and through it the HIP path) against regressions; the generator inputs themselves are pinned
against the real reference generator binary (oracle/_ref/matgen) by tests/test_oracle.py.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O  # noqa: E402

cases = []


def add(n, r, step=2, func="exp", sparsity=0.0):
    A = O.matgen(n, step, func, sparsity)
    LU, ip = O.mpf(A, r)
    mx, fro = O.check_plu(A, LU, ip)
    if not np.isfinite(mx):
        return
    cases.append(dict(n=n, r=r, step=step, func=func, sparsity=sparsity, ipiv=ip.tolist(),
                      lu_sha256=hashlib.sha256(np.ascontiguousarray(LU.T).tobytes()).hexdigest(),
                      max_abs_err=mx, fro_err=fro))


for n in (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024):
    for r in (32, 128, 256):
        if r == 32 or n > r // 2:
            add(n, r)
for n, step in ((3, 1), (31, 29), (33, 31), (65, 63), (127, 125), (129, 127), (257, 255), (513, 511), (1000, 998)):
    for r in (32, 128):
        add(n, r, step, "lin")
for n, sp in ((64, 0.3), (128, 0.5), (256, 0.3)):
    add(n, 32, 2, "exp", sp)
add(256, 7)      # odd panel width
add(100 + 2, 32, 100, "lin")

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mpf_golden.json")
with open(out, "w") as f:
    json.dump(dict(contract="MPF-AMD contract v1 (oracle/mpf_oracle.c header)", cases=cases), f)
print(f"wrote {len(cases)} cases to {out}, {os.path.getsize(out)} bytes")

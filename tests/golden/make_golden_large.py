"""Parity-at-scale fixtures: runs the CPU oracle (oracle/mpf_oracle.c) on the reference generator's own stream
(`matgen f N (N-2) lin` == oracle.matgen_skip(N): glibc rand(), 4 draws skipped, matrix_generator.cpp:55-80) at
BASELINE sizes and writes, per (N, nb):

    tests/golden/large_N{N}_nb{nb}_ipiv.npy     IPIV (int32, N entries, 1-based)
    tests/golden/large_N{N}_nb{nb}_colsum.npy   per-column position-weighted checksum of the LU bits (uint64, N entries):
                                                cs[j] = sum_i bits(LU[i, j]) * (2 i + 1)  mod 2^64
    tests/golden/mpf_golden_large.json          sha256 of IPIV, of the checksum vector and of all N^2 LU values
                                                (column-major byte order), oracle wall time

The checksum vector lets the GPU test name the first column / panel that differs without shipping 8 GiB; the full sha256
is compared as well.  N = 32768, nb = 256 is BASELINE config C3 (about 20 minutes of CPU here).

    python tests/golden/make_golden_large.py 4096:256 8192:128 8192:256 32768:256
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(HERE, "mpf_golden_large.json")


def column_checksums(LU):
    """cs[j] = sum_i bits(LU[i, j]) * (2 i + 1) mod 2^64, LU column-major float64."""
    n = LU.shape[0]
    w = (2 * np.arange(n, dtype=np.uint64) + np.uint64(1))[:, None]
    cs = np.empty(LU.shape[1], dtype=np.uint64)
    bits = LU.view(np.uint64)
    step = max(1, (1 << 26) // n)
    for c0 in range(0, LU.shape[1], step):
        cs[c0:c0 + step] = (bits[:, c0:c0 + step] * w).sum(axis=0, dtype=np.uint64)
    return cs


def sha_colmajor(LU):
    h = hashlib.sha256()
    step = max(1, (1 << 27) // LU.shape[0])
    for c0 in range(0, LU.shape[1], step):
        h.update(np.ascontiguousarray(LU[:, c0:c0 + step].T).tobytes())
    return h.hexdigest()


def main():
    cases = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            cases = json.load(f)["cases"]
    for arg in sys.argv[1:]:
        n, nb = (int(v) for v in arg.split(":"))
        t0 = time.time()
        A = O.matgen_skip(n)
        ipiv = np.arange(1, n + 1, dtype=np.int32)
        rc = O.lib().orc_mpf(O._dp(A), n, nb, O._ip(ipiv), 0)   # in place: no second N^2 copy at N = 32768
        assert rc == 0
        dt = time.time() - t0
        cs = column_checksums(A)
        key = f"N{n}_nb{nb}"
        np.save(os.path.join(HERE, f"large_{key}_ipiv.npy"), ipiv)
        np.save(os.path.join(HERE, f"large_{key}_colsum.npy"), cs)
        cases[key] = dict(n=n, nb=nb, input="oracle.matgen_skip(n, skip=4) == `matgen f n (n-2) lin` (matrix_generator.cpp:55-80)",
                          ipiv_sha256=hashlib.sha256(ipiv.tobytes()).hexdigest(),
                          colsum_sha256=hashlib.sha256(cs.tobytes()).hexdigest(),
                          lu_sha256=sha_colmajor(A), oracle_seconds=round(dt, 1),
                          swaps=int((ipiv != np.arange(1, n + 1)).sum()))
        with open(OUT, "w") as f:
            json.dump(dict(contract="MPF-AMD contract v1 (oracle/mpf_oracle.c header)", cases=cases), f, indent=1)
        print(key, cases[key], flush=True)


if __name__ == "__main__":
    main()

"""CPU tests of the oracle (oracle/mpf_oracle.c): pinned against the real reference generator binary,
the committed golden vectors, LAPACK on the sizes where fp16 and fp64 partial pivoting coincide, and the
reference's own acceptance test (benchmark.cpp:97-134)."""
import ctypes
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN_DIR


def test_rand_matches_glibc_and_survey_kat(oracle):
    kat = [1804289383, 846930886, 1681692777, 1714636915, 1957747793, 424238335]
    assert oracle.rand_stream(6).tolist() == kat
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    assert oracle.rand_stream(5000).tolist() == [libc.rand() for _ in range(5000)]


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "matgen")),
                    reason="oracle/_ref/matgen (built from the reference tree) not present")
@pytest.mark.parametrize("args", [("128", "2", "exp"), ("40", "7", "lin", "0.3"), ("30", "3", "lin")])
def test_matgen_matches_reference_binary(oracle, tmp_path, args):
    """The in-memory generator restatement equals the reference's own matrix_generator.cpp output."""
    f = tmp_path / "m.txt"
    subprocess.run([oracle.REF_MATGEN, str(f), *args], check=True, capture_output=True)
    step = int(args[1]); func = args[2]; sp = float(args[3]) if len(args) > 3 else 0.0
    mats = oracle.read_matgen_file(str(f))
    assert len(mats) >= 3
    for M in mats:
        assert np.array_equal(M, oracle.matgen(M.shape[0], step, func, sp))


def test_matgen_skip_equals_lin_sequence(oracle):
    # `matgen f N (N-2) lin` emits sizes 2 then N  
    assert np.array_equal(oracle.matgen_skip(50), oracle.matgen(50, 48, "lin"))


def test_fp16_conversion_portable_vs_f16c(oracle):
    L = oracle.lib()
    if not L.orc_has_f16c():
        pytest.skip("no F16C on this host")
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.exp2(rng.integers(-30, 18, 200000)).astype(np.float32),
                         np.array([0, -0.0, 65504, 65519.99, 65520, 1e9, np.inf, -np.inf, 2.0 ** -14, 2.0 ** -24, 2.0 ** -25,
                                   1.5 * 2.0 ** -24, 2.5 * 2.0 ** -24, 1 + 2.0 ** -11, 1 + 3 * 2.0 ** -11], dtype=np.float32)])
    for x in xs.tolist():
        assert L.orc_f32_to_f16(x) == L.orc_f32_to_f16_hw(x), x
    # every fp16 value survives a round trip, and matches numpy's float16
    allh = np.arange(65536, dtype=np.uint16)
    f = allh.view(np.float16).astype(np.float32)
    ok = ~np.isnan(f)
    back = np.array([L.orc_f32_to_f16(float(v)) for v in f[ok]], dtype=np.uint16)
    assert np.array_equal(back, allh[ok])


def test_double_to_fp16_semantics(oracle):
    m = float(np.float32(6.10352e-05))
    x = np.array([1e9, -1e9, 65504.0, 2.0 ** -14, -2.0 ** -14, float(np.nextafter(np.float32(m), np.float32(0))), m, 0.1, 9.9])
    got = oracle.double_to_fp16(x).view(np.float16).astype(np.float64)
    # clamp to +-65504 (fp16_utils.h:19-20); everything with |xf| < 6.10352e-05f = 2^-14 + 6 ulp, +-2^-14
    # included, flushes to zero (:21); the threshold itself survives as 2^-14
    assert got[0] == 65504 and got[1] == -65504 and got[2] == 65504
    assert got[3] == 0 and got[4] == 0 and got[5] == 0 and got[6] == 2.0 ** -14
    assert got[7] == np.float64(np.float16(0.1)) and got[8] == np.float64(np.float16(9.9))


def test_hgetf2_portable_equals_vectorised(oracle):
    rng = np.random.default_rng(5)
    P = np.asfortranarray(rng.integers(0, 100, (700, 64)) / 10.0)
    b1 = oracle.double_to_fp16(P)
    b2 = b1.copy(order="F")
    p1 = oracle.hgetf2(b1)
    oracle.lib().orc_force_portable_fp16(1)
    try:
        p2 = oracle.hgetf2(b2)
    finally:
        oracle.lib().orc_force_portable_fp16(0)
    assert np.array_equal(p1, p2) and np.array_equal(b1, b2)


def test_tie_break_is_bit_reversed_lane_then_lowest_block(oracle):
    """hgetf2_kernel.cu:47-56,73-78: among equal maxima the strict-'>' tree keeps the smallest
    bit-reversed lane of the lowest 256-row block (lanes 64 and 128 tie -> 128 wins)."""
    rows = 600
    P = np.zeros((rows, 1), order="F")
    P[64, 0] = P[128, 0] = 3.0
    bits = oracle.double_to_fp16(P)
    assert oracle.hgetf2(bits)[0] == 129
    P[:] = 0
    P[300, 0] = P[255, 0] = 5.0   # block 1 lane 44 vs block 0 lane 255 -> lowest block wins
    bits = oracle.double_to_fp16(P)
    assert oracle.hgetf2(bits)[0] == 256
    P[:] = 0                       # all-zero column: pivot = j (block 0, lane 0)
    bits = oracle.double_to_fp16(P)
    assert oracle.hgetf2(bits)[0] == 1


def test_golden_vectors(oracle):
    with open(os.path.join(GOLDEN_DIR, "mpf_golden.json")) as f:
        gold = json.load(f)
    assert len(gold["cases"]) >= 30
    for case in gold["cases"]:
        A = oracle.matgen(case["n"], case["step"], case["func"], case["sparsity"])
        LU, ip = oracle.mpf(A, case["r"])
        assert ip.tolist() == case["ipiv"], (case["n"], case["r"])
        assert hashlib.sha256(np.ascontiguousarray(LU.T).tobytes()).hexdigest() == case["lu_sha256"]
        mx, _ = oracle.check_plu(A, LU, ip)
        assert mx <= 1e-10  # the reference's own pass/fail (benchmark.cpp:97,134)


def test_survey_kats(oracle):
    """Pivot KATs derived independently (throw-away numpy restatement) during the survey, SURVEY 8c."""
    kat = {2: [2, 2], 4: [1, 4, 3, 4], 8: [5, 5, 7, 7, 7, 8, 7, 8],
           16: [6, 7, 9, 9, 5, 6, 12, 16, 9, 15, 15, 14, 13, 16, 15, 16]}
    kat[32] = [16, 11, 32, 17, 5, 25, 19, 32, 13, 22, 26, 26, 19, 30, 17, 24, 17, 19, 24, 22, 21, 26, 29, 25, 30, 27, 31, 31,
               32, 30, 31, 32]
    for n, want in kat.items():
        _, ip = oracle.mpf(oracle.matgen(n), 32)
        assert ip.tolist() == want
    # n = 128: sha256 prefix of the int32 IPIV bytes, panel widths 32 and 128 (IPIV depends on r, D5)
    A = oracle.matgen(128)
    for r, want in ((32, "a39dd2e822bf8366"), (128, "50efa7bd13e35bd5")):
        ip = oracle.mpf(A, r)[1].astype(np.int32)
        assert hashlib.sha256(ip.tobytes()).hexdigest()[:16] == want, r


def test_lapack_agreement_on_tiny_sizes(oracle):
    import scipy.linalg as sl
    for n in (2, 4, 8, 16, 64):   # sizes where fp16 pre-pivoting picks LAPACK's pivots
        A = oracle.matgen(n)
        _, ip = oracle.mpf(A, 32)
        _, piv = sl.lu_factor(A)
        assert np.array_equal(ip, piv + 1)


def test_ipiv_depends_on_panel_width_and_tail_is_skipped(oracle):
    A = oracle.matgen(128)
    assert not np.array_equal(oracle.mpf(A, 32)[1], oracle.mpf(A, 128)[1])
    A = oracle.matgen_skip(33)
    _, ip = oracle.mpf(A, 32)
    assert ip[-1] == 33   # 1x1 tail skipped: identity initialisation survives (MPF.cu:104)


def test_blocked_pieces_compose_to_mpf(oracle):
    """The step operators, chained the way MPF.cu:100-242 chains them, reproduce orc_mpf()."""
    n, r = 200, 64
    A = oracle.matgen_skip(n, skip=77)
    W = A.copy(order="F")
    ipiv = np.arange(1, n + 1, dtype=np.int32)
    for k in range(0, n, r):
        pc, pr = min(r, n - k), n - k
        if pr <= 1:
            break
        piv = oracle.panel_pivots(W, k, pc) + k
        ipiv[k:k + pc] = piv
        oracle.laswp(W, k, pc, piv)
        oracle.dgetf2_npv(W[k:, k:k + pc])
        if k + pc < n:
            oracle.dtrsm_llnu(W[k:k + pc, k:k + pc], W[k:k + pc, k + pc:])
            oracle.dgemm_minus(W[k + pc:, k + pc:], W[k + pc:, k:k + pc], W[k:k + pc, k + pc:])
    LU, ip = oracle.mpf(A, r)
    assert np.array_equal(ip, ipiv) and np.array_equal(LU.view(np.uint64), W.view(np.uint64))


def test_solve_helpers(oracle):
    n = 300
    A = oracle.matgen_skip(n, skip=5)
    LU, ip = oracle.mpf(A, 32)
    b = A @ np.ones(n)
    x = oracle.lu_solve(LU, ip, b)
    rel, _ = oracle.residual(A, x, b)
    assert rel < 1e-12 and np.allclose(x, 1.0, atol=1e-7)

"""GPU parity of the whole hot path: mpf_factor_dev / mpf_factor_host / MPF() against the CPU oracle
(bit-exact IPIV and LU), against the committed golden vectors, against the reference's own acceptance
test (max|A - P L U| <= 1e-10, benchmark.cpp:97-134), plus size-independent properties at larger N."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def _factor_gpu(ctx, A, r, **kw):
    dA = ctx.from_numpy_f(A)
    ipiv, info = ctx.factor(dA, r, **kw)
    ctx.synchronize()
    return ctx.to_numpy_f(dA), ipiv.cpu().numpy(), info


CASES = [(2, 32), (3, 32), (4, 2), (31, 32), (33, 32), (64, 32), (65, 32), (128, 128), (129, 128), (257, 256),
         (512, 32), (513, 128), (1000, 256), (1024, 32), (1024, 256), (2048, 128)]


@pytest.mark.parametrize("n,r", CASES)
def test_factor_matches_oracle_bit_exact(ctx, oracle, n, r):
    A = oracle.matgen_skip(n, skip=4 + n)  # generator-distributed entries, a different stream per n
    LU_o, ip_o = oracle.mpf(A, r)
    LU_g, ip_g, info = _factor_gpu(ctx, A, r)
    assert info == 0
    assert np.array_equal(ip_g, ip_o), f"{int((ip_g != ip_o).sum())} pivots differ, first at {np.argmax(ip_g != ip_o)}"
    assert np.array_equal(LU_g.view(np.uint64), LU_o.view(np.uint64)), "LU bits differ from the oracle"
    mx, fro = oracle.check_plu(A, LU_g, ip_g)
    assert mx <= 1e-10, mx  # the reference's own criterion (benchmark.cpp:97)


@pytest.mark.parametrize("n,r", [(513, 128), (1000, 256), (1024, 64), (2048, 256), (3000, 256), (4096, 256)])
def test_factor_with_window_pivots_bit_exact(mpf, oracle, n, r):
    """The pivot kernel's column-window form (hgetf2_win_kernel), which the schedule only picks for panels of 20 000 rows and more,
    forced on at sizes the oracle finishes quickly, with the chain pipelined at every size: IPIV and all LU bits equal the oracle's."""
    A = oracle.matgen_skip(n, skip=11 + n)
    LU_o, ip_o = oracle.mpf(A, r)
    c = mpf.MPFContext(0)
    try:
        for k, v in (("hp_window", 1), ("chain_pipeline_below", 1 << 30)):
            c.set_option(k, v)
        LU_g, ip_g, info = _factor_gpu(c, A, r)
        assert c.stats().hpanel_timeouts == 0
    finally:
        c.close()
    assert info == 0
    assert np.array_equal(ip_g, ip_o)
    assert np.array_equal(LU_g.view(np.uint64), LU_o.view(np.uint64)), "LU bits differ from the oracle"


def test_golden_vectors(ctx, oracle):
    with open(os.path.join(GOLDEN_DIR, "mpf_golden.json")) as f:
        gold = json.load(f)
    checked = 0
    for case in gold["cases"]:
        n, r = case["n"], case["r"]
        if n > 1024:
            continue
        A = oracle.matgen(n, case["step"], case["func"], case["sparsity"])
        LU_g, ip_g, info = _factor_gpu(ctx, A, r)
        assert ip_g.tolist() == case["ipiv"], (n, r)
        assert hashlib.sha256(np.ascontiguousarray(LU_g.T).tobytes()).hexdigest() == case["lu_sha256"], (n, r)
        checked += 1
    assert checked >= 20


def test_fused_panel_switch(ctx, oracle):
    A = oracle.matgen(256)
    LU_o, ip_o = oracle.mpf(A, 32, fused_panel=True)
    LU_g, ip_g, _ = _factor_gpu(ctx, A, 32, fused_panel=True)
    assert np.array_equal(ip_g, ip_o)
    assert np.array_equal(LU_g.view(np.uint64), LU_o.view(np.uint64))


def test_factor_host_and_ipiv_tail_untouched(ctx, oracle):
    # N = 33, r = 32: the last panel is 1x1 and is skipped (MPF.cu:104): IPIV[N-1] keeps the caller's value
    A = oracle.matgen_skip(33, skip=99)
    LU_o, ip_o = oracle.mpf(A, 32)
    Ah = A.copy(order="F")
    ip = np.arange(1, 34, dtype=np.int32)
    ip[-1] = 777
    ctx.factor_host(Ah, 32, ip)
    assert ip[-1] == 777
    assert np.array_equal(ip[:-1], ip_o[:-1])
    assert np.array_equal(Ah.view(np.uint64), LU_o.view(np.uint64))


def test_MPF_cxx_symbol_drop_in(mpf, oracle):
    """Call the reference's own symbol `void MPF(double*, int, int, int*)` (MPF.h:3) exactly as
    benchmark.cpp:212-222 does: host buffers, identity-initialised IPIV, r = 32."""
    L = mpf.load_library()
    f = getattr(L, mpf.CXX_SYMBOL_MPF)
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    for n in (8, 128, 512):
        A = oracle.matgen(n)
        LU_o, ip_o = oracle.mpf(A, 32)
        Ah = A.copy(order="F")
        ip = np.arange(1, n + 1, dtype=np.int32)
        f(Ah.ctypes.data, n, 32, ip.ctypes.data)
        assert np.array_equal(ip, ip_o)
        assert np.array_equal(Ah.view(np.uint64), LU_o.view(np.uint64))
        assert oracle.check_plu(A, Ah, ip)[0] <= 1e-10


def test_MPF_keeps_its_context_between_calls(mpf, oracle):
    """benchmark.cpp:181-266 calls MPF() once per matrix of a file: the symbol keeps one process-lifetime context (streams,
    workspace, grow-only device copies), so a larger matrix after a smaller one, and a smaller one after that, must all come out
    as the oracle's -- and a context of the C ABI gives its cached buffers back on mpf_trim and still works."""
    L = mpf.load_library()
    f = getattr(L, mpf.CXX_SYMBOL_MPF)
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    for n, r in ((300, 32), (1200, 64), (500, 32), (1200, 64)):
        A = oracle.matgen_skip(n, skip=n % 7)
        LU_o, ip_o = oracle.mpf(A, r)
        Ah = A.copy(order="F")
        ip = np.arange(1, n + 1, dtype=np.int32)
        f(Ah.ctypes.data, n, r, ip.ctypes.data)
        assert np.array_equal(ip, ip_o)
        assert np.array_equal(Ah.view(np.uint64), LU_o.view(np.uint64))
    c2 = mpf.MPFContext(0)
    for n in (900, 400):
        A = oracle.matgen_skip(n, skip=3)
        LU_o, ip_o = oracle.mpf(A, 64)
        Ah = A.copy(order="F")
        ip, _ = c2.factor_host(Ah, 64)
        assert np.array_equal(ip, ip_o) and np.array_equal(Ah.view(np.uint64), LU_o.view(np.uint64))
        c2.trim()
    c2.close()


def test_MPF_retries_on_the_generic_path_when_the_pivot_hand_off_gives_up(mpf, oracle, tmp_path):
    """The reference's MPF() has no failure mode of its own for the pivot search (a cooperative launch, MPF.cu:126-140).  Here the
    LDS pivot kernel's bounded hand-off can give up (-4) when its workgroups are not all resident; the caller's host buffers are
    then untouched and MPF() runs the call again on the generic pivot path.  Forced in a child process with a hand-off limit of
    one poll (the symbol's context reads MPF_HP_SPIN_LIMIT once, when the first call creates it)."""
    import os, subprocess, sys
    n, r = 2048, 128
    A = oracle.matgen_skip(n, skip=2)
    LU_o, ip_o = oracle.mpf(A, r)
    np.save(tmp_path / "A.npy", A)
    code = (
        "import ctypes as C, importlib, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "mpf = importlib.import_module('mixed-precision_lu_factorization_amd')\n"
        "L = mpf.load_library(); f = getattr(L, mpf.CXX_SYMBOL_MPF); f.restype = None\n"
        "f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]\n"
        "A = np.asfortranarray(np.load(%r)); n = A.shape[0]\n"
        "for rep in range(2):\n"
        "    Ah = A.copy(order='F'); ip = np.arange(1, n + 1, dtype=np.int32)\n"
        "    f(Ah.ctypes.data, n, %d, ip.ctypes.data)\n"
        "    np.save(%r + str(rep) + '.npy', Ah); np.save(%r + str(rep) + '.npy', ip)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "A.npy"), r, str(tmp_path / "LU"), str(tmp_path / "ip"))
    env = dict(os.environ, MPF_HP_SPIN_LIMIT="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "retrying on the generic pivot path" in out.stderr, out.stderr[-2000:]
    assert "MPF error" not in out.stdout
    for rep in range(2):
        assert np.array_equal(np.load(str(tmp_path / "ip") + f"{rep}.npy"), ip_o)
        assert np.array_equal(np.load(str(tmp_path / "LU") + f"{rep}.npy").view(np.uint64), LU_o.view(np.uint64))


@pytest.mark.parametrize("n,nb,parts,first,qpct", [(8192 + 72, 128, 2, 40, 60), (8192 + 72, 128, 3, 25, 40), (9000, 64, 1, 50, 100), (10240, 256, 4, 20, 30)])
def test_factor_host_starts_before_the_whole_matrix_is_up(mpf, n, nb, parts, first, qpct):
    """mpf_factor_host (MPF.cu:82 uploads the whole matrix first): only the first columns go up before the factorization starts,
    the rest follows in segments that receive the panels they missed one after the other (LatePlan, factor_lookahead_rm) -- per
    element the same operations in the same order, so the same bits as the device entry point, whatever the cut."""
    c = mpf.MPFContext(0)
    try:
        rng = np.random.default_rng(n * 7 + parts)
        A = np.asfortranarray(rng.standard_normal((n, n)))
        dA = c.from_numpy_f(A)
        ipiv_d, info = c.factor(dA, nb)
        c.synchronize()
        LU_d, ip_d = c.to_numpy_f(dA), ipiv_d.cpu().numpy()
        c.set_option("host_sink_min_n", 0); c.set_option("host_late_min_n", 0)
        c.set_option("host_late_parts", parts); c.set_option("host_first_pct", first); c.set_option("host_late_q_pct", qpct)
        for rep in range(2):
            Ah = A.copy(order="F")
            ip, _ = c.factor_host(Ah, nb)
            assert c.stats().host_late_segments == parts
            assert np.array_equal(ip, ip_d)
            assert np.array_equal(Ah.view(np.uint64), LU_d.view(np.uint64)), int((Ah.view(np.uint64) != LU_d.view(np.uint64)).sum())
    finally:
        c.close()


def test_factor_host_random_shapes_and_plans(mpf):
    """Eight random cases of the host entry point with its overlapped transfers forced on at small sizes -- panel widths that are
    not multiples of 32, ragged last panels, 0-4 late segments, any first-part share, the sink on or off -- against the device
    entry point, bit for bit (tools/host_path_fuzz.py runs the longer sweep: profiles/r05_host_path_fuzz.log)."""
    rng = np.random.default_rng(2025)
    c = mpf.MPFContext(0)
    try:
        c.set_option("host_sink_min_n", 0); c.set_option("host_late_min_n", 0); c.set_option("fp64_rowmajor_min_n", 0)
        for i in range(8):
            nb = int(rng.choice([32, 48, 64, 96, 128]))
            n = nb * int(rng.integers(33, 41)) - int(rng.integers(0, nb))
            parts, first, qpct, sink = int(rng.integers(0, 5)), int(rng.integers(10, 90)), int(rng.integers(20, 300)), int(rng.integers(0, 4) != 0)
            A = np.asfortranarray(rng.standard_normal((n, n)))
            dA = c.from_numpy_f(A)
            ipiv_d, info = c.factor(dA, nb)
            c.synchronize()
            LU_d, ip_d = c.to_numpy_f(dA), ipiv_d.cpu().numpy()
            for k, v in (("host_sink", sink), ("host_late_parts", parts), ("host_first_pct", first), ("host_late_q_pct", qpct)):
                c.set_option(k, v)
            Ah = A.copy(order="F")
            ip, _ = c.factor_host(Ah, nb)
            assert np.array_equal(ip, ip_d), (n, nb, sink, parts, first, qpct)
            assert np.array_equal(Ah.view(np.uint64), LU_d.view(np.uint64)), (n, nb, sink, parts, first, qpct)
    finally:
        c.close()


def test_factor_host_defaults_overlap_both_transfers_from_16k_on(mpf):
    """No option touched: at N = 16384 + 136 (ragged last panel), nb = 256, mpf_factor_host sends the first quarter up, three late
    segments behind it and every block row home while it factors -- and returns the bits of the device entry point; MPF() itself
    (the reference's symbol, process-lifetime context) does the same on the same input."""
    n, nb = 16384 + 136, 256
    c = mpf.MPFContext(0)
    try:
        A = c.to_numpy_f(c.matgen(n))                                    # the reference generator's stream
        dA = c.from_numpy_f(A)
        ipiv_d, info = c.factor(dA, nb)
        c.synchronize()
        LU_d, ip_d = c.to_numpy_f(dA), ipiv_d.cpu().numpy()
        del dA
        Ah = A.copy(order="F")
        ip, _ = c.factor_host(Ah, nb)
        st = c.stats()
        assert st.host_rows_streamed == (n + nb - 1) // nb and st.host_late_segments == 3, (st.host_rows_streamed, st.host_late_segments)
        assert np.array_equal(ip, ip_d) and np.array_equal(Ah.view(np.uint64), LU_d.view(np.uint64))
    finally:
        c.close()
    L = mpf.load_library()
    f = getattr(L, mpf.CXX_SYMBOL_MPF)
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    Ah = A.copy(order="F")
    ip = np.arange(1, n + 1, dtype=np.int32)
    f(Ah.ctypes.data, n, nb, ip.ctypes.data)
    assert np.array_equal(ip, ip_d) and np.array_equal(Ah.view(np.uint64), LU_d.view(np.uint64))


@pytest.mark.parametrize("n,nb", [(1000, 32), (2500, 64), (3000, 256), (8192 + 72, 256), (9000, 128)])
def test_factor_host_sends_block_rows_while_it_factors(mpf, oracle, n, nb):
    """mpf_factor_host / MPF() (MPF.cu:245-247 copies the matrix back after the last panel): here finished block rows leave while
    the factorization runs (rowsink.hip), the left-hand interchanges applied on the way out instead of by the deferred pass on the
    device.  Same bits as the device entry point (which the oracle pins), on both look-ahead schedules (in place below N = 8192,
    row-major working copy from there on), ragged last panels included; and the same again with the sink switched off."""
    import torch
    c = mpf.MPFContext(0)
    try:
        rng = np.random.default_rng(n + nb)
        A = np.asfortranarray(rng.standard_normal((n, n)))          # a matrix that pivots in every column
        dA = c.from_numpy_f(A)
        ipiv_d, info = c.factor(dA, nb)
        c.synchronize()
        LU_d, ip_d = c.to_numpy_f(dA), ipiv_d.cpu().numpy()
        if n <= 1000:
            LU_o, ip_o = oracle.mpf(A, nb)
            assert np.array_equal(ip_d, ip_o) and np.array_equal(LU_d.view(np.uint64), LU_o.view(np.uint64))
        npanels = (n + nb - 1) // nb
        for sink in (1, 0):
            c.set_option("host_sink", sink)
            c.set_option("host_sink_min_n", 0)
            Ah = A.copy(order="F")
            ip, _ = c.factor_host(Ah, nb)
            st = c.stats()
            assert st.host_rows_streamed == (npanels if sink else 0), (sink, st.host_rows_streamed, npanels)
            assert np.array_equal(ip, ip_d)
            assert np.array_equal(Ah.view(np.uint64), LU_d.view(np.uint64)), (sink, int((Ah.view(np.uint64) != LU_d.view(np.uint64)).sum()))
    finally:
        c.close()


def test_factor_host_repeats_itself_when_rows_have_left_and_the_hand_off_gives_up(mpf, oracle, tmp_path):
    """With block rows leaving during the factorization the caller's buffer is partly results when a pivot kernel's hand-off gives
    up (-4): mpf_factor_host keeps the uploaded matrix on the device and repeats the call on the generic pivot path by itself.
    Forced in a child process with a hand-off limit of one poll."""
    import os, subprocess, sys
    n, r = 2048, 128
    A = oracle.matgen_skip(n, skip=2)
    LU_o, ip_o = oracle.mpf(A, r)
    np.save(tmp_path / "A.npy", A)
    code = (
        "import ctypes as C, importlib, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "mpf = importlib.import_module('mixed-precision_lu_factorization_amd')\n"
        "L = mpf.load_library(); f = getattr(L, mpf.CXX_SYMBOL_MPF); f.restype = None\n"
        "f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]\n"
        "A = np.asfortranarray(np.load(%r)); n = A.shape[0]\n"
        "Ah = A.copy(order='F'); ip = np.arange(1, n + 1, dtype=np.int32)\n"
        "f(Ah.ctypes.data, n, %d, ip.ctypes.data)\n"
        "np.save(%r, Ah); np.save(%r, ip)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "A.npy"), r, str(tmp_path / "LU.npy"), str(tmp_path / "ip.npy"))
    env = dict(os.environ, MPF_HP_SPIN_LIMIT="1", MPF_HOST_SINK_MIN_N="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "generic pivot path" in out.stderr, out.stderr[-2000:]
    assert "MPF error" not in out.stdout
    assert np.array_equal(np.load(tmp_path / "ip.npy"), ip_o)
    assert np.array_equal(np.load(tmp_path / "LU.npy").view(np.uint64), LU_o.view(np.uint64))


def test_large_properties(ctx, oracle):
    """N = 4096, nb = 256: too big for an element-by-element oracle run in a unit test, so check
    size-independent properties: panel-0 pivots equal the oracle's, IPIV is a valid swap list, the solve
    built on the factors reproduces a known solution, repeated runs are bit-identical."""
    import torch
    n, r = 4096, 256
    A = oracle.matgen_skip(n)
    dA = ctx.from_numpy_f(A)
    W = dA.clone()
    ipiv, info = ctx.factor(W, r)
    ctx.synchronize()
    ip = ipiv.cpu().numpy()
    assert info == 0
    assert np.array_equal(ip[:r], oracle.panel_pivots(A, 0, r))           # panel 0: exact
    assert np.all(ip >= np.arange(1, n + 1)) and np.all(ip <= n)          # LAPACK-style swap list
    W2 = dA.clone()
    ipiv2, _ = ctx.factor(W2, r)
    ctx.synchronize()
    assert torch.equal(ipiv, ipiv2) and torch.equal(W, W2)                 # deterministic
    xs = torch.ones(n, dtype=torch.float64, device=ctx.device)
    b = dA @ xs
    x, st = ctx.solve_ir(dA, W, ipiv, b, max_iter=5, tol=1e-12)
    assert st.converged == 1 and st.rel_residual <= 1e-12, (st.rel_residual, list(st.history)[:6])
    assert float((x - xs).abs().max()) < 1e-6


def test_solve_ir_small(ctx, oracle):
    import torch
    n, r = 700, 64
    A = oracle.matgen_skip(n, skip=1234)
    dA = ctx.from_numpy_f(A)
    W = dA.clone()
    ipiv, _ = ctx.factor(W, r)
    rng = np.random.default_rng(3)
    b = torch.from_numpy(rng.uniform(-1, 1, n)).to(ctx.device)
    x, st = ctx.solve_ir(dA, W, ipiv, b, max_iter=5, tol=1e-13)
    xo = oracle.lu_solve(ctx.to_numpy_f(W), ipiv.cpu().numpy(), b.cpu().numpy())
    rel, _ = oracle.residual(A, x.cpu().numpy(), b.cpu().numpy())
    assert rel <= 1e-12
    assert np.allclose(x.cpu().numpy(), xo, rtol=1e-6, atol=1e-9)


def test_no_silent_fallback(mpf):
    """The product path is the HIP library: it must be the thing loaded, and it must not know the oracle."""
    L = mpf.load_library()
    assert os.path.samefile(L._name, mpf.LIB_PATH)
    with open("/proc/self/maps") as f:
        maps = f.read()
    assert "libmpf_amd.so" in maps


def _diag_dominant(oracle, n, skip):
    A = oracle.matgen_skip(n, skip=skip)
    A[np.arange(n), np.arange(n)] += A.sum(axis=1)     # generator matrix + diag(rowsum): kappa ~ 2
    return np.asfortranarray(A)


def test_fp16_trailing_mode_with_refinement(ctx, oracle, mpf):
    """Speed mode: fp16-in/fp32-acc MFMA trailing update, fp64 panels; the factors are only fp16-accurate, the
    fp64 refinement sweep restores ||b - A x|| / ||b|| <= 1e-12 (BASELINE.json tolerance)."""
    import torch
    n, r = 2048, 256
    A = _diag_dominant(oracle, n, 31)
    dA = ctx.from_numpy_f(A)
    W = dA.clone()
    ipiv, info = ctx.factor(W, r, trailing=mpf.TRAIL_FP16)
    ctx.synchronize()
    assert info == 0
    ip = ipiv.cpu().numpy()
    assert np.array_equal(ip[:r], oracle.panel_pivots(A, 0, r))       # panel 0 sees no low-precision update yet
    LU = ctx.to_numpy_f(W)
    mx, fro = oracle.check_plu(A, LU, ip)
    assert fro < 5e-3, fro                                            # fp16-level factorization ...
    xs = torch.ones(n, dtype=torch.float64, device=ctx.device)
    b = dA @ xs
    x, st = ctx.solve_ir(dA, W, ipiv, b, max_iter=10, tol=1e-12)
    assert st.converged == 1 and st.rel_residual <= 1e-12, list(st.history)[:8]   # ... refined to fp64
    assert 1 <= st.iterations <= 6
    assert float((x - xs).abs().max()) < 1e-9


def test_fp16_mode_matches_its_own_single_stream_schedule(ctx, oracle, mpf):
    """Look-ahead must not change results in the fp16 mode either."""
    import torch
    A = _diag_dominant(oracle, 1500, 77)
    dA = ctx.from_numpy_f(A)
    W1, W2 = dA.clone(), dA.clone()
    p1, _ = ctx.factor(W1, 128, trailing=mpf.TRAIL_FP16)
    p2, _ = ctx.factor(W2, 128, trailing=mpf.TRAIL_FP16, no_lookahead=True)
    ctx.synchronize()
    assert torch.equal(p1, p2) and torch.equal(W1, W2)


def test_lookahead_equals_single_stream_fp64(ctx, oracle):
    import torch
    A = oracle.matgen_skip(1800, skip=5)
    dA = ctx.from_numpy_f(A)
    W1, W2 = dA.clone(), dA.clone()
    p1, _ = ctx.factor(W1, 256)
    p2, _ = ctx.factor(W2, 256, no_lookahead=True)
    ctx.synchronize()
    assert ctx.stats().lookahead == 0
    assert torch.equal(p1, p2) and torch.equal(W1, W2)


def test_gesv_picks_the_path(ctx, oracle, mpf):
    """mpf_gesv: the diagonally dominant input is solved on the fp16-trailing path; the raw generator matrix
    (kappa ~ 1e6 at this size: plain refinement on fp16-accurate factors stalls) and a kappa ~ 1e8 row-scaled
    matrix (BASELINE config 5; its U12 overflows fp16) fall back to the fp64 trailing update.  All reach 1e-12."""
    import torch
    n, nb = 8192, 128
    g = torch.Generator(device=ctx.device); g.manual_seed(3)
    G = (torch.randint(0, 100, (n, n), generator=g, device=ctx.device, dtype=torch.int32).to(torch.float64) / 10.0).t()
    xs = torch.ones(n, dtype=torch.float64, device=ctx.device)
    idx = torch.arange(n, device=ctx.device)
    work = ctx.colmajor(n, n)
    # (a) diagonally dominant
    Ad = G.clone(); Ad[idx, idx] += G.sum(dim=1)
    x, st, _, _ = ctx.gesv(Ad, Ad @ xs, nb, work=work)
    assert st.path == 1 and st.ir_final.converged == 1 and st.ir_final.rel_residual <= 1e-12
    assert float((x - xs).abs().max()) < 1e-8
    # (b) raw generator distribution
    x, st, _, _ = ctx.gesv(G, G @ xs, nb, work=work)
    assert st.path == 2 and st.ir_fp16.converged == 0 and st.ir_final.converged == 1 and st.ir_final.rel_residual <= 1e-12
    # (c) kappa ~ 1e8: rows of the dominant matrix scaled by logspace(0, 8)
    D = torch.logspace(0, 8, n, dtype=torch.float64, device=ctx.device)
    Ak = Ad * D[:, None]
    Ak = Ak.t().contiguous().t()
    b = Ak @ xs
    x, st, _, _ = ctx.gesv(Ak, b, nb, work=work)
    assert st.ir_final.converged == 1 and st.ir_final.rel_residual <= 1e-12
    assert st.path in (1, 2)
    print("kappa~1e8 case solved on path", st.path, "fp16 history", list(st.ir_fp16.history)[:4])
    # (d) the raw generator matrix again, split operands (try_fp16 = 2): stays on the low-precision path
    x, st, _, _ = ctx.gesv(G, G @ xs, nb, work=work, try_fp16=2)
    assert st.path == 1 and st.ir_final.converged == 1 and st.ir_final.rel_residual <= 1e-12
    assert st.ir_final.iterations <= 6, list(st.ir_final.history)[:8]
    assert float((x - xs).abs().max()) < 1e-6


def test_fp16x3_split_mode_factors_are_fp32_class(ctx, oracle, mpf):
    """Split mode (hi + 2^-11 lo operands, three MFMA products): the factorization error drops by ~three orders of
    magnitude against the plain fp16 mode on the same generator-distribution matrix, pivots of panel 0 are the
    contract's, look-ahead does not change results, and refinement reaches 1e-12 in a few sweeps."""
    import torch
    n, r = 2048, 128
    A = oracle.matgen_skip(n, skip=2)
    dA = ctx.from_numpy_f(A)
    W3, W3b, W1 = dA.clone(), dA.clone(), dA.clone()
    p3, info = ctx.factor(W3, r, trailing=mpf.TRAIL_FP16X3)
    p3b, _ = ctx.factor(W3b, r, trailing=mpf.TRAIL_FP16X3, no_lookahead=True)
    p1, _ = ctx.factor(W1, r, trailing=mpf.TRAIL_FP16)
    ctx.synchronize()
    assert info == 0 and torch.equal(p3, p3b) and torch.equal(W3, W3b)
    assert np.array_equal(p3.cpu().numpy()[:r], oracle.panel_pivots(A, 0, r))
    _, fro3 = oracle.check_plu(A, ctx.to_numpy_f(W3), p3.cpu().numpy())
    _, fro1 = oracle.check_plu(A, ctx.to_numpy_f(W1), p1.cpu().numpy())
    assert fro3 < 1e-5 and fro3 * 500 < fro1, (fro3, fro1)
    xs = torch.ones(n, dtype=torch.float64, device=ctx.device)
    x, st = ctx.solve_ir(dA, W3, p3, dA @ xs, max_iter=10, tol=1e-12)
    assert st.converged == 1 and st.iterations <= 4, list(st.history)[:8]


@pytest.mark.parametrize("n,r", [(1, 32), (2, 1), (5, 1), (7, 3), (40, 256), (100, 17), (300, 255), (1030, 96)])
def test_odd_shapes_match_oracle(ctx, oracle, n, r):
    """Degenerate and ragged shapes: N = 1 (nothing to do), panel width 1, width > N, widths that do not divide N."""
    A = oracle.matgen_skip(n, skip=3 * n + r)
    LU_o, ip_o = oracle.mpf(A, r)
    LU_g, ip_g, info = _factor_gpu(ctx, A, r)
    assert np.array_equal(ip_g, ip_o)
    assert np.array_equal(LU_g.view(np.uint64), LU_o.view(np.uint64))


def test_leading_dimension_larger_than_n(ctx, oracle):
    n, r, lda = 500, 64, 777
    A = oracle.matgen_skip(n, skip=11)
    big = np.asfortranarray(np.full((lda, n + 2), 123.0))
    big[:n, :n] = A
    d = ctx.from_numpy_f(big)
    ipiv, info = ctx.factor(d[:n, :n], r)
    ctx.synchronize()
    out = ctx.to_numpy_f(d)
    LU_o, ip_o = oracle.mpf(A, r)
    assert np.array_equal(ipiv.cpu().numpy(), ip_o)
    assert np.array_equal(np.asfortranarray(out[:n, :n]).view(np.uint64), LU_o.view(np.uint64))
    assert np.all(out[n:, :] == 123.0) and np.all(out[:, n:] == 123.0)   # nothing outside the matrix is touched


@pytest.mark.parametrize("n,r,pad", [(2500, 128, 88), (2501, 100, 0), (1999, 96, 5)])
def test_leading_dimension_larger_than_n_on_the_rowmajor_two_lane_schedule(mpf, oracle, n, r, pad):
    """The same through the fp64 mode's row-major working copy with the two update lanes forced on at a small size (the copy is
    N x N whatever lda is; the panels, the U rows and the final L go back into the caller's padded matrix), look-ahead pipelined
    below 1024 columns: bit-exact against the oracle, padding untouched."""
    lda = n + pad      # (ragged sizes too: panel widths that are no multiple of 32 take the unpipelined chain, odd edges the guarded update kernel)
    A = oracle.matgen_skip(n, skip=23)
    big = np.asfortranarray(np.full((lda, n + 3), -7.5))
    big[:n, :n] = A
    c = mpf.MPFContext(0, options={"fp64_rowmajor_min_n": 0, "fp64_two_lanes": 256, "chain_pipeline_below": 1024})
    try:
        d = c.from_numpy_f(big)
        ipiv, info = c.factor(d[:n, :n], r)
        c.synchronize()
        out = c.to_numpy_f(d)
    finally:
        c.close()
    LU_o, ip_o = oracle.mpf(A, r)
    assert info == 0 and np.array_equal(ipiv.cpu().numpy(), ip_o)
    assert np.array_equal(np.asfortranarray(out[:n, :n]).view(np.uint64), LU_o.view(np.uint64))
    assert np.all(out[n:, :] == -7.5) and np.all(out[:, n:] == -7.5)


def test_singular_matrix_reports_info_and_does_not_hang(ctx, oracle):
    """A zero column: the reference divides by zero silently (hgetf2_kernel.cu:108, dgetf2_native_npv.cu:24);
    this build still returns, and reports the first zero fp64 pivot LAPACK-style."""
    n = 300
    A = oracle.matgen_skip(n, skip=5)
    A[:, 40] = 0.0
    dA = ctx.from_numpy_f(A)
    ipiv, info = ctx.factor(dA, 64)
    ctx.synchronize()
    assert info == 41
    assert ctx.stats().hpanel_timeouts == 0


def test_random_sizes_fuzz(ctx, oracle):
    rng = np.random.default_rng(2024)
    for _ in range(12):
        n = int(rng.integers(2, 700))
        r = int(rng.integers(1, 257))
        A = oracle.matgen_skip(n, skip=int(rng.integers(0, 5000)))
        LU_o, ip_o = oracle.mpf(A, r)
        LU_g, ip_g, info = _factor_gpu(ctx, A, r)
        assert np.array_equal(ip_g, ip_o), (n, r)
        assert np.array_equal(LU_g.view(np.uint64), LU_o.view(np.uint64)), (n, r)


@pytest.mark.parametrize("n,r", [(2048, 128), (1536, 96), (2048, 256), (1100, 64), (777, 32), (640, 128), (513, 128),
                                 (1, 32), (2, 1), (5, 1), (7, 3), (40, 256), (100, 17), (300, 255)])
def test_fp16_modes_two_level_schedule_on_generator_matrices(ctx, oracle, mpf, n, r):
    """The fp16 trailing modes run a two-level schedule (super-panels of 4 panels, one K = 4 r update of the matrix
    right of each, worked off in pieces under the next super-panel's chains) whenever N > 4 r.  Generator matrices
    pivot in nearly every column, so a misplaced interchange shows up as an O(1) factorization error; sizes cover a
    ragged last panel, a ragged last super-panel and N just above / below the switch-over.  The single-stream run must
    give the same bits."""
    import torch
    A = oracle.matgen_skip(n, skip=3)
    dA = ctx.from_numpy_f(A)
    for mode, tol in ((mpf.TRAIL_FP16, 2e-2), (mpf.TRAIL_FP16X3, 1e-5)):
        W, W2 = dA.clone(), dA.clone()
        p, info = ctx.factor(W, r, trailing=mode)
        p2, _ = ctx.factor(W2, r, trailing=mode, no_lookahead=True)
        ctx.synchronize()
        assert info == 0
        assert torch.equal(p, p2) and torch.equal(W, W2)
        _, fro = oracle.check_plu(A, ctx.to_numpy_f(W), p.cpu().numpy())
        assert fro < tol, (mode, fro)


def test_n65536_on_one_gpu(ctx, mpf):
    """BASELINE config 4's matrix size on a single GPU (34 GB, 64-bit offsets everywhere, 256 pivot workgroups = one per
    CU): the fp64-mode factors solve A x = b to 1e-12 without refinement help beyond one sweep."""
    import torch
    n = 65536
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 90e9:
        pytest.skip("needs ~75 GB of free HBM")
    g = torch.Generator(device=ctx.device); g.manual_seed(11)
    A = torch.empty((n, n), dtype=torch.float64, device=ctx.device).t()
    for c0 in range(0, n, 8192):
        A[:, c0:c0 + 8192] = (torch.randint(0, 100, (8192, n), generator=g, device=ctx.device, dtype=torch.int32).to(torch.float64) / 10.0).t()
    xs = torch.ones(n, dtype=torch.float64, device=ctx.device)
    b = A @ xs
    W = A.clone()
    ipiv, info = ctx.factor(W, 256)
    st = ctx.stats()
    assert info == 0 and st.hpanel_timeouts == 0
    x, ir = ctx.solve_ir(A, W, ipiv, b, max_iter=2, tol=1e-12)
    assert ir.converged == 1 and ir.rel_residual <= 1e-12 and ir.iterations <= 1, list(ir.history)[:3]
    assert float((x - xs).abs().max()) < 1e-6
    del A, W
    torch.cuda.empty_cache()


def test_n61440_fp16_mode_on_one_gpu(ctx, mpf):
    """A panel of 240 x 256 rows would take 240 of the 256 CUs in the full-slab form of the pivot kernel, and the 64 workgroups of
    the pipelined chain's gated interchange kernel -- waiting for its progress -- kept its last workgroups from ever becoming
    resident (-4 after the bounded spin, from N = 53 248 up).  Panels that leave fewer than 72 CUs free run the column-window
    form (two per CU) in the fp16 modes too."""
    import torch
    n = 61440
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 110e9:
        pytest.skip("needs ~95 GB of free HBM")
    A = ctx.matgen(n)
    idx = torch.arange(n, device=ctx.device)
    A[idx, idx] += A.sum(dim=1)
    xs = torch.ones(n, dtype=torch.float64, device=ctx.device)
    b = A @ xs
    W = A.clone()
    ipiv, info = ctx.factor(W, 256, trailing=mpf.TRAIL_FP16)
    st = ctx.stats()
    assert info == 0 and st.hpanel_timeouts == 0
    x, ir = ctx.solve_ir(A, W, ipiv, b, max_iter=6, tol=1e-12)
    assert ir.converged == 1 and ir.rel_residual <= 1e-12, list(ir.history)[:6]
    del A, W
    torch.cuda.empty_cache()


@pytest.mark.parametrize("side", ["last_full_slab", "first_window"])
@pytest.mark.parametrize("where", ["one_gpu", "dist_world1_loop"])
def test_fp16_mode_on_both_sides_of_the_pivot_kernels_room(mpf, side, where):
    """The pipelined chain's gated interchange kernel waits for the pivot kernel while its workgroups sit on CUs; the pivot kernel's
    workgroups must all be resident at once.  The room each form has beside them is DERIVED (kernel footprints + occupancy API,
    mpf_hgetf2_capacity_rows); a factorization whose first panel is the tallest the full-slab form takes, and one a block taller
    (column-window form), both finish with no give-up -- on one GPU and with one rank in the distributed loop."""
    import torch
    ctx = mpf.MPFContext(0, options={"dist_world1_loop": 1} if where == "dist_world1_loop" else None)
    try:
        cap_full = ctx.hgetf2_capacity_rows(-256, 1)
        cap_any = ctx.hgetf2_capacity_rows(-256, 0)
        assert 0 < cap_full <= cap_any and ctx.hgetf2_capacity_rows(0, 1) >= cap_full
        n = cap_full if side == "last_full_slab" else cap_full + 256
        if n > cap_any:
            pytest.skip("no panel on this device is too tall for the full-slab form and fits the window form")
        free_b, _ = torch.cuda.mem_get_info()
        if free_b < 3.3 * 8 * n * n:
            pytest.skip("not enough free HBM")
        A = ctx.matgen(n)
        idx = torch.arange(n, device=ctx.device)
        A[idx, idx] += A.sum(dim=1)
        xs = torch.ones(n, dtype=torch.float64, device=ctx.device)
        b = A @ xs
        W = A.clone()
        if where == "one_gpu":
            ipiv, info = ctx.factor(W, 256, trailing=mpf.TRAIL_FP16)
        else:
            ipiv, info = ctx.factor_dist(W, n, 256, mpf.MpfDist(rank=0, world=1), trailing=mpf.TRAIL_FP16)
        st = ctx.stats()
        assert info == 0 and st.hpanel_timeouts == 0
        x, ir = ctx.solve_ir(A, W, ipiv, b, max_iter=6, tol=1e-12)
        assert ir.converged == 1 and ir.rel_residual <= 1e-12, list(ir.history)[:6]
        del A, W
    finally:
        ctx.close()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("n,r,sb", [(2048, 128, 2), (2048, 128, 4), (1536, 96, 3), (1100, 64, 8)])
def test_fp64_two_level_schedule_is_bit_identical(ctx, oracle, n, r, sb):
    """mpf_opts.superpanel in the fp64 mode: sb panels per super-panel, one K = sb*r update of the matrix right of it.
    Every element still receives its fma chain with k ascending across the panels (contract C5), the U block-row the
    same TRSM on the same data (C4): IPIV and all N^2 fp64 values equal the one-level schedule's, which equals the oracle."""
    import torch
    A = oracle.matgen_skip(n, skip=1)
    dA = ctx.from_numpy_f(A)
    W1, W2, W3 = dA.clone(), dA.clone(), dA.clone()
    p1, i1 = ctx.factor(W1, r)
    p2, i2 = ctx.factor(W2, r, superpanel=sb)
    p3, i3 = ctx.factor(W3, r, superpanel=sb, no_lookahead=True)
    ctx.synchronize()
    assert i1 == i2 == i3 == 0
    assert torch.equal(p1, p2) and torch.equal(W1, W2)
    assert torch.equal(p1, p3) and torch.equal(W1, W3)


def test_config1_n8192_fp16_panel128_three_step_ir(ctx, mpf):
    """BASELINE.json configs[1]: N=8192, nb=128, fp16-MFMA trailing update, fp64 refinement: <= 3 sweeps to 1e-12 on the
    IR-friendly input (generator distribution + diag(rowsum))."""
    import torch
    n, nb = 8192, 128
    g = torch.Generator(device=ctx.device); g.manual_seed(8192)
    A = (torch.randint(0, 100, (n, n), generator=g, device=ctx.device, dtype=torch.int32).to(torch.float64) / 10.0).t()
    idx = torch.arange(n, device=ctx.device)
    A[idx, idx] += A.sum(dim=1)
    xs = torch.ones(n, dtype=torch.float64, device=ctx.device)
    b = A @ xs
    W = A.clone()
    ipiv, info = ctx.factor(W, nb, trailing=mpf.TRAIL_FP16)
    x, st = ctx.solve_ir(A, W, ipiv, b, max_iter=3, tol=1e-12)
    assert info == 0 and st.converged == 1 and st.iterations <= 3 and st.rel_residual <= 1e-12, list(st.history)[:5]
    assert float((x - xs).abs().max()) < 1e-9


def test_device_side_plu_check_agrees_with_the_oracle(ctx, oracle):
    """mpf_check_plu_dev (benchmark.cpp:106-144 with L * U on the MFMA GEMM) against the oracle's host restatement."""
    for n, r in ((5, 2), (257, 64), (1000, 128)):
        A = oracle.matgen_skip(n, skip=n)
        dA = ctx.from_numpy_f(A)
        W = dA.clone()
        ipiv, _ = ctx.factor(W, r)
        mx_o, fro_o = oracle.check_plu(A, ctx.to_numpy_f(W), ipiv.cpu().numpy())
        mx, fro = ctx.check_plu(dA, W, ipiv)
        # same quantity, different summation order inside L * U (MFMA fma chains vs the host loop): equal to rounding noise
        assert mx <= 1e-10 and abs(mx - mx_o) <= 1e-12 + 0.5 * mx_o and fro_o / 3 - 1e-17 <= fro <= 3 * fro_o + 1e-17, (mx, mx_o, fro, fro_o)
    # a wrong pivot must be noticed
    bad = ipiv.clone()
    bad[3], bad[4] = ipiv[4], ipiv[3] + 1
    assert ctx.check_plu(dA, W, bad)[0] > 1e-3


def test_device_report_has_the_fields_the_design_depends_on(mpf):
    """HIP analogue of check_cooperative_groups.cu:4-48 on the box: CU count, LDS sizes, cooperative launch, RCCL version."""
    n, txt = mpf.device_report()
    print(txt)
    assert n >= 1 and "HIP devices:" in txt
    assert "gfx950" in txt and "CUs 256" in txt and "cooperativeLaunch 1" in txt
    assert "LDS/CU 163840 B" in txt and "warpSize 64" in txt
    assert "RCCL version 2." in txt and "pivot-kernel rows/launch 65536" in txt


def test_multiple_right_hand_sides(ctx, oracle):
    """mpf_solve_ir_nrhs: several right-hand sides on one set of factors, each refined to the tolerance."""
    import torch
    n, r, nrhs = 900, 128, 5
    A = oracle.matgen_skip(n, skip=21)
    dA = ctx.from_numpy_f(A)
    W = dA.clone()
    ipiv, _ = ctx.factor(W, r)
    rng = np.random.default_rng(4)
    B = np.asfortranarray(rng.uniform(-1, 1, (n, nrhs)))
    X, sts = ctx.solve_ir_nrhs(dA, W, ipiv, ctx.from_numpy_f(B), max_iter=5, tol=1e-13)
    Xh = ctx.to_numpy_f(X)
    LU, ip = ctx.to_numpy_f(W), ipiv.cpu().numpy()
    for j in range(nrhs):
        assert sts[j].converged == 1 and sts[j].rel_residual <= 1e-13
        rel, _ = oracle.residual(A, np.ascontiguousarray(Xh[:, j]), np.ascontiguousarray(B[:, j]))
        assert rel <= 1e-12
        assert np.allclose(Xh[:, j], oracle.lu_solve(LU, ip, np.ascontiguousarray(B[:, j])), rtol=1e-6, atol=1e-9)


def test_gmres_ir_converges_where_plain_refinement_does_not(ctx, mpf):
    """The generator's own matrix at N = 8192 with plain fp16 operands (kappa ~ 1e6, factors accurate to ~1e-2): classical
    refinement stalls or diverges; GMRES preconditioned with the same factors reaches 1e-12.  mpf_gesv(try_fp16 = 3) then stays
    on the low-precision factorization instead of paying for the fp64 one."""
    import torch
    n, nb = 8192, 128
    A = ctx.matgen(n)
    xs = torch.ones(n, dtype=torch.float64, device=ctx.device)
    b = A @ xs
    W = A.clone()
    ipiv, info = ctx.factor(W, nb, trailing=mpf.TRAIL_FP16)
    assert info == 0
    x, st = ctx.solve_ir(A, W, ipiv, b, max_iter=10, tol=1e-12)
    assert st.converged == 0, list(st.history)[:6]
    x, gm = ctx.solve_gmres_ir(A, W, ipiv, b, max_outer=10, restart=40, tol=1e-12)
    print("GMRES-IR: outer", gm.outer_iterations, "inner", gm.inner_iterations, "history", [f"{v:.1e}" for v in list(gm.history)[:gm.outer_iterations + 1]], f"{gm.ms_total:.0f} ms")
    assert gm.converged == 1 and gm.rel_residual <= 1e-12, list(gm.history)[:8]
    assert float((x - xs).abs().max()) < 1e-5
    # mpf_gesv gives GMRES-IR the time an fp64 refactorization would take, then falls back to it: either way the answer is refined
    x, gs, _, _ = ctx.gesv(A, b, nb, try_fp16=3)
    print("gesv(try_fp16 = 3) took path", gs.path, f"{gs.ms_total:.0f} ms (fp16 factor {gs.ms_factor_fp16:.0f}, fp16-side solves {gs.ms_ir_fp16:.0f}, fp64 factor {gs.ms_factor_fp64:.0f})")
    assert gs.path in (2, 3) and gs.ir_final.converged == 1 and gs.ir_final.rel_residual <= 1e-12
    if gs.path == 2:   # the budget: GMRES-IR stopped within ~2 x the fp16 factorization's time
        assert gs.ms_ir_fp16 <= 4.0 * max(gs.ms_factor_fp16, 1.0) + 50.0


def _switch_results(mpf, options, probe=False):
    """fp64 factors' digest at two shapes + refinement outcome of the two fp16 modes, on a context with the given options"""
    import hashlib
    import torch
    ctx = mpf.MPFContext(0, options=options, probe=probe)
    out = []
    try:
        for n, nb, mode in ((4096, 256, 0), (3000, 128, 0), (4096, 256, 1), (4096, 256, 2)):
            A = ctx.matgen(n)
            if mode == 1:      # the plain fp16 mode needs a well-conditioned input to refine
                idx = torch.arange(n, device=ctx.device)
                A[idx, idx] += A.sum(dim=1)
            W = A.clone()
            ipiv, info = ctx.factor(W, nb, trailing=mode)
            assert info == 0 and ctx.stats().hpanel_timeouts == 0
            if mode == 0:
                out.append(hashlib.sha256(ipiv.cpu().numpy().tobytes() + W.t().contiguous().cpu().numpy().tobytes()).hexdigest())
            else:
                b = A @ torch.ones(n, dtype=torch.float64, device=ctx.device)
                x, st = ctx.solve_ir(A, W, ipiv, b, max_iter=12, tol=1e-12)
                out.append((int(st.converged), int(st.iterations)))
    finally:
        ctx.close()
    return out


def test_every_ab_switch_gives_the_same_factors(mpf):
    """The library's A/B switches select other kernels / schedules for the same arithmetic: the fp64 factors (IPIV and all LU
    bits) must not depend on any of them, and the fp16 modes must still refine to 1e-12.  The switches are per-context
    options (mpf_set_option): every setting gets a context of its own in this one process."""
    base = _switch_results(mpf, {})
    assert base[2][0] == 1 and base[3][0] == 1, base
    for sw in ({"chain_pipeline": 0}, {"dgemm_dma": 0}, {"dpanel_fused_form": 0}, {"lazy_gather": 0}, {"no_lookahead": 1},
               {"safe_pivots": 1}, {"superpanel_fp64": 2}, {"superpanel_fp64": 4}, {"superpanel_fp64": 3}, {"fp16_work32": 0}, {"chain_pipeline_below": 0}, {"superpanel_fp16": 4},
               {"trsm_laswp_fused": 0}, {"safe_pivots": 1, "generic_fused": 0}, {"fp64_rowmajor": 0}, {"fp64_rowmajor_min_n": 0},
               # the row-major schedule's two update lanes at these small sizes: always (chain never pipelined), with the hand-over to one
               # lane + pipelined chain half way, and switched off
               {"fp64_rowmajor_min_n": 0, "fp64_two_lanes": 512, "chain_pipeline_below": 0},
               {"fp64_rowmajor_min_n": 0, "fp64_two_lanes": 256, "chain_pipeline_below": 2048},
               {"fp64_rowmajor_min_n": 0, "fp64_two_lanes": 0},
               # pivot kernel in its column-window form / its full-slab form at every size
               {"hp_window": 1}, {"hp_window": 0}, {"hp_window": 1, "fp64_rowmajor_min_n": 0},
               # the pipelined chain's gate as a stream wait on the progress word (no CU held) instead of a spinning kernel; other super-panel widths
               {"gate_wait_value": 1}, {"gate_wait_value": 1, "chain_pipeline_below": 1 << 30}, {"superpanel_fp16": 6}, {"hgemm_mfma16": 0},
               # the pivot kernel's single-XCD form for short panels: never / in every schedule (default: fp16 modes only)
               {"hp_local_xcd": 0}, {"hp_local_xcd": 2}, {"hp_local_xcd": 2, "fp64_rowmajor_min_n": 0}, {"hp_half_slabs": 0}):
        got = _switch_results(mpf, sw)
        assert got[0] == base[0] and got[1] == base[1], (sw, "fp64 factors differ from the default context's")
        assert got[2][0] == 1 and got[3][0] == 1, (sw, got)
        assert got[2][1] <= 3 and got[3][1] <= 8, (sw, got)
    # the probe library's extra variants (four-wave update kernel) compute the same bits
    got = _switch_results(mpf, {"dgemm_w8": 0}, probe=True)
    assert got[0] == base[0] and got[1] == base[1]


def test_options_are_per_context_and_thread_safe(mpf, oracle):
    """Two contexts on two host threads with different options run at the same time: each keeps its own settings (nothing
    is process-global or read lazily from the environment) and each gets the oracle's bits."""
    import threading
    import torch
    n, nb = 2048, 128
    A = oracle.matgen_skip(n, skip=77)
    LU_o, ip_o = oracle.mpf(A, nb)
    res = {}

    def run(tag, options):
        s = torch.cuda.Stream(device=0)
        ctx = mpf.MPFContext(0, stream=s, options=options)
        try:
            with torch.cuda.stream(s):
                for rep in range(3):
                    W = ctx.from_numpy_f(A)
                    s.synchronize()
                    ipiv, info = ctx.factor(W, nb)
                    st = ctx.stats()
                    res[(tag, rep)] = (ipiv.cpu().numpy(), ctx.to_numpy_f(W), st.superpanel, st.lookahead, {k: ctx.get_option(k) for k in options})
        finally:
            ctx.close()

    oa = {"superpanel_fp64": 2, "chain_pipeline": 0, "lazy_gather": 0}
    ob = {"superpanel_fp64": 1, "no_lookahead": 1, "dgemm_dma": 0}
    ta = threading.Thread(target=run, args=("a", oa)); tb = threading.Thread(target=run, args=("b", ob))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert len(res) == 6
    for (tag, rep), (ip, LU, sb, la, seen) in res.items():
        assert np.array_equal(ip, ip_o) and np.array_equal(LU.view(np.uint64), LU_o.view(np.uint64)), (tag, rep)
        assert seen == (oa if tag == "a" else ob)
        assert (sb, la) == ((2, 1) if tag == "a" else (1, 0)), (tag, sb, la)


def test_unknown_option_is_an_error(ctx, mpf):
    with pytest.raises(mpf.MPFError):
        ctx.set_option("no_such_switch", 1)
    assert "superpanel_fp16" in mpf.option_names() and "hp_gate_ticks" in mpf.option_names()


def test_event_timers_option_levels(mpf):
    """Option event_timers: 1 (default) brackets the trailing-update launches only -- what the roofline needs, measured in every step --
    2 every timed region of mpf_stats, 0 nothing; the factors do not depend on it."""
    import hashlib
    import torch
    res = {}
    for lvl in (None, 2, 0):
        c = mpf.MPFContext(0, options=None if lvl is None else {"event_timers": lvl})
        try:
            assert c.get_option("event_timers") == (1 if lvl is None else lvl)
            for mode in (0, 1):
                A = c.matgen(8192)
                if mode:
                    idx = torch.arange(8192, device=c.device)
                    A[idx, idx] += A.sum(dim=1)
                ipiv, info = c.factor(A, 256, trailing=mode)
                st = c.stats()
                assert info == 0 and st.ms_total > 0
                h = hashlib.sha256(ipiv.cpu().numpy().tobytes() + A.t().contiguous().cpu().numpy().tobytes()).hexdigest()
                assert res.setdefault(mode, h) == h          # same factors whatever is timed
                small = (st.ms_hpanel, st.ms_dpanel, st.ms_trsm, st.ms_laswp)
                if lvl == 2:
                    assert st.ms_gemm > 0 and all(v > 0 for v in small)
                elif lvl is None:
                    assert st.ms_gemm > 0 and all(v == 0 for v in small)
                    if mode:
                        assert st.ms_gemm_big > 0 and st.gemm_big_launches > 0
                else:
                    assert st.ms_gemm == 0 and all(v == 0 for v in small)
        finally:
            c.close()


def test_gate_expiry_is_a_failure_not_a_silent_pass(mpf):
    """a gate that stops waiting lets the interchange and the fp64 panel through on pivots that are not final.
    That must surface as a failure (-4, like a give-up inside the pivot kernel), never as rc = 0.
    (1) Deterministic: one gate launch with nothing behind it (probe library's mpf_debug_gate, same kernel) and a zero
    time-out must flag the context's time-out counter -- the counter mpf_factor_dev / mpf_factor_dist turn into -4
    (test_spin_limit_gives_a_clean_error_not_a_hang covers that half).  With a generous time-out and progress never coming it
    flags too, after waiting; a later gate then leaves at once.
    (2) End to end: a factorization whose gates have a zero time-out either completes correctly (every gate found its pivots
    already final) or fails with -4 -- it never returns wrong factors silently."""
    import ctypes as C
    import time
    import torch
    pc = mpf.MPFContext(0, probe=True, options={"hp_gate_ticks": 0})
    pc.L.mpf_debug_gate.argtypes = [C.c_void_p, C.c_int]
    pc.L.mpf_debug_gate.restype = C.c_int
    assert pc.L.mpf_debug_gate(pc.h, 32) == 1
    pc.set_option("hp_gate_ticks", 2_000_000)      # 20 ms of the 100 MHz clock
    t0 = time.perf_counter()
    assert pc.L.mpf_debug_gate(pc.h, 32) == 1
    assert time.perf_counter() - t0 >= 0.015
    pc.close()
    c2 = mpf.MPFContext(0, options={"hp_gate_ticks": 0})
    n, nb = 8192, 256
    A = c2.matgen(n)
    idx = torch.arange(n, device=c2.device)
    A[idx, idx] += A.sum(dim=1)
    W = A.clone()
    try:
        ipiv, info = c2.factor(W, nb, trailing=mpf.TRAIL_FP16)     # the fp16 modes pipeline every panel's chain behind gates
        assert c2.stats().hpanel_timeouts == 0
        b = A @ torch.ones(n, dtype=torch.float64, device=c2.device)
        x, st = c2.solve_ir(A, W, ipiv, b, max_iter=5, tol=1e-12)
        assert st.converged == 1
    except mpf.MPFError as e:
        assert "(-4)" in str(e) and c2.stats().hpanel_timeouts > 0
    c2.close()
    torch.cuda.synchronize()


def test_fp32_working_copy_hands_back_the_right_windows(mpf):
    """The two-level schedule keeps the matrix right of the inner region in an fp32 copy and hands windows back to fp64 as
    they join the inner region (and block rows before their TRSM).  A window converted at the wrong time or place misses a
    K = sb * nb update or a row interchange: O(1) errors in the factors.  On a diagonally dominant matrix (no row moves, so
    both runs pivot identically) the factors with the copy must equal the factors of the in-place fp64 run (option
    fp16_work32 = 0) up to the copy's rounding: every element within 2^-20 of the largest factor element, at several
    super-panel widths and with real row interchanges switched on through a second, row-permuted input."""
    import torch
    n, nb = 4096, 128
    for perm in (False, True):
        ref = None
        for opts in ({"fp16_work32": 0, "superpanel_fp16": 4}, {"superpanel_fp16": 2}, {"superpanel_fp16": 3}, {"superpanel_fp16": 4},
                     {"superpanel_fp16": 8}):
            c2 = mpf.MPFContext(0, options=opts)
            A = c2.matgen(n)
            idx = torch.arange(n, device=c2.device)
            A[idx, idx] += A.sum(dim=1)
            if perm:     # rows reversed inside every 16-row group: every panel must interchange, the pivots stay unambiguous
                g = (idx // 16) * 16 + (15 - idx % 16)
                A = A[g, :].t().contiguous().t()
            W = A.clone()
            ipiv, info = c2.factor(W, nb, trailing=mpf.TRAIL_FP16X3)
            st = c2.stats()
            assert info == 0 and st.superpanel == opts["superpanel_fp16"]
            if perm:
                assert int((ipiv != idx.to(torch.int32) + 1).sum()) > n // 4
            if ref is None:
                ref = (ipiv.clone(), W.clone())
            else:
                assert torch.equal(ipiv, ref[0]), opts
                scale = float(ref[1].abs().max())
                assert float((W - ref[1]).abs().max()) <= 2.0 ** -20 * scale * 8, opts
            c2.close()

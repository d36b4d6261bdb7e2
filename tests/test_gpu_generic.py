"""The generic path (fp16_panel_generic.hip + the generic schedule in mpf_host.cpp): every shape the reference accepts and
the tuned LDS-resident design does not cover -- panels wider than 256 columns (MPF() takes any r, MPF.cu:66,100-102),
panels taller than 256 rows x #CUs (hgetf2_kernel.cu:6 allows 262 144 rows) -- and the opt-in no-spin mode for shared GPUs.
Everything is compared bit for bit with the CPU oracle, and with the tuned path where both apply."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _factor_gpu(ctx, A, r, **kw):
    dA = ctx.from_numpy_f(A)
    ipiv, info = ctx.factor(dA, r, **kw)
    ctx.synchronize()
    return ctx.to_numpy_f(dA), ipiv.cpu().numpy(), info


@pytest.mark.parametrize("n,r", [(600, 300), (1100, 512), (700, 700), (520, 513), (2048, 1024), (900, 257), (1537, 384)])
def test_wide_panels_match_oracle_bit_exact(ctx, oracle, n, r):
    """r > 256: MPF(A, n, 512, ipiv) and friends.  Wide panels cannot be split into sub-panels (the fp16 elimination runs
    across all r columns), so the whole fp16 panel goes through the global-memory kernels; TRSM is blocked by 256 rows."""
    A = oracle.matgen_skip(n, skip=7 * n + r)
    LU_o, ip_o = oracle.mpf(A, r)
    LU_g, ip_g, info = _factor_gpu(ctx, A, r)
    assert info == 0 and ctx.stats().pivot_path == 1
    assert np.array_equal(ip_g, ip_o), f"{int((ip_g != ip_o).sum())} pivots differ, first at {np.argmax(ip_g != ip_o)}"
    assert np.array_equal(LU_g.view(np.uint64), LU_o.view(np.uint64))
    assert oracle.check_plu(A, LU_g, ip_g)[0] <= 1e-10


@pytest.mark.parametrize("n,r", [(2, 32), (33, 32), (129, 128), (513, 128), (1000, 256), (1024, 32), (300, 255), (100, 17), (1, 32)])
def test_no_spin_mode_equals_oracle_and_tuned_path(ctx, oracle, n, r):
    """mpf_opts.pivot_path = 1: generic pivots + generic schedule on shapes the tuned path also covers: same bits."""
    A = oracle.matgen_skip(n, skip=11 * n + r)
    LU_o, ip_o = oracle.mpf(A, r)
    LU_g, ip_g, info = _factor_gpu(ctx, A, r, pivot_path=1)
    assert ctx.stats().pivot_path == (1 if n > 1 else 0)
    LU_t, ip_t, _ = _factor_gpu(ctx, A, r)
    assert np.array_equal(ip_g, ip_o) and np.array_equal(ip_t, ip_o)
    assert np.array_equal(LU_g.view(np.uint64), LU_o.view(np.uint64))
    assert np.array_equal(LU_t.view(np.uint64), LU_o.view(np.uint64))


def test_panel_taller_than_the_lds_design(ctx, oracle):
    """A 70 000-row panel (more than 256 rows x 256 CUs): pivots and fp16 factors through the step operator."""
    import torch
    rows, cols = 70000, 24
    rng = np.random.default_rng(5)
    P = np.asfortranarray(rng.integers(0, 100, (rows, cols)) / 10.0)
    P[rows - 3, 0] = 60.0         # late rows win the first columns: exercises candidate blocks beyond index 255
    P[66000, 1] = 120.0
    want = oracle.double_to_fp16(P)
    ip_o = oracle.hgetf2(want)
    dP = ctx.from_numpy_f(P)
    ip_g, out = ctx.hgetf2_pivots(dP, ipiv_offset=0, want_panel=True)
    assert np.array_equal(ip_g.cpu().numpy(), ip_o)
    got = np.asfortranarray(out.t().contiguous().cpu().numpy().T).view(np.uint16)
    assert np.array_equal(got, want)
    assert (ip_o > 65536).any()


@pytest.mark.parametrize("mode_name,tol", [("TRAIL_FP16", 2e-2), ("TRAIL_FP16X3", 1e-5)])
def test_wide_panels_in_the_fp16_modes(ctx, oracle, mpf, mode_name, tol):
    n, r = 1536, 384
    A = oracle.matgen_skip(n, skip=3)
    dA = ctx.from_numpy_f(A)
    p, info = ctx.factor(dA, r, trailing=getattr(mpf, mode_name))
    ctx.synchronize()
    assert info == 0
    _, fro = oracle.check_plu(A, ctx.to_numpy_f(dA), p.cpu().numpy())
    assert fro < tol, fro
    assert np.array_equal(p.cpu().numpy()[:r], oracle.panel_pivots(A, 0, r))


def test_MPF_symbol_with_r_512(mpf, oracle):
    """The reference's own entry point with a panel width it accepts and round 1 refused."""
    L = mpf.load_library()
    f = getattr(L, mpf.CXX_SYMBOL_MPF)
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    n = 1200
    A = oracle.matgen_skip(n, skip=9)
    LU_o, ip_o = oracle.mpf(A, 512)
    Ah = A.copy(order="F")
    ip = np.arange(1, n + 1, dtype=np.int32)
    f(Ah.ctypes.data, n, 512, ip.ctypes.data)
    assert np.array_equal(ip, ip_o) and np.array_equal(Ah.view(np.uint64), LU_o.view(np.uint64))


def test_step_operators_beyond_256(ctx, oracle):
    """mpf_laswp with more than 256 swaps, mpf_dtrsm_llnu with more than 256 rows (blocked), against the oracle."""
    n, cols, k = 900, 300, 100
    rng = np.random.default_rng(1)
    A = np.asfortranarray(rng.standard_normal((n, 64)))
    piv = (rng.integers(k, n, cols) + 1).astype(np.int32)
    want = A.copy(order="F")
    oracle.laswp(want, k, cols, piv)
    import torch
    dA = ctx.from_numpy_f(A)
    ctx.laswp(dA, k, cols, torch.from_numpy(piv).to(ctx.device))
    assert np.array_equal(ctx.to_numpy_f(dA).view(np.uint64), want.view(np.uint64))
    m, nn = 520, 200
    Lm = np.asfortranarray(np.tril(rng.standard_normal((m, m)) * 0.05, -1) + np.eye(m))
    B = np.asfortranarray(rng.standard_normal((m, nn)))
    want = B.copy(order="F")
    oracle.dtrsm_llnu(Lm, want)
    dB = ctx.from_numpy_f(B)
    ctx.dtrsm_llnu(ctx.from_numpy_f(Lm), dB)
    assert np.array_equal(ctx.to_numpy_f(dB).view(np.uint64), want.view(np.uint64))


def test_spin_limit_gives_a_clean_error_not_a_hang(mpf, oracle):
    """The LDS pivot kernel's workgroups wait for each other's candidates; every wait is bounded.  With the bound forced
    to one poll the hand-off must fail -- cleanly: error code -4, a message, no hang, no out-of-range access (the
    buffers are documented as invalid after -4), and the next context works."""
    import torch
    n, r = 16384, 256
    c2 = mpf.MPFContext(0, options={"hp_spin_limit": 1})
    A = c2.matgen(n)
    with pytest.raises(mpf.MPFError) as ei:
        c2.factor(A, r)
    assert "(-4)" in str(ei.value) and "hand-off timed out" in str(ei.value)
    assert c2.stats().hpanel_timeouts > 0
    c2.close()
    c3 = mpf.MPFContext(0)
    A = oracle.matgen_skip(700, skip=1)
    dA = c3.from_numpy_f(A)
    ip, info = c3.factor(dA, 64)
    c3.synchronize()
    LU_o, ip_o = oracle.mpf(A, 64)
    assert info == 0 and np.array_equal(ip.cpu().numpy(), ip_o)
    c3.close()
    torch.cuda.synchronize()

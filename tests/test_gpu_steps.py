"""GPU parity tests, one per reference kernel / library call, through the C ABI (ctypes -> libmpf_amd.so).
Every comparison against the CPU oracle is BIT-EXACT (integer pivots, fp16 bit patterns, fp64 bit
patterns): the HIP kernels and the oracle implement the same numeric contract (oracle/mpf_oracle.c)."""
import numpy as np
import pytest

from conftest import bits16

pytestmark = pytest.mark.gpu


def _bits64(a):
    return np.ascontiguousarray(a).view(np.uint64)


def _same_f64(a, b):
    return np.array_equal(_bits64(np.asfortranarray(a)), _bits64(np.asfortranarray(b)))


# ---- double_to_fp16_block (reference MPF.cu:20-25, fp16_utils.h:15-23) -------------------------------
def test_double_to_fp16_bits(ctx, oracle):
    import torch
    rng = np.random.default_rng(1)
    m = np.float32(6.10352e-05)
    special = np.array([0.0, -0.0, 1.0, -1.0, 65504.0, 65519.9, 65520.0, 1e6, -1e6, 1e300, -1e300, np.inf, -np.inf,
                        2.0 ** -14, -2.0 ** -14, float(m), float(np.nextafter(m, np.float32(0))), float(np.nextafter(m, np.float32(1))),
                        2.0 ** -24, 2.0 ** -25, 6.0e-5, 6.2e-5, 1.0 + 2.0 ** -11, 1.0 + 2.0 ** -11 + 2.0 ** -30,
                        1.0 + 3 * 2.0 ** -11, 0.1, 9.9, 4.95, 1e-300, 5e-324])
    x = np.concatenate([special, rng.standard_normal(100000) * np.exp2(rng.integers(-30, 20, 100000)),
                        (rng.integers(0, 100, 50000) / 10.0)])
    got = bits16(ctx.double_to_fp16(torch.from_numpy(x).to(ctx.device)))
    want = oracle.double_to_fp16(x)
    assert np.array_equal(got, want)


# ---- the '/' of hgetf2_kernel.cu:108 -----------------------------------------------------------------
def test_hdiv_ieee(ctx, oracle):
    import torch
    rng = np.random.default_rng(2)
    a = rng.integers(0, 65536, 1 << 22, dtype=np.uint16)
    b = rng.integers(0, 65536, 1 << 22, dtype=np.uint16)
    # every numerator against a handful of denominators, incl. subnormal / max / zero
    alla = np.arange(65536, dtype=np.uint16)
    for den in (0x3C00, 0x3555, 0x7BFF, 0x0001, 0x03FF, 0x0400, 0x4900, 0xC8F3, 0x0000):
        a = np.concatenate([a, alla])
        b = np.concatenate([b, np.full(65536, den, dtype=np.uint16)])
    ta = torch.from_numpy(a.view(np.int16)).to(ctx.device)
    tb = torch.from_numpy(b.view(np.int16)).to(ctx.device)
    got = bits16(ctx.hdiv(ta, tb))
    want = oracle.hdiv(a, b)
    gf, wf = got.view(np.float16), want.view(np.float16)
    nan = np.isnan(gf) & np.isnan(wf)  # NaN payloads are outside the contract
    assert np.array_equal(got[~nan], want[~nan])


def test_hdiv_ieee_all_pairs(ctx, oracle):
    """the fp16 division of the pivot kernel against the contract on ALL 2^32 operand pairs (the GPU divides
    2^26 pairs per launch; the oracle checks them with every host core)."""
    import torch
    nb = 1024
    alla = torch.arange(65536, dtype=torch.int32, device=ctx.device).to(torch.int16).repeat(nb)        # numerator fastest
    bad_total = 0
    for b0 in range(0, 65536, nb):
        tb = torch.arange(b0, b0 + nb, dtype=torch.int32, device=ctx.device).to(torch.int16).repeat_interleave(65536)
        got = bits16(ctx.hdiv(alla, tb))
        bad, first = oracle.hdiv_check_all(got, b0, nb)
        assert bad == 0, f"{bad} of {nb * 65536} quotients differ for denominators {b0:#x}..; first a, b = {first >> 16:#06x}, {first & 0xFFFF:#06x}"
        bad_total += bad
    assert bad_total == 0


# ---- fp16 pivot panel (MPF.cu:108-159 + hgetf2_kernel.cu:15-120) ----------------------------------------
def _panel_case(oracle, kind, rows, cols, seed):
    rng = np.random.default_rng(seed)
    if kind == "gen":      # matrix_generator values: {0.0 .. 9.9}, many fp16 ties
        return np.asfortranarray(rng.integers(0, 100, (rows, cols)) / 10.0)
    if kind == "ties":     # tiny alphabet: the tie-break decides most pivots
        return np.asfortranarray(rng.integers(-2, 3, (rows, cols)).astype(np.float64))
    if kind == "sparse":   # matgen sparsity-style zeros
        a = rng.integers(0, 100, (rows, cols)) / 10.0
        a[rng.random((rows, cols)) < 0.7] = 0.0
        a[np.arange(cols), np.arange(cols)] += 1.0  # keep the fp16 pivots non-zero
        return np.asfortranarray(a)
    return np.asfortranarray(rng.standard_normal((rows, cols)) * np.exp2(rng.integers(-6, 6, (rows, cols))))


PANEL_SHAPES = [(2, 2), (3, 2), (8, 8), (33, 32), (256, 32), (257, 32), (300, 128), (511, 256), (512, 256),
                (1000, 256), (2049, 256), (5000, 200), (9000, 256)]


@pytest.mark.parametrize("kind", ["gen", "ties", "sparse", "normal"])
@pytest.mark.parametrize("rows,cols", PANEL_SHAPES)
def test_hgetf2_pivots_and_panel(ctx, oracle, kind, rows, cols):
    P = _panel_case(oracle, kind, rows, cols, rows * 131 + cols)
    want_bits = oracle.double_to_fp16(P)
    want_piv = oracle.hgetf2(want_bits)  # factors want_bits in place
    if not np.all(np.isfinite(want_bits.view(np.float16).astype(np.float32))):
        pytest.skip("fp16 panel hit a zero pivot (inf/NaN): outside the parity contract")
    dP = ctx.from_numpy_f(P)
    ipiv, panel = ctx.hgetf2_pivots(dP, ipiv_offset=7, want_panel=True)
    ctx.synchronize()
    assert ctx.stats().hpanel_timeouts == 0
    got_piv = ipiv.cpu().numpy() - 7
    assert np.array_equal(got_piv, want_piv), f"first diff at column {np.argmax(got_piv != want_piv)}"
    got_bits = np.asfortranarray(bits16(panel.t().contiguous()).reshape(cols, rows).T)
    gf, wf = got_bits.view(np.float16), want_bits.view(np.float16)
    nan = np.isnan(gf) & np.isnan(wf)
    assert np.array_equal(got_bits[~nan], want_bits[~nan])


# The column-window form of the pivot kernel (hgetf2_win_kernel: 136 of the 256 columns in LDS, the rest in registers until
# column 120) is what the factorization chain runs; the step operator reaches it when no fp16 copy of the panel is asked for.
# Shapes sit on both sides of every boundary of that layout: window width 136, swap column 120, one/several workgroups.
WINDOW_SHAPES = [(119, 119), (300, 120), (300, 121), (136, 136), (700, 136), (137, 137), (700, 137), (256, 140), (1000, 140),
                 (600, 199), (257, 255), (256, 256), (513, 256), (3000, 256), (30000, 256), (65536, 256)]


@pytest.mark.parametrize("kind", ["gen", "ties", "sparse", "normal"])
@pytest.mark.parametrize("rows,cols", WINDOW_SHAPES)
def test_hgetf2_pivots_window_form(ctx, oracle, kind, rows, cols):
    if rows > 10000 and kind not in ("gen", "ties"):
        pytest.skip("tall panels: two kinds are enough")
    P = _panel_case(oracle, kind, rows, cols, rows * 17 + cols)
    want_bits = oracle.double_to_fp16(P)
    want_piv = oracle.hgetf2(want_bits)
    if not np.all(np.isfinite(want_bits.view(np.float16).astype(np.float32))):
        pytest.skip("fp16 panel hit a zero pivot (inf/NaN): outside the parity contract")
    dP = ctx.from_numpy_f(P)
    before = ctx.get_option("hp_window")            # the shared context's setting (library default: -1, automatic) is put back
    for window in (1, 0):
        ctx.set_option("hp_window", window)
        try:
            ipiv, _ = ctx.hgetf2_pivots(dP, ipiv_offset=3, want_panel=False)
            ctx.synchronize()
        finally:
            ctx.set_option("hp_window", before)
        assert ctx.stats().hpanel_timeouts == 0
        got_piv = ipiv.cpu().numpy() - 3
        assert np.array_equal(got_piv, want_piv), f"hp_window={window}: first diff at column {np.argmax(got_piv != want_piv)}"


def test_hgetf2_inplace_fp16_entry(ctx, oracle):
    """mpf_hgetf2 = HGETF2_kernel's own signature: fp16 panel in, factored in place."""
    import torch
    rows, cols = 700, 64
    P = _panel_case(oracle, "gen", rows, cols, 5)
    bits = oracle.double_to_fp16(P)
    want = bits.copy(order="F")
    want_piv = oracle.hgetf2(want)
    d = torch.from_numpy(np.ascontiguousarray(bits.T).view(np.int16)).to(ctx.device).t()
    ipiv = ctx.hgetf2(d)
    ctx.synchronize()
    assert np.array_equal(ipiv.cpu().numpy(), want_piv)
    got = np.asfortranarray(bits16(d.t().contiguous()).reshape(cols, rows).T)
    assert np.array_equal(got, want)


def test_hgetf2_deterministic(ctx, oracle):
    P = _panel_case(oracle, "ties", 6000, 256, 99)
    dP = ctx.from_numpy_f(P)
    first = None
    for _ in range(10):
        ipiv, _ = ctx.hgetf2_pivots(dP)
        ctx.synchronize()
        p = ipiv.cpu().numpy()
        first = p if first is None else first
        assert np.array_equal(p, first)


# ---- LASWP_kernel (MPF.cu:42-59) ----------------------------------------------------------------------
@pytest.mark.parametrize("n,k,cols", [(64, 0, 32), (300, 32, 32), (1000, 256, 256), (1000, 744, 256), (517, 500, 17)])
def test_laswp(ctx, oracle, n, k, cols):
    import torch
    rng = np.random.default_rng(n + k)
    A = np.asfortranarray(rng.standard_normal((n, n)))
    piv = np.array([rng.integers(k + j, n) + 1 for j in range(cols)], dtype=np.int32)
    piv[::5] = k + np.arange(cols)[::5] + 1           # some no-op swaps
    if cols > 8:
        piv[3] = piv[7]                               # a far row picked twice
    want = A.copy(order="F")
    oracle.laswp(want, k, cols, piv)
    dA = ctx.from_numpy_f(A)
    ctx.laswp(dA, k, cols, torch.from_numpy(piv).to(ctx.device))
    ctx.synchronize()
    assert _same_f64(ctx.to_numpy_f(dA), want)


# ---- dgetf2_native_npv (dgetf2_native_npv.cu:11-36) ------------------------------------------------------
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("rows,cols", [(2, 2), (32, 32), (33, 32), (100, 40), (256, 256), (700, 100), (3000, 256), (4500, 128)])
def test_dgetf2_npv(ctx, oracle, rows, cols, fused):
    rng = np.random.default_rng(rows + cols)
    P = rng.standard_normal((rows, cols))
    P[np.arange(cols), np.arange(cols)] += 4.0 * np.sign(P[np.arange(cols), np.arange(cols)])
    # embed in a taller allocation so the leading dimension differs from rows
    big = np.asfortranarray(rng.standard_normal((rows + 5, cols + 3)))
    big[2:2 + rows, 1:1 + cols] = P
    want = big.copy(order="F")
    oracle.dgetf2_npv(want[2:2 + rows, 1:1 + cols], fused=fused)
    d = ctx.from_numpy_f(big)
    ctx.dgetf2_npv(d[2:2 + rows, 1:1 + cols], fused=fused)
    ctx.synchronize()
    assert _same_f64(ctx.to_numpy_f(d), want)


# ---- cublasDtrsm call site (MPF.cu:215-225) ------------------------------------------------------------
@pytest.mark.parametrize("m,n", [(1, 5), (32, 100), (40, 129), (128, 1000), (256, 3000), (200, 77)])
def test_dtrsm(ctx, oracle, m, n):
    rng = np.random.default_rng(m * 7 + n)
    L = np.asfortranarray(rng.standard_normal((m + 3, m)) * 0.3)
    B = np.asfortranarray(rng.standard_normal((m + 1, n)))
    want = B.copy(order="F")
    oracle.dtrsm_llnu(L[:m, :], want[:m, :])
    dB = ctx.from_numpy_f(B)
    ctx.dtrsm_llnu(ctx.from_numpy_f(L)[:m, :], dB[:m, :])
    ctx.synchronize()
    assert _same_f64(ctx.to_numpy_f(dB), want)


# ---- cublasDgemm call site (MPF.cu:230-239) ------------------------------------------------------------
@pytest.mark.parametrize("m,n,k", [(16, 16, 4), (128, 128, 16), (128, 128, 256), (200, 130, 32), (129, 257, 100),
                                   (1024, 768, 256), (1000, 1000, 128), (31, 1, 7)])
def test_dgemm_minus(ctx, oracle, m, n, k):
    rng = np.random.default_rng(m + n + k)
    A = np.asfortranarray(rng.standard_normal((m + 2, k)))
    B = np.asfortranarray(rng.standard_normal((k + 1, n)))
    Cm = np.asfortranarray(rng.standard_normal((m + 4, n)))
    want = Cm.copy(order="F")
    oracle.dgemm_minus(want[:m, :], A[:m, :], B[:k, :])
    dC = ctx.from_numpy_f(Cm)
    ctx.dgemm_minus(dC[:m, :], ctx.from_numpy_f(A)[:m, :], ctx.from_numpy_f(B)[:k, :])
    ctx.synchronize()
    got = ctx.to_numpy_f(dC)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    assert _same_f64(got, want), "summation order differs from contract C5 (fma chain, k ascending)"


@pytest.mark.parametrize("m,n,k,pa,pb,off", [(256, 384, 256, 0, 0, 0), (256, 384, 256, 1, 0, 0), (256, 384, 256, 0, 1, 0),
                                             (256, 384, 64, 2, 2, 1), (300, 520, 48, 0, 0, 0), (1280, 1152, 512, 0, 0, 0)])
def test_dgemm_minus_every_kernel_form(ctx, oracle, m, n, k, pa, pb, off):
    """The update picks its kernel from the operands: full tiles with 16-byte-aligned operands go to the LDS-DMA eight-wave
    kernel, odd leading dimensions / offsets to the register-staged one, ragged edges to the guarded four-wave kernel.  All of
    them must give the fma chain of contract C5, bit for bit."""
    rng = np.random.default_rng(m + n + k + pa + 2 * pb + 4 * off)
    A = np.asfortranarray(rng.standard_normal((m + pa + off, k)))
    B = np.asfortranarray(rng.standard_normal((k + pb + off, n)))
    Cm = np.asfortranarray(rng.standard_normal((m + off, n)))
    want = Cm.copy(order="F")
    oracle.dgemm_minus(want[off:off + m, :], A[off:off + m, :], B[off:off + k, :])
    dC = ctx.from_numpy_f(Cm)
    ctx.dgemm_minus(dC[off:off + m, :], ctx.from_numpy_f(A)[off:off + m, :], ctx.from_numpy_f(B)[off:off + k, :])
    ctx.synchronize()
    assert _same_f64(ctx.to_numpy_f(dC), want)


# ---- build-added speed mode of the trailing update: fp16 in, fp32 accumulate (north_star) -----------------
@pytest.mark.parametrize("m,n,k", [(64, 64, 16), (128, 128, 256), (200, 130, 32), (129, 257, 100), (1000, 900, 256),
                                   (300, 200, 512), (129, 257, 768), (1000, 900, 1024), (2000, 1500, 1000)])  # K >= 512: LDS-ring kernel
def test_hgemm_minus_fp16_fp32(ctx, oracle, m, n, k):
    rng = np.random.default_rng(m + n + k)
    A = np.asfortranarray(rng.standard_normal((m, k)))
    B = np.asfortranarray(rng.standard_normal((k, n)) * 4.0)
    Cm = np.asfortranarray(rng.standard_normal((m + 3, n)) * 10.0)
    Ah = oracle.double_to_fp16(A).view(np.float16).astype(np.float64)   # operands as the kernel rounds them
    Bh = oracle.double_to_fp16(B).view(np.float16).astype(np.float64)
    want = Cm.copy(order="F")
    want[:m, :] -= Ah @ Bh
    dC = ctx.from_numpy_f(Cm)
    ctx.hgemm_minus(dC[:m, :], ctx.from_numpy_f(A), ctx.from_numpy_f(B))
    ctx.synchronize()
    got = ctx.to_numpy_f(dC)
    # fp32 accumulation of exact fp16 x fp16 products: error <= ~k * 2^-24 * sum|a||b| (tolerance stated here)
    bound = 4.0 * k * 2.0 ** -24 * (np.abs(Ah) @ np.abs(Bh)) + 1e-12
    assert np.all(np.abs(got[:m, :] - want[:m, :]) <= bound)
    assert np.array_equal(got[m:, :], Cm[m:, :])


@pytest.mark.parametrize("m,n,k", [(64, 64, 16), (200, 130, 32), (129, 257, 100), (1000, 900, 256), (300, 200, 512), (1000, 900, 1024)])
def test_hgemm_minus_split_is_fp32_class(ctx, m, n, k):
    """fp16x3 mode: operands a = hi + 2^-11 lo (two fp16 numbers, ~22 bits), products hi*hi + 2^-11 (hi*lo + lo*hi)
    accumulated in fp32.  Tolerance: operand truncation 2 * 2^-22 plus fp32 accumulation k * 2^-24, both relative to
    sum |a||b| -- three decimal digits tighter than the plain fp16 mode's operand rounding (2^-11)."""
    rng = np.random.default_rng(m * 7 + n + k)
    A = np.asfortranarray(rng.standard_normal((m, k)))
    B = np.asfortranarray(rng.standard_normal((k, n)) * 4.0)
    A[0, 0] = 1e-3; A[1, 0] = 3e-6; B[0, 0] = 2e-4        # small magnitudes: lo term lands in fp16's subnormal range
    Cm = np.asfortranarray(rng.standard_normal((m + 3, n)) * 10.0)
    want = Cm.copy(order="F")
    want[:m, :] -= A @ B
    dC = ctx.from_numpy_f(Cm)
    ctx.hgemm_minus(dC[:m, :], ctx.from_numpy_f(A), ctx.from_numpy_f(B), split=True)
    ctx.synchronize()
    got = ctx.to_numpy_f(dC)
    bound = (2.0 ** -20 + 4.0 * k * 2.0 ** -24) * (np.abs(A) @ np.abs(B)) + 1e-9
    err = np.abs(got[:m, :] - want[:m, :])
    assert np.all(err <= bound)
    # and it really is much better than plain fp16 operands on the same data
    dC1 = ctx.from_numpy_f(Cm)
    ctx.hgemm_minus(dC1[:m, :], ctx.from_numpy_f(A), ctx.from_numpy_f(B))
    ctx.synchronize()
    err1 = np.abs(ctx.to_numpy_f(dC1)[:m, :] - want[:m, :])
    assert err.max() * 50 < err1.max()
    assert np.array_equal(got[m:, :], Cm[m:, :])


# ---- the big-K update on 256-row tiles (hgemm_big_kernel) and the fp32 working copy ---------------------------------------
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("c32", [False, True])
@pytest.mark.parametrize("m,n,k", [(1024, 1024, 256), (1100, 1030, 512), (2048, 2304, 1024), (1500, 1300, 320), (1025, 2047, 2048),
                                   (300, 200, 512), (129, 1500, 100),   # these two: 128-tile kernel on fp32 too
                                   # m + 3 a multiple of 4: the fp32 copy's block allows 16-byte accesses (hgemm_pp_kernel)
                                   (1025, 1024, 256), (1101, 1030, 512), (2049, 2304, 1024), (1501, 1300, 320), (4093, 3333, 768)])
def test_hgemm_minus_big_tiles_and_fp32_copy(ctx, oracle, m, n, k, c32, split):
    """Step-level check of what the two-level schedule of the fp16 modes runs: C -= fp16(A) fp16(B) on an fp64 matrix and
    on the fp32 working copy, full and ragged 256-row tiles, against the oracle's operand rounding and a stated tolerance:
    fp32 accumulation 4 k 2^-24 sum|a||b| (+ operand truncation 2^-20 in the split mode, + one fp32 rounding of the result on
    the fp32 copy).  Rows below m (the matrix is m + 3 rows tall) must come back untouched."""
    import torch
    rng = np.random.default_rng(m + 3 * n + k + int(c32) + 2 * int(split))
    A = np.asfortranarray(rng.standard_normal((m, k)))
    B = np.asfortranarray(rng.standard_normal((k, n)) * 4.0)
    Cm = np.asfortranarray(rng.standard_normal((m + 3, n)) * 10.0)
    if c32:
        Cm = np.asfortranarray(Cm.astype(np.float32).astype(np.float64))
    if split:
        Ah, Bh, optol = A, B, 2.0 ** -20
    else:
        Ah = oracle.double_to_fp16(A).view(np.float16).astype(np.float64)   # operands as the kernel rounds them
        Bh = oracle.double_to_fp16(B).view(np.float16).astype(np.float64)
        optol = 0.0
    want = Cm.copy(order="F")
    want[:m, :] -= Ah @ Bh
    dC = ctx.from_numpy_f(Cm)
    if c32:
        dC32 = torch.empty((n, m + 3), dtype=torch.float32, device=ctx.device).t()
        dC32.copy_(dC)
        ctx.hgemm_minus_f32(dC32[:m, :], ctx.from_numpy_f(A), ctx.from_numpy_f(B), split=split)
        ctx.synchronize()
        got = dC32.to(torch.float64).t().contiguous().cpu().numpy().T
    else:
        ctx.hgemm_minus(dC[:m, :], ctx.from_numpy_f(A), ctx.from_numpy_f(B), split=split)
        ctx.synchronize()
        got = ctx.to_numpy_f(dC)
    bound = (optol + 4.0 * k * 2.0 ** -24) * (np.abs(Ah) @ np.abs(Bh)) + 1e-9
    if c32:
        bound = bound + 2.0 ** -23 * (np.abs(want[:m, :]) + np.abs(Ah) @ np.abs(Bh))
    assert np.all(np.abs(got[:m, :] - want[:m, :]) <= bound)
    assert np.array_equal(got[m:, :], Cm[m:, :])


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("m,n,k", [(1100, 1030, 512), (2048, 2304, 1024), (1500, 1300, 320)])
def test_hgemm_minus_half_tile_variant(mpf, oracle, m, n, k, tile):
    """Option hgemm_big_tile: the tile forms of the big-K update (0 = 256 x 256, one workgroup per CU; 1 = 128 x 256, ring of
    four stages; 2 = 128 x 256, ring of three stages, TWO workgroups per CU): same result contract whatever the tile."""
    import torch
    c2 = mpf.MPFContext(0, options={"hgemm_big_tile": tile})
    rng = np.random.default_rng(m + n + k)
    A = np.asfortranarray(rng.standard_normal((m, k)))
    B = np.asfortranarray(rng.standard_normal((k, n)) * 4.0)
    Cm = np.asfortranarray((rng.standard_normal((m + 3, n)) * 10.0).astype(np.float32).astype(np.float64))
    Ah = oracle.double_to_fp16(A).view(np.float16).astype(np.float64)
    Bh = oracle.double_to_fp16(B).view(np.float16).astype(np.float64)
    want = Cm.copy(order="F")
    want[:m, :] -= Ah @ Bh
    dC32 = torch.empty((n, m + 3), dtype=torch.float32, device=c2.device).t()
    dC32.copy_(c2.from_numpy_f(Cm))
    c2.hgemm_minus_f32(dC32[:m, :], c2.from_numpy_f(A), c2.from_numpy_f(B))
    c2.synchronize()
    got = dC32.to(torch.float64).t().contiguous().cpu().numpy().T
    bound = 4.0 * k * 2.0 ** -24 * (np.abs(Ah) @ np.abs(Bh)) + 1e-9 + 2.0 ** -23 * (np.abs(want[:m, :]) + np.abs(Ah) @ np.abs(Bh))
    assert np.all(np.abs(got[:m, :] - want[:m, :]) <= bound)
    assert np.array_equal(got[m:, :], Cm[m:, :])
    c2.close()


def test_w32_window_conversions_and_interchange(ctx, oracle):
    """The fp32 working copy of the fp16 modes, step by step: a window of the column-major fp64 matrix goes to the row-major
    copy as (float)x and comes back as (double)(float)x, bit for bit, at offsets and ragged sizes (64 x 64 LDS transposes);
    LASWP on the copy moves exactly the rows the oracle's LASWP (MPF.cu:42-59) moves."""
    import torch
    rng = np.random.default_rng(5)
    n = 700
    A = np.asfortranarray(rng.standard_normal((n, n)) * 1e3)
    dA = ctx.from_numpy_f(A)
    W = torch.zeros((n, n), dtype=torch.float32, device=ctx.device)
    for r0, c0, rows, cols in ((0, 0, n, n), (3, 5, 130, 257), (64, 128, 65, 1), (699, 0, 1, 700), (10, 690, 333, 10)):
        W.zero_()
        ctx.w32_from_f64(dA[r0:r0 + rows, c0:c0 + cols], W[r0:r0 + rows, c0:c0 + cols])
        ctx.synchronize()
        want = np.zeros((n, n), dtype=np.float32)
        want[r0:r0 + rows, c0:c0 + cols] = A[r0:r0 + rows, c0:c0 + cols].astype(np.float32)
        assert np.array_equal(W.cpu().numpy(), want)
        back = ctx.from_numpy_f(np.asfortranarray(np.full((n, n), -7.0)))
        ctx.w32_to_f64(W[r0:r0 + rows, c0:c0 + cols], back[r0:r0 + rows, c0:c0 + cols])
        ctx.synchronize()
        wb = np.full((n, n), -7.0)
        wb[r0:r0 + rows, c0:c0 + cols] = want[r0:r0 + rows, c0:c0 + cols].astype(np.float64)
        assert np.array_equal(ctx.to_numpy_f(back), wb)
    # interchange: pivots of a real panel (generator matrix: nearly every column pivots), applied to columns [40, 700)
    G = oracle.matgen_skip(n, skip=11)
    k, cols = 128, 64
    ip = (oracle.panel_pivots(G, k, cols) + k).astype(np.int32)   # global 1-based pivots of the panel at k (MPF.cu:152)
    Wf = torch.from_numpy(G.astype(np.float32)).to(ctx.device).contiguous()
    ctx.w32_laswp(Wf[:, 40:], k, cols, torch.from_numpy(ip).to(ctx.device))
    ctx.synchronize()
    want = np.asfortranarray(G.astype(np.float32).astype(np.float64))
    sub = np.asfortranarray(want[:, 40:].copy())
    oracle.laswp(sub, k, cols, ip)
    want[:, 40:] = sub
    assert np.array_equal(Wf.cpu().numpy().astype(np.float64), want)

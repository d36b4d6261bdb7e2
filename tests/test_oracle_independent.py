"""A second, independent restatement of the part of the reference the repository cannot pin with reference-made vectors (the
fp16 pivot panel and the fp64 no-pivot panel: SURVEY 8c "parity unpinned"), written the way the CUDA kernels are written --
a grid of 256-thread blocks, shared arrays, the strict-'>' reduction tree over strides 128 .. 1, the serial block scan by
thread 0 of block 0, the all-column swap, per-thread elimination -- with numpy's IEEE binary16 / binary64 element operations
(each operation rounded once).  It shares no code with oracle/mpf_oracle.c (which restates the same semantics sequentially, with
software binary16 and closed-form tie rules) and must agree with it bit for bit: pivots, the factored fp16 panel, the fp64
panel.  This does not replace reference-generated vectors (there are none, and the CUDA cannot be built here); it removes the
risk that ONE restatement misread the kernels."""
import numpy as np
import pytest

BLOCK = 256


def ref_double_to_fp16(x):
    """fp16_utils.h:15-23"""
    xf = np.float32(x)
    fmax, fmin = np.float32(65504.0), np.float32(6.10352e-05)
    if xf > fmax:
        xf = fmax
    elif xf < -fmax:
        xf = -fmax
    if -fmin < xf < fmin:
        xf = np.float32(0.0)
    return np.float16(xf)


def cuda_style_hgetf2(panel):
    """hgetf2_kernel.cu:15-120 on a (rows, cols) float16 array, in place; returns the 1-based pivots.  Thread (bid, tid) is
    simulated literally; `>` on float16 is numpy's ordered compare (false for NaN), as __hgt."""
    rows, cols = panel.shape
    nblocks = (rows + BLOCK - 1) // BLOCK
    ipiv = np.zeros(cols, dtype=np.int32)
    with np.errstate(all="ignore"):
        for j in range(cols):
            g_vals = np.zeros(nblocks, dtype=np.float16)
            g_idx = np.zeros(nblocks, dtype=np.int64)
            for bid in range(nblocks):                                  # :32-62, one block at a time
                max_vals = np.zeros(BLOCK, dtype=np.float16)            # :34
                piv = np.full(BLOCK, j, dtype=np.int64)                 # :35
                for tid in range(BLOCK):
                    row_idx = bid * BLOCK + tid + j                      # :39
                    if row_idx < rows:
                        max_vals[tid] = np.abs(panel[row_idx, j])        # :41
                        piv[tid] = row_idx
                stride = BLOCK // 2
                while stride > 0:                                        # :47-56
                    for tid in range(stride):                            # (reads of [tid + stride] never alias writes of [tid])
                        if max_vals[tid + stride] > max_vals[tid]:
                            max_vals[tid] = max_vals[tid + stride]
                            piv[tid] = piv[tid + stride]
                    stride //= 2
                g_vals[bid], g_idx[bid] = max_vals[0], piv[0]            # :59-62
            gmax, gidx = g_vals[0], g_idx[0]                             # :70-78
            for b in range(1, nblocks):
                if g_vals[b] > gmax:
                    gmax, gidx = g_vals[b], g_idx[b]
            ipiv[j] = gidx + 1                                           # :80-81
            p = gidx
            if p != j:                                                   # :92-98: all columns
                panel[[j, p], :] = panel[[p, j], :]
            if j + 1 < rows:                                             # :104-115, every row_idx > j
                pivot_val = panel[j, j]
                mult = (panel[j + 1:, j] / pivot_val).astype(np.float16)                 # one rounding: the fp16 '/'
                panel[j + 1:, j] = mult
                for k in range(j + 1, cols):
                    t = (mult * panel[j, k]).astype(np.float16)          # '*' rounded ...
                    panel[j + 1:, k] = (panel[j + 1:, k] - t).astype(np.float16)   # ... then '-=' rounded (no fma)
    return ipiv


def cuda_style_dgetf2_npv(panel):
    """dgetf2_native_npv.cu:18-35 (the committed build recipe does not contract: separate multiply and subtract)"""
    m, n = panel.shape
    with np.errstate(all="ignore"):
        for j in range(n):
            if j + 1 < m:
                mult = panel[j + 1:, j] / panel[j, j]
                panel[j + 1:, j] = mult
                for k in range(j + 1, n):
                    panel[j + 1:, k] = panel[j + 1:, k] - mult * panel[j, k]


def _panels():
    rng = np.random.default_rng(2026)
    out = []
    for rows, cols in ((1, 1), (2, 2), (5, 3), (37, 7), (64, 64), (256, 32), (257, 32), (300, 17), (600, 33), (1030, 24)):
        out.append(("generator", (rng.integers(0, 100, (rows, cols)) / 10.0)))
    # ties everywhere: few distinct magnitudes, both signs, zeros -> the reduction tree and the block scan decide
    for rows, cols in ((256, 16), (513, 20), (700, 8), (40, 40)):
        out.append(("ties", rng.choice(np.array([0.0, 1.0, -1.0, 2.0, -2.0, 0.5]), (rows, cols))))
    # conversion edges: overflow clamp, the flush threshold (exactly 2^-14 and its neighbours), tiny and huge values
    edge = np.array([65504.0, 65520.0, 1e6, -1e6, 6.103515625e-05, 6.1035156e-05, 6.104e-05, 6.2e-05, -6.103515625e-05, 1e-7, 0.0, 3.0, -7.25])
    out.append(("edges", rng.choice(edge, (300, 12))))
    # a zero column in the middle: pivot = the column's own row, then 0 / 0 -> NaN spreads (compared as bits)
    z = rng.integers(0, 100, (130, 9)) / 10.0
    z[:, 4] = 0.0
    out.append(("zero column", z))
    return out


@pytest.mark.parametrize("idx", range(16))
def test_fp16_panel_matches_an_independent_cuda_style_restatement(oracle, idx):
    name, A = _panels()[idx]
    rows, cols = A.shape
    cols = min(cols, rows)
    A = np.asfortranarray(A[:, :cols].astype(np.float64))
    # conversion (fp16_utils.h:15-23): element by element against the oracle's block conversion
    conv = np.array([[ref_double_to_fp16(A[i, c]) for c in range(cols)] for i in range(rows)], dtype=np.float16)
    bits_o = oracle.double_to_fp16(A)
    assert np.array_equal(conv.view(np.uint16), bits_o), name
    # the factorization: pivots and every bit of the factored fp16 panel
    P = conv.copy()
    ip = cuda_style_hgetf2(P)
    panel_o = np.asfortranarray(bits_o.copy())
    ip_o = oracle.hgetf2(panel_o)
    assert np.array_equal(ip, ip_o), (name, ip, ip_o)
    assert np.array_equal(P.view(np.uint16), panel_o), name
    assert np.array_equal(oracle.panel_pivots(A, 0, cols), ip)


@pytest.mark.parametrize("m,n", [(1, 1), (5, 5), (40, 7), (300, 32), (257, 64)])
def test_fp64_panel_matches_an_independent_restatement(oracle, m, n):
    rng = np.random.default_rng(m * 131 + n)
    A = np.asfortranarray(rng.integers(1, 100, (m, n)) / 10.0 + np.eye(m, n) * 50.0)
    want = A.copy(order="F")
    cuda_style_dgetf2_npv(want)
    got = A.copy(order="F")
    oracle.dgetf2_npv(got)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))

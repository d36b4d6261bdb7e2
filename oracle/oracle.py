"""ctypes front-end for the CPU oracle (oracle/mpf_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.

All matrices are numpy float64, column-major (Fortran order), as the reference stores them
(benchmark.cpp:19: mat[col * n + row]).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmpf_oracle.so")
REF_MATGEN = os.path.join(_HERE, "_ref", "matgen")


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "mpf_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    elif os.path.isdir("/root/reference") and not os.path.exists(REF_MATGEN):
        subprocess.run(["make", "-C", _HERE, "-s", "ref"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        try:
            _lib = C.CDLL(_SO)
        except OSError:
            build(force=True)
            _lib = C.CDLL(_SO)
        L = _lib
        dp, ip, hp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint16)
        i64 = C.c_int64
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_f32_to_f16_hw.restype = C.c_uint16
        L.orc_f32_to_f16_hw.argtypes = [C.c_float]
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_double_to_fp16_block.argtypes = [dp, hp, i64]
        L.orc_fp16_to_double_block.argtypes = [hp, dp, i64]
        L.orc_hdiv_block.argtypes = [hp, hp, hp, i64]
        L.orc_hdiv_check_all.restype = i64
        L.orc_hdiv_check_all.argtypes = [hp, C.c_int, C.c_int, C.POINTER(i64)]
        L.orc_hgetf2.argtypes = [hp, i64, C.c_int, C.c_int, ip]
        L.orc_panel_pivots.argtypes = [dp, i64, C.c_int, C.c_int, ip]
        L.orc_laswp.argtypes = [dp, i64, i64, C.c_int, C.c_int, ip]
        L.orc_dgetf2_npv.argtypes = [C.c_int, C.c_int, dp, i64, C.c_int]
        L.orc_dtrsm_llnu.argtypes = [C.c_int, i64, dp, i64, dp, i64]
        L.orc_dgemm_minus.argtypes = [i64, i64, C.c_int, dp, i64, dp, i64, dp, i64]
        L.orc_mpf.argtypes = [dp, C.c_int, C.c_int, ip, C.c_int]
        L.orc_check_plu.restype = C.c_double
        L.orc_check_plu.argtypes = [dp, dp, ip, C.c_int, dp]
        L.orc_rand_stream.argtypes = [C.c_uint, ip, C.c_int]
        L.orc_matgen.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_double]
        L.orc_matgen_skip.argtypes = [dp, C.c_int, i64]
        L.orc_lu_solve.argtypes = [dp, ip, C.c_int, dp]
        L.orc_residual.restype = C.c_double
        L.orc_residual.argtypes = [dp, dp, dp, C.c_int, dp]
        L.orc_force_portable_fp16.argtypes = [C.c_int]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _hp(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint16))


def _fcol(a):
    assert a.dtype == np.float64 and a.flags.f_contiguous, "need float64 column-major"
    return a


# ---- fp16 helpers -------------------------------------------------------------------
def double_to_fp16(x):
    """fp16_utils.h:15-23 element-wise; returns uint16 bit patterns with x's shape AND memory layout."""
    x = np.asarray(x, dtype=np.float64)
    if not (x.flags.c_contiguous or x.flags.f_contiguous):
        x = np.ascontiguousarray(x)
    out = np.empty_like(x, dtype=np.uint16)  # keeps C / Fortran order
    xin = x.ravel(order="K")
    o = out.ravel(order="K")
    assert np.shares_memory(o, out)
    lib().orc_double_to_fp16_block(_dp(xin), _hp(o), x.size)
    return out


def hdiv(a_bits, b_bits):
    a = np.ascontiguousarray(a_bits, dtype=np.uint16)
    b = np.ascontiguousarray(b_bits, dtype=np.uint16)
    q = np.empty_like(a)
    lib().orc_hdiv_block(_hp(a), _hp(b), _hp(q), a.size)
    return q


def hdiv_check_all(got_bits, b0, nb):
    """got_bits[ib * 65536 + a] = q(a, b0 + ib) for all 65536 numerators: returns (mismatches, first bad a * 65536 + b or -1)."""
    g = np.ascontiguousarray(got_bits, dtype=np.uint16)
    assert g.size == nb * 65536
    first = C.c_int64(-1)
    bad = lib().orc_hdiv_check_all(_hp(g), b0, nb, C.byref(first))
    return int(bad), int(first.value)


# ---- kernels ------------------------------------------------------------------------
def hgetf2(panel_bits):
    """hgetf2_kernel.cu:15-120 on a (rows, cols) uint16 column-major panel, in place.
    Returns 1-based panel-local pivots."""
    assert panel_bits.dtype == np.uint16 and panel_bits.flags.f_contiguous
    rows, cols = panel_bits.shape
    ipiv = np.zeros(cols, dtype=np.int32)
    lib().orc_hgetf2(_hp(panel_bits), rows, rows, cols, _ip(ipiv))
    return ipiv


def panel_pivots(A, k, cols):
    """Pivots (1-based, panel-local) the reference would pick for panel A[k:, k:k+cols]."""
    _fcol(A)
    n = A.shape[0]
    ipiv = np.zeros(cols, dtype=np.int32)
    sub = A[k:, k:k + cols]
    lib().orc_panel_pivots(C.cast(sub.ctypes.data, C.POINTER(C.c_double)), A.strides[1] // 8, n - k, cols,
                           _ip(ipiv))
    return ipiv


def laswp(A, k, cols, ipiv_global):
    """MPF.cu:42-59, in place on all columns of A."""
    _fcol(A)
    p = np.ascontiguousarray(ipiv_global, dtype=np.int32)
    lib().orc_laswp(_dp(A), A.strides[1] // 8, A.shape[1], k, cols, _ip(p))


def dgetf2_npv(P, fused=False):
    """dgetf2_native_npv.cu:11-36, in place on the (m, n) column-major view P."""
    assert P.dtype == np.float64 and P.strides[0] == 8
    m, n = P.shape
    lib().orc_dgetf2_npv(m, n, C.cast(P.ctypes.data, C.POINTER(C.c_double)), P.strides[1] // 8, int(fused))


def dtrsm_llnu(L, B):
    assert L.strides[0] == 8 and B.strides[0] == 8
    m, n = B.shape
    lib().orc_dtrsm_llnu(m, n, C.cast(L.ctypes.data, C.POINTER(C.c_double)), L.strides[1] // 8,
                         C.cast(B.ctypes.data, C.POINTER(C.c_double)), B.strides[1] // 8)


def dgemm_minus(Cm, A, B):
    """Cm -= A @ B with the contract-C5 summation order."""
    assert Cm.strides[0] == 8 and A.strides[0] == 8 and B.strides[0] == 8
    m, n = Cm.shape
    kk = A.shape[1]
    lib().orc_dgemm_minus(m, n, kk, C.cast(A.ctypes.data, C.POINTER(C.c_double)), A.strides[1] // 8,
                          C.cast(B.ctypes.data, C.POINTER(C.c_double)), B.strides[1] // 8,
                          C.cast(Cm.ctypes.data, C.POINTER(C.c_double)), Cm.strides[1] // 8)


def mpf(A, r, fused_panel=False):
    """MPF.cu:66-256.  Returns (LU, ipiv); A is not modified.  ipiv starts as identity
    (benchmark.cpp:215-217)."""
    LU = np.array(A, dtype=np.float64, order="F", copy=True)
    n = LU.shape[0]
    ipiv = np.arange(1, n + 1, dtype=np.int32)
    rc = lib().orc_mpf(_dp(LU), n, r, _ip(ipiv), int(fused_panel))
    assert rc == 0
    return LU, ipiv


def check_plu(A, LU, ipiv):
    """benchmark.cpp:106-144.  Returns (max|A-PLU|, ||A-PLU||_F/||A||_F)."""
    _fcol(A), _fcol(LU)
    fro = C.c_double(0)
    p = np.ascontiguousarray(ipiv, dtype=np.int32)
    mx = lib().orc_check_plu(_dp(A), _dp(LU), _ip(p), A.shape[0], C.byref(fro))
    return mx, fro.value


# ---- generator ----------------------------------------------------------------------
def rand_stream(n, seed=1):
    out = np.empty(n, dtype=np.int32)
    lib().orc_rand_stream(seed, _ip(out), n)
    return out


def matgen(n, step=2, func="exp", sparsity=0.0):
    """The n x n matrix `matgen f maxSize step func sparsity` emits, as benchmark.cpp
    interprets it (column-major)."""
    out = np.empty((n, n), dtype=np.float64, order="F")
    rc = lib().orc_matgen(_dp(out), n, step, 1 if func == "exp" else 0, sparsity)
    if rc != 0:
        raise ValueError(f"size {n} is not in the {func}/{step} sequence")
    return out


def matgen_skip(n, skip=4):
    """`matgen f n (n-2) lin` -> sizes 2 then n: the n x n matrix after `skip` draws."""
    out = np.empty((n, n), dtype=np.float64, order="F")
    lib().orc_matgen_skip(_dp(out), n, skip)
    return out


def read_matgen_file(path):
    """benchmark.cpp:171-199 reader: returns the list of matrices in the file."""
    with open(path) as f:
        toks = f.read().split()
    cnt = int(toks[0])
    pos, mats = 1, []
    for _ in range(cnt):
        n = int(toks[pos]); pos += 1
        vals = np.array(toks[pos:pos + n * n], dtype=np.float64); pos += n * n
        mats.append(np.asfortranarray(vals.reshape((n, n), order="F")))
    return mats


# ---- solve --------------------------------------------------------------------------
def lu_solve(LU, ipiv, b):
    x = np.array(b, dtype=np.float64, copy=True)
    p = np.ascontiguousarray(ipiv, dtype=np.int32)
    lib().orc_lu_solve(_dp(_fcol(LU)), _ip(p), LU.shape[0], _dp(x))
    return x


def residual(A, x, b):
    r = np.empty_like(b)
    rel = lib().orc_residual(_dp(_fcol(A)), _dp(np.ascontiguousarray(x)), _dp(np.ascontiguousarray(b)),
                             A.shape[0], _dp(r))
    return rel, r

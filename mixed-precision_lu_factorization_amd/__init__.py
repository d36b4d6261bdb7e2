"""MI355X-native MPF hot path: Python host mirror of the C ABI in include/mpf_c.h.

The product is lib/libmpf_amd.so (hand-written HIP kernels for gfx950 + C++ host driver); this
module only binds it with ctypes and uses torch for device memory and streams.  There is no CPU
fallback: importing works anywhere (so the build can be checked without a GPU), but every compute
entry point raises if the library is missing or no GPU is present.

The directory name contains '-', so import it with
    importlib.import_module("mixed-precision_lu_factorization_amd")
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPF_LIB") or os.path.join(_HERE, "lib", "libmpf_amd.so")   # (MPF_LIB: another build of the product library, for A/B runs of tools/)
# the same sources built with -DMPF_PROBE: + microbenchmarks and measured-slower kernel variants; tools/ and bench.py's
# on-box peak measurements load it, a product run never does (include/mpf_probe.h)
PROBE_LIB_PATH = os.environ.get("MPF_PROBE_LIB") or os.path.join(_HERE, "lib", "libmpf_probe.so")   # (tools: another build of the probe library for A/B runs)

TRAIL_FP64 = 0
TRAIL_FP16 = 1
TRAIL_FP16X3 = 2

# every symbol include/mpf_c.h declares (tests check the library exports all of them)
C_ABI_SYMBOLS = [
    "mpf_create", "mpf_destroy", "mpf_set_stream", "mpf_synchronize", "mpf_last_error", "mpf_get_stats",
    "mpf_device_report", "mpf_factor_host", "mpf_factor_dev", "mpf_double_to_fp16", "mpf_hdiv",
    "mpf_hgetf2_pivots", "mpf_hgetf2", "mpf_laswp", "mpf_dgetf2_npv", "mpf_dtrsm_llnu", "mpf_dgemm_minus",
    "mpf_solve_ir", "mpf_hgetf2_capacity_rows", "mpf_set_option", "mpf_get_option", "mpf_option_name", "mpf_hgemm_minus", "mpf_hgemm_minus_f32", "mpf_w32_from_f64", "mpf_w32_to_f64", "mpf_w32_laswp", "mpf_gesv", "mpf_matgen_dev", "mpf_matgen_cols_dev",
    "mpf_matgen_state", "mpf_rccl_unique_id", "mpf_rccl_init", "mpf_rccl_destroy", "mpf_rccl_version", "mpf_rccl_info", "mpf_rccl_bcast_probe", "mpf_factor_dist",
    "mpf_solve_ir_dist", "mpf_rccl_selftest", "mpf_check_plu_dev", "mpf_check_plu_host", "mpf_solve_ir_nrhs",
    "mpf_solve_gmres_ir", "mpf_trim", "mpf_dist_set_p2p",
]
PROBE_ONLY_SYMBOLS = ["mpf_microbench", "mpf_debug_mfma4", "mpf_debug_gate", "mpf_debug_hgemm_again"]   # include/mpf_probe.h
CXX_SYMBOL_MPF = "_Z3MPFPdiiPi"  # void MPF(double*, int, int, int*)  (reference MPF.h:3)


class MpfOpts(C.Structure):
    _fields_ = [("trailing", C.c_int32), ("verbose", C.c_int32), ("fused_panel", C.c_int32),
                ("sync_timing", C.c_int32), ("no_lookahead", C.c_int32), ("superpanel", C.c_int32), ("pivot_path", C.c_int32), ("reserved", C.c_int32)]


class MpfStats(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_h2d", C.c_double), ("ms_d2h", C.c_double),
                ("ms_hpanel", C.c_double), ("ms_laswp", C.c_double), ("ms_dpanel", C.c_double),
                ("ms_trsm", C.c_double), ("ms_gemm", C.c_double), ("n", C.c_int64), ("nb", C.c_int32),
                ("panels", C.c_int32), ("hpanel_timeouts", C.c_int32), ("info", C.c_int32),
                ("gemm_launches", C.c_int32), ("lookahead", C.c_int32), ("superpanel", C.c_int32),
                ("pivot_path", C.c_int32), ("gemm_flops", C.c_double), ("gemm_bytes", C.c_double),
                ("ms_gemm_big", C.c_double), ("gemm_big_flops", C.c_double), ("gemm_big_bytes", C.c_double),
                ("ms_cvt", C.c_double), ("ms_blockrow", C.c_double), ("gemm_big_launches", C.c_int32), ("host_rows_streamed", C.c_int32),
                ("host_late_segments", C.c_int32), ("reserved", C.c_int32)]


class MpfIrStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("rel_residual", C.c_double),
                ("history", C.c_double * 32), ("ms_total", C.c_double), ("stalled", C.c_int32), ("reserved", C.c_int32)]


class MpfGmresStats(C.Structure):
    _fields_ = [("outer_iterations", C.c_int32), ("inner_iterations", C.c_int32), ("converged", C.c_int32), ("budget_expired", C.c_int32),
                ("rel_residual", C.c_double), ("history", C.c_double * 32), ("ms_total", C.c_double)]


class MpfRcclInfo(C.Structure):
    _fields_ = [("version", C.c_int32), ("has_comm", C.c_int32), ("comm_count", C.c_int32), ("comm_rank", C.c_int32), ("has_p2p", C.c_int32),
                ("device", C.c_int32), ("visible_devices", C.c_int32), ("reserved", C.c_int32),
                ("bcast_calls", C.c_int64), ("bcast_bytes", C.c_int64), ("allreduce_calls", C.c_int64), ("p2p_calls", C.c_int64), ("p2p_bytes", C.c_int64),
                ("link_type", C.c_int32 * 16), ("link_hops", C.c_int32 * 16), ("peer_access", C.c_int32 * 16)]


class MpfGesvStats(C.Structure):
    _fields_ = [("path", C.c_int32), ("info", C.c_int32), ("ms_factor_fp16", C.c_double), ("ms_ir_fp16", C.c_double),
                ("ms_factor_fp64", C.c_double), ("ms_ir_fp64", C.c_double), ("ms_total", C.c_double),
                ("ir_fp16", MpfIrStats), ("ir_final", MpfIrStats), ("gmres_budget_ms", C.c_double), ("gmres_budget_expired", C.c_int32),
                ("reserved", C.c_int32)]


BCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
P2P_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p)


class MpfDist(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("bcast", BCAST_FN), ("allreduce", ALLREDUCE_FN), ("user", C.c_void_p)]


class MPFError(RuntimeError):
    pass


def build(force=False):
    """Compile lib/libmpf_amd.so (product) and lib/libmpf_probe.so (tools) for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs += [os.path.join(_HERE, "..", "include", h) for h in ("mpf_c.h", "MPF.h", "mpf_probe.h")] + [os.path.join(_HERE, "Makefile")]
    stale = force or any(not os.path.exists(lp) or any(os.path.getmtime(s) > os.path.getmtime(lp) for s in srcs)
                         for lp in (LIB_PATH, PROBE_LIB_PATH))
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s", "-j8", "all", "probe"], check=True)
    return LIB_PATH


_libs = {}


def load_library(probe=False):
    """dlopen the product library (probe=True: the probe build, tools only); raises MPFError (never falls back) when absent."""
    if probe in _libs:
        return _libs[probe]
    # torch first: libmpf_amd.so must bind to the HIP runtime torch already loaded (one runtime per process;
    # loading the system libamdhip64 before torch's bundled one leaves the process without devices)
    import torch  # noqa: F401
    path = PROBE_LIB_PATH if probe else LIB_PATH
    if not os.path.exists(path):
        raise MPFError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    L = C.CDLL(path)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.mpf_create.argtypes = [C.POINTER(vp), C.c_int]
    L.mpf_destroy.argtypes = [vp]
    L.mpf_set_stream.argtypes = [vp, vp]
    L.mpf_synchronize.argtypes = [vp]
    L.mpf_last_error.argtypes = [vp]
    L.mpf_last_error.restype = C.c_char_p
    L.mpf_get_stats.argtypes = [vp, C.POINTER(MpfStats)]
    L.mpf_set_option.argtypes = [vp, C.c_char_p, i64]
    L.mpf_get_option.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
    L.mpf_option_name.argtypes = [i32, C.c_char_p, i64]
    L.mpf_device_report.argtypes = [C.c_char_p, i64]
    L.mpf_factor_host.argtypes = [vp, vp, i64, i32, vp, C.POINTER(MpfOpts)]
    L.mpf_trim.argtypes = [vp]
    L.mpf_dist_set_p2p.argtypes = [vp, P2P_FN, vp]
    L.mpf_factor_dev.argtypes = [vp, vp, i64, i64, i32, vp, C.POINTER(MpfOpts)]
    L.mpf_double_to_fp16.argtypes = [vp, vp, vp, i64]
    L.mpf_hdiv.argtypes = [vp, vp, vp, vp, i64]
    L.mpf_hgetf2_pivots.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp]
    L.mpf_hgetf2.argtypes = [vp, vp, i64, i32, i32, vp]
    L.mpf_laswp.argtypes = [vp, vp, i64, i64, i32, i32, vp]
    L.mpf_dgetf2_npv.argtypes = [vp, vp, i64, i32, i32, i32]
    L.mpf_dtrsm_llnu.argtypes = [vp, i32, i64, vp, i64, vp, i64]
    L.mpf_dgemm_minus.argtypes = [vp, i64, i64, i32, vp, i64, vp, i64, vp, i64]
    L.mpf_hgemm_minus.argtypes = [vp, i64, i64, i32, vp, i64, vp, i64, vp, i64, i32]
    L.mpf_hgemm_minus_f32.argtypes = [vp, i64, i64, i32, vp, i64, vp, i64, vp, i64, i32]
    L.mpf_w32_from_f64.argtypes = [vp, vp, i64, vp, i64, i64, i64]
    L.mpf_w32_to_f64.argtypes = [vp, vp, i64, vp, i64, i64, i64]
    L.mpf_w32_laswp.argtypes = [vp, vp, i64, i64, i32, i32, vp]
    L.mpf_solve_ir.argtypes = [vp, vp, i64, vp, i64, vp, i64, vp, vp, i32, dbl, C.POINTER(MpfIrStats)]
    L.mpf_solve_ir_nrhs.argtypes = [vp, vp, i64, vp, i64, vp, i64, i32, vp, i64, vp, i64, i32, dbl, C.POINTER(MpfIrStats)]
    L.mpf_solve_gmres_ir.argtypes = [vp, vp, i64, vp, i64, vp, i64, vp, vp, i32, i32, dbl, C.POINTER(MpfGmresStats)]
    L.mpf_gesv.argtypes = [vp, vp, i64, i64, i32, vp, vp, vp, vp, i32, dbl, i32, C.POINTER(MpfGesvStats)]
    L.mpf_matgen_dev.argtypes = [vp, vp, i64, i64, i64]
    L.mpf_matgen_cols_dev.argtypes = [vp, vp, i64, i64, i64, i64, i64]
    L.mpf_matgen_state.argtypes = [i64, C.POINTER(C.c_uint32)]
    L.mpf_rccl_unique_id.argtypes = [vp]
    L.mpf_rccl_init.argtypes = [vp, vp, i32, i32]
    L.mpf_rccl_destroy.argtypes = [vp]
    L.mpf_rccl_version.argtypes = []
    L.mpf_rccl_selftest.argtypes = [vp]
    L.mpf_check_plu_dev.argtypes = [vp, vp, i64, vp, i64, vp, i64, C.POINTER(dbl), C.POINTER(dbl)]
    L.mpf_check_plu_host.argtypes = [vp, vp, vp, i64, C.POINTER(dbl), C.POINTER(dbl)]
    L.mpf_factor_dist.argtypes = [vp, vp, i64, i64, i32, vp, C.POINTER(MpfDist), C.POINTER(MpfOpts)]
    L.mpf_solve_ir_dist.argtypes = [vp, vp, i64, vp, i64, vp, i64, i32, vp, vp, i32, dbl, C.POINTER(MpfDist), C.POINTER(MpfIrStats)]
    for name in C_ABI_SYMBOLS:
        if name != "mpf_last_error":
            getattr(L, name).restype = C.c_int
    if probe:
        L.mpf_microbench.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
        L.mpf_microbench.restype = C.c_int
    _libs[probe] = L
    return L


def option_names():
    L = load_library()
    buf = C.create_string_buffer(64)
    n = L.mpf_option_name(-1, buf, 64)
    out = []
    for i in range(n):
        L.mpf_option_name(i, buf, 64)
        out.append(buf.value.decode())
    return out


def device_report():
    """HIP analogue of the reference's check_cooperative_groups.cu probe."""
    L = load_library()
    buf = C.create_string_buffer(4096)
    n = L.mpf_device_report(buf, 4096)
    return n, buf.value.decode()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _colmajor_ld(t):
    """Leading dimension of a 2-D torch tensor that is a column-major view (stride(0) == 1)."""
    assert t.dim() == 2 and t.stride(0) == 1, "need a column-major matrix (e.g. X.t() of a contiguous tensor)"
    return t.stride(1) if t.shape[1] > 1 else max(t.shape[0], 1)


class MPFContext:
    """Owns a mpf_ctx.  Matrices are torch float64 CUDA tensors in COLUMN-MAJOR layout, i.e. a
    tensor `A` with A.stride() == (1, lda) -- create one with `colmajor(n, m)` or `from_numpy_f`."""

    def __init__(self, device=0, use_torch_stream=True, stream=None, probe=False, options=None):
        import torch
        if not torch.cuda.is_available():
            raise MPFError("no GPU visible: the MPF hot path is HIP-only (no CPU fallback)")
        self.torch = torch
        self.probe = probe
        self.L = load_library(probe)
        self.h = C.c_void_p()
        rc = self.L.mpf_create(C.byref(self.h), device)
        if rc != 0:
            raise MPFError("mpf_create failed: " + self.L.mpf_last_error(None).decode())
        self.device = torch.device("cuda", device)
        self.stream = stream  # a torch.cuda.Stream this context always launches on; None: torch's CURRENT stream,
        self._follow = stream is None and use_torch_stream  # re-read at every call (so `with torch.cuda.stream(s):` works)
        self._bound = None
        if stream is not None:
            self.L.mpf_set_stream(self.h, C.c_void_p(stream.cuda_stream))
        self._bind()
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def set_option(self, name, value):
        """Per-context behaviour switch (include/mpf_c.h mpf_set_option); defaults came from MPF_* at construction."""
        self._check(self.L.mpf_set_option(self.h, name.encode(), int(value)), "mpf_set_option")

    def rccl_info(self):
        """mpf_rccl_info as a dict (the multi-GPU bench line's `rccl` object is built from it)."""
        inf = MpfRcclInfo()
        self.L.mpf_rccl_info.argtypes = [C.c_void_p, C.POINTER(MpfRcclInfo)]
        self._check(self.L.mpf_rccl_info(self.h, C.byref(inf)), "mpf_rccl_info")
        nd = min(int(inf.visible_devices), 16)
        return {"version": int(inf.version), "has_comm": bool(inf.has_comm), "comm_count": int(inf.comm_count), "comm_rank": int(inf.comm_rank),
                "has_p2p": bool(inf.has_p2p), "device": int(inf.device), "visible_devices": int(inf.visible_devices),
                "bcast_calls": int(inf.bcast_calls), "bcast_bytes": int(inf.bcast_bytes), "allreduce_calls": int(inf.allreduce_calls),
                "p2p_calls": int(inf.p2p_calls), "p2p_bytes": int(inf.p2p_bytes),
                "link_type_to_device": [int(inf.link_type[d]) for d in range(nd)], "link_hops_to_device": [int(inf.link_hops[d]) for d in range(nd)],
                "peer_access_to_device": [int(inf.peer_access[d]) for d in range(nd)]}

    def rccl_bcast_probe(self, nbytes, root=0, reps=5):
        """ms per ncclBroadcast of nbytes on the context's communicator (every rank calls it)."""
        self._bind()
        ms = C.c_double(0)
        self.L.mpf_rccl_bcast_probe.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
        self._check(self.L.mpf_rccl_bcast_probe(self.h, int(nbytes), int(root), int(reps), C.byref(ms)), "mpf_rccl_bcast_probe")
        return ms.value

    def hgetf2_capacity_rows(self, waiters=0, form=0):
        """Tallest panel (rows) the LDS pivot kernel takes beside `waiters` waiting workgroups (-w: beside the pipelined chain on a
        w-column panel); form 0 = either form, 1 = full slab, 2 = column window."""
        self.L.mpf_hgetf2_capacity_rows.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        self.L.mpf_hgetf2_capacity_rows.restype = C.c_int64
        return int(self.L.mpf_hgetf2_capacity_rows(self.h, waiters, form))

    def get_option(self, name):
        v = C.c_int64(0)
        self._check(self.L.mpf_get_option(self.h, name.encode(), C.byref(v)), "mpf_get_option")
        return v.value

    def _bind(self):
        """Launch on the stream the caller's torch operations are ordered on: torch's current stream, looked up at every
        call unless an explicit stream was given at construction."""
        if self._follow:
            cur = self.torch.cuda.current_stream(self.device).cuda_stream
            if cur != self._bound:
                self.L.mpf_set_stream(self.h, C.c_void_p(cur))
                self._bound = cur

    def close(self):
        if self.h:
            self.L.mpf_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise MPFError(f"{what} failed ({rc}): " + self.L.mpf_last_error(self.h).decode())
        return rc

    # ---- helpers ---------------------------------------------------------------------------
    def colmajor(self, rows, cols, dtype=None):
        t = self.torch
        return t.empty((cols, rows), dtype=dtype or t.float64, device=self.device).t()

    def from_numpy_f(self, a):
        """numpy (rows, cols) array -> column-major device tensor of the same logical shape."""
        import numpy as np
        t = self.torch
        af = np.asfortranarray(a)
        return t.from_numpy(np.ascontiguousarray(af.T)).to(self.device).t()

    def to_numpy_f(self, x):
        import numpy as np
        return np.asfortranarray(x.t().contiguous().cpu().numpy().T)

    def synchronize(self):
        self._bind()
        self._check(self.L.mpf_synchronize(self.h), "synchronize")

    def stats(self):
        s = MpfStats()
        self.L.mpf_get_stats(self.h, C.byref(s))
        return s

    def microbench(self, which):
        """0: f64 MFMA TFLOP/s, 1: f16 MFMA TFLOP/s, 2: HBM copy TB/s (measured on this box).  Probe library only:
        construct the context with probe=True."""
        if not self.probe:
            raise MPFError("microbench lives in libmpf_probe.so: MPFContext(device, probe=True)")
        self._bind()
        r = C.c_double(0)
        self._check(self.L.mpf_microbench(self.h, which, C.byref(r)), "microbench")
        return r.value

    def matgen(self, n, skip=4, out=None, col0=0, ncols=None):
        """The reference generator's matrix (matrix_generator.cpp:55-80 as benchmark.cpp reads it: `matgen f n (n-2) lin`
        for skip = 4), produced on the device; columns [col0, col0 + ncols) of it when given.  Column-major tensor."""
        self._bind()
        ncols = n - col0 if ncols is None else ncols
        if out is None:
            out = self.colmajor(n, ncols)
        assert out.shape == (n, ncols) and out.dtype == self.torch.float64
        rc = self.L.mpf_matgen_cols_dev(self.h, _ptr(out), _colmajor_ld(out), n, skip, col0, ncols)
        self._check(rc, "mpf_matgen_cols_dev")
        return out

    # ---- whole path ------------------------------------------------------------------------
    def factor(self, A, nb, ipiv=None, trailing=TRAIL_FP64, fused_panel=False, sync_timing=False, verbose=False,
               no_lookahead=False, superpanel=0, pivot_path=0):
        """mpf_factor_dev: in-place MPF of the column-major device matrix A (N x N).
        Returns (ipiv int32 device tensor, info)."""
        self._bind()
        t = self.torch
        n = A.shape[0]
        assert A.shape[1] == n and A.dtype == t.float64
        if ipiv is None:
            ipiv = t.arange(1, n + 1, dtype=t.int32, device=self.device)  # benchmark.cpp:215-217
        o = MpfOpts(trailing=trailing, verbose=int(verbose), fused_panel=int(fused_panel), sync_timing=int(sync_timing),
                    no_lookahead=int(no_lookahead), superpanel=int(superpanel), pivot_path=int(pivot_path))
        rc = self.L.mpf_factor_dev(self.h, _ptr(A), _colmajor_ld(A), n, nb, _ptr(ipiv), C.byref(o))
        return ipiv, self._check(rc, "mpf_factor_dev")

    def trim(self):
        """mpf_trim: give the context's large cached buffers (host-path copies, working copies) back to the device."""
        self._bind()
        self._check(self.L.mpf_trim(self.h), "mpf_trim")

    def factor_host(self, A_np, nb, ipiv_np=None, **kw):
        """mpf_factor_host: numpy column-major matrix in host memory, factored in place."""
        import numpy as np
        assert A_np.dtype == np.float64 and A_np.flags.f_contiguous
        n = A_np.shape[0]
        if ipiv_np is None:
            ipiv_np = np.arange(1, n + 1, dtype=np.int32)
        o = MpfOpts(trailing=kw.get("trailing", TRAIL_FP64), fused_panel=int(kw.get("fused_panel", False)))
        rc = self.L.mpf_factor_host(self.h, C.c_void_p(A_np.ctypes.data), n, nb, C.c_void_p(ipiv_np.ctypes.data),
                                    C.byref(o))
        return ipiv_np, self._check(rc, "mpf_factor_host")

    def solve_ir(self, A, LU, ipiv, b, max_iter=10, tol=1e-12):
        self._bind()
        t = self.torch
        n = A.shape[0]
        x = t.empty(n, dtype=t.float64, device=self.device)
        st = MpfIrStats()
        rc = self.L.mpf_solve_ir(self.h, _ptr(A), _colmajor_ld(A), _ptr(LU), _colmajor_ld(LU), _ptr(ipiv), n,
                                 _ptr(b), _ptr(x), max_iter, tol, C.byref(st))
        self._check(rc, "mpf_solve_ir")
        return x, st

    def solve_ir_nrhs(self, A, LU, ipiv, B, max_iter=10, tol=1e-12):
        """mpf_solve_ir_nrhs: B is N x nrhs column-major; returns (X, list of per-column stats)."""
        self._bind()
        n, nrhs = B.shape
        X = self.colmajor(n, nrhs)
        st = (MpfIrStats * nrhs)()
        rc = self.L.mpf_solve_ir_nrhs(self.h, _ptr(A), _colmajor_ld(A), _ptr(LU), _colmajor_ld(LU), _ptr(ipiv), n, nrhs,
                                      _ptr(B), _colmajor_ld(B), _ptr(X), _colmajor_ld(X), max_iter, tol, st)
        self._check(rc, "mpf_solve_ir_nrhs")
        return X, list(st)

    def solve_gmres_ir(self, A, LU, ipiv, b, max_outer=10, restart=30, tol=1e-12):
        """mpf_solve_gmres_ir: refinement with GMRES (preconditioned by the factors) on the correction equation."""
        self._bind()
        t = self.torch
        n = A.shape[0]
        x = t.empty(n, dtype=t.float64, device=self.device)
        st = MpfGmresStats()
        rc = self.L.mpf_solve_gmres_ir(self.h, _ptr(A), _colmajor_ld(A), _ptr(LU), _colmajor_ld(LU), _ptr(ipiv), n, _ptr(b), _ptr(x),
                                       max_outer, restart, tol, C.byref(st))
        self._check(rc, "mpf_solve_gmres_ir")
        return x, st

    def check_plu(self, A, LU, ipiv):
        """The reference's acceptance test on the device (benchmark.cpp:106-144): returns (max|A - P L U|, ||.||_F / ||A||_F)."""
        self._bind()
        mx, fro = C.c_double(0), C.c_double(0)
        rc = self.L.mpf_check_plu_dev(self.h, _ptr(A), _colmajor_ld(A), _ptr(LU), _colmajor_ld(LU), _ptr(ipiv), A.shape[0],
                                      C.byref(mx), C.byref(fro))
        self._check(rc, "mpf_check_plu_dev")
        return mx.value, fro.value

    # ---- multi-GPU (one process per GPU, 1-D block-cyclic columns) ---------------------------
    def rccl_init(self, rank, world, group=None):
        """Create this context's RCCL communicator; the 128-byte unique id travels through torch.distributed."""
        import torch.distributed as tdist
        ident = [None]
        if world == 1:
            buf = C.create_string_buffer(128)
            self._check(self.L.mpf_rccl_unique_id(buf), "mpf_rccl_unique_id")
            self._check(self.L.mpf_rccl_init(self.h, buf, 0, 1), "mpf_rccl_init")
            return
        # every rank must be able to load librccl BEFORE any rank enters ncclCommInitRank (a rank that cannot would return
        # without entering the collective and leave its peers waiting there): agree on that over torch.distributed first
        loaded = [int(self.L.mpf_rccl_version() > 0)]
        everyone = [None] * world
        tdist.all_gather_object(everyone, loaded[0], group=group)
        if not all(everyone):
            raise RuntimeError(f"librccl not loadable on ranks {[r for r, ok in enumerate(everyone) if not ok]}")
        if rank == 0:   # a failure here must still reach the other ranks, or they would wait in the broadcast for ever
            buf = C.create_string_buffer(128)
            if self.L.mpf_rccl_unique_id(buf) == 0:
                ident[0] = bytes(buf.raw)
        tdist.broadcast_object_list(ident, src=0, group=group)
        if ident[0] is None:
            raise RuntimeError("mpf_rccl_unique_id failed on rank 0 (librccl not loadable?)")
        self._check(self.L.mpf_rccl_init(self.h, C.c_char_p(ident[0]), rank, world), "mpf_rccl_init")

    def dist_set_p2p(self, fn):
        """mpf_dist_set_p2p: point-to-point callback (a P2P_FN the caller keeps alive, or None) for the distributed solves' chain."""
        self._bind()
        self._check(self.L.mpf_dist_set_p2p(self.h, fn if fn is not None else C.cast(None, P2P_FN), None), "mpf_dist_set_p2p")

    def factor_dist(self, Aloc, n, nb, dist, ipiv=None, trailing=TRAIL_FP64, no_lookahead=False, pivot_path=0, verbose=False, superpanel=0):
        """mpf_factor_dist: Aloc = this rank's column blocks (n x local columns, column-major); returns (ipiv, info).
        superpanel: panels per super-panel in the fp16 modes (0 = the context's default, 1 = one-level schedule)."""
        self._bind()
        t = self.torch
        if ipiv is None:
            ipiv = t.arange(1, n + 1, dtype=t.int32, device=self.device)
        o = MpfOpts(trailing=trailing, no_lookahead=int(no_lookahead), pivot_path=int(pivot_path), verbose=int(verbose), superpanel=int(superpanel))
        ld = _colmajor_ld(Aloc) if Aloc.shape[1] > 0 else n
        rc = self.L.mpf_factor_dist(self.h, _ptr(Aloc) if Aloc.shape[1] > 0 else C.c_void_p(0), max(ld, n), n, nb, _ptr(ipiv),
                                    C.byref(dist), C.byref(o))
        return ipiv, self._check(rc, "mpf_factor_dist")

    def solve_ir_dist(self, Aloc, LUloc, ipiv, b, n, nb, dist, max_iter=10, tol=1e-12):
        self._bind()
        t = self.torch
        x = t.empty(n, dtype=t.float64, device=self.device)
        st = MpfIrStats()
        has = Aloc.shape[1] > 0
        rc = self.L.mpf_solve_ir_dist(self.h, _ptr(Aloc) if has else C.c_void_p(0), max(_colmajor_ld(Aloc), n) if has else n,
                                      _ptr(LUloc) if has else C.c_void_p(0), max(_colmajor_ld(LUloc), n) if has else n, _ptr(ipiv),
                                      n, nb, _ptr(b), _ptr(x), max_iter, tol, C.byref(dist), C.byref(st))
        self._check(rc, "mpf_solve_ir_dist")
        return x, st

    # ---- step operators --------------------------------------------------------------------
    def double_to_fp16(self, x):
        self._bind()
        t = self.torch
        out = t.empty(x.numel(), dtype=t.int16, device=self.device)
        self._check(self.L.mpf_double_to_fp16(self.h, _ptr(x), _ptr(out), x.numel()), "double_to_fp16")
        return out

    def hdiv(self, a_bits, b_bits):
        self._bind()
        t = self.torch
        q = t.empty_like(a_bits)
        self._check(self.L.mpf_hdiv(self.h, _ptr(a_bits), _ptr(b_bits), _ptr(q), a_bits.numel()), "hdiv")
        return q

    def hgetf2_pivots(self, P, ipiv_offset=0, want_panel=False):
        """P: column-major fp64 view rows x cols.  Returns (ipiv int32[cols], fp16 panel bits or None)."""
        self._bind()
        t = self.torch
        rows, cols = P.shape
        ipiv = t.zeros(cols, dtype=t.int32, device=self.device)
        out = self.colmajor(rows, cols, dtype=t.int16) if want_panel else None
        rc = self.L.mpf_hgetf2_pivots(self.h, _ptr(P), _colmajor_ld(P), rows, cols, ipiv_offset, _ptr(ipiv), _ptr(out))
        self._check(rc, "hgetf2_pivots")
        return ipiv, out

    def hgetf2(self, P16):
        self._bind()
        t = self.torch
        rows, cols = P16.shape
        ipiv = t.zeros(cols, dtype=t.int32, device=self.device)
        self._check(self.L.mpf_hgetf2(self.h, _ptr(P16), _colmajor_ld(P16), rows, cols, _ptr(ipiv)), "hgetf2")
        return ipiv

    def laswp(self, A, k, cols, ipiv_global):
        self._bind()
        self._check(self.L.mpf_laswp(self.h, _ptr(A), _colmajor_ld(A), A.shape[1], k, cols, _ptr(ipiv_global)), "laswp")

    def dgetf2_npv(self, P, fused=False):
        self._bind()
        rows, cols = P.shape
        self._check(self.L.mpf_dgetf2_npv(self.h, _ptr(P), _colmajor_ld(P), rows, cols, int(fused)), "dgetf2_npv")

    def dtrsm_llnu(self, Lm, B):
        self._bind()
        m, n = B.shape
        assert Lm.shape[0] >= m and Lm.shape[1] >= m, "dtrsm_llnu: L smaller than m x m"
        self._check(self.L.mpf_dtrsm_llnu(self.h, m, n, _ptr(Lm), _colmajor_ld(Lm), _ptr(B), _colmajor_ld(B)), "dtrsm")

    def dgemm_minus(self, Cm, A, B):
        self._bind()
        m, n = Cm.shape
        k = A.shape[1]
        assert A.shape == (m, k) and B.shape == (k, n), "dgemm_minus: operand shapes do not match C"
        self._check(self.L.mpf_dgemm_minus(self.h, m, n, k, _ptr(A), _colmajor_ld(A), _ptr(B), _colmajor_ld(B),
                                           _ptr(Cm), _colmajor_ld(Cm)), "dgemm")

    def hgemm_minus(self, Cm, A, B, split=False):
        """fp16-in / fp32-accumulate variant of dgemm_minus (speed mode of the trailing update); split=True uses
        hi + 2^-11 lo operands (three MFMA products, fp32-class accuracy)."""
        self._bind()
        m, n = Cm.shape
        k = A.shape[1]
        assert A.shape == (m, k) and B.shape == (k, n), "hgemm_minus: operand shapes do not match C"
        self._check(self.L.mpf_hgemm_minus(self.h, m, n, k, _ptr(A), _colmajor_ld(A), _ptr(B), _colmajor_ld(B),
                                           _ptr(Cm), _colmajor_ld(Cm), int(split)), "hgemm")

    def hgemm_minus_f32(self, Cm, A, B, split=False):
        """mpf_hgemm_minus_f32: the fp16 update on an fp32 column-major matrix (the fp16 modes' working copy)."""
        self._bind()
        assert Cm.dtype == self.torch.float32
        m, n = Cm.shape
        k = A.shape[1]
        assert A.shape == (m, k) and B.shape == (k, n), "hgemm_minus_f32: operand shapes do not match C"
        self._check(self.L.mpf_hgemm_minus_f32(self.h, m, n, k, _ptr(A), _colmajor_ld(A), _ptr(B), _colmajor_ld(B),
                                               _ptr(Cm), _colmajor_ld(Cm), int(split)), "hgemm_f32")

    # the fp32 working copy of the fp16 modes (row-major torch float32 tensors: W[i, j] contiguous in j)
    def w32_from_f64(self, A, W):
        self._bind()
        rows, cols = A.shape
        assert W.shape == (rows, cols) and W.dtype == self.torch.float32 and W.stride(1) == 1
        self._check(self.L.mpf_w32_from_f64(self.h, _ptr(A), _colmajor_ld(A), _ptr(W), W.stride(0), rows, cols), "w32_from_f64")

    def w32_to_f64(self, W, A):
        self._bind()
        rows, cols = A.shape
        assert W.shape == (rows, cols) and W.dtype == self.torch.float32 and W.stride(1) == 1
        self._check(self.L.mpf_w32_to_f64(self.h, _ptr(W), W.stride(0), _ptr(A), _colmajor_ld(A), rows, cols), "w32_to_f64")

    def w32_laswp(self, W, k, cols, ipiv_global):
        self._bind()
        assert W.dtype == self.torch.float32 and W.stride(1) == 1
        self._check(self.L.mpf_w32_laswp(self.h, _ptr(W), W.stride(0), W.shape[1], k, cols, _ptr(ipiv_global)), "w32_laswp")

    def gesv(self, A, b, nb=256, max_iter=10, tol=1e-12, try_fp16=True, work=None):
        """mpf_gesv: x with ||b - A x|| / ||b|| <= tol by the fastest path (fp16 trailing + refinement, else fp64)."""
        self._bind()
        t = self.torch
        n = A.shape[0]
        if work is None:
            work = self.colmajor(n, n)
        ipiv = t.empty(n, dtype=t.int32, device=self.device)
        x = t.empty(n, dtype=t.float64, device=self.device)
        st = MpfGesvStats()
        rc = self.L.mpf_gesv(self.h, _ptr(A), _colmajor_ld(A), n, nb, _ptr(work), _ptr(ipiv), _ptr(b), _ptr(x), max_iter, tol,
                             int(try_fp16), C.byref(st))  # try_fp16: 0 fp64 only, 1/True fp16, 2 fp16x3
        self._check(rc, "mpf_gesv")
        return x, st, work, ipiv

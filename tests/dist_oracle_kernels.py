"""Test double for the step operators of MPFContext, backed by the CPU oracle, working on column-major CPU
torch tensors.  Lets the distributed schedule (mixed-precision_lu_factorization_amd/dist.py) run under
gloo without a GPU.  TEST CODE: the product never imports this."""
import numpy as np
import torch

from oracle import oracle as O


def _np(t):
    assert t.device.type == "cpu" and t.stride(0) == 1
    return t.numpy()  # shares memory; F-ordered view


class OracleKernels:
    def hgetf2_pivots(self, P, ipiv_offset=0, want_panel=False):
        a = _np(P)
        rows, cols = a.shape
        piv = np.zeros(cols, dtype=np.int32)
        import ctypes as C
        O.lib().orc_panel_pivots(C.cast(a.ctypes.data, C.POINTER(C.c_double)), a.strides[1] // 8 if cols > 1 else rows,
                                 rows, cols, piv.ctypes.data_as(C.POINTER(C.c_int)))
        return torch.from_numpy(piv + ipiv_offset), None

    def laswp(self, A, k, cols, ipiv_global):
        a = _np(A)
        if a.shape[1] == 0:
            return
        import ctypes as C
        p = np.ascontiguousarray(ipiv_global.numpy(), dtype=np.int32)
        O.lib().orc_laswp(C.cast(a.ctypes.data, C.POINTER(C.c_double)), a.strides[1] // 8 if a.shape[1] > 1 else a.shape[0],
                          a.shape[1], k, cols, p.ctypes.data_as(C.POINTER(C.c_int)))

    def dgetf2_npv(self, P, fused=False):
        O.dgetf2_npv(_np(P), fused)

    def dtrsm_llnu(self, L, B):
        if B.shape[1]:
            O.dtrsm_llnu(_np(L), _np(B))

    def dgemm_minus(self, Cm, A, B):
        if Cm.shape[0] and Cm.shape[1]:
            O.dgemm_minus(_np(Cm), _np(A), _np(B))

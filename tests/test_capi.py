"""CPU checks of the drop-in boundary: the library builds, loads, and exports exactly what include/*.h
declare (no compute calls here -- those need a GPU and live in the -m gpu tests)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(mpf):
    mpf.build()
    L = mpf.load_library()
    hdr = open(os.path.join(ROOT, "include", "mpf_c.h")).read()
    declared = sorted(set(re.findall(r"\b(mpf_[a-z0-9_]+)\s*\(", hdr)))
    assert sorted(mpf.C_ABI_SYMBOLS) == declared
    for name in declared:
        assert hasattr(L, name), name
    # the reference's own C++ symbol: void MPF(double*, int, int, int*)  (MPF.h:3)
    assert hasattr(L, mpf.CXX_SYMBOL_MPF)
    assert "void MPF(double *h_A, int N, int r, int *IPIV);" in open(os.path.join(ROOT, "include", "MPF.h")).read()


def test_struct_layouts_match_header(mpf):
    assert C.sizeof(mpf.MpfOpts) == 32
    assert C.sizeof(mpf.MpfStats) == 8 * 8 + 8 + 4 * 6
    assert C.sizeof(mpf.MpfIrStats) == 8 + 8 + 32 * 8 + 8 + 8
    assert C.sizeof(mpf.MpfGesvStats) == 8 + 5 * 8 + 2 * C.sizeof(mpf.MpfIrStats)


def test_no_gpu_means_loud_failure(mpf):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mpf.MPFError):
        mpf.MPFContext(0)
    n, txt = mpf.device_report()
    assert n == 0 and "HIP devices: 0" in txt
    # mpf_create itself reports the reference's message (MPF.cu:72-75 analogue)
    L = mpf.load_library()
    h = C.c_void_p()
    assert L.mpf_create(C.byref(h), 0) < 0
    assert b"No HIP devices" in L.mpf_last_error(None)


def test_product_never_touches_the_oracle(mpf):
    """The shipped path may not import, link, load or call anything under oracle/ (comments may cite it)."""
    pkg = os.path.dirname(mpf.__file__)
    bad = [r"^\s*(from|import)\s+oracle", r"libmpf_oracle", r"#include\s+\"[^\"]*oracle", r"oracle\.py", r"orc_[a-z0-9_]+\s*\("]
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp, f)).read()
                for pat in bad:
                    assert not re.search(pat, txt, flags=re.M), (f, pat)

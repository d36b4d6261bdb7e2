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


def test_probes_are_not_in_the_product_library(mpf):
    """libmpf_amd.so keeps what a default run can reach; microbenchmarks, the cycle-stamped pivot kernel and the four-wave A/B
    switch live in libmpf_probe.so only (include/mpf_probe.h), which exports the whole C ABI plus those.  (The 128-row pivot
    kernel is a product kernel since round 5: option hp_half_slabs.)"""
    import subprocess
    mpf.build()
    prod = subprocess.run(["nm", "-D", "--defined-only", mpf.LIB_PATH], capture_output=True, text=True, check=True).stdout
    probe = subprocess.run(["nm", "-D", "--defined-only", mpf.PROBE_LIB_PATH], capture_output=True, text=True, check=True).stdout
    for name in mpf.PROBE_ONLY_SYMBOLS:
        assert name not in prod and name in probe, name
    for frag in ("hgetf2_lds_kernelILi256ELb1", "mfma_f64_rate_kernel", "stream_copy_kernel"):
        assert frag not in prod, frag
    assert "hgetf2_lds_kernelILi128" in prod and "hgetf2_lds_kernelILi128" in probe and "hgetf2_lds_kernelILi256ELb1" in probe
    for name in mpf.C_ABI_SYMBOLS:
        assert name in probe, name
    hdr = open(os.path.join(ROOT, "include", "mpf_probe.h")).read()
    assert "mpf_microbench" in hdr


def test_no_lazy_environment_reads_in_the_library(mpf):
    """Behaviour switches are per-context options: the environment is read in ONE place (mpf_create's defaults)."""
    src = os.path.join(os.path.dirname(mpf.__file__), "csrc")
    hits = []
    for f in sorted(os.listdir(src)):
        if f == "microbench.hip":
            continue
        for i, line in enumerate(open(os.path.join(src, f)), 1):
            if "getenv(" in line:
                hits.append((f, i))
    assert len(hits) == 1 and hits[0][0] == "mpf_host.cpp", hits


def test_struct_layouts_match_header(mpf, tmp_path):
    """sizeof / offsetof as gcc sees include/mpf_c.h == the ctypes mirror in the Python host."""
    import subprocess
    src = tmp_path / "lay.c"
    src.write_text("""
#include <stdio.h>
#include <stddef.h>
#include "mpf_c.h"
int main(void) {
    printf("%zu %zu %zu %zu\\n", sizeof(mpf_opts), sizeof(mpf_stats), sizeof(mpf_ir_stats), sizeof(mpf_gesv_stats));
    printf("%zu %zu %zu %zu\\n", offsetof(mpf_opts, superpanel), offsetof(mpf_stats, gemm_bytes), offsetof(mpf_ir_stats, stalled),
           offsetof(mpf_gesv_stats, ir_final));
    return 0;
}
""")
    exe = tmp_path / "lay"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    sizes, offs = [int(v) for v in out[:4]], [int(v) for v in out[4:]]
    assert sizes == [C.sizeof(mpf.MpfOpts), C.sizeof(mpf.MpfStats), C.sizeof(mpf.MpfIrStats), C.sizeof(mpf.MpfGesvStats)]
    assert offs == [mpf.MpfOpts.superpanel.offset, mpf.MpfStats.gemm_bytes.offset, mpf.MpfIrStats.stalled.offset,
                    mpf.MpfGesvStats.ir_final.offset]


def test_generator_jump_ahead_matches_the_stream(mpf, oracle):
    """mpf_matgen_state (host-only part of the device generator): the 31-word ring in front of rand() call j, continued
    with o[k] = o[k-3] + o[k-31], reproduces glibc's stream as the oracle restates it (itself checked against libc)."""
    L = mpf.load_library()
    ref = oracle.rand_stream(1_000_200)
    for call in (0, 1, 4, 30, 31, 32, 1000, 123457, 1_000_000):
        st = (C.c_uint32 * 31)()
        assert L.mpf_matgen_state(call, st) == 0
        s = [int(x) for x in st]
        out = []
        for k in range(200):
            u = k % 31
            s[u] = (s[u] + s[(u + 28) % 31]) & 0xFFFFFFFF
            out.append(s[u] >> 1)
        assert out == ref[call:call + 200].tolist(), call


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

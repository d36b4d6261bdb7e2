"""CPU sanitizer build of the host-only code: the oracle restatement and the harness generator compiled with
-fsanitize=address,undefined and run on small inputs.  No GPU, no product library involved."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "oracle", "_san")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="2")


def _build():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "san"], check=True)


def _clean(out):
    txt = out.stdout + out.stderr
    assert out.returncode == 0, txt[-3000:]
    assert "AddressSanitizer" not in txt and "runtime error" not in txt and "LeakSanitizer" not in txt, txt[-3000:]


def test_oracle_under_asan_ubsan():
    _build()
    out = subprocess.run([os.path.join(SAN, "san_driver")], env=ENV, capture_output=True, text=True, timeout=600)
    _clean(out)
    assert "san driver ok" in out.stdout


def test_harness_generator_under_asan_ubsan(tmp_path):
    _build()
    for args in (("64", "2", "exp"), ("40", "7", "lin", "0.3"), ("1", "2", "exp"), ("33", "31", "lin", "0.9")):
        out = subprocess.run([os.path.join(SAN, "mpf_matgen_san"), str(tmp_path / "m.txt"), *args], env=ENV, capture_output=True, text=True, timeout=120)
        _clean(out)
    # the reference tool's argument errors (matrix_generator.cpp:8-31) must not trip the sanitizers either
    for args in ((), ("x.txt",), ("x.txt", "abc"), ("x.txt", "8", "0"), ("x.txt", "8", "2", "sideways")):
        out = subprocess.run([os.path.join(SAN, "mpf_matgen_san"), *args], env=ENV, cwd=tmp_path, capture_output=True, text=True, timeout=60)
        txt = out.stdout + out.stderr
        assert "AddressSanitizer" not in txt and "runtime error" not in txt, txt[-2000:]

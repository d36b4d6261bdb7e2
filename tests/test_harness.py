"""The reference's own workflow (generator -> bench -> benchmark_times.csv) on top of libmpf_amd.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = os.path.join(ROOT, "harness")


def _build():
    subprocess.run(["make", "-C", H, "-s"], check=True)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "matgen")), reason="oracle/_ref/matgen missing")
@pytest.mark.parametrize("args", [("64", "2", "exp"), ("40", "7", "lin", "0.3"), ("33", "31", "lin"), ("20", "5", "exp", "0.5")])
def test_matgen_cli_is_byte_identical_to_the_reference(mpf, tmp_path, args):
    mpf.build()
    _build()
    a, b = tmp_path / "a.txt", tmp_path / "b.txt"
    subprocess.run([os.path.join(H, "mpf_matgen"), str(a), *args], check=True, capture_output=True)
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "matgen"), str(b), *args], check=True, capture_output=True)
    assert a.read_bytes() == b.read_bytes()


def test_bench_cli_without_gpu_reports_like_the_reference(mpf, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    mpf.build()
    _build()
    f = tmp_path / "m.txt"
    subprocess.run([os.path.join(H, "mpf_matgen"), str(f), "16", "2", "exp"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(H, "mpf_bench"), str(f)], cwd=tmp_path, capture_output=True, text=True)
    # no device: MPF() prints on stderr and leaves the buffers untouched (MPF.cu:72-75), so the check complains
    assert "No HIP devices available." in r.stderr
    assert "MPF produced incorrect results." in r.stdout
    csv = (tmp_path / "benchmark_times.csv").read_text().splitlines()
    assert csv[0] == "matrix_size,mpf_time,lapack_time" and [l.split(",")[0] for l in csv[1:]] == ["2", "4", "8", "16"]


@pytest.mark.gpu
def test_bench_cli_on_gpu(mpf, tmp_path):
    mpf.build()
    _build()
    f = tmp_path / "m.txt"
    subprocess.run([os.path.join(H, "mpf_matgen"), str(f), "512", "2", "exp"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(H, "mpf_bench"), str(f)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "incorrect" not in r.stdout
    csv = (tmp_path / "benchmark_times.csv").read_text().splitlines()
    assert csv[0] == "matrix_size,mpf_time,lapack_time" and len(csv) == 1 + 9


@pytest.mark.gpu
def test_bench_cli_verbose_dumps_and_device_side_check(mpf, tmp_path):
    """-v prints what benchmark.cpp:27-57,114-139 prints for n < 10 (L, U, LU, PLU, Correctitude); from n = 2048 on the
    L * U of the check runs on the GPU (mpf_check_plu_host) -- the generator's 4096 file exercises both paths."""
    mpf.build()
    _build()
    f = tmp_path / "m.txt"
    subprocess.run([os.path.join(H, "mpf_matgen"), str(f), "8", "2", "exp"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(H, "mpf_bench"), str(f), "-v"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for key in ("Number of matrices: 3", "Original matrix:", "L matrix:", "U matrix:", "LU matrix:", "PLU matrix:", "Correctitude: True"):
        assert key in r.stdout, key
    assert "Correctitude: False" not in r.stdout and "incorrect" not in r.stdout
    subprocess.run([os.path.join(H, "mpf_matgen"), str(f), "4096", "2", "exp"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(H, "mpf_bench"), str(f), "-v", "-r", "128"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "incorrect" not in r.stdout and r.stdout.count("(L * U on the device)") == 4      # n = 2048, 4096 x (MPF, dgetrf)
    assert "Correctitude: False" not in r.stdout

"""bench.py command-line behaviour that needs no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how the driver calls bench.py) must launch torch.distributed.run as
    a child and relay its return code.  Without a GPU every rank stops at 'No HIP GPUs are available': that message coming from
    the ranks (twice: one per rank) shows the launch happened; the exit code is the launcher's (non-zero), not a usage error."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MPF_BENCH_REHEARSAL="1", MPF_BENCH_N="256", MPF_BENCH_NB="64", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "must run under torch.distributed.run" not in out.stderr
    assert out.stderr.count("No HIP GPUs are available") + out.stderr.count("bench.py needs a GPU") >= 2, out.stderr[-3000:]


def test_cpu_only_leg_prints_one_object():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-only", "--cpu-n", "512"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["value"] > 0 and d["kind"] == "reference" and d["cores"] >= 1

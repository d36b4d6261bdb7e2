"""GPU checks of the multi-GPU schedule with the real HIP step operators.  The box has ONE GPU, so the
2-rank case shares cuda:0 between two processes and moves the panel through gloo + host memory; the
RCCL path itself (bench.py --gpus N) is exercised by the driver on the 8-GPU node."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dist_schedule_world1_matches_oracle(ctx, oracle, mpf):
    D = importlib.import_module("mixed-precision_lu_factorization_amd.dist")
    for n, nb in ((300, 64), (1024, 256), (700, 128)):
        A = oracle.matgen_skip(n, skip=9 + n)
        lay = D.BlockCyclic(n, nb, 0, 1)
        loc = D.scatter_columns(ctx.from_numpy_f(A), lay, ctx.device)
        ipiv = D.factor(ctx, loc, lay)
        ctx.synchronize()
        LU_o, ip_o = oracle.mpf(A, nb)
        # depth-1 look-ahead with a side-stream context gives the same bits
        import torch
        side = mpf.MPFContext(0, stream=torch.cuda.Stream(device=ctx.device, priority=-1))
        loc2 = D.scatter_columns(ctx.from_numpy_f(A), lay, ctx.device)
        ipiv2 = D.factor_lookahead(ctx, side, loc2, lay)
        torch.cuda.synchronize()
        assert torch.equal(ipiv, ipiv2) and torch.equal(loc, loc2)
        side.close()
        assert np.array_equal(ipiv.cpu().numpy(), ip_o)
        assert np.array_equal(ctx.to_numpy_f(loc).view(np.uint64), LU_o.view(np.uint64))


def _worker(rank, world, port, n, nb, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
    D = importlib.import_module("mixed-precision_lu_factorization_amd.dist")
    from oracle import oracle as O
    ctx = mpf.MPFContext(0)
    A = O.matgen_skip(n, skip=4 + n)
    lay = D.BlockCyclic(n, nb, rank, world)
    loc = D.scatter_columns(ctx.from_numpy_f(A), lay, ctx.device)
    import torch as _t
    side = mpf.MPFContext(0, stream=_t.cuda.Stream(device=ctx.device, priority=-1))
    ipiv = D.factor_lookahead(ctx, side, loc, lay, host_staged_bcast=True)
    _t.cuda.synchronize()
    ctx.synchronize()
    full = torch.zeros((n, n), dtype=torch.float64).t()
    lc = loc.cpu()
    for b in lay.my_blocks:
        w = lay.width(b)
        full[:, b * nb:b * nb + w] = lc[:, lay.local_col(b):lay.local_col(b) + w]
    flat = full.t().contiguous()
    dist.all_reduce(flat)
    if rank == 0:
        np.save(out + "_lu.npy", np.asfortranarray(flat.numpy().T))
        np.save(out + "_ip.npy", ipiv.cpu().numpy())
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,nb", [(2, 1024, 128), (3, 900, 64)])
def test_dist_two_ranks_share_one_gpu(oracle, tmp_path, world, n, nb):
    port = 29700 + (os.getpid() % 1000) + world
    out = str(tmp_path / "g")
    mp.spawn(_worker, args=(world, port, n, nb, out), nprocs=world, join=True)
    LU_o, ip_o = oracle.mpf(oracle.matgen_skip(n, skip=4 + n), nb)
    assert np.array_equal(np.load(out + "_ip.npy"), ip_o)
    assert np.array_equal(np.asfortranarray(np.load(out + "_lu.npy")).view(np.uint64), LU_o.view(np.uint64))


def test_bench_multi_rank_path_rehearsal(tmp_path):
    """bench.py --gpus 2 end to end (torch.distributed.run, barrier + MAX-over-ranks timing, one JSON line with the
    contract's keys and a roofline object), rehearsed here with both ranks on the one visible GPU over gloo: the RCCL run
    on distinct GPUs is the driver's."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPF_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", MPF_BENCH_N="2048", MPF_BENCH_NB="128")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["pivots_consistent_across_ranks"] is True
    assert d["roofline"] is not None and d["roofline"]["achieved"] > 0 and "workload" in d["config"]

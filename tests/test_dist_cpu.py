"""The N>1 path on CPU: world_size 2 and 3 over gloo, the model of the distributed schedule (tests/dist_model.py: the
product's layout arithmetic, message sequence and step order) driven by oracle-backed step operators, must reproduce the
single-process oracle bit for bit (IPIV and LU).  The product schedule (C++, mpf_factor_dist) runs the same steps with the HIP
kernels: tests/test_gpu_dist.py."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, nb, out, lookahead):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = importlib.import_module("mixed-precision_lu_factorization_amd.dist")
    from dist_oracle_kernels import OracleKernels
    import dist_model as M
    from oracle import oracle as O
    A = O.matgen_skip(n, skip=4 + n)
    full = torch.from_numpy(np.ascontiguousarray(A.T)).t()
    lay = D.BlockCyclic(n, nb, rank, world)
    loc = D.scatter_columns(full, lay, torch.device("cpu"))
    K = OracleKernels()
    ipiv = M.factor_lookahead(K, K, loc, lay) if lookahead else M.factor(K, loc, lay)
    LU = D.gather_columns(loc, lay)
    if rank == 0:
        np.save(out + "_lu.npy", np.asfortranarray(LU.t().contiguous().numpy().T))
        np.save(out + "_ip.npy", ipiv.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lookahead", [False, True])
@pytest.mark.parametrize("world,n,nb", [(2, 200, 32), (3, 257, 64), (2, 96, 32), (2, 130, 128)])
def test_distributed_schedule_matches_single_process(oracle, tmp_path, world, n, nb, lookahead):
    port = 29500 + (os.getpid() % 2000) + world * 7 + n % 13
    out = str(tmp_path / "r")
    mp.spawn(_worker, args=(world, port + int(lookahead) * 101, n, nb, out, lookahead), nprocs=world, join=True)
    A = oracle.matgen_skip(n, skip=4 + n)
    LU_o, ip_o = oracle.mpf(A, nb)
    LU = np.load(out + "_lu.npy")
    ip = np.load(out + "_ip.npy")
    assert np.array_equal(ip, ip_o)
    assert np.array_equal(np.asfortranarray(LU).view(np.uint64), LU_o.view(np.uint64))


def _solve_worker(rank, world, port, n, nb, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = importlib.import_module("mixed-precision_lu_factorization_amd.dist")
    import dist_model as M
    from oracle import oracle as O
    A = O.matgen_skip(n, skip=11 + n)
    LU, ip = O.mpf(A, nb)
    perm = np.arange(n)
    for i in range(n):                                   # LAPACK-style swap list -> row permutation (benchmark.cpp:84-95 undone)
        p = int(ip[i]) - 1
        perm[i], perm[p] = perm[p], perm[i]
    lay = D.BlockCyclic(n, nb, rank, world)
    full = torch.from_numpy(np.ascontiguousarray(LU.T)).t()
    loc = D.scatter_columns(full, lay, torch.device("cpu"))
    b = torch.from_numpy(A @ np.ones(n))
    x, msgs = M.lu_solve_chain(loc, lay, torch.from_numpy(perm), b)
    if rank == 0:
        np.save(out + "_x.npy", x.numpy())
    tot = torch.tensor([msgs], dtype=torch.int64)
    dist.all_reduce(tot)
    if rank == 0:
        np.save(out + "_msgs.npy", tot.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,nb", [(2, 200, 32), (3, 257, 64), (3, 130, 128), (2, 96, 32)])
def test_owner_to_owner_triangular_solves(oracle, tmp_path, world, n, nb):
    """The point-to-point chain of the distributed solves (csrc/mpf_dist.cpp lu_solve_chain, modelled in tests/dist_model.py) over
    gloo with 2 and 3 ranks: the solution of A x = A 1 from the oracle's factors, 2 (nblocks - 1) messages per solve (each counted
    by its sender and its receiver), none of them a broadcast."""
    port = 29800 + (os.getpid() % 2000) + world * 5 + n % 11
    out = str(tmp_path / "s")
    mp.spawn(_solve_worker, args=(world, port, n, nb, out), nprocs=world, join=True)
    x = np.load(out + "_x.npy")
    assert np.max(np.abs(x - 1.0)) < 1e-8
    nblocks = (n + nb - 1) // nb
    assert int(np.load(out + "_msgs.npy")[0]) == 2 * 2 * (nblocks - 1)


def test_block_cyclic_layout():
    D = importlib.import_module("mixed-precision_lu_factorization_amd.dist")
    n, nb, world = 1000, 128, 3
    seen = []
    for r in range(world):
        lay = D.BlockCyclic(n, nb, r, world)
        seen += lay.my_blocks
        assert lay.local_cols() == sum(min(nb, n - b * nb) for b in lay.my_blocks)
        for b in range(lay.nblocks):
            t0 = lay.first_local_col_after(b)
            assert t0 == sum(min(nb, n - x * nb) for x in lay.my_blocks if x <= b)
    assert sorted(seen) == list(range((n + nb - 1) // nb))

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
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_model as M
    for n, nb in ((300, 64), (1024, 256), (700, 128)):
        A = oracle.matgen_skip(n, skip=9 + n)
        lay = D.BlockCyclic(n, nb, 0, 1)
        loc = D.scatter_columns(ctx.from_numpy_f(A), lay, ctx.device)
        ipiv = M.factor(ctx, loc, lay)
        ctx.synchronize()
        LU_o, ip_o = oracle.mpf(A, nb)
        # depth-1 look-ahead with a side-stream context gives the same bits
        import torch
        side = mpf.MPFContext(0, stream=torch.cuda.Stream(device=ctx.device, priority=-1))
        loc2 = D.scatter_columns(ctx.from_numpy_f(A), lay, ctx.device)
        ipiv2 = M.factor_lookahead(ctx, side, loc2, lay)
        torch.cuda.synchronize()
        assert torch.equal(ipiv, ipiv2) and torch.equal(loc, loc2)
        side.close()
        assert np.array_equal(ipiv.cpu().numpy(), ip_o)
        assert np.array_equal(ctx.to_numpy_f(loc).view(np.uint64), LU_o.view(np.uint64))


def test_cxx_dist_loop_world1_all_modes(ctx, oracle, mpf):
    """mpf_factor_dist (the C++ host loop) with a single rank: fp64 mode bit-exact against the oracle; the fp16 modes equal
    the single-GPU schedules (two-level default, one-level) bit for bit; mpf_solve_ir_dist refines on the same layout."""
    D = importlib.import_module("mixed-precision_lu_factorization_amd.dist")
    one = mpf.MpfDist(rank=0, world=1)
    # By default ONE rank hands over to mpf_factor_dev (the single-GPU schedules); dist_world1_loop = 1 keeps it in the distributed
    # loop, which is what this test is about.  fp64_rowmajor_min_n = 0: the loop's row-major working copy at these small sizes too.
    c1 = mpf.MPFContext(0, options={"dist_world1_loop": 1})
    c1r = mpf.MPFContext(0, options={"dist_world1_loop": 1, "fp64_rowmajor_min_n": 0})
    for n, nb in ((300, 64), (1024, 256), (700, 128), (130, 64)):
        A = oracle.matgen_skip(n, skip=9 + n)
        dA = ctx.from_numpy_f(A)
        W = dA.clone()
        ipiv, info = c1.factor_dist(W, n, nb, one)
        LU_o, ip_o = oracle.mpf(A, nb)
        assert info == 0 and np.array_equal(ipiv.cpu().numpy(), ip_o)
        assert np.array_equal(ctx.to_numpy_f(W).view(np.uint64), LU_o.view(np.uint64))
        Wr = dA.clone()
        ipr, _ = c1r.factor_dist(Wr, n, nb, one)                      # row-major working copy of the local columns
        assert torch.equal(ipiv, ipr) and torch.equal(W, Wr)
        Wd = dA.clone()
        ipd, _ = ctx.factor_dist(Wd, n, nb, one)                      # default: handed over to mpf_factor_dev
        assert torch.equal(ipiv, ipd) and torch.equal(W, Wd)
        W2 = dA.clone()
        ipiv2, _ = c1.factor_dist(W2, n, nb, one, no_lookahead=True)
        assert torch.equal(ipiv, ipiv2) and torch.equal(W, W2)
        W3 = dA.clone()
        ipiv3, _ = c1.factor_dist(W3, n, nb, one, pivot_path=1)     # generic pivots + planned interchange list
        assert torch.equal(ipiv, ipiv3) and torch.equal(W, W3)
        xs = torch.ones(n, dtype=torch.float64, device=ctx.device)
        x, st = ctx.solve_ir_dist(dA, W, ipiv, dA @ xs, n, nb, one, max_iter=3, tol=1e-12)
        assert st.converged == 1 and float((x - xs).abs().max()) < 1e-6
    # fp16 modes: the distributed loop runs the same two-level schedule as the single-GPU driver (super-panels, fp32 working copy
    # of the far columns, block-row tasks, K = sb * nb updates) -- same operations per element, same bits; likewise one-level
    for n, nb in ((1536, 128), (1500, 96), (2048, 256)):
        A = oracle.matgen_skip(n, skip=3)
        dA = ctx.from_numpy_f(A)
        for mode in (mpf.TRAIL_FP16, mpf.TRAIL_FP16X3):
            for sb in (0, 1, 2):
                W, V = dA.clone(), dA.clone()
                p1, _ = c1.factor_dist(W, n, nb, one, trailing=mode, superpanel=sb)
                assert c1.stats().superpanel == (sb if sb else 4)
                p2, _ = ctx.factor(V, nb, trailing=mode, superpanel=sb)
                assert ctx.stats().superpanel == (sb if sb else 4)
                assert torch.equal(p1, p2) and torch.equal(W, V), (n, nb, mode, sb)
            W, V = dA.clone(), dA.clone()
            p1, _ = c1.factor_dist(W, n, nb, one, trailing=mode, no_lookahead=True)
            p2, _ = ctx.factor(V, nb, trailing=mode)
            assert torch.equal(p1, p2) and torch.equal(W, V), (n, nb, mode, "one stream")
    c1.close(); c1r.close()


def test_cxx_dist_loop_wide_panels_world1(ctx, oracle, mpf):
    """Panels wider than 256 columns in the distributed loop (generic pivot kernels, the reference's sequential interchange of all
    local columns, one stream): fp64 bit-exact against the oracle, the fp16 modes equal mpf_factor_dev's generic schedule."""
    one = mpf.MpfDist(rank=0, world=1)
    c1 = mpf.MPFContext(0, options={"dist_world1_loop": 1})
    for n, nb in ((1536, 512), (1100, 320)):
        A = oracle.matgen_skip(n, skip=5 + n)
        dA = ctx.from_numpy_f(A)
        W = dA.clone()
        ipiv, info = c1.factor_dist(W, n, nb, one)
        LU_o, ip_o = oracle.mpf(A, nb)
        assert info == 0 and np.array_equal(ipiv.cpu().numpy(), ip_o)
        assert np.array_equal(ctx.to_numpy_f(W).view(np.uint64), LU_o.view(np.uint64))
        for mode in (mpf.TRAIL_FP16, mpf.TRAIL_FP16X3):
            W, V = dA.clone(), dA.clone()
            p1, _ = c1.factor_dist(W, n, nb, one, trailing=mode)
            p2, _ = ctx.factor(V, nb, trailing=mode)
            assert torch.equal(p1, p2) and torch.equal(W, V), (n, nb, mode)
    c1.close()


@pytest.mark.parametrize("world,n,nb", [(2, 1600, 320), (3, 2048, 512)])
def test_cxx_dist_loop_wide_panels_ranks_share_one_gpu(oracle, tmp_path, world, n, nb):
    port = 29200 + (os.getpid() % 1000) + world
    out = str(tmp_path / "w")
    mp.spawn(_cxx_worker, args=(world, port, n, nb, 0, out), nprocs=world, join=True)
    LU_o, ip_o = oracle.mpf(oracle.matgen_skip(n), nb)
    assert np.array_equal(np.load(out + "_ip.npy"), ip_o)
    assert np.array_equal(np.asfortranarray(np.load(out + "_lu.npy")).view(np.uint64), LU_o.view(np.uint64))
    conv, its, rel, err, info, msgs = np.load(out + "_ir.npy")
    assert conv == 1 and its <= 1 and rel <= 1e-12 and info == 0 and msgs == (n + nb - 1) // nb


def test_rccl_transport_loads_and_runs_on_one_rank(ctx, mpf):
    """The built-in transport: librccl resolved with dlopen, a communicator of one rank, one broadcast and one all-reduce
    through it (the multi-rank RCCL run is the driver's, on the 8-GPU node)."""
    L = mpf.load_library()
    v = L.mpf_rccl_version()
    assert v > 20000, v
    ctx.rccl_init(0, 1)
    assert L.mpf_rccl_selftest(ctx.h) == 0, L.mpf_last_error(ctx.h)
    assert L.mpf_rccl_destroy(ctx.h) == 0


def _torch_transport_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MPF_DIST_TRANSPORT"] = "torch"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
    D = importlib.import_module("mixed-precision_lu_factorization_amd.dist")
    ctx = mpf.MPFContext(0)
    cfg, name, keep = D.pick_transport(ctx, rank, world, ctx.device)
    ok = isinstance(keep, D.TorchDist) and "torch.distributed" in name
    # the callbacks on the library's kind of arguments: raw device pointers, byte / element counts, a HIP stream handle
    side = torch.cuda.Stream(device=ctx.device)
    buf = torch.arange(4096, dtype=torch.uint8, device=ctx.device)
    vec = torch.arange(1000, dtype=torch.float64, device=ctx.device)
    side.wait_stream(torch.cuda.current_stream())
    rc1 = keep._keep[0](None, buf.data_ptr(), buf.numel(), 0, side.cuda_stream)
    rc2 = keep._keep[1](None, vec.data_ptr(), vec.numel(), side.cuda_stream)
    side.synchronize()
    ok = ok and rc1 == 0 and rc2 == 0 and keep.messages == 1 and keep.bytes == 4096
    ok = ok and torch.equal(buf.cpu(), torch.arange(4096, dtype=torch.uint8)) and torch.equal(vec.cpu(), torch.arange(1000, dtype=torch.float64))
    # and the C++ loop driven through them (one rank: same result as mpf_factor_dev)
    n, nb = 768, 128
    A = ctx.matgen(n)
    W, V = A.clone(), A.clone()
    p1, i1 = ctx.factor_dist(W, n, nb, cfg)
    p2, i2 = ctx.factor(V, nb)
    ok = ok and torch.equal(p1, p2) and torch.equal(W, V) and i1 == 0
    b = A @ torch.ones(n, dtype=torch.float64, device=ctx.device)
    x, st = ctx.solve_ir_dist(A, W, p1, b, n, nb, cfg, max_iter=5, tol=1e-12)
    ok = ok and st.converged == 1
    np.save(out, np.array([1 if ok else 0]))
    ctx.close()
    dist.destroy_process_group()


def test_torch_distributed_transport_on_device_buffers(tmp_path):
    """The fall-back transport of bench.py --gpus N (TorchDist: the library's device buffers handed to torch.distributed /
    RCCL, ordered on the library's stream): a one-rank nccl group on the one GPU here."""
    out = str(tmp_path / "t.npy")
    mp.spawn(_torch_transport_worker, args=(1, 29700 + (os.getpid() % 200), out), nprocs=1, join=True)
    assert np.load(out)[0] == 1


def _cxx_worker(rank, world, port, n, nb, mode, out, options=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
    D = importlib.import_module("mixed-precision_lu_factorization_amd.dist")
    ctx = mpf.MPFContext(0, options={k: v for k, v in (options or {}).items() if not k.startswith("_")})
    lay = D.BlockCyclic(n, nb, rank, world)
    A0 = D.colmajor_empty(n, lay.local_cols(), ctx.device)
    for b in lay.my_blocks:                      # every rank generates its own blocks of the reference generator's matrix
        ctx.matgen(n, out=A0[:, lay.local_col(b):lay.local_col(b) + lay.width(b)], col0=b * nb, ncols=lay.width(b))
    loc = A0.clone()
    gd = D.GlooDist(rank, world)
    if options and options.get("_p2p"):       # (test switch, not a library option) the solves' point-to-point chain
        gd.attach_p2p(ctx)
    ipiv, info = ctx.factor_dist(loc, n, nb, gd.c, trailing=mode)
    factor_msgs = gd.messages      # messages of the factorization alone (the refinement below adds its own)
    xs = torch.ones(n, dtype=torch.float64, device=ctx.device)
    bl = (A0 @ torch.ones(lay.local_cols(), dtype=torch.float64, device=ctx.device)).cpu() if lay.local_cols() else torch.zeros(n, dtype=torch.float64)
    dist.all_reduce(bl)
    if nb % 64 == 0:
        x, st = ctx.solve_ir_dist(A0, loc, ipiv, bl.to(ctx.device), n, nb, gd.c, max_iter=10, tol=1e-12)
    else:               # (the distributed solve walks 64-column steps: factor-only check for other widths)
        x, st = xs, mpf.MpfIrStats(converged=-1)
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
        np.save(out + "_ir.npy", np.array([st.converged, st.iterations, st.rel_residual, float((x - xs).abs().max()), info, factor_msgs]))
        np.save(out + "_p2p.npy", np.array([gd.p2p_messages, gd.messages - factor_msgs]))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("rowmajor", [False, True])
@pytest.mark.parametrize("world,n,nb", [(2, 1024, 128), (3, 960, 64), (2, 1100, 128)])
def test_cxx_dist_loop_ranks_share_one_gpu(oracle, tmp_path, world, n, nb, rowmajor):
    """The C++ loop with 2 / 3 ranks (processes) on the one GPU, messages carried by the gloo callbacks: IPIV and all N^2
    values equal the oracle's (= the 1-GPU result), the distributed refinement converges, one message per panel (+ the round in
    which the ranks agree on the schedule's rank-local inputs: one small broadcast per rank).  rowmajor: every rank keeps its
    columns right of the panel in a row-major working copy (forced at this size with fp64_rowmajor_min_n = 0)."""
    port = 29900 + (os.getpid() % 1000) + world + (7 if rowmajor else 0)
    out = str(tmp_path / "c")
    mp.spawn(_cxx_worker, args=(world, port, n, nb, 0, out, {"fp64_rowmajor_min_n": 0} if rowmajor else {"fp64_rowmajor": 0}), nprocs=world, join=True)
    LU_o, ip_o = oracle.mpf(oracle.matgen_skip(n), nb)
    assert np.array_equal(np.load(out + "_ip.npy"), ip_o)
    assert np.array_equal(np.asfortranarray(np.load(out + "_lu.npy")).view(np.uint64), LU_o.view(np.uint64))
    conv, its, rel, err, info, msgs = np.load(out + "_ir.npy")
    assert conv == 1 and its <= 1 and rel <= 1e-12 and err < 1e-6 and info == 0
    npanels = (n + nb - 1) // nb
    assert msgs == npanels + world  # ONE message per panel + the agreement round


@pytest.mark.parametrize("world,n,nb", [(2, 1024, 128), (3, 1536, 128), (3, 1280, 256)])
def test_dist_solve_passes_the_vector_from_owner_to_owner(oracle, tmp_path, world, n, nb):
    """mpf_solve_ir_dist with a point-to-point transport registered (mpf_dist_set_p2p): the triangular solves send the running
    vector from the owner of a block to the owner of the next one -- this rank's share of 2 (nblocks - 1) sends per solve, and ONE
    all-reduce -- instead of a broadcast to every rank after every block; the refined solution is the same."""
    port = 29600 + (os.getpid() % 1000) + world
    out = str(tmp_path / "p")
    mp.spawn(_cxx_worker, args=(world, port, n, nb, 0, out, {"_p2p": 1}), nprocs=world, join=True)
    conv, its, rel, err, info, msgs = np.load(out + "_ir.npy")
    assert conv == 1 and its <= 1 and rel <= 1e-12 and err < 1e-6 and info == 0
    p2p_msgs, bcasts_in_solves = np.load(out + "_p2p.npy")
    nblocks = (n + nb - 1) // nb
    assert bcasts_in_solves == 0                      # no broadcast in the solves any more
    assert p2p_msgs > 0 and p2p_msgs <= 2 * (its + 1) * 2 * nblocks


@pytest.mark.parametrize("world,n,nb", [(2, 1024, 128), (3, 1280, 256), (2, 1100, 64)])
def test_cxx_dist_loop_message_in_instalments(oracle, tmp_path, world, n, nb):
    """The panel message in 32-column instalments (each leaves as soon as its sub-panel of the fp64 panel is done, carries its
    own pivots, and every rank applies the later sub-panels' interchanges to the instalments it already holds): forced for every
    panel here (dist_instalment_min_bytes = 0).  Same bits as the oracle, nb / 32 messages per pipelined panel."""
    port = 29400 + (os.getpid() % 1000) + world
    out = str(tmp_path / "i")
    mp.spawn(_cxx_worker, args=(world, port, n, nb, 0, out, {"dist_instalment_min_bytes": 0}), nprocs=world, join=True)
    LU_o, ip_o = oracle.mpf(oracle.matgen_skip(n), nb)
    assert np.array_equal(np.load(out + "_ip.npy"), ip_o)
    assert np.array_equal(np.asfortranarray(np.load(out + "_lu.npy")).view(np.uint64), LU_o.view(np.uint64))
    conv, its, rel, err, info, msgs = np.load(out + "_ir.npy")
    assert conv == 1 and rel <= 1e-12 and info == 0
    full_panels = sum(1 for b in range(1, (n + nb - 1) // nb) if n - b * nb >= nb)   # panel 0 goes in one piece
    assert msgs - world == ((n + nb - 1) // nb - full_panels) + full_panels * (nb // 32) - (0 if (n % nb) != 1 else 1) or msgs - world >= full_panels * (nb // 32)


def test_cxx_dist_loop_fp16x3_mode_two_ranks(oracle, tmp_path):
    import importlib as _il
    mpfm = _il.import_module("mixed-precision_lu_factorization_amd")
    n, nb, world = 1536, 128, 2
    port = 29950 + (os.getpid() % 1000)
    out = str(tmp_path / "h")
    mp.spawn(_cxx_worker, args=(world, port, n, nb, mpfm.TRAIL_FP16X3, out), nprocs=world, join=True)
    A = oracle.matgen_skip(n)
    _, fro = oracle.check_plu(A, np.asfortranarray(np.load(out + "_lu.npy")), np.load(out + "_ip.npy"))
    assert fro < 1e-5, fro
    conv, its, rel, err, info, msgs = np.load(out + "_ir.npy")
    assert conv == 1 and its <= 6 and rel <= 1e-12


@pytest.mark.parametrize("world,n,nb,mode", [(2, 1536, 128, 1), (3, 2048, 128, 2), (2, 1500, 96, 2)])
def test_cxx_dist_loop_two_level_fp16_ranks_share_one_gpu(ctx, oracle, tmp_path, world, n, nb, mode):
    """The two-level fp16 schedule across ranks: every rank keeps its far columns in an fp32 working copy and the current
    super-panel's panels in a store assembled from the messages.  Per element the operations are the single-GPU two-level
    schedule's, whatever the column split: IPIV and all N^2 values equal mpf_factor_dev's bit for bit."""
    port = 29300 + (os.getpid() % 1000) + world
    out = str(tmp_path / "t")
    mp.spawn(_cxx_worker, args=(world, port, n, nb, mode, out), nprocs=world, join=True)
    A = ctx.matgen(n)
    p2, info = ctx.factor(A, nb, trailing=mode)
    assert info == 0 and ctx.stats().superpanel == 4
    assert np.array_equal(np.load(out + "_ip.npy"), p2.cpu().numpy())
    assert np.array_equal(np.asfortranarray(np.load(out + "_lu.npy")).view(np.uint64), ctx.to_numpy_f(A).view(np.uint64))
    conv, its, rel, err, info, msgs = np.load(out + "_ir.npy")
    assert info == 0
    if mode == 2 and nb % 64 == 0:       # (plain fp16 products do not refine on the generator's matrix: config 2 uses a dominant one)
        assert conv == 1 and its <= 6 and rel <= 1e-12


def _worker(rank, world, port, n, nb, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
    D = importlib.import_module("mixed-precision_lu_factorization_amd.dist")
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_model as M
    ctx = mpf.MPFContext(0)
    A = O.matgen_skip(n, skip=4 + n)
    lay = D.BlockCyclic(n, nb, rank, world)
    loc = D.scatter_columns(ctx.from_numpy_f(A), lay, ctx.device)
    import torch as _t
    side = mpf.MPFContext(0, stream=_t.cuda.Stream(device=ctx.device, priority=-1))
    ipiv = M.factor_lookahead(ctx, side, loc, lay, host_staged_bcast=True)
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
    assert d["ir"] is not None and d["ir"]["converged"] and d["ir"]["rel_residual"] <= 1e-12      # the metric's second half
    assert d["cpu_baseline"] is None      # the contract times the CPU leg on rank 0 at N = 1 only
    # the fp16-mode leg on the same layout: the two-level schedule ran on both ranks and refines in a sweep or two
    assert d["mxp"]["superpanel"] == 4 and d["mxp"]["info"] == 0 and d["mxp"]["ir_converged"] and d["mxp"]["ir_iterations"] <= 3
    assert d["mxp"]["rank0_big_update"]["launches"] > 0


def test_bench_self_launch_plain_python_rehearsal():
    """`python bench.py --gpus 2 ...` with WORLD_SIZE unset -- the shape of the driver's N = 1 command -- starts
    torch.distributed.run itself as a child process (before torch is imported: no exec after a GPU call) and relays rank 0's
    one JSON line and the return code."""
    import json, subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MPF_BENCH_REHEARSAL="1", MPF_BENCH_N="2048", MPF_BENCH_NB="128")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["value"] > 0 and d["pivots_consistent_across_ranks"] is True

"""Multi-GPU MPF: 1-D block-cyclic column layout, one process per GPU, panel broadcast over RCCL/xGMI.

The product path is C++ behind the C ABI (csrc/mpf_dist.cpp: mpf_factor_dist, mpf_solve_ir_dist); this module holds
  * the layout arithmetic and scatter / gather helpers,
  * the transports the C++ loop is driven with: `rccl_dist` (the context's own RCCL communicator: ncclBroadcast /
    ncclAllReduce) and `gloo_dist` (ctypes callbacks that stage through host memory and torch.distributed's gloo backend:
    several ranks on ONE GPU in the tests, or CPU-side rehearsals of more ranks),
  * (the schedule itself exists ONCE, in C++; a Python model of it over step operators, which the CPU tests run on the
    oracle's step operators with gloo, lives with the tests: tests/dist_model.py),
  * `bench_main`: what `bench.py --gpus N` runs under torch.distributed.run.

The reference is single-device (MPF.cu:77); this partition is the build's extension.  A column block of `nb`
columns lives entirely on one rank, so the fp16 pivot panel, its pivot search and the fp64 no-pivot panel stay local to
the owner; per panel the ONLY exchange step is one broadcast, owner -> all.  Arithmetic per element is identical to the
1-GPU path, so IPIV and LU are bit-identical to mpf_factor_dev.  This module never imports the oracle.
"""
import time

import torch
import torch.distributed as dist


class BlockCyclic:
    """Column-block ownership: global block b -> rank b % world, local index b // world."""

    def __init__(self, n, nb, rank, world):
        self.n, self.nb, self.rank, self.world = n, nb, rank, world
        self.nblocks = (n + nb - 1) // nb
        self.my_blocks = list(range(rank, self.nblocks, world))

    def owner(self, b):
        return b % self.world

    def width(self, b):
        return min(self.nb, self.n - b * self.nb)

    def local_col(self, b):
        """first local column of (owned) global block b"""
        assert b % self.world == self.rank
        return (b // self.world) * self.nb

    def local_cols(self):
        return sum(self.width(b) for b in self.my_blocks)

    def first_local_col_after(self, b):
        """first local column whose global block index is > b (trailing part), == local_cols() if none"""
        cnt = len([x for x in self.my_blocks if x <= b])
        return min(cnt * self.nb, self.local_cols())


def colmajor_empty(rows, cols, device, dtype=torch.float64):
    return torch.empty((max(cols, 1), rows), dtype=dtype, device=device).t()[:, :cols]


def scatter_columns(A_full, layout, device):
    """Local column blocks of a full column-major matrix (every rank passes the same A_full, or builds its
    blocks some other way)."""
    loc = colmajor_empty(layout.n, layout.local_cols(), device)
    for b in layout.my_blocks:
        w = layout.width(b)
        lc = layout.local_col(b)
        loc[:, lc:lc + w] = A_full[:, b * layout.nb: b * layout.nb + w].to(device)
    return loc


def gather_columns(loc, layout, group=None):
    """Inverse of scatter_columns on rank 0 (testing aid): returns the full matrix on every rank."""
    n, nb = layout.n, layout.nb
    full = torch.zeros((n, n), dtype=loc.dtype, device=loc.device).t()
    for b in layout.my_blocks:
        w = layout.width(b)
        full[:, b * nb:b * nb + w] = loc[:, layout.local_col(b):layout.local_col(b) + w]
    flat = full.t().contiguous()
    dist.all_reduce(flat, group=group)  # blocks are disjoint: the sum assembles the matrix
    return flat.t()


# -------------------------------------------------------------------------------------------------------------
# transports for the C++ loop (mpf_factor_dist / mpf_solve_ir_dist)
# -------------------------------------------------------------------------------------------------------------
def rccl_dist(ctx, rank, world, group=None):
    """mpf_dist that uses the context's RCCL communicator (created here; id exchanged through torch.distributed)."""
    import importlib
    mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
    if world > 1:
        ctx.rccl_init(rank, world, group=group)
    return mpf.MpfDist(rank=rank, world=world)   # NULL callbacks: built-in RCCL transport


class GlooDist:
    """mpf_dist whose callbacks stage every message through host memory and a gloo process group.  For several ranks on
    one GPU (tests, rehearsals): slow by construction, same message sequence as the RCCL transport."""

    def __init__(self, rank, world, group=None):
        import ctypes as C
        import importlib
        import numpy as np
        mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
        hip = C.CDLL("libamdhip64.so")   # the runtime torch already loaded
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.messages = 0
        self.bytes = 0

        def bcast(user, d_buf, nbytes, root, stream):
            try:
                if hip.hipStreamSynchronize(stream) != 0:
                    return -2
                host = np.empty(nbytes, dtype=np.uint8)
                if rank == root and hip.hipMemcpy(host.ctypes.data, d_buf, nbytes, 2) != 0:   # hipMemcpyDeviceToHost
                    return -2
                th = torch.from_numpy(host)
                dist.broadcast(th, src=root, group=group)
                if rank != root and hip.hipMemcpy(d_buf, host.ctypes.data, nbytes, 1) != 0:   # hipMemcpyHostToDevice
                    return -2
                self.messages += 1
                self.bytes += int(nbytes)
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("gloo bcast callback failed:", e, flush=True)
                return -5

        def allreduce(user, d_buf, count, stream):
            try:
                if hip.hipStreamSynchronize(stream) != 0:
                    return -2
                host = np.empty(count, dtype=np.float64)
                if hip.hipMemcpy(host.ctypes.data, d_buf, count * 8, 2) != 0:
                    return -2
                th = torch.from_numpy(host)
                dist.all_reduce(th, group=group)
                if hip.hipMemcpy(d_buf, host.ctypes.data, count * 8, 1) != 0:
                    return -2
                return 0
            except Exception as e:
                print("gloo allreduce callback failed:", e, flush=True)
                return -5

        def p2p(user, d_buf, nbytes, peer, send, stream):
            try:
                if hip.hipStreamSynchronize(stream) != 0:
                    return -2
                host = np.empty(nbytes, dtype=np.uint8)
                th = torch.from_numpy(host)
                if send:
                    if hip.hipMemcpy(host.ctypes.data, d_buf, nbytes, 2) != 0:
                        return -2
                    dist.send(th, dst=peer, group=group)
                else:
                    dist.recv(th, src=peer, group=group)
                    if hip.hipMemcpy(d_buf, host.ctypes.data, nbytes, 1) != 0:
                        return -2
                self.p2p_messages += 1
                return 0
            except Exception as e:
                print("gloo p2p callback failed:", e, flush=True)
                return -5

        self.p2p_messages = 0
        self._keep = (mpf.BCAST_FN(bcast), mpf.ALLREDUCE_FN(allreduce), mpf.P2P_FN(p2p))   # the C side holds raw pointers to these
        self.c = mpf.MpfDist(rank=rank, world=world, bcast=self._keep[0], allreduce=self._keep[1], user=None)

    def attach_p2p(self, ctx):
        """Register the point-to-point callback with a context: its distributed solves then pass the running vector from owner
        to owner instead of broadcasting it after every block."""
        ctx.dist_set_p2p(self._keep[2])


class _DevBuf:
    """A raw device pointer dressed up for torch.as_tensor (zero copy)."""

    def __init__(self, ptr, nbytes, typestr="|u1", itemsize=1):
        self.__cuda_array_interface__ = {"shape": (int(nbytes) // itemsize,), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


class TorchDist:
    """mpf_dist whose callbacks hand the library's DEVICE buffers to torch.distributed (backend nccl = RCCL): no host staging.
    The collective is ordered on the HIP stream the library passes (made torch's current stream for the call).  Used by
    bench.py when the context's own RCCL communicator cannot be created (librccl not loadable outside torch, for instance),
    and selectable with MPF_DIST_TRANSPORT=torch."""

    def __init__(self, rank, world, device, group=None):
        import importlib
        mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
        self.messages = 0
        self.bytes = 0
        dev = torch.device(device)

        def bcast(user, d_buf, nbytes, root, stream):
            try:
                t = torch.as_tensor(_DevBuf(d_buf, nbytes), device=dev)
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                    dist.broadcast(t, src=root, group=group)
                self.messages += 1
                self.bytes += int(nbytes)
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("torch.distributed bcast callback failed:", e, flush=True)
                return -5

        def allreduce(user, d_buf, count, stream):
            try:
                t = torch.as_tensor(_DevBuf(d_buf, count * 8, "<f8", 8), device=dev)
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                    dist.all_reduce(t, group=group)
                return 0
            except Exception as e:
                print("torch.distributed allreduce callback failed:", e, flush=True)
                return -5

        def p2p(user, d_buf, nbytes, peer, send, stream):
            try:
                t = torch.as_tensor(_DevBuf(d_buf, nbytes), device=dev)
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                    if send:
                        dist.send(t, dst=peer, group=group)
                    else:
                        dist.recv(t, src=peer, group=group)
                return 0
            except Exception as e:
                print("torch.distributed p2p callback failed:", e, flush=True)
                return -5

        self._keep = (mpf.BCAST_FN(bcast), mpf.ALLREDUCE_FN(allreduce), mpf.P2P_FN(p2p))
        self.c = mpf.MpfDist(rank=rank, world=world, bcast=self._keep[0], allreduce=self._keep[1], user=None)

    def attach_p2p(self, ctx):
        ctx.dist_set_p2p(self._keep[2])


def combine_info(info, device, group=None):
    """LAPACK-style info of a distributed factorization: the smallest positive per-rank value (first zero pivot), 0 if none."""
    big = 2 ** 31 - 1
    t = torch.tensor([info if info > 0 else big], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    v = int(t.item())
    return 0 if v == big else v


def pick_transport(ctx, rank, world, device):
    """(mpf_dist, name, keep-alive object) for a real multi-GPU run: the context's own RCCL communicator unless it cannot be
    created on EVERY rank (or MPF_DIST_TRANSPORT=torch), then torch.distributed's process group on the same device buffers."""
    import os
    want = os.environ.get("MPF_DIST_TRANSPORT", "rccl")
    ok = 0
    if want == "rccl":
        try:
            cfg = rccl_dist(ctx, rank, world)
            # one small broadcast + all-reduce through the new communicator before the factorization relies on it
            ok = 1 if (world == 1 or ctx.L.mpf_rccl_selftest(ctx.h) == 0) else 0
            if not ok:
                print(f"rank {rank}: RCCL self-test failed; falling back to torch.distributed", flush=True)
        except Exception as e:
            print(f"rank {rank}: RCCL communicator not available ({e}); falling back to torch.distributed", flush=True)
    flag = torch.tensor([ok], dtype=torch.int32, device=torch.device(device))
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return cfg, "RCCL (communicator owned by the library)", None
    td = TorchDist(rank, world, device)
    return td.c, "RCCL through torch.distributed (device buffers)", td


# -------------------------------------------------------------------------------------------------------------
# bench.py --gpus N entry (launched by torch.distributed.run, one rank per GPU, backend nccl = RCCL)
# -------------------------------------------------------------------------------------------------------------
def bench_main(args, rank, world, local_rank, rehearsal=False):
    import importlib
    import json
    import os
    mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
    dev = torch.device("cuda", local_rank)
    ctx = mpf.MPFContext(local_rank)
    # The distributed solves' point-to-point chain (ncclSend / ncclRecv, round 4) has only ever run through the gloo callbacks of
    # the tests: no multi-GPU box was available to the builder.  The driver's scaling run keeps the broadcast form, whose
    # primitive the factorization has exercised by then, unless MPF_BENCH_SOLVE_P2P=1 asks for the chain.
    if os.environ.get("MPF_BENCH_SOLVE_P2P", "0") != "1":
        ctx.set_option("dist_solve_p2p", 0)
    n, nb = args.n, args.nb
    layout = BlockCyclic(n, nb, rank, world)
    if rehearsal:
        gd = GlooDist(rank, world)
        dcfg, transport = gd.c, "gloo (host-staged rehearsal)"
    else:
        dcfg, transport, _keep = pick_transport(ctx, rank, world, dev)
    # this rank's column blocks of the reference generator's matrix (`matgen f N (N-2) lin`), produced on the device
    A0 = colmajor_empty(n, layout.local_cols(), dev)
    for b in layout.my_blocks:
        w = layout.width(b)
        lc = layout.local_col(b)
        ctx.matgen(n, out=A0[:, lc:lc + w], col0=b * nb, ncols=w)
    work = colmajor_empty(n, layout.local_cols(), dev)
    ipiv = None
    for _ in range(args.warmup):
        work.copy_(A0)
        ipiv, info = ctx.factor_dist(work, n, nb, dcfg)
    times = []
    st = None
    for _ in range(args.steps):
        work.copy_(A0)  # restore is outside the timed region
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        ipiv, info = ctx.factor_dist(work, n, nb, dcfg)
        torch.cuda.synchronize()
        dist.barrier()
        times.append(time.perf_counter() - t0)
        st = ctx.stats()
    # per-phase timers of rank 0: one extra step with all of them on (the timed steps keep the update timers only)
    ctx.set_option("event_timers", 2)
    work.copy_(A0)
    ctx.factor_dist(work, n, nb, dcfg)
    sd = ctx.stats()
    ctx.set_option("event_timers", 1)
    work.copy_(A0)
    ipiv, info = ctx.factor_dist(work, n, nb, dcfg)      # (the factors the refinement below uses)
    rdev = torch.device("cpu") if rehearsal else dev
    t = torch.tensor([sum(times)], dtype=torch.float64, device=rdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    total = float(t.item())
    ms_per_step = total * 1e3 / args.steps
    value = 2.0 / 3.0 * n ** 3 / (ms_per_step * 1e-3) / 1e9
    # a rank's info covers the panels it owns: the run's info is the first zero pivot over all ranks (0 = none)
    info = combine_info(info, rdev)
    # every rank must hold the same pivots
    chk = ipiv.to(torch.float64).sum().reshape(1).clone().to(rdev)
    mx = chk.clone(); mn = chk.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    # the metric's second half: refinement sweeps to ||b - A x|| / ||b|| < 1e-12 on the distributed factors
    ir = None
    if not args.no_ir and nb % 64 == 0:
        xs = torch.ones(n, dtype=torch.float64, device=dev)
        bl = A0 @ torch.ones(layout.local_cols(), dtype=torch.float64, device=dev) if layout.local_cols() > 0 else torch.zeros(n, dtype=torch.float64, device=dev)
        bl = bl.to(rdev)
        dist.all_reduce(bl)                      # b = A 1, assembled from the ranks' column blocks
        bvec = bl.to(dev)
        x, irs = ctx.solve_ir_dist(A0, work, ipiv, bvec, n, nb, dcfg, max_iter=10, tol=1e-12)
        ir = {"iterations": int(irs.iterations), "rel_residual": float(irs.rel_residual), "converged": bool(irs.converged),
              "ms": round(float(irs.ms_total), 2), "max_abs_err_vs_ones": float((x - xs).abs().max())}
    # speed mode on the same layout (BASELINE config 2: diagonally dominant input): the two-level fp16 schedule of mpf_factor_dist
    # (super-panels, fp32 working copy of each rank's far columns) + distributed refinement; one timed factorization
    mxp = None
    if not args.no_mxp and nb % 64 == 0:
        try:
            rs = A0.sum(dim=1).to(rdev) if layout.local_cols() > 0 else torch.zeros(n, dtype=torch.float64, device=rdev)
            dist.all_reduce(rs)                      # row sums of the whole matrix
            rs = rs.to(dev)
            Ad = A0.clone()
            for b in layout.my_blocks:
                w = layout.width(b)
                lc = layout.local_col(b)
                idx = torch.arange(w, device=dev)
                Ad[b * nb + idx, lc + idx] += rs[b * nb:b * nb + w]
            tm = 0.0
            for rep in range(2):                     # one warm-up (buffers, images), one timed
                work.copy_(Ad)
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                ipiv_h, info_h = ctx.factor_dist(work, n, nb, dcfg, trailing=mpf.TRAIL_FP16)
                torch.cuda.synchronize()
                dist.barrier()
                tm = time.perf_counter() - t0
            sth = ctx.stats()
            tt = torch.tensor([tm], dtype=torch.float64, device=rdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            info_h = combine_info(info_h, rdev)
            bl = Ad @ torch.ones(layout.local_cols(), dtype=torch.float64, device=dev) if layout.local_cols() > 0 else torch.zeros(n, dtype=torch.float64, device=dev)
            bl = bl.to(rdev)
            dist.all_reduce(bl)
            xh, irh = ctx.solve_ir_dist(Ad, work, ipiv_h, bl.to(dev), n, nb, dcfg, max_iter=10, tol=1e-12)
            fms = float(tt.item()) * 1e3
            mxp = {"trailing": "fp16-in/fp32-acc MFMA, two-level schedule per rank", "matrix": "generator + diag(rowsum) (diagonally dominant)",
                   "factor_ms": round(fms, 2), "factor_gflops": round(2.0 / 3.0 * n ** 3 / (fms * 1e-3) / 1e9, 1), "superpanel": int(sth.superpanel),
                   "ir_iterations": int(irh.iterations), "ir_rel_residual": float(irh.rel_residual), "ir_converged": bool(irh.converged),
                   "ir_ms": round(float(irh.ms_total), 2), "info": int(info_h),
                   "rank0_big_update": {"launches": int(sth.gemm_big_launches), "ms": round(sth.ms_gemm_big, 2),
                                        "tflops": round(sth.gemm_big_flops / max(sth.ms_gemm_big, 1e-9) / 1e9, 1)}}
            del Ad
        except Exception as ex:   # (a failure here must not cost the run its line: every rank fails alike, none is left in a collective)
            mxp = {"error": repr(ex)[:300]}
    if rank == 0:
        line = {
            "metric": "LU GFLOP/s at N=32768 (1/2/4/8 GPUs); IR iterations to ||r||/||b||<1e-12",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic: the reference generator's own stream (`matgen f N (N-2) lin`, matrix_generator.cpp:55-80), "
                                    "each rank produces its own column blocks on the device (mpf_matgen_cols_dev)",
            "config": {"workload": f"N={n} nb={nb} MPF LU, 1-D block-cyclic columns over {world} MI355X (C++ host loop mpf_factor_dist), one "
                                   f"{transport} broadcast of the factored panel per panel step, "
                                   f"depth-1 look-ahead, fp64 trailing update", "n": n, "nb": nb, "trailing": "fp64",
                       "parallelism": f"1-D block-cyclic columns x{world}"},
            "pivots_consistent_across_ranks": bool(mx.item() == mn.item()), "info": int(info), "ir": ir,
            "rank0_events": {"what": "one extra step with every timer on", "panel_chain_ms": round(sd.ms_hpanel + sd.ms_dpanel, 2),
                             "trsm_ms": round(sd.ms_trsm, 2), "laswp_ms": round(sd.ms_laswp, 2), "gemm_ms": round(sd.ms_gemm, 2),
                             "device_ms": round(sd.ms_total, 2), "device_ms_timed_step": round(st.ms_total, 2)},
            "mxp": mxp, "roofline": None, "cpu_baseline": None,
        }
        if st.ms_gemm > 0:
            ach = st.gemm_flops / (st.ms_gemm * 1e-3) / 1e12
            line["roofline"] = {"kernel": "dgemm_minus_kernel (rank 0's share of the trailing updates, last timed step)", "bound": "mfma",
                                "achieved": round(ach, 2), "peak": 78.6, "unit": "TFLOP/s", "frac": round(ach / 78.6, 4), "traffic": None,
                                "launches": int(st.gemm_launches), "avg_launch_ms": round(st.ms_gemm / max(st.gemm_launches, 1), 4)}
        if not args.no_cpu and os.environ.get("MPF_BENCH_CPU_AT_N") == "1":   # the contract times the CPU leg at N = 1 only
            import sys
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            if root not in sys.path:
                sys.path.insert(0, root)
            line["cpu_baseline"] = importlib.import_module("bench").cpu_baseline(min(args.cpu_n, n))
        print(json.dumps(line))
    dist.barrier()
    dist.destroy_process_group()

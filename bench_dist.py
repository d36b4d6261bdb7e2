"""`bench.py --gpus N` under torch.distributed.run (one rank per GPU, backend nccl = RCCL): the multi-rank leg of the benchmark.
Benchmark code, not product code: it lives beside bench.py, outside the package, and is the only multi-rank caller
of bench.cpu_baseline (the CPU leg, which times LAPACK and the oracle)."""
import importlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def bench_main(args, rank, world, local_rank, rehearsal=False):
    D = importlib.import_module("mixed-precision_lu_factorization_amd.dist")
    BlockCyclic, GlooDist, colmajor_empty, pick_transport, combine_info = D.BlockCyclic, D.GlooDist, D.colmajor_empty, D.pick_transport, D.combine_info
    mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
    dev = torch.device("cuda", local_rank)
    ctx = mpf.MPFContext(local_rank)
    # The distributed solves' point-to-point chain (ncclSend / ncclRecv, round 4) has only ever run through the gloo callbacks of
    # the tests: no multi-GPU box was available to the builder.  It is off by default in the library (round 5; the ranks vote on
    # it); MPF_BENCH_SOLVE_P2P=1 asks for it here.
    solve_p2p = os.environ.get("MPF_BENCH_SOLVE_P2P", "0") == "1"
    ctx.set_option("dist_solve_p2p", 1 if solve_p2p else 0)
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
    # ---- what RCCL itself says about this job: communicator size and rank from ncclCommCount /
    #      ncclCommUserRank, version, link types from the device to its peers, the library's own count of what it sent over the
    #      communicator during the LAST timed fp64 factorization, and a timed broadcast of the first panel's size ------------------
    rccl = None
    try:
        info0 = ctx.rccl_info()
        rccl = {"communicator_owned_by_the_library": bool(info0["has_comm"]), "ncclCommCount": info0["comm_count"], "ncclCommUserRank": info0["comm_rank"],
                "version": info0["version"], "has_send_recv": info0["has_p2p"], "solve_chain_p2p": bool(solve_p2p),
                "transport_of_the_factorization": transport,
                "visible_devices": info0["visible_devices"], "link_type_to_device": info0["link_type_to_device"],
                "link_hops_to_device": info0["link_hops_to_device"], "peer_access_to_device": info0["peer_access_to_device"],
                "link_type_legend": "hipExtGetLinkTypeAndHopCount: 0 = none/self, 1 = HyperTransport, 2 = QPI, 3 = PCIe, 4 = InfiniBand, 5 = xGMI",
                "env": {k: os.environ.get(k) for k in ("NCCL_P2P_DISABLE", "NCCL_SHM_DISABLE", "NCCL_ALGO", "NCCL_PROTO", "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY")}}
        if info0["has_comm"] and not rehearsal:
            # the library's counters over one more fp64 factorization (counters are per context, reset here by difference)
            work.copy_(A0)
            before = ctx.rccl_info()
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            ctx.factor_dist(work, n, nb, dcfg)
            torch.cuda.synchronize(); dist.barrier()
            t_f = time.perf_counter() - t0
            after = ctx.rccl_info()
            rccl["one_factorization"] = {"ms": round(t_f * 1e3, 2), "broadcast_calls": after["bcast_calls"] - before["bcast_calls"],
                                         "broadcast_bytes": after["bcast_bytes"] - before["bcast_bytes"],
                                         "allreduce_calls": after["allreduce_calls"] - before["allreduce_calls"]}
            # one panel message of the first panel's size, alone on the wire: every rank calls the probe
            pbytes = (n + 16) * nb * 8
            ms_b = ctx.rccl_bcast_probe(pbytes, root=0, reps=5)
            rccl["panel_broadcast_probe"] = {"bytes": pbytes, "ms": round(ms_b, 4), "GBps": round(pbytes / (ms_b * 1e-3) / 1e9, 1) if ms_b > 0 else None,
                                             "what": "ncclBroadcast of the first panel's message size from rank 0, 5 repetitions after a warm-up, HIP events on rank 0's stream"}
            work.copy_(A0)
            ipiv, _ = ctx.factor_dist(work, n, nb, dcfg)
    except Exception as ex:   # (diagnostic object: never costs the run its line; every rank takes the same path)
        rccl = {"error": repr(ex)[:300]}
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
            "mxp": mxp, "rccl": rccl, "roofline": None, "cpu_baseline": None,
        }
        if st.ms_gemm > 0:
            ach = st.gemm_flops / (st.ms_gemm * 1e-3) / 1e12
            line["roofline"] = {"kernel": "dgemm_minus_kernel (rank 0's share of the trailing updates, last timed step)", "bound": "mfma",
                                "achieved": round(ach, 2), "peak": 78.6, "unit": "TFLOP/s", "frac": round(ach / 78.6, 4), "traffic": None,
                                "launches": int(st.gemm_launches), "avg_launch_ms": round(st.ms_gemm / max(st.gemm_launches, 1), 4)}
        if not args.no_cpu and os.environ.get("MPF_BENCH_CPU_AT_N") == "1":   # the contract times the CPU leg at N = 1 only
            line["cpu_baseline"] = importlib.import_module("bench").cpu_baseline(min(args.cpu_n, n))
        print(json.dumps(line))
    dist.barrier()
    dist.destroy_process_group()

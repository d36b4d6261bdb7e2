"""Multi-GPU MPF: 1-D block-cyclic column layout, one process per GPU, panel broadcast over RCCL/xGMI.

The reference is single-device (MPF.cu:77); this partition is the build's extension.  A column
block of `nb` columns lives entirely on one rank, so the fp16 pivot panel, its pivot search and the fp64
no-pivot panel stay local to the owner -- there is no cross-GPU argmax.  Per panel the ONLY exchange step
is one broadcast, owner -> all, of the factored panel (rows k..N x nb, fp64) with the panel's pivots
appended; every rank then applies the row interchanges, the TRSM and the GEMM to the columns it owns.
The arithmetic per element is identical to the 1-GPU path (the partition only changes WHO computes a
column block), so IPIV and LU are bit-identical to mpf_factor_dev.

`kernels` is any object with the step operators of MPFContext (hgetf2_pivots, laswp, dgetf2_npv,
dtrsm_llnu, dgemm_minus) working on column-major torch tensors.  The product passes an MPFContext (HIP);
the CPU tests pass an object of their own -- this module never imports the oracle.
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


def factor(kernels, Aloc, layout, ipiv=None, group=None, timers=None, host_staged_bcast=False):
    """In-place distributed MPF of the local column blocks `Aloc` (n x local_cols, column-major).
    Returns the full IPIV (int32, 1-based, replicated on every rank).  Panel loop = MPF.cu:100-242."""
    n, nb, rank = layout.n, layout.nb, layout.rank
    dev = Aloc.device
    if ipiv is None:
        ipiv = torch.arange(1, n + 1, dtype=torch.int32, device=dev)  # benchmark.cpp:215-217
    buf = torch.empty(n * nb + nb, dtype=torch.float64, device=dev)  # packed panel + pivots
    t_bcast = 0.0
    for b in range(layout.nblocks):
        k = b * nb
        pc = layout.width(b)
        pr = n - k
        if pr <= 1:  # MPF.cu:104: a 1x1 tail is skipped
            break
        owner = layout.owner(b)
        P = buf[:pr * pc].view(pc, pr).t()  # packed panel, column-major, ld = pr
        tail = buf[pr * pc: pr * pc + pc]
        if rank == owner:
            lc = layout.local_col(b)
            Ap = Aloc[k:, lc:lc + pc]
            piv, _ = kernels.hgetf2_pivots(Ap, ipiv_offset=k)              # steps 1.1-3.2, pivots global 1-based
            kernels.laswp(Aloc[:, lc:lc + pc], k, pc, piv)                 # step 3.1 on the panel's own columns
            kernels.dgetf2_npv(Ap)                                         # step 4
            P.copy_(Ap)
            tail.copy_(piv.to(torch.float64))
        # ---- the one exchange step per panel: owner -> all ------------------------------------------------
        if layout.world > 1:
            t0 = time.perf_counter() if timers is not None else 0.0
            if host_staged_bcast:  # rehearsal on one GPU shared by several ranks: gloo through host memory
                hb = buf[:pr * pc + pc].cpu()
                dist.broadcast(hb, src=owner, group=group)
                if rank != owner:
                    buf[:pr * pc + pc].copy_(hb)
            else:
                dist.broadcast(buf[:pr * pc + pc], src=owner, group=group)
            if timers is not None:
                t_bcast += time.perf_counter() - t0
        piv = tail.to(torch.int32)
        ipiv[k:k + pc] = piv
        # ---- row interchanges on every column this rank owns except the (already swapped) panel ------------
        if rank == owner:
            lc = layout.local_col(b)
            if lc > 0:
                kernels.laswp(Aloc[:, :lc], k, pc, piv)
            if lc + pc < Aloc.shape[1]:
                kernels.laswp(Aloc[:, lc + pc:], k, pc, piv)
        elif Aloc.shape[1] > 0:
            kernels.laswp(Aloc, k, pc, piv)
        # ---- trailing update of the local columns right of the panel (MPF.cu:203-239) ----------------------
        if k + pc < n:
            t0c = layout.first_local_col_after(b)
            if t0c < Aloc.shape[1]:
                U12 = Aloc[k:k + pc, t0c:]
                kernels.dtrsm_llnu(P[:pc, :], U12)
                kernels.dgemm_minus(Aloc[k + pc:, t0c:], P[pc:, :], U12)
    if timers is not None:
        timers["bcast_s"] = t_bcast
    return ipiv


def factor_lookahead(kernels, kernels_side, Aloc, layout, ipiv=None, group=None, host_staged_bcast=False, gemm_timer=None):
    """Same result as factor(), scheduled with depth-1 look-ahead: the owner of panel b+1 updates that block
    first, then runs the panel chain (pivots, interchange, fp64 panel, pack) and the broadcast of panel b+1
    on a side stream, under everybody's trailing update of panel b on the main stream.
    `kernels` launches on the current (main) stream, `kernels_side` on `kernels_side.stream` (a torch stream, or
    None on CPU where the two are the same object and everything runs in order)."""
    n, nb, rank = layout.n, layout.nb, layout.rank
    dev = Aloc.device
    gpu = dev.type == "cuda"
    main = torch.cuda.current_stream(dev) if gpu else None
    side = getattr(kernels_side, "stream", None) if gpu else None

    class _Side:
        def __enter__(self_inner):
            if side is not None:
                self_inner.ctx = torch.cuda.stream(side); self_inner.ctx.__enter__()
        def __exit__(self_inner, *a):
            if side is not None:
                self_inner.ctx.__exit__(*a)

    def side_waits_main():
        if side is not None:
            side.wait_stream(main)

    def main_waits_side():
        if side is not None:
            main.wait_stream(side)

    if ipiv is None:
        ipiv = torch.arange(1, n + 1, dtype=torch.int32, device=dev)
    bufs = [torch.empty(n * nb + nb, dtype=torch.float64, device=dev) for _ in range(2)]

    def views(b):
        k = b * nb; pc = layout.width(b); pr = n - k
        buf = bufs[b % 2]
        return buf, buf[:pr * pc].view(pc, pr).t(), buf[pr * pc: pr * pc + pc]

    def panel_chain(b, K):
        """owner only: factor panel b from the local matrix into its packed buffer (stream of K)"""
        k = b * nb; pc = layout.width(b)
        buf, P, tail = views(b)
        lc = layout.local_col(b)
        Ap = Aloc[k:, lc:lc + pc]
        piv, _ = K.hgetf2_pivots(Ap, ipiv_offset=k)
        K.laswp(Aloc[:, lc:lc + pc], k, pc, piv)
        K.dgetf2_npv(Ap)
        P.copy_(Ap)
        tail.copy_(piv.to(torch.float64))

    def bcast(b):
        k = b * nb; pc = layout.width(b); pr = n - k
        buf = bufs[b % 2]
        if layout.world > 1:
            if host_staged_bcast:
                hb = buf[:pr * pc + pc].cpu()
                dist.broadcast(hb, src=layout.owner(b), group=group)
                if rank != layout.owner(b):
                    buf[:pr * pc + pc].copy_(hb)
            else:
                dist.broadcast(buf[:pr * pc + pc], src=layout.owner(b), group=group)

    def live(b):
        return b < layout.nblocks and n - b * nb > 1

    # panel 0: nothing to hide under
    if live(0):
        if rank == layout.owner(0):
            panel_chain(0, kernels)
        bcast(0)
    b = 0
    while live(b):
        k = b * nb; pc = layout.width(b)
        owner = layout.owner(b)
        buf, P, tail = views(b)
        piv = tail.to(torch.int32)
        ipiv[k:k + pc] = piv
        # interchanges of panel b on every local column except the owner's (already swapped) panel columns
        if rank == owner:
            lc = layout.local_col(b)
            if lc > 0:
                kernels.laswp(Aloc[:, :lc], k, pc, piv)
            if lc + pc < Aloc.shape[1]:
                kernels.laswp(Aloc[:, lc + pc:], k, pc, piv)
        elif Aloc.shape[1] > 0:
            kernels.laswp(Aloc, k, pc, piv)
        t0c = layout.first_local_col_after(b)
        nxt = b + 1
        has_next = live(nxt)
        i_own_next = has_next and rank == layout.owner(nxt)
        rest0 = t0c
        if k + pc < n and i_own_next:
            # my block of panel b+1 first ("strip"), then its panel chain on the side stream
            lcn = layout.local_col(nxt); wn = layout.width(nxt)
            U12s = Aloc[k:k + pc, lcn:lcn + wn]
            kernels.dtrsm_llnu(P[:pc, :], U12s)
            kernels.dgemm_minus(Aloc[k + pc:, lcn:lcn + wn], P[pc:, :], U12s)
            rest0 = lcn + wn
            side_waits_main()
            with _Side():
                panel_chain(nxt, kernels_side)
        elif has_next:
            side_waits_main()  # the receive buffer of panel b+1 was last read by the update of panel b-1
        if has_next:
            with _Side():
                bcast(nxt)
        # the rest of the trailing update of panel b on the main stream
        if k + pc < n and rest0 < Aloc.shape[1]:
            U12 = Aloc[k:k + pc, rest0:]
            kernels.dtrsm_llnu(P[:pc, :], U12)
            if gemm_timer is not None:  # bench: event pair around this rank's dominant launch (torch's current stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            kernels.dgemm_minus(Aloc[k + pc:, rest0:], P[pc:, :], U12)
            if gemm_timer is not None:
                e1.record()
                gemm_timer.append((2.0 * (n - k - pc) * (Aloc.shape[1] - rest0) * pc, e0, e1))
        main_waits_side()
        b = nxt
    return ipiv


# -------------------------------------------------------------------------------------------------------------
# bench.py --gpus N entry (launched by torch.distributed.run, one rank per GPU, backend nccl = RCCL)
# -------------------------------------------------------------------------------------------------------------
def synth_block(n, width, b, device, seed=1234):
    """Column block b of the synthetic bench matrix: i.i.d. uniform {0.0..9.9} (matrix_generator.cpp:66
    distribution); seeded per block so the matrix does not depend on the number of ranks."""
    g = torch.Generator(device=device)
    g.manual_seed(seed * 1000003 + b)
    return (torch.randint(0, 100, (width, n), generator=g, device=device, dtype=torch.int32).to(torch.float64) / 10.0).t()


def bench_main(args, rank, world, local_rank, rehearsal=False):
    import importlib
    import json
    mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
    dev = torch.device("cuda", local_rank)
    ctx = mpf.MPFContext(local_rank)
    side_stream = torch.cuda.Stream(device=dev, priority=-1)
    ctx_side = mpf.MPFContext(local_rank, stream=side_stream)
    n, nb = args.n, args.nb
    layout = BlockCyclic(n, nb, rank, world)
    A0 = colmajor_empty(n, layout.local_cols(), dev)
    for b in layout.my_blocks:
        w = layout.width(b)
        A0[:, layout.local_col(b):layout.local_col(b) + w] = synth_block(n, w, b, dev)
    work = colmajor_empty(n, layout.local_cols(), dev)
    ipiv = None
    for _ in range(args.warmup):
        work.copy_(A0)
        ipiv = factor_lookahead(ctx, ctx_side, work, layout, host_staged_bcast=rehearsal)
    times = []
    last_gemm_events = []
    for _ in range(args.steps):
        work.copy_(A0)  # restore is outside the timed region
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        gemm_events = [] if (rank == 0 and _ == args.steps - 1) else None
        ipiv = factor_lookahead(ctx, ctx_side, work, layout, host_staged_bcast=rehearsal, gemm_timer=gemm_events)
        torch.cuda.synchronize()
        dist.barrier()
        times.append(time.perf_counter() - t0)
        if gemm_events is not None:
            last_gemm_events = gemm_events
    rdev = torch.device("cpu") if rehearsal else dev
    t = torch.tensor([sum(times)], dtype=torch.float64, device=rdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    total = float(t.item())
    ms_per_step = total * 1e3 / args.steps
    value = 2.0 / 3.0 * n ** 3 / (ms_per_step * 1e-3) / 1e9
    # cheap cross-rank sanity: every rank must hold the same pivots
    chk = ipiv.to(torch.float64).sum().reshape(1).clone().to(rdev)
    mx = chk.clone(); mn = chk.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    if rank == 0:
        line = {
            "metric": "LU GFLOP/s at N=32768 (1/2/4/8 GPUs); IR iterations to ||r||/||b||<1e-12",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic: i.i.d. uniform {0.0..9.9} (matrix_generator.cpp:66 distribution), per-block torch seeds",
            "config": {"workload": f"N={n} nb={nb} MPF LU, 1-D block-cyclic columns over {world} MI355X, one RCCL broadcast of the "
                                   f"factored panel per panel step (depth-1 look-ahead: chain + broadcast of panel k+1 under update k), fp64 trailing update", "n": n, "nb": nb, "trailing": "fp64",
                       "parallelism": f"1-D block-cyclic columns x{world}"},
            "pivots_consistent_across_ranks": bool(mx.item() == mn.item()),
            "roofline": None, "cpu_baseline": None,
        }
        if last_gemm_events:
            gf = sum(f for f, _, _ in last_gemm_events)
            ms = sum(a.elapsed_time(b) for _, a, b in last_gemm_events)
            ach = gf / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            line["roofline"] = {"kernel": "dgemm_minus_kernel (rank 0's share of the trailing updates, last timed step)", "bound": "mfma",
                                "achieved": round(ach, 2), "peak": 78.6, "unit": "TFLOP/s", "frac": round(ach / 78.6, 4), "traffic": None,
                                "launches": len(last_gemm_events), "avg_launch_ms": round(ms / len(last_gemm_events), 4)}
        print(json.dumps(line))
    dist.barrier()
    dist.destroy_process_group()

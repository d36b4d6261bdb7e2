"""A Python MODEL of the distributed schedule (test infrastructure; the product schedule is the C++ loop mpf_factor_dist in
csrc/mpf_dist.cpp).  Same partition, same message sequence and the same per-element arithmetic, spelled out over step
operators so that it runs on any `kernels` object: the CPU tests drive it with the oracle's step operators over gloo
(world 2 / 3), the GPU tests with the HIP step operators, and both must give the single-process oracle's bits -- which pins
the layout arithmetic (BlockCyclic, shared with the product) and the order of the steps the C++ loop restates.
Panel loop = MPF.cu:100-242."""
import time

import torch
import torch.distributed as dist


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




# ---- the distributed triangular solves' point-to-point chain (csrc/mpf_dist.cpp: lu_solve_chain) as a numpy / gloo model ----------
def lu_solve_chain(LUloc, layout, perm, rhs, group=None):
    """x = U^-1 L^-1 P rhs with the factors' columns block-cyclic over the ranks, the way mpf_solve_ir_dist does it with a
    point-to-point transport: lower sweep -- the running tail out[k + w:] goes from the owner of block b to the owner of block
    b + 1; upper sweep -- every rank starts from its OWN segments of y (zero elsewhere), adds what arrives, applies its block, sends
    the head [0, k) on and zeroes it locally; one all-reduce at the end replicates x.  LUloc: (N, local columns) float64 tensor.
    Returns (x, number of point-to-point messages this rank took part in)."""
    import torch
    import torch.distributed as dist
    n, nb, rank, world = layout.n, layout.nb, layout.rank, layout.world
    nblocks = (n + nb - 1) // nb
    out = rhs[perm].clone()
    msgs = 0

    def owner(b):
        return b % world

    for b in range(nblocks):
        if owner(b) != rank:
            continue
        k, w = b * nb, layout.width(b)
        lc = layout.local_col(b)
        if b > 0 and owner(b - 1) != rank:
            buf = torch.empty(n - k, dtype=torch.float64)
            dist.recv(buf, src=owner(b - 1), group=group); msgs += 1
            out[k:] = buf
        L = LUloc[:, lc:lc + w]
        for j in range(w):                                   # unit-lower block, then the rows below
            out[k + j + 1:] -= L[k + j + 1:, j] * out[k + j]
        if b + 1 < nblocks and owner(b + 1) != rank and k + w < n:
            dist.send(out[k + w:].contiguous(), dst=owner(b + 1), group=group); msgs += 1
    up = torch.zeros(n, dtype=torch.float64)
    for b in range(rank, nblocks, world):
        k, w = b * nb, layout.width(b)
        up[k:k + w] = out[k:k + w]
    for b in range(nblocks - 1, -1, -1):
        if owner(b) != rank:
            continue
        k, w = b * nb, layout.width(b)
        lc = layout.local_col(b)
        if b + 1 < nblocks and owner(b + 1) != rank:
            buf = torch.empty(k + w, dtype=torch.float64)
            dist.recv(buf, src=owner(b + 1), group=group); msgs += 1
            up[:k + w] += buf
        U = LUloc[:, lc:lc + w]
        for j in range(w - 1, -1, -1):
            up[k + j] /= U[k + j, j]
            up[:k + j] -= U[:k + j, j] * up[k + j]
        if b > 0 and owner(b - 1) != rank:
            dist.send(up[:k].contiguous(), dst=owner(b - 1), group=group); msgs += 1
            up[:k] = 0
    if world > 1:
        dist.all_reduce(up, group=group)
    return up, msgs

"""Multi-GPU MPF: 1-D block-cyclic column layout, one process per GPU, panel broadcast over RCCL/xGMI.

The product path is C++ behind the C ABI (csrc/mpf_dist.cpp: mpf_factor_dist, mpf_solve_ir_dist); this module holds
  * the layout arithmetic and scatter / gather helpers,
  * the transports the C++ loop is driven with: `rccl_dist` (the context's own RCCL communicator: ncclBroadcast /
    ncclAllReduce) and `gloo_dist` (ctypes callbacks that stage through host memory and torch.distributed's gloo backend:
    several ranks on ONE GPU in the tests, or CPU-side rehearsals of more ranks),
  * (the schedule itself exists ONCE, in C++; a Python model of it over step operators, which the CPU tests run on the
    oracle's step operators with gloo, lives with the tests: tests/dist_model.py),
  * (the multi-rank benchmark entry lives in bench_dist.py at the repository root: benchmark code is not part of the package)

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

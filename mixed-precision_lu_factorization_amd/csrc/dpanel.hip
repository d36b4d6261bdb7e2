// fp64 no-pivot panel: replaces dgetf2_native_npv (reference dgetf2_native_npv.cu:11-36) and the
// extract / write-back memcpy loops around it (MPF.cu:168-200) -- the panel is factored in place
// with its leading dimension, no packed copy.
//
// The reference synchronises the whole grid once per column.  Here the per-element operation
// order of the reference (updates j ascending, m = a/p, then separate multiply and subtract --
// contract C3) is kept, but the schedule is blocked by DP_IB = 32 columns, which is bit-identical
// because every element still receives the same operations in the same order:
//   dpanel_sub    one launch per 32-column sub-panel.  EVERY workgroup re-factors the 32x32
//                 diagonal tile in LDS (identical arithmetic => identical bits, no inter-workgroup
//                 dependency).  Because every workgroup reads the UNfactored tile from the matrix,
//                 workgroup 0 parks the factored tile in the workspace (dpanel_tiles_store puts all
//                 tiles of the panel back at the end) and solves the U row-block right of it;
//                 workgroups >= 1 each finish 256 rows below it, one row per thread, the row's 32
//                 values in registers (row-independent recurrence).
//   dpanel_update rank-32 update of the rest of the panel, one row per thread, 32 columns per
//                 workgroup, j ascending.
#include "mpf_internal.h"
#include <limits.h>
#include <cstdlib>

constexpr int DP_IB = 32;

template <bool FUSED>
__device__ __forceinline__ double mulsub(double x, double m, double u) {
    if (FUSED) return __builtin_fma(-m, u, x);
    const double t = m * u; // file is compiled with -ffp-contract=off: stays mul + sub
    return x - t;
}

// An LDS pointer the optimiser cannot see through: stops it from hoisting every T[][] read of the fully
// unrolled recurrences to the top of the loop nest (which costs ~1000 VGPRs and spills).
typedef __attribute__((address_space(3))) const double lds_cdouble;
__device__ __forceinline__ lds_cdouble *opaque_lds(lds_cdouble *p) {
    asm volatile("" : "+v"(p));
    return p;
}

// opaque_lds tied to a value of the running recurrence: the LDS reads of step j cannot be issued before `dep` (a result of step
// j - 1) exists, so the scheduler cannot pile the reads of all 32 steps up in front of the arithmetic
__device__ __forceinline__ lds_cdouble *opaque_lds_after(lds_cdouble *p, double &dep) {
    asm volatile("" : "+v"(p), "+v"(dep));
    return p;
}

// A global pointer the optimiser cannot see through: pins the loads that use it BEHIND this point of the program.  Without
// it the 32 loads of a later phase are hoisted to the top of the kernel and stay live through every phase before.
template <typename T>
__device__ __forceinline__ T *opaque_ptr(T *p) {
    asm volatile("" : "+v"(p) : : "memory");
    return p;
}

// FULL = true: the sub-panel is a full DP_IB columns wide (w == DP_IB): no guards anywhere.
template <bool FUSED, bool FULL>
__global__ __launch_bounds__(256) void dpanel_sub_kernel(double *P, long long ld, int rows, int cols, int j0,
                                                        int w, int *info, int info_base, double *tile_out) {
    __shared__ double T[DP_IB][DP_IB + 1];
    const int tid = threadIdx.x;
    __builtin_amdgcn_s_setprio(3); // part of the look-ahead latency chain (see fp16_panel.hip)
    // ---- diagonal tile: load (identity padding outside w x w) and factor in LDS ---------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, r = e & 31, c = e >> 5;
        T[r][c] = (FULL || (r < w && c < w)) ? P[(j0 + r) + (long long)(j0 + c) * ld] : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int j = 0; j < w; ++j) {
        const double piv = T[j][j];
        if (tid < w && tid > j) T[tid][j] = T[tid][j] / piv; // dgetf2_native_npv.cu:24-25
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, r = e & 31, c = e >> 5;
            if (r > j && c > j && r < w && c < w) T[r][c] = mulsub<FUSED>(T[r][c], T[r][j], T[j][c]); // :29
        }
        __syncthreads();
    }

    if (blockIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, r = e & 31, c = e >> 5;
            tile_out[r * DP_IB + c] = T[r][c]; // NOT into P: other workgroups may not have read the tile yet
        }
        if (tid == 0 && info)
            for (int j = 0; j < w; ++j)
                if (T[j][j] == 0.0) { atomicMin(info, info_base + j0 + j + 1); break; }
        // U row-block: columns right of the sub-panel, rows j0..j0+w-1; one column per thread.
        // (T is identity-padded, so the padded rows of a narrow tail tile are harmless no-ops.)
        for (int c = j0 + w + tid; c < cols; c += 256) {
            double *pc = P + j0 + (long long)c * ld;
            double x[DP_IB];
#pragma unroll
            for (int i = 0; i < DP_IB; ++i) { // clamped address + select: no branch per element
                const double v = pc[FULL ? i : (i < w ? i : w - 1)];
                x[i] = (FULL || i < w) ? v : 0.0;
            }
#pragma unroll
            for (int j = 0; j < DP_IB; ++j) {
                lds_cdouble *tj = opaque_lds((lds_cdouble *)&T[0][j]);
#pragma unroll
                for (int i = j + 1; i < DP_IB; ++i) x[i] = mulsub<FUSED>(x[i], tj[i * (DP_IB + 1)], x[j]);
            }
#pragma unroll
            for (int i = 0; i < DP_IB; ++i)
                if (FULL || i < w) pc[i] = x[i];
        }
    } else {
        // rows below the tile: thread = row, the row's sub-panel entries live in registers
        const long long r = (long long)j0 + w + (long long)(blockIdx.x - 1) * 256 + tid;
        if (r < rows) {
            double *pr = P + r + (long long)j0 * ld;
            double x[DP_IB];
#pragma unroll
            for (int c = 0; c < DP_IB; ++c) {
                const double v = pr[(long long)(FULL ? c : (c < w ? c : w - 1)) * ld];
                x[c] = (FULL || c < w) ? v : 0.0;
            }
#pragma unroll
            for (int j = 0; j < DP_IB; ++j) {
                // identity padding makes the steps j >= w of a narrow tile no-ops (m = 0/1 = 0)
                lds_cdouble *tj = opaque_lds((lds_cdouble *)&T[j][0]);
                const double m = x[j] / tj[j];
                x[j] = m;
#pragma unroll
                for (int c = j + 1; c < DP_IB; ++c) x[c] = mulsub<FUSED>(x[c], m, tj[c]);
            }
#pragma unroll
            for (int c = 0; c < DP_IB; ++c)
                if (FULL || c < w) pr[(long long)c * ld] = x[c];
        }
    }
}

// put the factored diagonal tiles of a finished panel back into the matrix (one workgroup per tile)
__global__ __launch_bounds__(256) void dpanel_tiles_store_kernel(double *P, long long ld, int cols, const double *tiles) {
    const int j0 = blockIdx.x * DP_IB;
    const int w = cols - j0 < DP_IB ? cols - j0 : DP_IB;
    const double *t = tiles + (long long)blockIdx.x * DP_IB * DP_IB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = threadIdx.x + 256 * i, r = e & 31, c = e >> 5;
        if (r < w && c < w) P[(j0 + r) + (long long)(j0 + c) * ld] = t[r * DP_IB + c];
    }
}

template <bool FUSED>
__global__ __launch_bounds__(256) void dpanel_update_kernel(double *P, long long ld, int rows, int cols, int j0,
                                                           int w) {
    __shared__ double Ut[DP_IB][DP_IB]; // Ut[j][cc], read as a broadcast (all lanes one address)
    const int tid = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    const int c0 = j0 + w + blockIdx.y * DP_IB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, j = e & 31, cc = e >> 5;
        Ut[j][cc] = (j < w && c0 + cc < cols) ? P[(j0 + j) + (long long)(c0 + cc) * ld] : 0.0;
    }
    __syncthreads();
    const long long r = (long long)j0 + w + (long long)blockIdx.x * 256 + tid;
    if (r >= rows) return;
    double x[DP_IB], mv[DP_IB];
    // all of the row's 32 multipliers and 32 targets are requested up front (64 loads in flight)
#pragma unroll
    for (int j = 0; j < DP_IB; ++j) mv[j] = P[r + (long long)(j0 + (j < w ? j : w - 1)) * ld];
#pragma unroll
    for (int cc = 0; cc < DP_IB; ++cc) x[cc] = (c0 + cc < cols) ? P[r + (long long)(c0 + cc) * ld] : 0.0;
#pragma unroll
    for (int j = 0; j < DP_IB; ++j) {
        if (j < w) {
            lds_cdouble *uj = opaque_lds((lds_cdouble *)&Ut[j][0]);
#pragma unroll
            for (int cc = 0; cc < DP_IB; ++cc) x[cc] = mulsub<FUSED>(x[cc], mv[j], uj[cc]);
        }
    }
#pragma unroll
    for (int cc = 0; cc < DP_IB; ++cc)
        if (c0 + cc < cols) P[r + (long long)(c0 + cc) * ld] = x[cc];
}

// ---------------------------------------------------------------------------------------------------------------------
// Panels up to 256 columns wide (the tuned schedules): the same blocked recurrence in 8 + 2 launches instead of 17, and
// only the LAST of them is wide.  Under the look-ahead schedule every launch of the chain waits for trailing-update
// workgroups to retire before its own can start (a 128-workgroup launch needs 128 free slots: ~35 us each time); the
// launches of the top block below need at most 7.
//   dpanel_step   one launch per 32-column sub-panel, on the TOP block only (the first 256 rows, where L11 \ U11 live):
//                 workgroup `by` owns one 32-column chunk right of the sub-panel.  Each re-factors the diagonal tile (identical
//                 bits everywhere), solves ITS chunk of the U row-block (thread = column), runs the recurrence of the rows
//                 under the tile (thread = row; identical in every workgroup, the multipliers stay in registers) and applies
//                 the rank-32 update to its chunk.  Outputs that another workgroup of the same launch still reads as input --
//                 the factored tile, the multipliers -- are parked (workspace) and put into the matrix by dpanel_top_store.
//   dpanel_below  ONE launch for all rows under the top block, left-looking over the sub-panels: a thread owns a row, brings
//                 its 32 entries of sub-panel s up to date with the multipliers it computed for the sub-panels before
//                 (x -= m_j u_j, j ascending over ALL earlier columns: the reference's order), then runs the recurrence
//                 against tile s.  U comes from the finished top block through LDS.
// Per element the operations and their order are those of dgetf2_native_npv.cu:18-35 (contract C3): bit-identical to the
// multi-launch form above, which stays in use for panels wider than 256 columns.
// ---------------------------------------------------------------------------------------------------------------------
template <bool FUSED, bool FULL>
__global__ __launch_bounds__(256) void dpanel_step_kernel(double *P, long long ld, int top, int cols, int j0, int w, int *info,
                                                         int info_base, double *tile_out, double *mscr) {
    __shared__ double T[DP_IB][DP_IB + 1];
    __shared__ double Uc[DP_IB][DP_IB];      // this workgroup's chunk of the U row-block: Uc[j][cc]
    const int tid = threadIdx.x, by = blockIdx.y;
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, r = e & 31, c = e >> 5;
        T[r][c] = (FULL || (r < w && c < w)) ? P[(j0 + r) + (long long)(j0 + c) * ld] : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int j = 0; j < w; ++j) {
        const double piv = T[j][j];
        if (tid < w && tid > j) T[tid][j] = T[tid][j] / piv; // dgetf2_native_npv.cu:24-25
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, r = e & 31, c = e >> 5;
            if (r > j && c > j && r < w && c < w) T[r][c] = mulsub<FUSED>(T[r][c], T[r][j], T[j][c]); // :29
        }
        __syncthreads();
    }
    const int c0 = j0 + w + by * DP_IB;      // first column of this workgroup's chunk (>= cols: no chunk)
    if (by == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, r = e & 31, c = e >> 5;
            tile_out[r * DP_IB + c] = T[r][c]; // NOT into P: the other workgroups may not have read the tile yet
        }
        if (tid == 0 && info)
            for (int j = 0; j < w; ++j)
                if (T[j][j] == 0.0) { atomicMin(info, info_base + j0 + j + 1); break; }
    }
    // ---- U row-block of this chunk: thread = column (identity padding makes a narrow tile's padded rows no-ops) ----
    if (tid < DP_IB) {
        const int c = c0 + tid;
        const bool live = c0 < cols && c < cols;
        double *pc = opaque_ptr(P + j0 + (long long)(live ? c : j0) * ld);
        double x[DP_IB];
#pragma unroll
        for (int i = 0; i < DP_IB; ++i) {
            const double v = pc[FULL ? i : (i < w ? i : w - 1)];
            x[i] = (live && (FULL || i < w)) ? v : 0.0;
        }
#pragma unroll
        for (int j = 0; j < DP_IB; ++j) {
            lds_cdouble *tj = opaque_lds_after((lds_cdouble *)&T[0][j], x[j]);
#pragma unroll
            for (int i = j + 1; i < DP_IB; ++i) x[i] = mulsub<FUSED>(x[i], tj[i * (DP_IB + 1)], x[j]);
        }
#pragma unroll
        for (int i = 0; i < DP_IB; ++i) {
            Uc[i][tid] = x[i];
            if (live && (FULL || i < w)) pc[i] = x[i];
        }
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    // ---- rows of the top block under the tile: recurrence on the sub-panel's columns, then the chunk's update ----
    const int r = j0 + w + tid;
    if (r >= top) return;
    double mv[DP_IB];
    {
        const double *pr = opaque_ptr(P + r + (long long)j0 * ld);
#pragma unroll
        for (int c = 0; c < DP_IB; ++c) {
            const double v = pr[(long long)(FULL ? c : (c < w ? c : w - 1)) * ld];
            mv[c] = (FULL || c < w) ? v : 0.0;
        }
#pragma unroll
        for (int j = 0; j < DP_IB; ++j) {
            lds_cdouble *tj = opaque_lds_after((lds_cdouble *)&T[j][0], mv[j]);
            const double m = mv[j] / tj[j]; // identity padding: steps j >= w are no-ops (0 / 1)
            mv[j] = m;
#pragma unroll
            for (int c = j + 1; c < DP_IB; ++c) mv[c] = mulsub<FUSED>(mv[c], m, tj[c]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (by == 0) { // parked: the other workgroups read the un-eliminated entries from the matrix
            double *ms = mscr + r + (long long)j0 * top;
#pragma unroll
            for (int c = 0; c < DP_IB; ++c)
                if (FULL || c < w) ms[(long long)c * top] = mv[c];
        }
    }
    __builtin_amdgcn_sched_barrier(0); // keep the update's loads out of the recurrence above: two unrolled 32 x 32 passes
                                       // interleaved by the scheduler do not fit the register file
    if (c0 < cols) {
        double x[DP_IB];
        double *px = opaque_ptr(P + r + (long long)c0 * ld);
#pragma unroll
        for (int cc = 0; cc < DP_IB; ++cc) x[cc] = (c0 + cc < cols) ? px[(long long)cc * ld] : 0.0;
#pragma unroll
        for (int j = 0; j < DP_IB; ++j) {
            if (FULL || j < w) {
                lds_cdouble *uj = opaque_lds_after((lds_cdouble *)&Uc[j][0], x[DP_IB - 1]);
#pragma unroll
                for (int cc = 0; cc < DP_IB; ++cc) x[cc] = mulsub<FUSED>(x[cc], mv[j], uj[cc]);
            }
        }
#pragma unroll
        for (int cc = 0; cc < DP_IB; ++cc)
            if (c0 + cc < cols) px[(long long)cc * ld] = x[cc];
    }
}

// factored diagonal tiles and the multipliers of the top block: workspace -> matrix (one workgroup per sub-panel)
__global__ __launch_bounds__(256) void dpanel_top_store_kernel(double *P, long long ld, int top, int cols, const double *tiles,
                                                              const double *mscr) {
    const int j0 = blockIdx.x * DP_IB;
    const int w = cols - j0 < DP_IB ? cols - j0 : DP_IB;
    const double *t = tiles + (long long)blockIdx.x * DP_IB * DP_IB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = threadIdx.x + 256 * i, r = e & 31, c = e >> 5;
        if (r < w && c < w) P[(j0 + r) + (long long)(j0 + c) * ld] = t[r * DP_IB + c];
    }
    const int r = j0 + w + threadIdx.x;
    if (r < top)
        for (int c = 0; c < w; ++c) P[r + (long long)(j0 + c) * ld] = mscr[r + (long long)(j0 + c) * top];
}

template <bool FUSED>
__global__ __launch_bounds__(256) void dpanel_below_kernel(double *P, long long ld, int rows, int cols, int top) {
    __shared__ double Ub[DP_IB][DP_IB];
    const int tid = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    const long long r = (long long)top + (long long)blockIdx.x * 256 + tid;
    const bool active = r < rows;
    double *prow = P + (active ? r : 0);
    const int nsub = (cols + DP_IB - 1) / DP_IB;
#pragma unroll 1
    for (int sidx = 0; sidx < nsub; ++sidx) {
        const int j0 = sidx * DP_IB;
        const int w = cols - j0 < DP_IB ? cols - j0 : DP_IB;
        double x[DP_IB];
        {
            const double *px = opaque_ptr(prow);
#pragma unroll
            for (int c = 0; c < DP_IB; ++c) {
                const double v = active ? px[(long long)(j0 + (c < w ? c : w - 1)) * ld] : 0.0;
                x[c] = c < w ? v : 0.0;
            }
        }
#pragma unroll 1
        for (int sp = 0; sp < sidx; ++sp) { // rank-32 update by every earlier sub-panel, in column order
            __syncthreads();                 // the previous Ub is no longer read
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = tid + 256 * i, j = e & 31, cc = e >> 5;
                Ub[j][cc] = cc < w ? P[(sp * DP_IB + j) + (long long)(j0 + cc) * ld] : 0.0;
            }
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            double mv[DP_IB];                // this row's multipliers of sub-panel sp (stored by this thread earlier)
            const double *pm = opaque_ptr(prow);
#pragma unroll
            for (int j = 0; j < DP_IB; ++j) mv[j] = active ? pm[(long long)(sp * DP_IB + j) * ld] : 0.0;
#pragma unroll
            for (int j = 0; j < DP_IB; ++j) {
                lds_cdouble *uj = opaque_lds_after((lds_cdouble *)&Ub[j][0], x[DP_IB - 1]);
#pragma unroll
                for (int cc = 0; cc < DP_IB; ++cc) x[cc] = mulsub<FUSED>(x[cc], mv[j], uj[cc]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) { // tile sidx of U11 (upper part; identity padding outside w x w)
            const int e = tid + 256 * i, j = e & 31, cc = e >> 5;
            Ub[j][cc] = (j < w && cc < w) ? P[(j0 + j) + (long long)(j0 + cc) * ld] : (j == cc ? 1.0 : 0.0);
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < DP_IB; ++j) {
            lds_cdouble *tj = opaque_lds_after((lds_cdouble *)&Ub[j][0], x[j]);
            const double m = x[j] / tj[j];   // dgetf2_native_npv.cu:24-25
            x[j] = m;
#pragma unroll
            for (int c = j + 1; c < DP_IB; ++c) x[c] = mulsub<FUSED>(x[c], m, tj[c]); // :29
        }
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
#pragma unroll
            for (int c = 0; c < DP_IB; ++c)
                if (c < w) prow[(long long)(j0 + c) * ld] = x[c];
        }
    }
}

static int launch_dpanel_narrow(mpf_ctx *c, double *P, int64_t ld, int rows, int cols, int fused, int info_base) {
    int *info = &c->ws->info;
    const int top = rows < 256 ? rows : 256;
    const int nsub = (cols + DP_IB - 1) / DP_IB;
    if (!c->dp_mscr) MPF_HIP_TRY(c, hipMalloc((void **)&c->dp_mscr, (size_t)256 * 256 * sizeof(double)));
    for (int s = 0; s < nsub; ++s) {
        const int j0 = s * DP_IB;
        const int w = cols - j0 < DP_IB ? cols - j0 : DP_IB;
        const int right = cols - j0 - w;
        dim3 grid(1, (unsigned)(right > 0 ? (right + DP_IB - 1) / DP_IB : 1));
        double *tile = c->dtiles + (size_t)s * DP_IB * DP_IB;
#define DPSTEP(F, U) dpanel_step_kernel<F, U><<<grid, 256, 0, c->stream>>>(P, ld, top, cols, j0, w, info, info_base, tile, c->dp_mscr)
        if (fused) DPSTEP(true, false); else DPSTEP(false, false); // (the guard-free instantiation spills 7 KB per thread: not used)
#undef DPSTEP
    }
    dpanel_top_store_kernel<<<nsub, 256, 0, c->stream>>>(P, ld, top, cols, c->dtiles, c->dp_mscr);
    if (rows > top) {
        const int gb = (rows - top + 255) / 256;
        if (fused) dpanel_below_kernel<true><<<gb, 256, 0, c->stream>>>(P, ld, rows, cols, top);
        else dpanel_below_kernel<false><<<gb, 256, 0, c->stream>>>(P, ld, rows, cols, top);
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

int launch_dgetf2_npv(mpf_ctx *c, double *P, int64_t ld, int rows, int cols, int fused, int info_base) {
    if (rows < 1 || cols < 1) return 0;
    if (cols > rows) { c->err = "dgetf2_npv: cols > rows"; return -1; }
    int *info = &c->ws->info;
    const int ntiles = (cols + DP_IB - 1) / DP_IB;
    if (ntiles > c->dtiles_cap) {
        if (c->dtiles) hipFree(c->dtiles);
        c->dtiles = nullptr; c->dtiles_cap = 0;
        const int cap = ntiles < 8 ? 8 : ntiles;
        MPF_HIP_TRY(c, hipMalloc((void **)&c->dtiles, (size_t)cap * DP_IB * DP_IB * sizeof(double)));
        c->dtiles_cap = cap;
    }
    static int multi = -1; // MPF_DPANEL_MULTI=1: the 17-launch form for narrow panels too (A/B measurements)
    if (multi < 0) { const char *e = getenv("MPF_DPANEL_MULTI"); multi = (e && e[0] == '1') ? 1 : 0; }
    if (cols <= 256 && !multi) return launch_dpanel_narrow(c, P, ld, rows, cols, fused, info_base);
    for (int j0 = 0; j0 < cols; j0 += DP_IB) {
        const int w = cols - j0 < DP_IB ? cols - j0 : DP_IB;
        const long long below = (long long)rows - j0 - w;
        const int gb = (int)((below + 255) / 256);
        double *tile = c->dtiles + (size_t)(j0 / DP_IB) * DP_IB * DP_IB;
        if (w == DP_IB) {
            if (fused) dpanel_sub_kernel<true, true><<<1 + gb, 256, 0, c->stream>>>(P, ld, rows, cols, j0, w, info, info_base, tile);
            else dpanel_sub_kernel<false, true><<<1 + gb, 256, 0, c->stream>>>(P, ld, rows, cols, j0, w, info, info_base, tile);
        } else {
            if (fused) dpanel_sub_kernel<true, false><<<1 + gb, 256, 0, c->stream>>>(P, ld, rows, cols, j0, w, info, info_base, tile);
            else dpanel_sub_kernel<false, false><<<1 + gb, 256, 0, c->stream>>>(P, ld, rows, cols, j0, w, info, info_base, tile);
        }
        const int right = cols - j0 - w;
        if (right > 0 && below > 0) {
            dim3 grid(gb, (right + DP_IB - 1) / DP_IB);
            if (fused) dpanel_update_kernel<true><<<grid, 256, 0, c->stream>>>(P, ld, rows, cols, j0, w);
            else dpanel_update_kernel<false><<<grid, 256, 0, c->stream>>>(P, ld, rows, cols, j0, w);
        }
    }
    dpanel_tiles_store_kernel<<<ntiles, 256, 0, c->stream>>>(P, ld, cols, c->dtiles);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

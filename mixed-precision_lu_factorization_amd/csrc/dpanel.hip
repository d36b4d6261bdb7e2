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


// Fused form for full 32-column sub-panels s >= 1: ONE launch applies the rank-32 update of sub-panel s-1 to everything right
// of it AND factors sub-panel s (9 launches per 256-column panel instead of 17).  No workgroup depends on another one of the
// same launch: every workgroup rebuilds the diagonal tile of s itself -- raw tile, minus L(s-1) U(s-1) (j ascending, the
// order the row threads use), then the in-LDS factorization -- from data the previous launch left in the matrix.
//   column group 0 (the sub-panel): rows below the tile, one per thread: update, then the row recurrence; workgroup (0, 0)
//                  parks the factored tile in the workspace (others may still be reading the raw one);
//   column groups >= 1 (32 columns each): rows from the tile's first row on: update; the tile's own 32 rows then go through
//                  LDS to one thread per column for the U row-block solve with the tile's unit-lower part.
// Per element the operations and their order are those of dpanel_sub / dpanel_update: bit-identical.
template <bool FUSED>
__global__ __launch_bounds__(256) void dpanel_fused_kernel(double *P, long long ld, int rows, int cols, int j0, int *info,
                                                          int info_base, double *tile_out) {
    __shared__ double T[DP_IB][DP_IB + 1];   // tile of sub-panel s: raw -> updated -> factored
    __shared__ double Lt[DP_IB][DP_IB + 1];  // multipliers of the tile's rows in sub-panel s-1: Lt[r][j]
    __shared__ double U0[DP_IB][DP_IB];      // U(s-1)[j][c], c over the sub-panel's columns
    __shared__ double U1[DP_IB][DP_IB];      // U(s-1)[j][cc], cc over this workgroup's columns (column groups >= 1)
    __shared__ double S[DP_IB][DP_IB + 1];   // the tile rows of a column group >= 1 on their way to one thread per column
    const int tid = threadIdx.x, cg = blockIdx.y, rb = blockIdx.x;
    const int jp = j0 - DP_IB, c0 = j0 + DP_IB * cg;
    __builtin_amdgcn_s_setprio(3);
    // The factored tile of s is needed by the sub-panel's own rows (column group 0: the row recurrence) and by the U row-block solve
    // (row block 0 of every column group).  Every other workgroup -- most of a tall panel's -- only applies the rank-32 update: it
    // skips the tile altogether (the 32-step factorization is 64 barriers of latency it would otherwise sit through, holding a
    // slot on a CU the chain's next launches are waiting for).
    const bool need_tile = cg == 0 || rb == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, r = e & 31, c = e >> 5;
        if (need_tile) {
            T[r][c] = P[(j0 + r) + (long long)(j0 + c) * ld];
            Lt[r][c] = P[(j0 + r) + (long long)(jp + c) * ld];
            U0[r][c] = P[(jp + r) + (long long)(j0 + c) * ld];
        }
        if (cg > 0) U1[r][c] = P[(jp + r) + (long long)(c0 + c) * ld];
    }
    // this thread's row: requested before the tile work so that the loads are in flight under it
    const long long r = (long long)j0 + (cg == 0 ? DP_IB : 0) + (long long)rb * 256 + tid;
    const bool live = r < rows;
    double x[DP_IB], mv[DP_IB];
    {
        const long long rr = live ? r : (long long)rows - 1;
#pragma unroll
        for (int j = 0; j < DP_IB; ++j) mv[j] = P[rr + (long long)(jp + j) * ld];
#pragma unroll
        for (int cc = 0; cc < DP_IB; ++cc) x[cc] = P[rr + (long long)(c0 + cc) * ld];
    }
    __syncthreads();
    // ---- tile: rank-32 update from sub-panel s-1, then the factorization (dgetf2_native_npv.cu:24-29) -------------
    if (need_tile) {
        double t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int e = tid + 256 * i; t[i] = T[e & 31][e >> 5]; }
#pragma unroll
        for (int j = 0; j < DP_IB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int e = tid + 256 * i; t[i] = mulsub<FUSED>(t[i], Lt[e & 31][j], U0[j][e >> 5]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int e = tid + 256 * i; T[e & 31][e >> 5] = t[i]; }
    }
    __syncthreads();
    for (int j = 0; need_tile && j < DP_IB; ++j) {
        const double piv = T[j][j];
        if (tid < DP_IB && tid > j) T[tid][j] = T[tid][j] / piv;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, rr = e & 31, c = e >> 5;
            if (rr > j && c > j) T[rr][c] = mulsub<FUSED>(T[rr][c], T[rr][j], T[j][c]);
        }
        __syncthreads();
    }
    if (rb == 0 && cg == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, rr = e & 31, c = e >> 5;
            tile_out[rr * DP_IB + c] = T[rr][c];
        }
        if (tid == 0 && info)
            for (int j = 0; j < DP_IB; ++j)
                if (T[j][j] == 0.0) { atomicMin(info, info_base + j0 + j + 1); break; }
    }
    // ---- this thread's row: update from sub-panel s-1 ------------------------------------------------------------------
    {
        lds_cdouble *ub = (lds_cdouble *)(cg == 0 ? &U0[0][0] : &U1[0][0]);
#pragma unroll
        for (int j = 0; j < DP_IB; ++j) {
            lds_cdouble *uj = opaque_lds(ub + j * DP_IB);
#pragma unroll
            for (int cc = 0; cc < DP_IB; ++cc) x[cc] = mulsub<FUSED>(x[cc], mv[j], uj[cc]);
        }
    }
    if (cg == 0) {
        // rows below the tile: the row recurrence of dpanel_sub
        if (live) {
#pragma unroll
            for (int j = 0; j < DP_IB; ++j) {
                lds_cdouble *tj = opaque_lds((lds_cdouble *)&T[j][0]);
                const double m = x[j] / tj[j];
                x[j] = m;
#pragma unroll
                for (int c = j + 1; c < DP_IB; ++c) x[c] = mulsub<FUSED>(x[c], m, tj[c]);
            }
#pragma unroll
            for (int c = 0; c < DP_IB; ++c) P[r + (long long)(c0 + c) * ld] = x[c];
        }
        return;
    }
    if (rb == 0) {
        // the tile's rows (threads 0..31) hand their updated values to one thread per column for the U row-block solve
        if (tid < DP_IB) {
#pragma unroll
            for (int cc = 0; cc < DP_IB; ++cc) S[tid][cc] = x[cc];
        }
        __syncthreads();
        if (tid < DP_IB) {
            double y[DP_IB];
#pragma unroll
            for (int i = 0; i < DP_IB; ++i) y[i] = S[i][tid];
#pragma unroll
            for (int j = 0; j < DP_IB; ++j) {
                lds_cdouble *tj = opaque_lds((lds_cdouble *)&T[0][j]);
#pragma unroll
                for (int i = j + 1; i < DP_IB; ++i) y[i] = mulsub<FUSED>(y[i], tj[i * (DP_IB + 1)], y[j]);
            }
            double *pc = P + j0 + (long long)(c0 + tid) * ld;
#pragma unroll
            for (int i = 0; i < DP_IB; ++i) pc[i] = y[i];
            return;
        }
    }
    if (live) {
#pragma unroll
        for (int cc = 0; cc < DP_IB; ++cc) P[r + (long long)(c0 + cc) * ld] = x[cc];
    }
}


// The fused form in pieces (one launch each), so that a caller can interleave them with other work of the same stream: the
// look-ahead chain runs piece s as soon as the pivot kernel has fixed the pivots of columns < 32 (s + 1).
int dgetf2_npv_pieces(mpf_ctx *c, int cols) { return (c->tune.dpanel_fused_form && cols % DP_IB == 0 && cols >= 2 * DP_IB) ? cols / DP_IB : 0; }
int launch_dgetf2_npv_piece(mpf_ctx *c, double *P, int64_t ld, int rows, int cols, int fused, int info_base, int piece) {
    const int np = dgetf2_npv_pieces(c, cols);
    if (np == 0 || piece < 0 || piece >= np || cols > rows) { c->err = "dgetf2_npv_piece: shape not covered"; return -1; }
    int *info = &c->ws->info;
    const int ntiles = np;
    if (ntiles > c->dtiles_cap) {
        if (c->dtiles) (void)hipFree(c->dtiles);
        c->dtiles = nullptr; c->dtiles_cap = 0;
        const int cap = ntiles < 8 ? 8 : ntiles;
        MPF_HIP_TRY(c, hipMalloc((void **)&c->dtiles, (size_t)cap * DP_IB * DP_IB * sizeof(double)));
        c->dtiles_cap = cap;
    }
    if (piece == 0) {
        const int gb0 = (int)(((long long)rows - DP_IB + 255) / 256);
        if (fused) dpanel_sub_kernel<true, true><<<1 + gb0, 256, 0, c->stream>>>(P, ld, rows, cols, 0, DP_IB, info, info_base, c->dtiles);
        else dpanel_sub_kernel<false, true><<<1 + gb0, 256, 0, c->stream>>>(P, ld, rows, cols, 0, DP_IB, info, info_base, c->dtiles);
    } else {
        const int j0 = piece * DP_IB;
        dim3 grid((unsigned)(((long long)rows - j0 + 255) / 256), (unsigned)((cols - j0) / DP_IB));
        double *tile = c->dtiles + (size_t)piece * DP_IB * DP_IB;
        if (fused) dpanel_fused_kernel<true><<<grid, 256, 0, c->stream>>>(P, ld, rows, cols, j0, info, info_base, tile);
        else dpanel_fused_kernel<false><<<grid, 256, 0, c->stream>>>(P, ld, rows, cols, j0, info, info_base, tile);
    }
    if (piece == np - 1) dpanel_tiles_store_kernel<<<ntiles, 256, 0, c->stream>>>(P, ld, cols, c->dtiles);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

// The factored 32 x 32 diagonal tile of sub-panel `piece` of the panel being factored in pieces, written to dst (column-major,
// leading dimension ld).  Until the last piece the matrix itself still holds the UNfactored tile (see dpanel_sub above): whoever
// copies a finished sub-panel out of the matrix before then (mpf_factor_dist's message instalments) takes the tile from here.
int launch_dpanel_tile_copy(mpf_ctx *c, double *dst, int64_t ld, int piece) {
    if (!c->dtiles || piece < 0 || piece >= c->dtiles_cap) { c->err = "dpanel tile copy: no such parked tile"; return -1; }
    dpanel_tiles_store_kernel<<<1, 256, 0, c->stream>>>(dst, ld, DP_IB, c->dtiles + (size_t)piece * DP_IB * DP_IB);
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
    const int fused_form = c->tune.dpanel_fused_form;
    if (fused_form && cols % DP_IB == 0 && cols >= 2 * DP_IB) {
        // sub-panel 0 as below (its U row-block solve covers all columns right of it), then one fused launch per sub-panel
        const int gb0 = (int)(((long long)rows - DP_IB + 255) / 256);
        if (fused) dpanel_sub_kernel<true, true><<<1 + gb0, 256, 0, c->stream>>>(P, ld, rows, cols, 0, DP_IB, info, info_base, c->dtiles);
        else dpanel_sub_kernel<false, true><<<1 + gb0, 256, 0, c->stream>>>(P, ld, rows, cols, 0, DP_IB, info, info_base, c->dtiles);
        for (int j0 = DP_IB; j0 < cols; j0 += DP_IB) {
            dim3 grid((unsigned)(((long long)rows - j0 + 255) / 256), (unsigned)((cols - j0) / DP_IB));
            double *tile = c->dtiles + (size_t)(j0 / DP_IB) * DP_IB * DP_IB;
            if (fused) dpanel_fused_kernel<true><<<grid, 256, 0, c->stream>>>(P, ld, rows, cols, j0, info, info_base, tile);
            else dpanel_fused_kernel<false><<<grid, 256, 0, c->stream>>>(P, ld, rows, cols, j0, info, info_base, tile);
        }
        dpanel_tiles_store_kernel<<<ntiles, 256, 0, c->stream>>>(P, ld, cols, c->dtiles);
        MPF_HIP_TRY(c, hipGetLastError());
        return 0;
    }
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

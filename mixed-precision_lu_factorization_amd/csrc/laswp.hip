// Row interchanges on the fp64 matrix: replaces LASWP_kernel (reference MPF.cu:42-59, launch :162).
//
// The reference gives every column to one thread that walks the `cols` swaps one after the other
// (a dependent chain of uncoalesced 8-byte accesses).  Here the chain is resolved once per panel:
//   laswp_plan   -- one workgroup turns the sequential swap list into a net permutation of the
//                   <= 2*cols rows it touches: a list (src -> dst) of rows that really move.
//   laswp_apply  -- every column reads all its moving elements into registers, waits for them, and
//                   writes them to their destinations: two independent passes instead of `cols`
//                   dependent round trips.  A workgroup handles LASWP_CPB columns per pass.
#include "mpf_internal.h"

constexpr int LASWP_CPB = 16; // columns per workgroup pass (32 independent 8-byte gathers in flight per thread)

__global__ __launch_bounds__(256) void laswp_plan_kernel(const int *ipiv, int k, int cols, MovedList *out) {
    // slots 0..cols-1 <-> rows k..k+cols-1; a pivot row beyond the panel's top block gets slot
    // cols + (index of its first occurrence in ipiv)
    __shared__ int slot[HP_MAXCOLS];
    __shared__ int rowof[2 * HP_MAXCOLS];
    __shared__ int content[2 * HP_MAXCOLS];
    __shared__ int used[2 * HP_MAXCOLS];
    __shared__ int count;
    const int t = threadIdx.x;
    for (int s = t; s < 2 * cols; s += 256) { content[s] = s; used[s] = s < cols; rowof[s] = s < cols ? k + s : -1; }
    if (t == 0) count = 0;
    __syncthreads();
    if (t < cols) {
        const int p = ipiv[t] - 1; // 0-based global row, MPF.cu:49
        int s;
        if (p < k + cols) s = p - k;
        else {
            int first = t;
            for (int i = 0; i < t; ++i)
                if (ipiv[i] - 1 == p) { first = i; break; }
            s = cols + first;
            if (first == t) { rowof[s] = p; used[s] = 1; }
        }
        slot[t] = s;
    }
    __syncthreads();
    if (t == 0) { // the sequential part: cols swaps on a 2*cols-entry label array (MPF.cu:47-57)
        for (int j = 0; j < cols; ++j) {
            const int s = slot[j];
            if (s != j) { const int tmp = content[j]; content[j] = content[s]; content[s] = tmp; }
        }
    }
    __syncthreads();
    for (int s = t; s < 2 * cols; s += 256)
        if (used[s] && content[s] != s) {
            const int i = atomicAdd(&count, 1);
            out->src[i] = rowof[content[s]];
            out->dst[i] = rowof[s];
        }
    __syncthreads();
    if (t == 0) out->n = count;
}

// hole: the columns [hole_from, hole_from + hole_len) of A are skipped (ncols counts the columns that ARE processed): one launch
// covers the column ranges left and right of a panel whose own columns already have the interchange.
__global__ __launch_bounds__(256) void laswp_apply_kernel(double *A, long long lda, long long ncols,
                                                         const MovedList *ml, long long hole_from, long long hole_len) {
    int n = ml->n;
    if (n > LASWP_MAXMOVED) n = LASWP_MAXMOVED;
    if (n == 0) return;
    const int t = threadIdx.x;
    const int i0 = t, i1 = t + 256;
    const int s0 = i0 < n ? ml->src[i0] : -1, d0 = i0 < n ? ml->dst[i0] : -1;
    const int s1 = i1 < n ? ml->src[i1] : -1, d1 = i1 < n ? ml->dst[i1] : -1;
    for (long long cb = (long long)blockIdx.x * LASWP_CPB; cb < ncols; cb += (long long)gridDim.x * LASWP_CPB) {
        double v0[LASWP_CPB], v1[LASWP_CPB];
#pragma unroll
        for (int c = 0; c < LASWP_CPB; ++c) {
            const long long col = cb + c, pcol = col + (col >= hole_from ? hole_len : 0);
            v0[c] = (s0 >= 0 && col < ncols) ? A[s0 + pcol * lda] : 0.0;
            v1[c] = (s1 >= 0 && col < ncols) ? A[s1 + pcol * lda] : 0.0;
        }
        // all gathers of this pass must have RETURNED before any scatter of the pass is issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int c = 0; c < LASWP_CPB; ++c) {
            const long long col = cb + c, pcol = col + (col >= hole_from ? hole_len : 0);
            if (d0 >= 0 && col < ncols) A[d0 + pcol * lda] = v0[c];
            if (d1 >= 0 && col < ncols) A[d1 + pcol * lda] = v1[c];
        }
        // columns of the next pass are disjoint from this one: no barrier needed here
    }
}


// Plan + apply in one launch, for a few swaps on a few columns (the pipelined panel chain applies the pivots of 32 columns at a
// time to the panel's own 256 columns): every workgroup resolves the sequential swap list itself (LDS, as laswp_plan does) and
// moves its 4 columns' elements with one gather and one scatter.  ~5 us instead of the ~50 us a thread-per-column walk of 32
// dependent swaps takes.  The pivots are read with device-scope loads: they may come from a pivot kernel that is still running.
// GATED: the launch first waits (bounded) until the pivot kernel with launch sequence `seq` has published `target` columns -- the
// gate of the pipelined chain (fp16_panel.hip) folded into the launch it guards: one launch per 32-column piece less.  A gate that
// expires flags the factorization as failed (hp_timeouts, -4) and applies nothing.
struct LaswpGate { const unsigned long long *progress; int *timeouts; unsigned seq, target; unsigned long long max_ticks; };
template <bool GATED>
__global__ __launch_bounds__(256) void laswp_block_kernel(double *A, long long lda, long long ncols, int k, int cols, const int *ipiv,
                                                         long long nrows, LaswpGate g) {
    if (GATED) {
        __shared__ int go;
        if (threadIdx.x == 0) {
            int ok = 1;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                const unsigned long long v = __hip_atomic_load(g.progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(v >> 32) == g.seq && (unsigned)v >= g.target) break;
                if (__hip_atomic_load(g.timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
                if (__builtin_amdgcn_s_memrealtime() - t0 > g.max_ticks) { atomicAdd(g.timeouts, 1); ok = 0; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            go = ok;
        }
        __syncthreads();
        if (!go) return;
    }
    __shared__ int piv[HP_MAXCOLS];
    __shared__ int slot[HP_MAXCOLS];
    __shared__ int rowof[2 * HP_MAXCOLS];
    __shared__ int content[2 * HP_MAXCOLS];
    __shared__ int used[2 * HP_MAXCOLS];
    __shared__ int msrc[2 * HP_MAXCOLS], mdst[2 * HP_MAXCOLS];
    __shared__ int count;
    const int t = threadIdx.x;
    for (int s = t; s < 2 * cols; s += 256) { content[s] = s; used[s] = s < cols; rowof[s] = s < cols ? k + s : -1; }
    if (t < cols) {
        int p = __hip_atomic_load(&ipiv[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1;
        if (p < k + t || p >= nrows) p = k + t;   // (never with valid pivots: keeps every index inside the matrix)
        piv[t] = p;
    }
    if (t == 0) count = 0;
    __syncthreads();
    if (t < cols) {
        const int p = piv[t];
        int s;
        if (p < k + cols) s = p - k;
        else {
            int first = t;
            for (int i = 0; i < t; ++i)
                if (piv[i] == p) { first = i; break; }
            s = cols + first;
            if (first == t) { rowof[s] = p; used[s] = 1; }
        }
        slot[t] = s;
    }
    __syncthreads();
    if (t == 0) {
        for (int j = 0; j < cols; ++j) {
            const int s = slot[j];
            if (s != j) { const int tmp = content[j]; content[j] = content[s]; content[s] = tmp; }
        }
    }
    __syncthreads();
    for (int s = t; s < 2 * cols; s += 256)
        if (used[s] && content[s] != s) {
            const int i = atomicAdd(&count, 1);
            msrc[i] = rowof[content[s]];
            mdst[i] = rowof[s];
        }
    __syncthreads();
    const int n = count;
    // columns: 256 / 64 = 4 per pass when n <= 64; in general every thread walks entries e, e + 64, ...
    const int cl = t >> 6, e0 = t & 63;
    for (long long cb = (long long)blockIdx.x * 4; cb < ncols; cb += (long long)gridDim.x * 4) {
        const long long col = cb + cl;
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + 64 * u; v[u] = (e < n && col < ncols) ? A[msrc[e] + col * lda] : 0.0; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // every gather of the pass has returned before any scatter is issued
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + 64 * u; if (e < n && col < ncols) A[mdst[e] + col * lda] = v[u]; }
        // the next pass works on other columns: no barrier needed here
    }
}
int launch_laswp_block(mpf_ctx *c, double *A, int64_t lda, int64_t ncols, int k, int cols, const int *d_ipiv, int64_t nrows) {
    if (cols < 1 || ncols < 1) return 0;
    if (cols > HP_MAXCOLS) { c->err = "laswp block: more than 256 swaps per call"; return -1; }
    long long blocks = (ncols + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    laswp_block_kernel<false><<<(int)blocks, 256, 0, c->stream>>>(A, lda, ncols, k, cols, d_ipiv, nrows, LaswpGate{});
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
// what one workgroup of the gated interchange kernel holds of its CU while it waits for the pivot kernel (fp16_panel.hip derives from it
// how many pivot workgroups still fit beside it), and how many of them a call on `ncols` columns launches
int laswp_gated_footprint(int *lds_bytes, int *vgprs, int *threads) {
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, (const void *)laswp_block_kernel<true>) != hipSuccess) return -1;
    *lds_bytes = (int)fa.sharedSizeBytes; *vgprs = fa.numRegs; *threads = 256;
    return 0;
}
int laswp_gated_grid(int64_t ncols) {
    long long blocks = (ncols + 3) / 4;
    return (int)(blocks > 4096 ? 4096 : blocks);
}
// the same behind a gate on the running pivot kernel's progress (`target` columns published)
int launch_laswp_block_gated(mpf_ctx *c, double *A, int64_t lda, int64_t ncols, int k, int cols, const int *d_ipiv, int64_t nrows, int target) {
    if (cols < 1 || ncols < 1) return 0;
    if (cols > HP_MAXCOLS) { c->err = "laswp block: more than 256 swaps per call"; return -1; }
    long long blocks = (ncols + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (c->tune.gate_wait_value && c->hp_signal) {
        // the STREAM waits for the progress word (command processor, no CU held), then the ungated kernel.  No bound on this wait: a
        // pivot kernel that never progresses hangs the stream -- but nothing sits on a CU it needs either.  (Measured: DESIGN 4.1.)
        MPF_HIP_TRY(c, hipStreamWaitValue64(c->stream, c->hp_signal, ((unsigned long long)c->hp_seq << 32) | (unsigned)target, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
        laswp_block_kernel<false><<<(int)blocks, 256, 0, c->stream>>>(A, lda, ncols, k, cols, d_ipiv, nrows, LaswpGate{});
        MPF_HIP_TRY(c, hipGetLastError());
        return 0;
    }
    const LaswpGate g{&c->ws->hp_progress, &c->ws->hp_timeouts, c->hp_seq, (unsigned)target, (unsigned long long)c->tune.hp_gate_ticks};
    laswp_block_kernel<true><<<(int)blocks, 256, 0, c->stream>>>(A, lda, ncols, k, cols, d_ipiv, nrows, g);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

int launch_laswp(mpf_ctx *c, double *A, int64_t lda, int64_t ncols, int k, int cols, const int *d_ipiv) {
    if (cols < 1 || ncols < 1) return 0;
    if (cols > HP_MAXCOLS) { c->err = "laswp: more than 256 swaps per call"; return -1; }
    laswp_plan_kernel<<<1, 256, 0, c->stream>>>(d_ipiv, k, cols, &c->ws->list0);
    MPF_HIP_TRY(c, hipGetLastError());
    long long blocks = (ncols + LASWP_CPB - 1) / LASWP_CPB;
    if (blocks > 8192) blocks = 8192;
    laswp_apply_kernel<<<(int)blocks, 256, 0, c->stream>>>(A, lda, ncols, &c->ws->list0, ncols, 0);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

int launch_laswp_plan(mpf_ctx *c, const int *d_ipiv, int k, int cols, MovedList *out) {
    if (cols < 1 || cols > HP_MAXCOLS) { c->err = "laswp plan: 1 <= swaps <= 256"; return -1; }
    laswp_plan_kernel<<<1, 256, 0, c->stream>>>(d_ipiv, k, cols, out);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

int launch_laswp_from_list(mpf_ctx *c, double *A, int64_t lda, int64_t ncols, const MovedList *ml) {
    return launch_laswp_from_list_hole(c, A, lda, ncols, ml, ncols, 0);
}
// the columns [0, hole_from) and [hole_from + hole_len, ncols + hole_len) of A in one launch
int launch_laswp_from_list_hole(mpf_ctx *c, double *A, int64_t lda, int64_t ncols, const MovedList *ml, int64_t hole_from, int64_t hole_len) {
    if (ncols < 1) return 0;
    long long blocks = (ncols + LASWP_CPB - 1) / LASWP_CPB;
    if (blocks > 8192) blocks = 8192;
    laswp_apply_kernel<<<(int)blocks, 256, 0, c->stream>>>(A, lda, ncols, ml, hole_from, hole_len);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}


// ---- fp32 working copy of the trailing matrix (two-level schedule of the fp16 trailing modes) -----------------------------------
// The copy is ROW-major: W[row * ldw + col] (the fp64 matrix is column-major, the reference's layout; the copy is the
// library's own and only the fp16 update kernel -- symmetric in its two operands -- the interchanges and these conversions
// ever see it).  A row interchange then moves contiguous row segments instead of one 4-byte element per 64-byte sector: with
// real pivoting (the generator's matrices) the interchanges right of a super-panel took 40 ms of an fp16-mode factorization
// at N = 32768 in the column-major copy of round 2.
//   launch_laswp_from_list_f32: rows src[i] -> dst[i] of the columns [0, ncols) at W (two passes through a scratch image:
//                               the list is a permutation, sources must be read before any destination is written);
//   launch_cvt_f64_f32 / _f32_f64: A (column-major fp64) <-> W (row-major fp32) through 64 x 64 LDS transposes.
template <typename T_>
__global__ __launch_bounds__(256) void wt_rows_gather_kernel(const T_ *__restrict__ W, long long ldw, long long ncols,
                                                            const MovedList *__restrict__ ml, T_ *__restrict__ T) {
    int n = ml->n;
    if (n > LASWP_MAXMOVED) n = LASWP_MAXMOVED;
    const int i = blockIdx.y;
    if (i >= n) return;
    const T_ *src = W + (long long)ml->src[i] * ldw;
    T_ *dst = T + (long long)i * ncols;
    const long long c0 = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const long long c = c0 + 256 * j; if (c < ncols) dst[c] = src[c]; }
}
template <typename T_>
__global__ __launch_bounds__(256) void wt_rows_scatter_kernel(T_ *__restrict__ W, long long ldw, long long ncols,
                                                             const MovedList *__restrict__ ml, const T_ *__restrict__ T) {
    int n = ml->n;
    if (n > LASWP_MAXMOVED) n = LASWP_MAXMOVED;
    const int i = blockIdx.y;
    if (i >= n) return;
    T_ *dst = W + (long long)ml->dst[i] * ldw;
    const T_ *src = T + (long long)i * ncols;
    const long long c0 = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const long long c = c0 + 256 * j; if (c < ncols) dst[c] = src[c]; }
}
// W = the copy's element (row 0, first column of the range); scratch = c->perm_tmp (N x nb doubles >= 2 nb rows x N floats)
int launch_laswp_from_list_f32(mpf_ctx *c, float *W, int64_t ldw, int64_t ncols, const MovedList *ml) {
    if (ncols < 1) return 0;
    if (!c->perm_tmp || (size_t)c->perm_cap * sizeof(double) < (size_t)LASWP_MAXMOVED * (size_t)ncols * sizeof(float)) {
        c->err = "laswp (fp32 copy): scratch too small"; return -1;
    }
    float *T = (float *)c->perm_tmp;
    dim3 grid((unsigned)((ncols + 1023) / 1024), LASWP_MAXMOVED);
    wt_rows_gather_kernel<float><<<grid, 256, 0, c->stream>>>(W, ldw, ncols, ml, T);
    wt_rows_scatter_kernel<float><<<grid, 256, 0, c->stream>>>(W, ldw, ncols, ml, T);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
// the same on an fp64 row-major copy (fp64 mode's working copy of the trailing matrix, mpf_host.cpp factor_lookahead_rm);
// scratch = c->rm_tmp (2 * HP_MAXCOLS rows x N doubles)
// scratch_off: first element of the scratch this launch may use (two lanes of the schedule interchange disjoint column ranges
// at the same time: each gets its own part of the scratch)
int launch_laswp_from_list_rm64(mpf_ctx *c, double *R, int64_t ldr, int64_t ncols, const MovedList *ml, int64_t scratch_off) {
    if (ncols < 1) return 0;
    if (!c->rm_tmp || scratch_off < 0 || c->rm_tmp_cap < scratch_off + (int64_t)LASWP_MAXMOVED * ncols) {
        c->err = "laswp (fp64 row-major copy): scratch too small"; return -1;
    }
    double *T = c->rm_tmp + scratch_off;
    dim3 grid((unsigned)((ncols + 1023) / 1024), LASWP_MAXMOVED);
    wt_rows_gather_kernel<double><<<grid, 256, 0, c->stream>>>(R, ldr, ncols, ml, T);
    wt_rows_scatter_kernel<double><<<grid, 256, 0, c->stream>>>(R, ldr, ncols, ml, T);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
// fp64 window, column-major (A, lda) <-> row-major (R, ldr): 64 x 64 LDS transposes, values copied bit for bit
template <bool TO_ROWMAJOR>
__global__ __launch_bounds__(256) void transpose64_kernel(double *__restrict__ A, long long lda, double *__restrict__ R, long long ldr,
                                                         long long rows, long long cols) {
    __shared__ double t[64][65];
    const long long r0 = (long long)blockIdx.x * 64, c0 = (long long)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (TO_ROWMAJOR) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int cc = ty + 4 * i; if (r0 + tx < rows && c0 + cc < cols) t[cc][tx] = A[r0 + tx + (c0 + cc) * lda]; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int rr = ty + 4 * i; if (r0 + rr < rows && c0 + tx < cols) R[(r0 + rr) * ldr + c0 + tx] = t[tx][rr]; }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int rr = ty + 4 * i; if (r0 + rr < rows && c0 + tx < cols) t[rr][tx] = R[(r0 + rr) * ldr + c0 + tx]; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int cc = ty + 4 * i; if (r0 + tx < rows && c0 + cc < cols) A[r0 + tx + (c0 + cc) * lda] = t[tx][cc]; }
    }
}
int launch_transpose64(mpf_ctx *c, double *A, int64_t lda, double *R, int64_t ldr, int64_t rows, int64_t cols, bool to_rowmajor) {
    if (rows <= 0 || cols <= 0) return 0;
    if ((cols + 63) / 64 > 65535) { c->err = "transpose: too many columns"; return -1; }
    dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((cols + 63) / 64));
    if (to_rowmajor) transpose64_kernel<true><<<grid, 256, 0, c->stream>>>(A, lda, R, ldr, rows, cols);
    else transpose64_kernel<false><<<grid, 256, 0, c->stream>>>(A, lda, R, ldr, rows, cols);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
__global__ __launch_bounds__(256) void cvt_a64_wt32_kernel(const double *__restrict__ A, long long lda, float *__restrict__ W, long long ldw,
                                                          long long rows, long long cols) {
    __shared__ float t[64][65];
    const long long r0 = (long long)blockIdx.x * 64, c0 = (long long)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int cc = ty + 4 * i;
        if (r0 + tx < rows && c0 + cc < cols) t[cc][tx] = (float)A[r0 + tx + (c0 + cc) * lda];   // lanes along a column of A
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int rr = ty + 4 * i;
        if (r0 + rr < rows && c0 + tx < cols) W[(r0 + rr) * ldw + c0 + tx] = t[tx][rr];           // lanes along a row of W
    }
}
__global__ __launch_bounds__(256) void cvt_wt32_a64_kernel(const float *__restrict__ W, long long ldw, double *__restrict__ A, long long lda,
                                                          long long rows, long long cols) {
    __shared__ float t[64][65];
    const long long r0 = (long long)blockIdx.x * 64, c0 = (long long)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int rr = ty + 4 * i;
        if (r0 + rr < rows && c0 + tx < cols) t[rr][tx] = W[(r0 + rr) * ldw + c0 + tx];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int cc = ty + 4 * i;
        if (r0 + tx < rows && c0 + cc < cols) A[r0 + tx + (c0 + cc) * lda] = (double)t[tx][cc];
    }
}
// W[r * ldw + c] = (float) A[r + c * lda] for r < rows, c < cols   /   A = (double) W
int launch_cvt_f64_f32(mpf_ctx *c, const double *A, int64_t lda, float *W, int64_t ldw, int64_t rows, int64_t cols) {
    if (rows <= 0 || cols <= 0) return 0;
    if ((cols + 63) / 64 > 65535) { c->err = "cvt: too many columns"; return -1; }
    dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((cols + 63) / 64));
    cvt_a64_wt32_kernel<<<grid, 256, 0, c->stream>>>(A, lda, W, ldw, rows, cols);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
int launch_cvt_f32_f64(mpf_ctx *c, const float *W, int64_t ldw, double *A, int64_t lda, int64_t rows, int64_t cols) {
    if (rows <= 0 || cols <= 0) return 0;
    if ((cols + 63) / 64 > 65535) { c->err = "cvt: too many columns"; return -1; }
    dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((cols + 63) / 64));
    cvt_wt32_a64_kernel<<<grid, 256, 0, c->stream>>>(W, ldw, A, lda, rows, cols);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Deferred interchanges of the columns LEFT of each panel.  The reference swaps all N columns at every panel
// (MPF.cu:162); the columns left of the panel are finished L columns that nothing reads again before the end, so
// their swaps can wait and be composed: column block b must undergo the swaps of panels b+1, b+2, ..., last, i.e.
// the composite position map F_b = P_last o ... o P_{b+1}, and F_b = F_{b+1} o P_{b+1} differs from F_{b+1} only
// on the <= 512 rows panel b+1 moves.  Walking b downwards, every element of the lower triangle is moved ONCE
// (contiguous reads, writes absorbed by the caches) instead of up to 127 times as scattered 8-byte accesses.
// ---------------------------------------------------------------------------------------------------------------
__global__ void lazy_init_map_kernel(int *F, int *G, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { F[i] = (int)i; G[i] = (int)i; }
}
// F <- F o P  for the moved rows of one panel: F_new[src] = F_old[dst]; G = F^-1 is kept alongside (G[F[r]] = r)
__global__ __launch_bounds__(512) void lazy_update_map_kernel(int *F, int *G, const MovedList *ml) {
    int n = ml->n;
    if (n > LASWP_MAXMOVED) n = LASWP_MAXMOVED;
    const int t = threadIdx.x;
    int v = 0;
    if (t < n) v = F[ml->dst[t]];
    __syncthreads();
    if (t < n) { const int s = ml->src[t]; F[s] = v; G[v] = s; }
}
// T[d - r0, c] = A[G[d], c] for rows d >= r0 of `w` columns (thread = destination row: coalesced writes; the scattered 8-byte
// READS of a column segment find their sectors in the caches after the first touch, where scattered writes end as partial
// sectors in HBM)
__global__ __launch_bounds__(256) void lazy_gather_kernel(const double *__restrict__ A, long long lda, long long n, long long r0,
                                                         int w, const int *__restrict__ G, double *__restrict__ T, long long ldt) {
    const long long d = r0 + (long long)blockIdx.x * 256 + threadIdx.x;
    if (d >= n) return;
    const long long sr = G[d];
    const int c0 = blockIdx.y * 16;
#pragma unroll 8
    for (int c = c0; c < c0 + 16 && c < w; ++c) T[(d - r0) + (long long)c * ldt] = A[sr + (long long)c * lda];
}
// T[F[r] - r0, c] = A[r, c] for rows r >= r0 of `w` columns (thread = row: coalesced reads)
__global__ __launch_bounds__(256) void lazy_scatter_kernel(const double *__restrict__ A, long long lda, long long n, long long r0,
                                                          int w, const int *__restrict__ F, double *__restrict__ T, long long ldt) {
    const long long r = r0 + (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const long long d = (long long)F[r] - r0;
    const int c0 = blockIdx.y * 16;
#pragma unroll 4
    for (int c = c0; c < c0 + 16 && c < w; ++c) T[d + (long long)c * ldt] = A[r + (long long)c * lda];
}
__global__ __launch_bounds__(256) void lazy_copyback_kernel(double *__restrict__ A, long long lda, long long n, long long r0, int w,
                                                           const double *__restrict__ T, long long ldt) {
    const long long r = r0 + (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int c0 = blockIdx.y * 16;
#pragma unroll 4
    for (int c = c0; c < c0 + 16 && c < w; ++c) A[r + (long long)c * lda] = T[(r - r0) + (long long)c * ldt];
}

// sb > 1 (super-panels of sb panels, fp16 trailing modes): the schedule applies a panel's interchanges to the earlier
// columns of its own super-panel right away (the deferred K = sb * nb update reads them), so column block b is only
// owed the panels from the next super-panel on.
// world > 1 (1-D block-cyclic column layout, mpf_dist.cpp; sb == 1 there): A holds only the blocks b = rank (mod world), block b at
// local column (b / world) * nb; the composite map is built from every panel's list on every rank.
int launch_lazy_left_swaps(mpf_ctx *c, double *A, int64_t lda, int64_t N, int nb, int npanels, const MovedList *lists, int sb,
                           int world, int rank) {
    if (npanels < 2) return 0;
    if (sb < 1 || world > 1) sb = 1;
    const int gather = c->tune.lazy_gather;
    int *Gmap = c->Fmap + N;
    lazy_init_map_kernel<<<(int)((N + 255) / 256), 256, 0, c->stream>>>(c->Fmap, Gmap, N);
    for (int p = npanels - 1; p >= 1; --p) {
        lazy_update_map_kernel<<<1, 512, 0, c->stream>>>(c->Fmap, Gmap, lists + p);
        if (p % sb) continue;                      // F = composite of panels p..last: due for the blocks of the super-panel before p
        const int64_t r0 = (int64_t)p * nb;        // panels >= p only touch rows >= p*nb
        const int64_t rows = N - r0;
        if (rows <= 0) continue;
        for (int b = p - sb; b < p; ++b) {
            if (world > 1 && b % world != rank) continue;
            const int w = nb;                      // blocks left of a panel are never the (possibly narrower) last one
            dim3 grid((unsigned)((rows + 255) / 256), (unsigned)((w + 15) / 16));
            double *Ab = A + (int64_t)(world > 1 ? b / world : b) * nb * lda;
            if (gather) lazy_gather_kernel<<<grid, 256, 0, c->stream>>>(Ab, lda, N, r0, w, Gmap, c->perm_tmp, rows);
            else lazy_scatter_kernel<<<grid, 256, 0, c->stream>>>(Ab, lda, N, r0, w, c->Fmap, c->perm_tmp, rows);
            lazy_copyback_kernel<<<grid, 256, 0, c->stream>>>(Ab, lda, N, r0, w, c->perm_tmp, rows);
        }
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

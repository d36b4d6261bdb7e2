// Build-added iterative-refinement sweep (no reference counterpart; BASELINE north_star):
// fp64 residual r = b - A x (HBM-bound GEMV), triangular solves with the packed LU factors, axpy, norms.
// Everything here is bandwidth-bound streaming of column-major fp64 data: thread = row so that a wave
// reads 512 contiguous bytes of a column, x is staged in LDS and read as a broadcast.
#include "mpf_internal.h"

constexpr int RS_CCH = 512; // columns per workgroup of the residual GEMV
constexpr int TS_B = 64;    // triangular-solve block (one wave solves it with lane broadcasts)

__global__ __launch_bounds__(256) void gather_rows_kernel(const double *in, const int *perm, double *out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[perm[i]];
}

// part[chunk][row] = A[row, c0:c0+RS_CCH] * x[c0:...] for column chunk `chunk`; residual_reduce_kernel then forms
// r = b - sum over the chunks in ascending order: no atomics, the residual's bits are reproducible run to run
__global__ __launch_bounds__(256) void residual_kernel(const double *__restrict__ A, long long lda,
                                                       const double *__restrict__ x, double *__restrict__ part, long long n,
                                                       long long ncols) {
    __shared__ double xs[RS_CCH];
    const long long c0 = (long long)blockIdx.y * RS_CCH;
    const int nc = (int)((ncols - c0) < RS_CCH ? (ncols - c0) : RS_CCH);
    for (int i = threadIdx.x; i < RS_CCH; i += 256) xs[i] = i < nc ? x[c0 + i] : 0.0;
    __syncthreads();
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    const double *a = A + row + c0 * lda;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int c = 0;
    for (; c + 4 <= nc; c += 4) {
        s0 += a[(long long)(c + 0) * lda] * xs[c + 0];
        s1 += a[(long long)(c + 1) * lda] * xs[c + 1];
        s2 += a[(long long)(c + 2) * lda] * xs[c + 2];
        s3 += a[(long long)(c + 3) * lda] * xs[c + 3];
    }
    for (; c < nc; ++c) s0 += a[(long long)c * lda] * xs[c];
    part[(long long)blockIdx.y * n + row] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void residual_reduce_kernel(const double *__restrict__ part, const double *__restrict__ b,
                                                              double *__restrict__ r, long long n, int nchunks) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    double s = 0;
    for (int ch = 0; ch < nchunks; ++ch) s += part[(long long)ch * n + row];
    r[row] = (b ? b[row] : 0.0) - s;
}

// Inverses of the TS_B x TS_B diagonal blocks of L (unit lower) and U, built once per solve: afterwards a
// block step of a triangular solve is two small matrix-vector products and no dependent chain at all.
// grid = (number of blocks, 2): y = 0 -> L block, y = 1 -> U block.  Thread = column of the inverse; the
// block sits in LDS and is read as a broadcast.  Output layout inv[blk][j][i] (column-major, lane = row).
// k0 / blk0: LU's column 0 is global column k0 (a column block of a distributed matrix), the first block handled is the
// global 64-block blk0; rows are always global
__global__ __launch_bounds__(64) void trsv_invert_blocks_kernel(const double *__restrict__ LU, long long ld, long long n,
                                                               double *__restrict__ invL, double *__restrict__ invU, long long k0,
                                                               long long blk0) {
    __shared__ double D[TS_B][TS_B + 1];
    const long long kb = (blk0 + (long long)blockIdx.x) * TS_B;
    const int nb = (int)((n - kb) < TS_B ? (n - kb) : TS_B);
    const int c = threadIdx.x;
    const bool upper = blockIdx.y == 1;
    for (int j = 0; j < TS_B; ++j) // lane = row: coalesced column loads
        D[c][j] = (c < nb && j < nb) ? LU[(kb + c) + (kb - k0 + j) * ld] : (c == j ? 1.0 : 0.0);
    __syncthreads();
    double x[TS_B];
#pragma unroll
    for (int i = 0; i < TS_B; ++i) x[i] = (i == c) ? 1.0 : 0.0;
    double *out = (upper ? invU : invL) + (blk0 + (long long)blockIdx.x) * TS_B * TS_B + (long long)c * TS_B;
    if (!upper) {
#pragma unroll
        for (int j = 0; j < TS_B; ++j)
#pragma unroll
            for (int i = j + 1; i < TS_B; ++i) x[i] -= D[i][j] * x[j];
    } else {
#pragma unroll
        for (int j = TS_B - 1; j >= 0; --j) {
            x[j] = x[j] / D[j][j];
#pragma unroll
            for (int i = 0; i < j; ++i) x[i] -= D[i][j] * x[j];
        }
    }
#pragma unroll
    for (int i = 0; i < TS_B; ++i) out[i] = x[i];
}

// y_blk = inv * x_blk with all 256 threads (row = tid & 63, four 16-column parts), result in ys[] and y[]
__device__ __forceinline__ void block_apply_inverse(const double *__restrict__ inv, const double *x, double *y, long long kb,
                                                    int nb, double *ys, double *part, bool store) {
    const int tid = threadIdx.x, i = tid & 63, p = tid >> 6;
    double s = 0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = p * 16 + jj;
        const double xj = j < nb ? x[kb + j] : 0.0;
        s += inv[i + TS_B * j] * xj;
    }
    part[p * TS_B + i] = s;
    __syncthreads();
    if (tid < TS_B) {
        const double v = (part[tid] + part[TS_B + tid]) + (part[2 * TS_B + tid] + part[3 * TS_B + tid]);
        ys[tid] = tid < nb ? v : 0.0;
        if (store && tid < nb) y[kb + tid] = v;
    }
    __syncthreads();
}

// One block step of the forward solve  L y = x.  Every workgroup applies the block's inverse itself (identical
// arithmetic, no inter-workgroup dependency), then subtracts L[rows, kb:kb+TS_B] * y_blk from its 256 rows
// below.  x[kb:kb+TS_B] is read-only in this step; the finished values go to y.
__global__ __launch_bounds__(256) void trsv_lower_step_kernel(const double *__restrict__ LU, long long ld,
                                                              const double *__restrict__ invL, double *x, double *y,
                                                              long long n, long long kb, long long k0) {
    __shared__ double ys[TS_B];
    __shared__ double part[4 * TS_B];
    const int tid = threadIdx.x;
    const int nb = (int)((n - kb) < TS_B ? (n - kb) : TS_B);
    block_apply_inverse(invL + (kb / TS_B) * TS_B * TS_B, x, y, kb, nb, ys, part, blockIdx.x == 0);
    const long long row = kb + TS_B + (long long)blockIdx.x * 256 + tid;
    if (row >= n) return;
    const double *l = LU + row + (kb - k0) * ld;
    double s0 = 0, s1 = 0;
#pragma unroll 8
    for (int j = 0; j < TS_B; j += 2) {
        s0 += l[(long long)j * ld] * ys[j];
        s1 += l[(long long)(j + 1) * ld] * ys[j + 1];
    }
    x[row] -= s0 + s1;
}

// One block step of the backward solve  U y = x; rows above the block get the update.
__global__ __launch_bounds__(256) void trsv_upper_step_kernel(const double *__restrict__ LU, long long ld,
                                                              const double *__restrict__ invU, double *x, double *y,
                                                              long long n, long long kb, long long k0) {
    __shared__ double ys[TS_B];
    __shared__ double part[4 * TS_B];
    const int tid = threadIdx.x;
    const int nb = (int)((n - kb) < TS_B ? (n - kb) : TS_B);
    block_apply_inverse(invU + (kb / TS_B) * TS_B * TS_B, x, y, kb, nb, ys, part, blockIdx.x == 0);
    const long long row = (long long)blockIdx.x * 256 + tid;
    if (row >= kb) return;
    const double *u = LU + row + (kb - k0) * ld;
    double s0 = 0, s1 = 0;
    int j = 0;
    for (; j + 2 <= nb; j += 2) {
        s0 += u[(long long)j * ld] * ys[j];
        s1 += u[(long long)(j + 1) * ld] * ys[j + 1];
    }
    if (j < nb) s0 += u[(long long)j * ld] * ys[j];
    x[row] -= s0 + s1;
}

__global__ void axpy_kernel(double alpha, const double *x, double *y, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] += alpha * x[i];
}

__global__ __launch_bounds__(256) void sumsq_kernel(const double *x, long long n, double *part) {
    __shared__ double ws[4];
    double s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += x[i] * x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
// ordered sum of the per-block partials (one wave): reproducible, no atomics
__global__ __launch_bounds__(64) void sumsq_final_kernel(const double *part, int nparts, double *out) {
    double s = 0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) out[0] = s;
}

__global__ __launch_bounds__(256) void dot_kernel(const double *x, const double *y, long long n, double *part) {
    __shared__ double ws[4];
    double s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += x[i] * y[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
__global__ void scal_kernel(double alpha, double *x, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] *= alpha;
}
int launch_dot(mpf_ctx *c, const double *x, const double *y, int64_t n, double *d_out) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (1024 > (int)c->res_part_cap) {
        if (c->res_part) hipFree(c->res_part);
        c->res_part = nullptr; c->res_part_cap = 0;
        MPF_HIP_TRY(c, hipMalloc((void **)&c->res_part, 1024 * sizeof(double)));
        c->res_part_cap = 1024;
    }
    dot_kernel<<<blocks, 256, 0, c->stream>>>(x, y, n, c->res_part);
    sumsq_final_kernel<<<1, 64, 0, c->stream>>>(c->res_part, blocks, d_out);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
int launch_scal(mpf_ctx *c, double alpha, double *x, int64_t n) {
    scal_kernel<<<(int)((n + 255) / 256), 256, 0, c->stream>>>(alpha, x, n);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

int launch_gather_rows(mpf_ctx *c, const double *in, const int *perm, double *out, int64_t n) {
    gather_rows_kernel<<<(int)((n + 255) / 256), 256, 0, c->stream>>>(in, perm, out, n);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
// r = (b or 0) - A[:, 0:ncols] x[0:ncols], A n x ncols (the distributed residual: a rank's own columns; b on one rank only)
int launch_residual_rect(mpf_ctx *c, const double *A, int64_t lda, const double *x, const double *b, double *r, int64_t n, int64_t ncols) {
    const int nchunks = (int)((ncols + RS_CCH - 1) / RS_CCH);
    const size_t need = (size_t)(nchunks > 0 ? nchunks : 1) * (size_t)n;
    if (need > c->res_part_cap) {
        if (c->res_part) hipFree(c->res_part);
        c->res_part = nullptr; c->res_part_cap = 0;
        MPF_HIP_TRY(c, hipMalloc((void **)&c->res_part, need * sizeof(double)));
        c->res_part_cap = need;
    }
    if (nchunks > 0) {
        dim3 grid((unsigned)((n + 255) / 256), (unsigned)nchunks);
        residual_kernel<<<grid, 256, 0, c->stream>>>(A, lda, x, c->res_part, n, ncols);
    }
    residual_reduce_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(c->res_part, b, r, n, nchunks);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
int launch_residual(mpf_ctx *c, const double *A, int64_t lda, const double *x, const double *b, double *r, int64_t n) {
    return launch_residual_rect(c, A, lda, x, b, r, n, n);
}
// ---- 256-wide block steps (single-GPU solve).  A 64-wide step costs ~10 us of dependent round trips whatever it moves, so
// the solve took 512 x 10 us.  Here a step is 256 columns and two launches:
//   trsv_diag_kernel    ONE workgroup solves the 256 x 256 diagonal block: four 64-wide sub-steps (inverted 64 x 64 diagonal
//                       blocks, the block's own off-diagonal 64 x 64 blocks), the vector block held in LDS;
//   trsv_update_kernel  everyone: x[rows beyond the block] -= F[rows, block] * y_block, 64 rows x 4 column groups per
//                       workgroup (512 workgroups at N = 32768: every CU streams), partial sums combined in a fixed order.
constexpr int TW = 256;
template <bool UPPER>
__global__ __launch_bounds__(256) void trsv_diag_kernel(const double *__restrict__ LU, long long ld, const double *__restrict__ inv,
                                                        const double *__restrict__ x, double *__restrict__ y, long long n, long long kb) {
    __shared__ double xs[TW], ys[TW], part[4 * TS_B];
    const int tid = threadIdx.x;
    const int w = (int)((n - kb) < TW ? (n - kb) : TW);
    xs[tid] = tid < w ? x[kb + tid] : 0.0;
    __syncthreads();
    const int nq = (w + TS_B - 1) / TS_B;
    for (int qq = 0; qq < nq; ++qq) {
        const int q = UPPER ? nq - 1 - qq : qq;
        const long long sb = kb + (long long)q * TS_B;
        const int nbq = (int)((n - sb) < TS_B ? (n - sb) : TS_B);
        {   // y_q = inv_q * xs_q: row = tid & 63, four 16-column parts
            const double *iq = inv + (sb / TS_B) * TS_B * TS_B;
            const int i = tid & 63, p = tid >> 6;
            double sacc = 0;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = p * 16 + jj;
                sacc += iq[i + TS_B * j] * (j < nbq ? xs[q * TS_B + j] : 0.0);
            }
            part[p * TS_B + i] = sacc;
        }
        __syncthreads();
        if (tid < TS_B) ys[q * TS_B + tid] = tid < nbq ? (part[tid] + part[TS_B + tid]) + (part[2 * TS_B + tid] + part[3 * TS_B + tid]) : 0.0;
        __syncthreads();
        // the block's other rows lose F[rows, sub-block q] * y_q
        const int r0 = UPPER ? 0 : (q + 1) * TS_B, r1 = UPPER ? q * TS_B : w;
        const int r = r0 + tid;
        if (r < r1) {
            // 32 independent loads in flight per thread (clamped column + select: no branch per element): the block comes from
            // HBM on first touch, and a dependent load per iteration would cost a full memory round trip each
            const double *f = LU + (kb + r) + sb * ld;
            double sa[4] = {0, 0, 0, 0};
#pragma unroll
            for (int jb = 0; jb < TS_B; jb += 32) {
                double v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = f[(long long)((jb + u) < nbq ? (jb + u) : (nbq - 1)) * ld];
#pragma unroll
                for (int u = 0; u < 32; ++u) sa[u & 3] += ((jb + u) < nbq ? v[u] : 0.0) * ys[q * TS_B + jb + u];
            }
            xs[r] -= (sa[0] + sa[1]) + (sa[2] + sa[3]);
        }
        __syncthreads();
    }
    if (tid < w) y[kb + tid] = ys[tid];
}

template <bool UPPER>
__global__ __launch_bounds__(256) void trsv_update_kernel(const double *__restrict__ LU, long long ld, const double *__restrict__ y,
                                                          double *__restrict__ x, long long n, long long kb, int w) {
    __shared__ double ys[TW], part[4 * 64];
    const int tid = threadIdx.x, r = tid & 63, g = tid >> 6;
    ys[tid] = tid < w ? y[kb + tid] : 0.0;
    __syncthreads();
    const long long row = (UPPER ? 0 : kb + w) + (long long)blockIdx.x * 64 + r;
    const bool live = UPPER ? row < kb : row < n;
    double sa[4] = {0, 0, 0, 0};
    if (live) {
        const int j0 = g * 64;
        const double *f = LU + row + (kb + j0) * ld;
        const int cnt = (w - j0) < 64 ? (w - j0) : 64;        // columns of this group that exist (<= 0: none)
        if (cnt > 0) {
#pragma unroll
            for (int jb = 0; jb < 64; jb += 32) {             // 32 independent loads in flight per thread
                double v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = f[(long long)((jb + u) < cnt ? (jb + u) : (cnt - 1)) * ld];
#pragma unroll
                for (int u = 0; u < 32; ++u) sa[u & 3] += ((jb + u) < cnt ? v[u] : 0.0) * ys[j0 + jb + u];
            }
        }
    }
    part[g * 64 + r] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    __syncthreads();
    if (g == 0 && live) x[row] -= (part[r] + part[64 + r]) + (part[128 + r] + part[192 + r]);
}

// x is consumed (overwritten with intermediate values); the solution lands in y
static int trsv_lower_wide(mpf_ctx *c, const double *LU, int64_t ld, double *x, double *y, int64_t n) {
    for (int64_t kb = 0; kb < n; kb += TW) {
        const int w = (int)((n - kb) < TW ? (n - kb) : TW);
        trsv_diag_kernel<false><<<1, 256, 0, c->stream>>>(LU, ld, c->trsv_inv, x, y, n, kb);
        const int64_t below = n - kb - w;
        if (below > 0) trsv_update_kernel<false><<<(unsigned)((below + 63) / 64), 256, 0, c->stream>>>(LU, ld, y, x, n, kb, w);
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
static int trsv_upper_wide(mpf_ctx *c, const double *LU, int64_t ld, double *x, double *y, int64_t n) {
    const int64_t nblk = (n + TS_B - 1) / TS_B;
    const double *invU = c->trsv_inv + nblk * TS_B * TS_B;
    for (int64_t kb = ((n - 1) / TW) * TW; kb >= 0; kb -= TW) {
        const int w = (int)((n - kb) < TW ? (n - kb) : TW);
        trsv_diag_kernel<true><<<1, 256, 0, c->stream>>>(LU, ld, invU, x, y, n, kb);
        if (kb > 0) trsv_update_kernel<true><<<(unsigned)((kb + 63) / 64), 256, 0, c->stream>>>(LU, ld, y, x, n, kb, w);
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

int launch_trsv_lower_unit(mpf_ctx *c, const double *LU, int64_t ld, double *x, int64_t n) {
    // in: x, out: x (via the context's scratch vector)
    double *y = c->solve_buf + 3 * c->solve_n;
    int rc = trsv_lower_wide(c, LU, ld, x, y, n);
    if (rc) return rc;
    MPF_HIP_TRY(c, hipMemcpyAsync(x, y, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
int launch_trsv_upper(mpf_ctx *c, const double *LU, int64_t ld, double *x, int64_t n) {
    double *y = c->solve_buf + 3 * c->solve_n;
    int rc = trsv_upper_wide(c, LU, ld, x, y, n);
    if (rc) return rc;
    MPF_HIP_TRY(c, hipMemcpyAsync(x, y, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
int launch_axpy(mpf_ctx *c, double alpha, const double *x, double *y, int64_t n) {
    axpy_kernel<<<(int)((n + 255) / 256), 256, 0, c->stream>>>(alpha, x, y, n);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
int launch_norm2(mpf_ctx *c, const double *x, int64_t n, double *d_out) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (1024 > (int)c->res_part_cap) { // the residual's partial-sum buffer doubles as scratch here
        if (c->res_part) hipFree(c->res_part);
        c->res_part = nullptr; c->res_part_cap = 0;
        MPF_HIP_TRY(c, hipMalloc((void **)&c->res_part, 1024 * sizeof(double)));
        c->res_part_cap = 1024;
    }
    sumsq_kernel<<<blocks, 256, 0, c->stream>>>(x, n, c->res_part);
    sumsq_final_kernel<<<1, 64, 0, c->stream>>>(c->res_part, blocks, d_out);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

// build the inverted diagonal blocks of the factors for the solves that follow (once per mpf_solve_ir)
int launch_trsv_prepare(mpf_ctx *c, const double *LU, int64_t ld, int64_t n) {
    const int64_t nblk = (n + TS_B - 1) / TS_B;
    dim3 grid((unsigned)nblk, 2);
    trsv_invert_blocks_kernel<<<grid, 64, 0, c->stream>>>(LU, ld, n, c->trsv_inv, c->trsv_inv + nblk * TS_B * TS_B, 0, 0);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

// ---- column-block forms of the same steps (distributed solve, mpf_dist.cpp): LUb holds the global columns [k0, k0 + w) with
//      global rows, k0 a multiple of TS_B; the vector x is updated IN PLACE (finished values overwrite x[k0 .. k0 + w)) ----
int launch_trsv_prepare_cols(mpf_ctx *c, const double *LUb, int64_t ld, int64_t n, int64_t k0, int w) {
    if (k0 % TS_B) { c->err = "distributed solve: the panel width must be a multiple of 64"; return -1; }
    const int64_t nblk = (n + TS_B - 1) / TS_B;
    dim3 grid((unsigned)((w + TS_B - 1) / TS_B), 2);
    trsv_invert_blocks_kernel<<<grid, 64, 0, c->stream>>>(LUb, ld, n, c->trsv_inv, c->trsv_inv + nblk * TS_B * TS_B, k0, k0 / TS_B);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
int launch_trsv_lower_cols(mpf_ctx *c, const double *LUb, int64_t ld, double *x, int64_t n, int64_t k0, int w) {
    double *y = c->solve_buf + 3 * c->solve_n;
    for (int64_t kb = k0; kb < k0 + w; kb += TS_B) {
        const int64_t below = n - kb - TS_B;
        const int blocks = below > 0 ? (int)((below + 255) / 256) : 1;
        trsv_lower_step_kernel<<<blocks, 256, 0, c->stream>>>(LUb, ld, c->trsv_inv, x, y, n, kb, k0);
    }
    MPF_HIP_TRY(c, hipGetLastError());
    MPF_HIP_TRY(c, hipMemcpyAsync(x + k0, y + k0, (size_t)w * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
int launch_trsv_upper_cols(mpf_ctx *c, const double *LUb, int64_t ld, double *x, int64_t n, int64_t k0, int w) {
    double *y = c->solve_buf + 3 * c->solve_n;
    const int64_t nblk = (n + TS_B - 1) / TS_B;
    for (int64_t kb = k0 + ((w - 1) / TS_B) * TS_B; kb >= k0; kb -= TS_B) {
        const int blocks = kb > 0 ? (int)((kb + 255) / 256) : 1;
        trsv_upper_step_kernel<<<blocks, 256, 0, c->stream>>>(LUb, ld, c->trsv_inv + nblk * TS_B * TS_B, x, y, n, kb, k0);
    }
    MPF_HIP_TRY(c, hipGetLastError());
    MPF_HIP_TRY(c, hipMemcpyAsync(x + k0, y + k0, (size_t)w * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

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
// ---- 256-wide block steps (single-GPU solve), round 5 -----------------------------------------------------------------------------------
// Round 4's step was two launches: ONE workgroup solving the 256 x 256 diagonal block in four dependent sub-steps (17-20 us: eight
// dependent trips to memory), then everyone updating the rows beyond it (10 us): 8.8 ms per sweep at N = 32768 where the factors
// stream in 1.1-1.4 ms.  Now:
//   * trsv_prepare builds the FULL inverse of every 256 x 256 diagonal block of L and of U (from the 64 x 64 inverses: three levels
//     of 64-block products, 2.1 GFLOP per factorization): a diagonal step is a plain matrix-vector product, no dependent chain;
//   * one launch per step (trsv_step_kernel): the workgroups that update the rows of the NEXT diagonal block come first and count
//     themselves off; four workgroups, which fetched their 64 rows of the next block's inverse while they waited, then produce the
//     next block of the solution; everyone else streams the remaining rows meanwhile.  The waiting workgroups wait only for
//     lower-numbered ones (dispatched before them), and every wait is bounded.
// Per element the summation order is fixed (no atomics on data): a solve is reproducible bit for bit.
constexpr int TW = 256;
constexpr int TS_NEAR = TW / 64;   // update workgroups that cover the next diagonal block

// C (64 x 64, registers: thread t owns rows 4 (t & 15) .., columns 4 (t >> 4) ..) += At^T * Bs, At[m][i], Bs[m][j] in LDS
__device__ __forceinline__ void blk64_mma(const double (*At)[68], const double (*Bs)[68], double (&acc)[4][4], int r0, int c0) {
#pragma unroll 8
    for (int m = 0; m < 64; ++m) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = At[m][r0 + u]; b[u] = Bs[m][c0 + u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
    }
}
// Level d of the 256 x 256 inverses: 64-block (I, J) with |I - J| = d of every diagonal block K (grid: x = pair, y = K).
//   lower (L X = I, unit diagonal):  X[I][J] = -inv(L_II) * sum_{M = J .. I-1} L[I][M] X[M][J]
//   upper (U X = I):                 X[I][J] = -inv(U_II) * sum_{M = I+1 .. J} U[I][M] X[M][J]
// d = 0 copies the 64 x 64 inverses onto the diagonal.  inv64: [block][j][i]; inv256: [K] 256 x 256 column-major.  Entries of the
// factor beyond row / column n read as zero (the 64 x 64 inverses are padded with the identity there).
template <bool UPPER>
__device__ __forceinline__ void trsv_inv256_level(const double *__restrict__ LU, long long ld, long long n,
                                                  const double *__restrict__ inv64, double *__restrict__ inv256, int d, double (*At)[68], double (*Bs)[68]) {
    const int tid = threadIdx.x, li = tid & 63, lq = tid >> 6;
    const long long kb = (long long)blockIdx.y * TW;
    double *X = inv256 + (long long)blockIdx.y * TW * TW;
    const int q = blockIdx.x;
    const int I = UPPER ? q : q + d, J = UPPER ? q + d : q;
    if (kb + 64 * (I > J ? I : J) >= n) return;          // the block lies beyond the matrix: stays zero (workgroup-uniform)
    const double *invI = inv64 + (kb / 64 + I) * 64 * 64;
    if (d == 0) {
        for (int j = lq; j < 64; j += 4) X[(64 * I + li) + (long long)(64 * I + j) * TW] = invI[li + 64 * j];
        return;
    }
    const int r0 = (tid & 15) * 4, c0 = (tid >> 4) * 4;
    double acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
    for (int M = UPPER ? I + 1 : J; M <= (UPPER ? J : I - 1); ++M) {
        for (int m = lq; m < 64; m += 4) {                // lane = row: coalesced column reads of both blocks
            const long long row = kb + 64 * I + li, col = kb + 64 * M + m;
            At[m][li] = (row < n && col < n) ? LU[row + col * ld] : 0.0;
        }
        for (int j = lq; j < 64; j += 4) Bs[li][j] = X[(64 * M + li) + (long long)(64 * J + j) * TW];
        __syncthreads();
        blk64_mma(At, Bs, acc, r0, c0);
        __syncthreads();
    }
    // S -> Bs, inv(F_II) -> At (transposed: At[m][i] = inv[i][m]), X[I][J] = -At^T * Bs
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) Bs[r0 + u][c0 + v] = acc[u][v];
    for (int m = lq; m < 64; m += 4) At[m][li] = invI[li + 64 * m];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
    blk64_mma(At, Bs, acc, r0, c0);
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int u = 0; u < 4; ++u) X[(64 * I + r0 + u) + (long long)(64 * J + c0 + v) * TW] = -acc[u][v];
}
// both factors in one launch: blockIdx.z = 0 -> L, 1 -> U
__global__ __launch_bounds__(256) void trsv_inv256_level_kernel(const double *__restrict__ LU, long long ld, long long n, const double *__restrict__ inv64L,
                                                                const double *__restrict__ inv64U, double *__restrict__ inv256L, double *__restrict__ inv256U, int d) {
    __shared__ __attribute__((aligned(16))) double At[64][68], Bs[64][68];
    if (blockIdx.z == 0) trsv_inv256_level<false>(LU, ld, n, inv64L, inv256L, d, At, Bs);
    else trsv_inv256_level<true>(LU, ld, n, inv64U, inv256U, d, At, Bs);
}

// One step of a triangular solve with the factor F = L (unit lower; blocks ascending) or U (blocks descending).  `kb`: first row of the
// block whose solution y[kb .. kb + 256) is already known (kb < 0: none yet -- the first launch only produces the first block);
// `kn`: first row of the next block to solve (kn < 0: none).  Workgroup roles by index:
//   update workgroups (64 rows each, nearest to the solved block first):  x[rows] -= F[rows, kb .. kb + w) y[kb ..]
//       the first `nnear` of them hold the rows of the next block: when their rows are stored they add 1 to *cnt (agent-scope release);
//   diag workgroups, indices nnear .. nnear + 3 when kn >= 0:  wait (bounded) for *cnt == nnear, then
//       y[kn + 64 g + r] = sum_j inv256_next[64 g + r][j] x[kn + j]   (their 128 KB of the inverse are in registers by then).
// A give-up of the wait flags *timeouts (the host turns it into an error) and still completes the launch.
template <bool UPPER>
__global__ __launch_bounds__(256) void trsv_step_kernel(const double *__restrict__ F, long long ld, const double *__restrict__ inv256n,
                                                        double *x, double *y, long long n, long long kb, int w, long long kn, int nnear,
                                                        int nupd, int *cnt, long long spin_limit, int *timeouts) {
    __shared__ double ys[TW], part[4 * 64];
    const int tid = threadIdx.x, r = tid & 63, g = tid >> 6;
    const int b = blockIdx.x;
    const bool diag = kn >= 0 && b >= nnear && b < nnear + 4;
    if (diag) {
        const int part_i = b - nnear;                    // rows 64 part_i .. of the next block
        const int wn = (int)((n - kn) < TW ? (n - kn) : TW);
        // the inverse's rows first: they depend on nothing
        double iv[64];
        const double *ip = inv256n + (64 * part_i + r) + (long long)(64 * g) * TW;
#pragma unroll
        for (int j = 0; j < 64; ++j) iv[j] = ip[(long long)j * TW];
        if (nnear > 0) {
            // Hand-off (MI355X_MICROARCH.md, cross-workgroup visibility): the producers store their rows write-through (sc1), every
            // storing wave waits for its stores, a workgroup barrier, ONE lane adds to the counter (relaxed, agent scope) -- no
            // release fence, whose L2 write-back costs 2-6 us beside the far rows' traffic.  Here: ONE lane polls (relaxed, sc1), then
            // the workgroup barrier; the loads of x are sc1 loads.
            if (tid == 0) {
                long long spins = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nnear) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > spin_limit) { atomicAdd(timeouts, 1); break; }
                }
            }
            __syncthreads();   // (no L1 invalidate: the rows are read with sc1 loads, which bypass L1, and no workgroup of this launch has
                               //  touched them before on this CU; the producers' stores were write-through)
        }
        ys[tid] = tid < wn ? __hip_atomic_load(&x[kn + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        __syncthreads();
        double sa[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 64; ++j) sa[j & 3] += iv[j] * ys[64 * g + j];
        part[g * 64 + r] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
        __syncthreads();
        if (g == 0 && 64 * part_i + r < wn) y[kn + 64 * part_i + r] = (part[r] + part[64 + r]) + (part[128 + r] + part[192 + r]);
        return;
    }
    if (kb < 0) return;
    const int u = (kn >= 0 && b >= nnear + 4) ? b - 4 : b;                // update chunk, nearest first
    if (u >= nupd) return;
    const double yv = tid < w ? y[kb + tid] : 0.0;
    const long long row = UPPER ? kb - 64ll * (u + 1) + r : kb + w + 64ll * u + r;
    const bool live = UPPER ? row >= 0 : row < n;
    const int j0 = g * 64;
    const int cnt_c = (w - j0) < 64 ? (w - j0) : 64;              // columns of this group that exist (<= 0: none)
    double sa[4] = {0, 0, 0, 0};
    double v[64];                                                 // all 64 loads of the thread in flight together with the y load: one trip to memory
    if (live && cnt_c > 0) {
        const double *f = F + row + (kb + j0) * ld;
#pragma unroll
        for (int q = 0; q < 64; ++q) v[q] = f[(long long)(q < cnt_c ? q : (cnt_c - 1)) * ld];
    }
    ys[tid] = yv;
    __syncthreads();
    if (live && cnt_c > 0) {
#pragma unroll
        for (int q = 0; q < 64; ++q) sa[q & 3] += (q < cnt_c ? v[q] : 0.0) * ys[j0 + q];
    }
    part[g * 64 + r] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    __syncthreads();
    const bool near = kn >= 0 && u < nnear;
    if (g == 0 && live) {
        const double v = x[row] - ((part[r] + part[64 + r]) + (part[128 + r] + part[192 + r]));
        if (near) __hip_atomic_store(&x[row], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read by another workgroup of this launch
        else x[row] = v;
    }
    if (near) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores of x have been acknowledged
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// x is consumed (overwritten with intermediate values); the solution lands in y
template <bool UPPER>
static int trsv_wide(mpf_ctx *c, const double *LU, int64_t ld, double *x, double *y, int64_t n) {
    const int64_t nblk = (n + TW - 1) / TW;
    const double *inv = c->trsv_inv256 + (UPPER ? nblk * TW * TW : 0);
    int *cnt = c->trsv_cnt + (UPPER ? nblk + 1 : 0);
    MPF_HIP_TRY(c, hipMemsetAsync(cnt, 0, (size_t)(nblk + 1) * sizeof(int), c->stream));
    const long long spin = c->tune.hp_spin_limit;
    // launch s = 0: the first block's solution; launch s >= 1: update from block s - 1 (in solve order) + the next block's solution
    for (int64_t s = 0; s <= nblk; ++s) {
        const int64_t kcur = s == 0 ? -1 : (UPPER ? nblk - s : s - 1);        // block whose solution is known
        const int64_t knext = s == nblk ? -1 : (UPPER ? nblk - 1 - s : s);   // block to solve in this launch
        const long long kb = kcur < 0 ? -1 : kcur * TW, kn = knext < 0 ? -1 : knext * TW;
        const int w = kcur < 0 ? 0 : (int)((n - kb) < TW ? (n - kb) : TW);
        const int64_t rows = kcur < 0 ? 0 : (UPPER ? kb : n - kb - w);         // rows the known block still has to be taken out of
        const int nupd = (int)((rows + 63) / 64);
        const int nnear = knext < 0 ? 0 : (nupd < TS_NEAR ? nupd : TS_NEAR);
        const int grid = nupd + (knext >= 0 ? 4 : 0);
        if (grid == 0) continue;
        trsv_step_kernel<UPPER><<<grid, 256, 0, c->stream>>>(LU, ld, knext >= 0 ? inv + knext * TW * TW : inv, x, y, n, kb, w, kn, nnear, nupd,
                                                             cnt + s, spin, &c->ws->flags[0]);
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
static int trsv_lower_wide(mpf_ctx *c, const double *LU, int64_t ld, double *x, double *y, int64_t n) { return trsv_wide<false>(c, LU, ld, x, y, n); }
static int trsv_upper_wide(mpf_ctx *c, const double *LU, int64_t ld, double *x, double *y, int64_t n) { return trsv_wide<true>(c, LU, ld, x, y, n); }

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

// build the inverted diagonal blocks of the factors for the solves that follow (once per mpf_solve_ir): the 64 x 64 inverses, then
// from them the full 256 x 256 ones (three levels of 64-block products)
int launch_trsv_prepare(mpf_ctx *c, const double *LU, int64_t ld, int64_t n) {
    const int64_t nblk = (n + TS_B - 1) / TS_B;
    dim3 grid((unsigned)nblk, 2);
    trsv_invert_blocks_kernel<<<grid, 64, 0, c->stream>>>(LU, ld, n, c->trsv_inv, c->trsv_inv + nblk * TS_B * TS_B, 0, 0);
    const int64_t nb256 = (n + TW - 1) / TW;
    if (!c->trsv_inv256 || !c->trsv_cnt) { c->err = "solve: the 256 x 256 inverses are not allocated"; return -1; }
    MPF_HIP_TRY(c, hipMemsetAsync(c->trsv_inv256, 0, (size_t)(2 * nb256) * TW * TW * sizeof(double), c->stream));
    for (int d = 0; d < 4; ++d) {
        dim3 g2((unsigned)(4 - d), (unsigned)nb256, 2);
        trsv_inv256_level_kernel<<<g2, 256, 0, c->stream>>>(LU, ld, n, c->trsv_inv, c->trsv_inv + nblk * TS_B * TS_B, c->trsv_inv256,
                                                           c->trsv_inv256 + nb256 * TW * TW, d);
    }
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

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

// r -= A[:, c0:c0+RS_CCH] * x[c0:...]   (r pre-loaded with b)
__global__ __launch_bounds__(256) void residual_kernel(const double *__restrict__ A, long long lda,
                                                       const double *__restrict__ x, double *r, long long n) {
    __shared__ double xs[RS_CCH];
    const long long c0 = (long long)blockIdx.y * RS_CCH;
    const int nc = (int)((n - c0) < RS_CCH ? (n - c0) : RS_CCH);
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
    unsafeAtomicAdd(&r[row], -((s0 + s1) + (s2 + s3)));
}

// One block step of the forward solve  L y = x  (L unit lower, packed in LU).  Every workgroup
// re-solves the TS_B x TS_B diagonal block with wave 0 (lane = row, x_j broadcast by readlane),
// then subtracts L[rows, kb:kb+TS_B] * y_blk from its 256 rows below.  x[kb:kb+TS_B] is read-only in
// this step; the finished values go to y.
__global__ __launch_bounds__(256) void trsv_lower_step_kernel(const double *__restrict__ LU, long long ld, double *x,
                                                              double *y, long long n, long long kb) {
    __shared__ double ys[TS_B];
    __shared__ double Dt[TS_B][TS_B + 1]; // the diagonal block, all 4096 loads in flight at once
    const int tid = threadIdx.x;
    const int nb = (int)((n - kb) < TS_B ? (n - kb) : TS_B);
    for (int e = tid; e < TS_B * TS_B; e += 256) {
        const int i = e & 63, j = e >> 6;
        Dt[i][j] = (i < nb && j < nb && i > j) ? LU[(kb + i) + (kb + j) * ld] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
        double v = tid < nb ? x[kb + tid] : 0.0;
        for (int j = 0; j < nb; ++j) {
            const double vj = __shfl(v, j);
            v -= Dt[tid][j] * vj; // zero on and above the diagonal
        }
        ys[tid] = v;
        if (blockIdx.x == 0 && tid < nb) y[kb + tid] = v;
    }
    __syncthreads();
    const long long row = kb + TS_B + (long long)blockIdx.x * 256 + tid;
    if (row >= n) return;
    const double *l = LU + row + kb * ld;
    double s0 = 0, s1 = 0;
#pragma unroll 8
    for (int j = 0; j < TS_B; j += 2) {
        s0 += l[(long long)j * ld] * ys[j];
        s1 += l[(long long)(j + 1) * ld] * ys[j + 1];
    }
    x[row] -= s0 + s1;
}

// One block step of the backward solve  U y = x  (U upper incl. diagonal).  kb is the first row of
// the block being solved; rows above it get the update.
__global__ __launch_bounds__(256) void trsv_upper_step_kernel(const double *__restrict__ LU, long long ld, double *x,
                                                              double *y, long long n, long long kb) {
    __shared__ double ys[TS_B];
    __shared__ double Dt[TS_B][TS_B + 1];
    const int tid = threadIdx.x;
    const int nb = (int)((n - kb) < TS_B ? (n - kb) : TS_B);
    for (int e = tid; e < TS_B * TS_B; e += 256) {
        const int i = e & 63, j = e >> 6;
        Dt[i][j] = (i < nb && j < nb && i <= j) ? LU[(kb + i) + (kb + j) * ld] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < 64) {
        double v = tid < nb ? x[kb + tid] : 0.0;
        for (int j = nb - 1; j >= 0; --j) {
            if (tid == j) v = v / Dt[j][j];
            const double vj = __shfl(v, j);
            if (tid < j) v -= Dt[tid][j] * vj;
        }
        ys[tid] = tid < nb ? v : 0.0;
        if (blockIdx.x == 0 && tid < nb) y[kb + tid] = v;
    }
    __syncthreads();
    const long long row = (long long)blockIdx.x * 256 + tid;
    if (row >= kb) return;
    const double *u = LU + row + kb * ld;
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

__global__ __launch_bounds__(256) void sumsq_kernel(const double *x, long long n, double *out) {
    __shared__ double part[4];
    double s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += x[i] * x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

int launch_gather_rows(mpf_ctx *c, const double *in, const int *perm, double *out, int64_t n) {
    gather_rows_kernel<<<(int)((n + 255) / 256), 256, 0, c->stream>>>(in, perm, out, n);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
int launch_residual(mpf_ctx *c, const double *A, int64_t lda, const double *x, const double *b, double *r, int64_t n) {
    MPF_HIP_TRY(c, hipMemcpyAsync(r, b, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)((n + RS_CCH - 1) / RS_CCH));
    residual_kernel<<<grid, 256, 0, c->stream>>>(A, lda, x, r, n);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
// x is consumed (overwritten with intermediate values); the solution lands in y
static int trsv_lower(mpf_ctx *c, const double *LU, int64_t ld, double *x, double *y, int64_t n) {
    for (int64_t kb = 0; kb < n; kb += TS_B) {
        const int64_t below = n - kb - TS_B;
        const int blocks = below > 0 ? (int)((below + 255) / 256) : 1;
        trsv_lower_step_kernel<<<blocks, 256, 0, c->stream>>>(LU, ld, x, y, n, kb);
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
static int trsv_upper(mpf_ctx *c, const double *LU, int64_t ld, double *x, double *y, int64_t n) {
    const int64_t nblk = (n + TS_B - 1) / TS_B;
    for (int64_t b = nblk - 1; b >= 0; --b) {
        const int64_t kb = b * TS_B;
        const int blocks = kb > 0 ? (int)((kb + 255) / 256) : 1;
        trsv_upper_step_kernel<<<blocks, 256, 0, c->stream>>>(LU, ld, x, y, n, kb);
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
int launch_trsv_lower_unit(mpf_ctx *c, const double *LU, int64_t ld, double *x, int64_t n) {
    // in: x, out: x (via the context's scratch vector)
    double *y = c->solve_buf + 3 * c->solve_n;
    int rc = trsv_lower(c, LU, ld, x, y, n);
    if (rc) return rc;
    MPF_HIP_TRY(c, hipMemcpyAsync(x, y, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
int launch_trsv_upper(mpf_ctx *c, const double *LU, int64_t ld, double *x, int64_t n) {
    double *y = c->solve_buf + 3 * c->solve_n;
    int rc = trsv_upper(c, LU, ld, x, y, n);
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
    MPF_HIP_TRY(c, hipMemsetAsync(d_out, 0, sizeof(double), c->stream));
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    sumsq_kernel<<<blocks, 256, 0, c->stream>>>(x, n, d_out);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

// fp16-in / fp32-accumulate trailing update for gfx950 (BASELINE north_star "MFMA fp16 -> fp32 for the trailing
// GEMM"; the reference's update is the fp64 cublasDgemm at MPF.cu:230-239 -- this is the build's speed mode,
// mpf_opts.trailing = MPF_TRAIL_FP16, made usable by the fp64 refinement sweep).
//
//   A22 (fp64, in place) -= fp16(L21) * fp16(U12),   products exact in fp32, accumulated in fp32 over K = nb,
//                                                     subtracted from the fp64 matrix once per panel.
// The matrix, the panels and the TRSM stay fp64 (the fp64 panel dgetf2_native_npv is part of the hot path), so
// everything else of the fp64 path is reused and the factors land where the solve expects them.
//
// This kernel is HBM-bound by construction: per element of A22 it moves 16 B (fp64 read + write) for 2*nb
// flops = 32 flop/B at nb = 256 against a ridge of ~300 flop/B.  So it is written as a STREAMING kernel:
//   * operands are pre-converted once per panel into MFMA-friendly fp16 images (cvt kernels below): U12 as
//     [n][Kp] and L21 as [m][Kp] (k contiguous), so a lane's 8-element fragment of v_mfma_f32_32x32x16_f16 is ONE
//     16-byte global load -- no LDS staging at all (the images are 16 MB each and L2/Infinity-Cache resident);
//   * the MFMA's row index is mapped to A22's column and its column index to A22's row, so each wave-level fp64
//     load/store of the accumulator tile is two 256-byte runs of the column-major matrix;
//   * a wave owns a 64 x 64 block (2 x 2 MFMA tiles); the MFMA phase comes first, then the fp64 block is streamed
//     through registers 32 elements per lane at a time.  ~130 VGPRs => several workgroups per CU, so some are
//     always in their HBM phase while others run operand loads and MFMAs.
#include "mpf_internal.h"

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f16_t __attribute__((ext_vector_type(16)));

// fp64 -> fp16 for the GEMM operands: round to nearest (through fp32), saturate at +-65504 like
// fp16_utils.h:19-20, but KEEP subnormals (no flush): multipliers of a diagonally dominant matrix sit below
// 2^-14 and flushing them would switch the update off.
__device__ __forceinline__ unsigned short d2h_sat(double x) {
    float xf = (float)x;
    const float FP16_MAX = 65504.0f;
    if (xf > FP16_MAX) xf = FP16_MAX;
    else if (xf < -FP16_MAX) xf = -FP16_MAX;
    return __builtin_bit_cast(unsigned short, (_Float16)xf);
}

// U12 (K x n fp64, column-major) -> Uh[n][Kp] fp16, rows K..Kp-1 zero
__global__ __launch_bounds__(256) void cvt_u12_kernel(const double *__restrict__ U, long long ldu, int K, int Kp, long long n,
                                                      unsigned short *__restrict__ Uh) {
    const int k = threadIdx.x;
    for (long long c = blockIdx.x; c < n; c += gridDim.x)
        if (k < Kp) Uh[c * Kp + k] = k < K ? d2h_sat(U[k + c * ldu]) : (unsigned short)0;
}

// L21 (m x K fp64, column-major) -> Lh[m][Kp] fp16 (row-major: k contiguous), via a 64 x 64 LDS transpose
__global__ __launch_bounds__(256) void cvt_l21_kernel(const double *__restrict__ Lm, long long ldl, long long m, int K, int Kp,
                                                      unsigned short *__restrict__ Lh) {
    __shared__ unsigned short t[64][66];
    const long long r0 = (long long)blockIdx.x * 64;
    const int k0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // ty 0..3
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int kk = ty + 4 * i;                    // column of the tile
        const long long r = r0 + tx;                  // row: consecutive lanes -> consecutive rows (coalesced)
        t[kk][tx] = (r < m && k0 + kk < K) ? d2h_sat(Lm[r + (long long)(k0 + kk) * ldl]) : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int rr = ty + 4 * i;                    // row of the tile
        const long long r = r0 + rr;
        if (r < m && k0 + tx < Kp) Lh[r * Kp + k0 + tx] = t[tx][rr]; // consecutive lanes -> consecutive k
    }
}

__global__ __launch_bounds__(256, 3) void hgemm_minus_kernel(long long m, long long n, int Kp, const unsigned short *__restrict__ Lh,
                                                             const unsigned short *__restrict__ Uh, double *__restrict__ C,
                                                             long long ldc, int tiles_m, int tiles_n) {
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rr_ = nwg & 7;
    const int lin = (xcd < rr_ ? xcd * (q + 1) : rr_ * (q + 1) + (xcd - rr_) * q) + (bid >> 3);
    // tiles are walked in groups of 8 tile-columns, tile-column fastest: the ~64 workgroups an XCD runs at a time
    // form an 8 x 8 block of tiles that shares 8 A and 8 B operand tiles (4 MB = one XCD's L2)
    const int grp = lin / (tiles_m * 8);
    const int gw = (tiles_n - grp * 8) < 8 ? (tiles_n - grp * 8) : 8;
    const int idx = lin - grp * tiles_m * 8;
    const int tm = idx / gw, tn = grp * 8 + idx % gw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long m0 = (long long)tm * 128 + (wave & 1) * 64, n0 = (long long)tn * 128 + (wave >> 1) * 64;
    const int r = lane & 31, h = lane >> 5;

    f16_t acc[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[nt][mt][g] = 0.f;

    const h8_t zero8 = (h8_t){0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned short *up[2], *lp[2];
    bool uok[2], lok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long long col = n0 + t * 32 + r, row = m0 + t * 32 + r;
        uok[t] = col < n; lok[t] = row < m;
        up[t] = Uh + (uok[t] ? col : 0) * Kp + 8 * h;
        lp[t] = Lh + (lok[t] ? row : 0) * Kp + 8 * h;
    }
#pragma unroll 4
    for (int k0 = 0; k0 < Kp; k0 += 16) {
        h8_t a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            a[t] = uok[t] ? *(const h8_t *)(up[t] + k0) : zero8; // A'[n][k] = U[k][n]
            b[t] = lok[t] ? *(const h8_t *)(lp[t] + k0) : zero8; // B'[k][m] = L[m][k]
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[nt], b[mt], acc[nt][mt], 0, 0, 0);
    }
    // ---- epilogue: stream the fp64 block through registers, 32 elements (two MFMA tiles) in flight per lane;
    //      the other workgroups on the CU (<= 128 VGPRs each) are in their operand / MFMA phase meanwhile --------
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        double cv[2][16];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const long long row = m0 + mt * 32 + r;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const long long col = n0 + nt * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                cv[mt][g] = (row < m && col < n) ? C[row + col * ldc] : 0.0;
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const long long row = m0 + mt * 32 + r;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const long long col = n0 + nt * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                if (row < m && col < n) C[row + col * ldc] = cv[mt][g] - (double)acc[nt][mt][g];
            }
        }
    }
}

// C[m x n] -= fp16(A[m x K]) * fp16(B[K x n]); A = L21, B = U12 (fp64, column-major).  Lh/Uh are scratch images.
int launch_cvt_l21(mpf_ctx *c, const double *A, int64_t lda, int64_t m, int K) {
    const int Kp = (K + 15) & ~15;
    dim3 grid((unsigned)((m + 63) / 64), (unsigned)((Kp + 63) / 64));
    cvt_l21_kernel<<<grid, 256, 0, c->stream>>>(A, lda, m, K, Kp, c->h_L);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
int launch_hgemm_minus(mpf_ctx *c, int64_t m, int64_t n, int K, const double *B, int64_t ldb, double *C, int64_t ldc,
                       int64_t u_col0) {
    if (m <= 0 || n <= 0 || K <= 0) return 0;
    const int Kp = (K + 15) & ~15;
    unsigned short *Uh = c->h_U + u_col0 * Kp;
    long long cb = n < 4096 ? n : 4096;
    cvt_u12_kernel<<<(int)cb, 256, 0, c->stream>>>(B, ldb, K, Kp, n, Uh);
    const long long tm = (m + 127) / 128, tn = (n + 127) / 128;
    hgemm_minus_kernel<<<(int)(tm * tn), 256, 0, c->stream>>>(m, n, Kp, c->h_L, Uh, C, ldc, (int)tm, (int)tn);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

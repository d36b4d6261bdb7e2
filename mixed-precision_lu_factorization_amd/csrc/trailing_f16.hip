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
#include <cstdlib>

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

// Split mode ("fp16x3"): a = hi + 2^-11 * lo with hi = fp16(a), lo = fp16((a - hi) * 2^11).  The three products
// hi*hi + 2^-11 (hi*lo + lo*hi) carry ~22 significant bits -- fp32-class accuracy from fp16 MFMAs.
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
constexpr double SPLIT_SCALE = 2048.0;
#ifndef MPF_C_AUX
#define MPF_C_AUX 2
#endif
constexpr int C_AUX = MPF_C_AUX; // cache policy of the streamed fp64 block (2 = nt: keep the operand images in L2 instead)
__device__ __forceinline__ unsigned short d2h_lo(double x, unsigned short hi_bits) {
    const double hi = (double)(float)__builtin_bit_cast(_Float16, hi_bits);
    double r = (x - hi) * SPLIT_SCALE;
    if (!(r == r) || r > 65504.0 || r < -65504.0) r = 0.0; // saturated / non-finite hi: no correction term
    return __builtin_bit_cast(unsigned short, (_Float16)(float)r);
}

// U12 (K x n fp64, column-major) -> Uh[n][Kp] fp16 (and Ul if given), rows K..Kp-1 zero
__global__ __launch_bounds__(256) void cvt_u12_kernel(const double *__restrict__ U, long long ldu, int K, int Kp, long long n,
                                                      unsigned short *__restrict__ Uh, unsigned short *__restrict__ Ul) {
    for (long long c = blockIdx.x; c < n; c += gridDim.x)
        for (int k = threadIdx.x; k < Kp; k += 256) {
            const double x = k < K ? U[k + c * ldu] : 0.0;
            const unsigned short hi = k < K ? d2h_sat(x) : (unsigned short)0;
            Uh[c * Kp + k] = hi;
            if (Ul) Ul[c * Kp + k] = k < K ? d2h_lo(x, hi) : (unsigned short)0;
        }
}

// L21 (m x K fp64, column-major) -> Lh[m][Kp] fp16 (row-major: k contiguous), via a 64 x 64 LDS transpose
__global__ __launch_bounds__(256) void cvt_l21_kernel(const double *__restrict__ Lm, long long ldl, long long m, int K, int Kp,
                                                      unsigned short *__restrict__ Lh, unsigned short *__restrict__ Ll) {
    __shared__ unsigned short t[64][66];
    __shared__ unsigned short tl[64][66];
    const long long r0 = (long long)blockIdx.x * 64;
    const int k0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // ty 0..3
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int kk = ty + 4 * i;                    // column of the tile
        const long long r = r0 + tx;                  // row: consecutive lanes -> consecutive rows (coalesced)
        const bool in = r < m && k0 + kk < K;
        const double x = in ? Lm[r + (long long)(k0 + kk) * ldl] : 0.0;
        const unsigned short hi = in ? d2h_sat(x) : (unsigned short)0;
        t[kk][tx] = hi;
        if (Ll) tl[kk][tx] = in ? d2h_lo(x, hi) : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int rr = ty + 4 * i;                    // row of the tile
        const long long r = r0 + rr;
        if (r < m && k0 + tx < Kp) {
            Lh[r * Kp + k0 + tx] = t[tx][rr]; // consecutive lanes -> consecutive k
            if (Ll) Ll[r * Kp + k0 + tx] = tl[tx][rr];
        }
    }
}

// ---- K loop through an LDS ring ------------------------------------------------------------------------------------
// K loop of the fp16 trailing update.  The workgroup's operand rows go global -> LDS directly (global_load_lds_dwordx4, no
// VGPR staging) into a ring of three stages; two stages are in flight while one is consumed, each operand byte is fetched
// once per workgroup, and all four waves read their MFMA fragments from LDS (ds_read_b128, conflict-free layout).
// Stage = 32 k (plain) or 16 k (split: hi and lo images), 16 KB; 48 KB per workgroup, three (plain) / two (split)
// workgroups per CU.  Waves 0-1 fetch the U-side rows (the tile's 128 columns), waves 2-3 the L-side rows.  Per output
// element: one fp32 MFMA accumulation chain over k ascending (contract C6), one fp64 subtraction at the end.
// (Round 1's register-fed variant -- every wave fetching its own fragments from global memory -- was latency-bound from
// K = 512 on and is gone.)
// C32 = true: the updated block is the fp32 working copy of the trailing matrix (8 instead of 16 bytes of HBM per element)
template <bool SPLIT, bool C32 = false>
__global__ __launch_bounds__(256, 3) void hgemm_ring_kernel(long long m, long long n, int Kp, const unsigned short *__restrict__ Lh,
                                                                        const unsigned short *__restrict__ Uh, const unsigned short *__restrict__ Ll,
                                                                        const unsigned short *__restrict__ Ul, void *__restrict__ Cv,
                                                                        long long ldc, int tiles_m, int tiles_n) {
    constexpr int KS = SPLIT ? 1 : 2;          // k-steps (of 16) per stage
    constexpr int RB = 32 * KS;                // bytes one operand row contributes to a stage
    constexpr int CPR = RB / 16;               // 16-byte chunks per row
    constexpr int NS = 3;                      // ring depth
    constexpr int NIMG = SPLIT ? 2 : 1;        // images per side (hi [, lo])
    constexpr int ARR = 128 * RB;              // bytes of one image's 128 rows in a stage
    constexpr int STAGE = 2 * NIMG * ARR;      // U-side images, then L-side images: 16 KB
    constexpr int RPI = 64 / CPR;              // rows one 64-lane issue covers
    constexpr int IPW = 64 / RPI;              // issues per image per wave (the wave owns 64 rows)
    constexpr int LPS = NIMG * IPW;            // loads per wave per stage (4)
    __shared__ __attribute__((aligned(16))) unsigned char ring[NS * STAGE];

    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rr_ = nwg & 7;
    const int lin = (xcd < rr_ ? xcd * (q + 1) : rr_ * (q + 1) + (xcd - rr_) * q) + (bid >> 3);
    const int grp = lin / (tiles_m * 8);
    const int gw = (tiles_n - grp * 8) < 8 ? (tiles_n - grp * 8) : 8;
    const int idx = lin - grp * tiles_m * 8;
    const int tm = idx / gw, tn = grp * 8 + idx % gw;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0t = (long long)tm * 128, n0t = (long long)tn * 128;
    const int r = lane & 31, h = lane >> 5;

    // ---- loader role ---------------------------------------------------------------------------------------------
    const bool lside = wave >= 2;
    const int half = wave & 1;                 // rows half*64 .. half*64+63 of the side's 128
    const int lr = lane / CPR, pc = lane % CPR; // row within an issue, physical chunk
    const unsigned short *img[NIMG];
    img[0] = lside ? Lh : Uh;
    if (SPLIT) img[NIMG - 1] = lside ? Ll : Ul;
    long long goff[IPW];                       // element offset of this lane's chunk at k0 = 0
#pragma unroll
    for (int ii = 0; ii < IPW; ++ii) {
        const int trow = half * 64 + ii * RPI + lr;
        const long long grow = (lside ? m0t : n0t) + trow, lim = lside ? m : n;
        const int c = SPLIT ? pc : (pc ^ ((trow >> 1) & 3));   // logical chunk stored at physical position pc
        goff[ii] = (grow < lim ? grow : 0) * Kp + c * 8;
    }
    const int side_off = lside ? NIMG * ARR : 0;
    auto issue = [&](int s) {
        unsigned char *st = ring + (s % NS) * STAGE + side_off;
        const int k0 = s * 16 * KS;
#pragma unroll
        for (int im = 0; im < NIMG; ++im)
#pragma unroll
            for (int ii = 0; ii < IPW; ++ii)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(img[im] + goff[ii] + k0),
                                                 (__attribute__((address_space(3))) void *)(st + im * ARR + (half * 64 + ii * RPI) * RB), 16, 0, 0);
    };

    // ---- consumer role: wave (wr, wc) owns the 64 x 64 block at (m0t + 64 wr, n0t + 64 wc) ------------------------------
    const int wr = wave & 1, wc = wave >> 1;
    f16_t acc[2][2], accx[SPLIT ? 2 : 1][SPLIT ? 2 : 1];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 16; ++g) { acc[nt][mt][g] = 0.f; if (SPLIT) accx[nt][mt][g] = 0.f; }
    int uo[2], lo_[2]; // byte offsets of this lane's rows inside an image's stage block
#pragma unroll
    for (int t = 0; t < 2; ++t) { uo[t] = (wc * 64 + t * 32 + r) * RB; lo_[t] = (wr * 64 + t * 32 + r) * RB; }
    auto frag = [&](const unsigned char *blk, int rowoff, int trow, int ks) -> h8_t {
        const int c = ks * 2 + h;
        const int p = SPLIT ? c : (c ^ ((trow >> 1) & 3));
        return *(const h8_t *)(blk + rowoff + p * 16);
    };
    const int nst = Kp / (16 * KS);
    for (int s = 0; s < NS - 1 && s < nst; ++s) issue(s);
    for (int i = 0; i < nst; ++i) {
        // stage i has landed once at most the loads of the later stages already issued are outstanding
        if (i + NS - 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // everyone's part of stage i is in LDS; everyone is done with stage i-1
        if (i + NS - 1 < nst) issue(i + NS - 1);
        const unsigned char *st = ring + (i % NS) * STAGE;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            h8_t a[2], b[2], al[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = frag(st, uo[t], wc * 64 + t * 32 + r, ks);
                b[t] = frag(st + NIMG * ARR, lo_[t], wr * 64 + t * 32 + r, ks);
                if (SPLIT) {
                    al[t] = frag(st + ARR, uo[t], wc * 64 + t * 32 + r, ks);
                    bl[t] = frag(st + NIMG * ARR + ARR, lo_[t], wr * 64 + t * 32 + r, ks);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[nt], b[mt], acc[nt][mt], 0, 0, 0);
                    if (SPLIT) {
                        accx[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[nt], bl[mt], accx[nt][mt], 0, 0, 0);
                        accx[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[nt], b[mt], accx[nt][mt], 0, 0, 0);
                    }
                }
        }
    }
    // ---- epilogue: as in hgemm_wave_pass ---------------------------------------------------------------------------
    const long long m0 = m0t + wr * 64, n0 = n0t + wc * 64;
    const long long mrem = m - m0, nrem = n - n0;
    if (mrem <= 0 || nrem <= 0) return; // wave-uniform, after the last barrier
    const long long ncl = nrem < 64 ? nrem : 64, mcl = mrem < 64 ? mrem : 64;
    constexpr unsigned ES = C32 ? 4u : 8u;    // bytes per element of the updated block
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((char *)Cv + (m0 + n0 * ldc) * (long long)ES), 0, (int)(((ncl - 1) * ldc + mcl) * ES), 0x00020000);
    const unsigned ldc8 = (unsigned)ldc * ES;
    unsigned voff[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) voff[mt] = (mt * 32 + r < mrem) ? (unsigned)(mt * 32 + r) * ES + (unsigned)(4 * h) * ldc8 : 0x80000000u;
    // the block through registers: two MFMA tiles (32 elements per lane) per batch in the plain kernel; one tile per batch
    // in the split kernel, whose second accumulator set would otherwise cost the third workgroup per CU
    constexpr int EBR = (SPLIT ? 1 : 2) * (C32 ? 2 : 1);   // (4-byte elements: twice the tiles in the same registers)
#pragma unroll
    for (int bt = 0; bt < 4 / EBR; ++bt) {
        double cv[C32 ? 1 : EBR][16];
        float cf[C32 ? EBR : 1][16];
#pragma unroll
        for (int e = 0; e < EBR; ++e) {
            const int tix = bt * EBR + e, nt = tix >> 1, mt = tix & 1;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const unsigned soff = (unsigned)(nt * 32 + (g & 3) + 8 * (g >> 2)) * ldc8;
                if (C32) cf[C32 ? e : 0][g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (int)voff[mt], (int)soff, C_AUX));
                else cv[C32 ? 0 : e][g] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rc, (int)voff[mt], (int)soff, C_AUX));
            }
        }
#pragma unroll
        for (int e = 0; e < EBR; ++e) {
            const int tix = bt * EBR + e, nt = tix >> 1, mt = tix & 1;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const unsigned soff = (unsigned)(nt * 32 + (g & 3) + 8 * (g >> 2)) * ldc8;
                if (C32) {
                    float pf = acc[nt][mt][g];
                    if (SPLIT) pf += accx[nt][mt][g] * (float)(1.0 / SPLIT_SCALE);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cf[C32 ? e : 0][g] - pf), rc, (int)voff[mt], (int)soff, C_AUX);
                } else {
                    double p = (double)acc[nt][mt][g];
                    if (SPLIT) p += (double)accx[nt][mt][g] * (1.0 / SPLIT_SCALE);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, cv[C32 ? 0 : e][g] - p), rc, (int)voff[mt], (int)soff, C_AUX);
                }
            }
        }
    }
}

// C[m x n] -= fp16(A[m x K]) * fp16(B[K x n]); A = L21, B = U12 (fp64, column-major).  Lh/Uh are scratch images.
static unsigned short *l_image(mpf_ctx *c, int img) { return img == 0 ? c->h_L : c->h_Lb[img - 1]; }

int launch_cvt_l21(mpf_ctx *c, const double *A, int64_t lda, int64_t m, int K, int split, int img, int64_t elem_off) {
    const int Kp = (K + 63) & ~63;
    dim3 grid((unsigned)((m + 63) / 64), (unsigned)((Kp + 63) / 64));
    unsigned short *Lh = l_image(c, img);
    if (!Lh) { c->err = "fp16 operand image not allocated"; return -1; }
    Lh += elem_off;
    cvt_l21_kernel<<<grid, 256, 0, c->stream>>>(A, lda, m, K, Kp, Lh, split ? Lh + c->h_rows * c->h_kmax : nullptr);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
static int hgemm_minus_any(mpf_ctx *c, int64_t m, int64_t n, int K, const double *B, int64_t ldb, void *C, int64_t ldc, bool c32,
                           int split, int img, int64_t elem_off) {
    if (m <= 0 || n <= 0 || K <= 0) return 0;
    // the kernel addresses a wave's 64 x 64 block of C with 32-bit byte offsets from the block's base (63 * ldc * 8 < 2^31)
    if (ldc > (1ll << 21)) { c->err = "hgemm: leading dimension too large for 32-bit tile offsets"; return -1; }
    if (((m + 127) / 128) * ((n + 127) / 128) > 0x7FFFFFFFll) { c->err = "hgemm: too many tiles"; return -1; }
    const int Kp = (K + 63) & ~63;
    unsigned short *Lh = l_image(c, img);
    if (!Lh) { c->err = "fp16 operand image not allocated"; return -1; }
    Lh += elem_off;
    unsigned short *Uh = c->h_U, *Ul = c->h_U + c->h_rows * c->h_kmax, *Ll = Lh + c->h_rows * c->h_kmax;
    long long cb = n < 4096 ? n : 4096;
    cvt_u12_kernel<<<(int)cb, 256, 0, c->stream>>>(B, ldb, K, Kp, n, Uh, split ? Ul : nullptr);
    const long long tm = (m + 127) / 128, tn = (n + 127) / 128;
    const int g = (int)(tm * tn);
    // The split-operand kernel needs 168 VGPRs: three workgroups per CU leave 8 registers per SIMD lane, and EVERY launch of
    // the panel chain (pivots 56, interchanges 80, fp64 panel 176-184) then waits for one of its long tiles to retire (fp64
    // panel 139 ms per factorization instead of 68).  32 KB of unused dynamic LDS cap it at two workgroups per CU: the update
    // itself loses 8 % (149 -> 162 ms), the factorization gains 3 % (318 -> 307 ms at N = 32768).  The plain kernel (136
    // VGPRs) leaves room for everything but the fp64 panel; capping it too measured slower (254 vs 249 ms).
    // Options hgemm_pad / hgemm_split_pad (bytes; defaults from MPF_HGEMM_PAD / MPF_HGEMM_SPLIT_PAD) override.
    const int pad_plain = c->tune.hgemm_pad, pad_split = c->tune.hgemm_split_pad;
    if (!(c->attr_done & ATTR_HGEMM)) {
        if (pad_plain > 0) {
            MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)hgemm_ring_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, pad_plain));
            MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)hgemm_ring_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, pad_plain));
        }
        if (pad_split > 0) {
            MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)hgemm_ring_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, pad_split));
            MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)hgemm_ring_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, pad_split));
        }
        c->attr_done |= ATTR_HGEMM;
    }
    if (split) {
        if (c32) hgemm_ring_kernel<true, true><<<g, 256, pad_split, c->stream>>>(m, n, Kp, Lh, Uh, Ll, Ul, C, ldc, (int)tm, (int)tn);
        else hgemm_ring_kernel<true, false><<<g, 256, pad_split, c->stream>>>(m, n, Kp, Lh, Uh, Ll, Ul, C, ldc, (int)tm, (int)tn);
    } else {
        if (c32) hgemm_ring_kernel<false, true><<<g, 256, pad_plain, c->stream>>>(m, n, Kp, Lh, Uh, nullptr, nullptr, C, ldc, (int)tm, (int)tn);
        else hgemm_ring_kernel<false, false><<<g, 256, pad_plain, c->stream>>>(m, n, Kp, Lh, Uh, nullptr, nullptr, C, ldc, (int)tm, (int)tn);
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
int launch_hgemm_minus(mpf_ctx *c, int64_t m, int64_t n, int K, const double *B, int64_t ldb, double *C, int64_t ldc,
                       int split, int img, int64_t elem_off) {
    return hgemm_minus_any(c, m, n, K, B, ldb, C, ldc, false, split, img, elem_off);
}
// the same update on the fp32 working copy of the trailing matrix (two-level schedule of the fp16 modes)
int launch_hgemm_minus_w32(mpf_ctx *c, int64_t m, int64_t n, int K, const double *B, int64_t ldb, float *C, int64_t ldc,
                           int split, int img, int64_t elem_off) {
    return hgemm_minus_any(c, m, n, K, B, ldb, C, ldc, true, split, img, elem_off);
}

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
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
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
// ks = row stride of the image in elements (>= Kp): a block of K rows can be written into a wider image at column offset
__global__ __launch_bounds__(256) void cvt_u12_kernel(const double *__restrict__ U, long long ldu, int K, int Kp, long long n,
                                                      unsigned short *__restrict__ Uh, unsigned short *__restrict__ Ul, long long ks) {
    for (long long c = blockIdx.x; c < n; c += gridDim.x)
        for (int k = threadIdx.x; k < Kp; k += 256) {
            const double x = k < K ? U[k + c * ldu] : 0.0;
            const unsigned short hi = k < K ? d2h_sat(x) : (unsigned short)0;
            Uh[c * ks + k] = hi;
            if (Ul) Ul[c * ks + k] = k < K ? d2h_lo(x, hi) : (unsigned short)0;
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
                                                                        long long ldc, int tiles_m, int tiles_n, int ksL, int ksU) {
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
        goff[ii] = (grow < lim ? grow : 0) * (lside ? ksL : ksU) + c * 8;   // image row strides (elements) >= Kp
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

// ---- big-K update: 256-row tiles, eight waves ---------------------------------------------------------------------------
// The K = sb * nb updates of the two-level schedule (m, n in the thousands, K = 512 .. 2048) carry > 95 % of the fp16 modes'
// flops.  hgemm_ring_kernel's 128 x 128 tile spends more cycles ISSUING its operand DMA (one 1-KB global_load_lds piece per
// 2 MFMAs per wave, ~100 cycles each beside MFMAs) than computing.  Here a workgroup of eight waves owns a
// (WM * MT * 32) x (WN * NT * 32) tile -- plain: 256 x 256, wave = 128 x 64 (8 accumulator tiles = 128 VGPRs, two waves per
// SIMD); split operands: 256 x 128, wave = 64 x 64 (4 + 4 accumulator tiles) -- so one DMA piece feeds 4 MFMAs (plain), a
// stage of 32 k is 16 MFMAs per wave between two barriers, and fragment reads are 0.75 ds_read_b128 per MFMA (the LDS array
// sustains 2).  Ring of NS stages filled by global_load_lds_dwordx4 exactly like the 128-tile kernel (same XOR-swizzled
// 64-byte rows, same fragment addressing); one workgroup per CU (96-128 KB of LDS), its C tile streamed through registers
// once at the end.  Per output element: ONE fp32 MFMA accumulation chain over k ascending (contract C6), one subtraction.
// Chunk swizzle of the big-tile kernel's LDS rows.  A ds_read_b128 is serviced in four groups of 16 lanes -- {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- each of which must hit 64 different banks.
// Fragment reads: lane = tile row (mod 32), chunk from the lane's half.  64-byte rows (plain): bank = 16 row + 4 chunk; the
// row quads of a group are quads {0, 3, 5, 6} or {1, 2, 4, 7}: chunk ^= f(quad), f(q) = (q ^ (q >> 1)) & 3 gives each quad of
// a group its own chunk.  32-byte rows (split): bank = 8 row + 4 chunk: chunk ^= (row >> 3) & 1.
template <bool SPLIT> __device__ __forceinline__ int big_swz(int trow) {
    if (SPLIT) return (trow >> 3) & 1;
    const int qd = (trow >> 2) & 7;
    return (qd ^ (qd >> 1)) & 3;
}

// EPI: 0 = the C tile one batch (MFMA tile-row) at a time, load -> subtract -> store (round 3); 1 = batches pipelined (up to
// three batches of loads in flight, subtraction in place in the accumulators); 2 = 1 + the first batch requested three stages
// before the K loop ends (fp32 copy, plain operands: its 32 registers fit beside the K loop's).
template <bool SPLIT, bool C32, int MT, int NT, int WM, int WN, int NS, bool REG, int WPE = 1, int EPI = 0>
__global__ __launch_bounds__(512, WPE) void hgemm_big_kernel(long long m, long long n, int Kp, const unsigned short *__restrict__ Lh,
                                                           const unsigned short *__restrict__ Uh, const unsigned short *__restrict__ Ll,
                                                           const unsigned short *__restrict__ Ul, void *__restrict__ Cv,
                                                           long long ldc, int tiles_m, int tiles_n, int ksL, int ksU, int dbg_) {
    static_assert(WM * WN == 8, "eight waves");
#ifdef MPF_PROBE   // probe library: 1 = K loop only (no C load / store), 2 = C stream only (no K loop); K loop only and
                   // 3 = no operand DMA (stale LDS), 4 = no fragment reads after the first, 5 = neither
    const int dbg = dbg_ >= 3 ? 1 : dbg_;
    const bool no_dma = dbg_ == 3 || dbg_ == 5, no_frag = dbg_ == 4 || dbg_ == 5;
#else
    constexpr int dbg = 0;
    constexpr bool no_dma = false, no_frag = false;
#endif
    constexpr int TM = WM * MT * 32, TN = WN * NT * 32;   // workgroup tile (rows of L / rows of U)
    constexpr int KS = SPLIT ? 1 : 2;          // k-steps (of 16) per stage
    constexpr int RB = 32 * KS;                // bytes one operand row contributes to a stage
    constexpr int CPR = RB / 16;               // 16-byte chunks per row
    constexpr int NIMG = SPLIT ? 2 : 1;        // images per side (hi [, lo])
    constexpr int UARR = TN * RB, LARR = TM * RB;          // bytes of one image's rows in a stage
    constexpr int STAGE = NIMG * (UARR + LARR);
    constexpr int RPI = 64 / CPR;              // rows one 64-lane DMA piece covers
    constexpr int UPIECES = TN / RPI, LPIECES = TM / RPI;  // pieces per image and stage
    constexpr int PIECES = NIMG * (UPIECES + LPIECES);
    static_assert(PIECES % 8 == 0, "pieces are dealt to the eight waves evenly");
    constexpr int LPS = PIECES / 8;            // DMA pieces per wave per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];

    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rr_ = nwg & 7;
    const int lin = (xcd < rr_ ? xcd * (q + 1) : rr_ * (q + 1) + (xcd - rr_) * q) + (bid >> 3);
    constexpr int GW = 4;                      // tile-columns walked together: an XCD's run of tiles shares few operand rows
    const int grp = lin / (tiles_m * GW);
    const int gw = (tiles_n - grp * GW) < GW ? (tiles_n - grp * GW) : GW;
    const int idx = lin - grp * tiles_m * GW;
    const int tm = idx / gw, tn = grp * GW + idx % gw;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0t = (long long)tm * TM, n0t = (long long)tn * TN;
    const int r = lane & 31, h = lane >> 5;

    // ---- loader role: piece pi of a stage = (image, side, 16 / 32 rows); wave w takes pieces w, w + 8, ... ------------------
    const int lr = lane / CPR, pc = lane % CPR; // row within a piece, physical chunk
    const unsigned short *gsrc[LPS];           // this lane's global source at k0 = 0, per piece
    int ldst[LPS];                             // LDS byte offset of the piece inside a stage
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const int pi = wave + 8 * i;
        const int im = pi / (UPIECES + LPIECES), pj = pi % (UPIECES + LPIECES);
        const bool lside = pj >= UPIECES;
        const int prow = (lside ? pj - UPIECES : pj) * RPI;
        const int trow = prow + lr;
        const long long grow = (lside ? m0t : n0t) + trow, lim = lside ? m : n;
        const int c = pc ^ big_swz<SPLIT>(trow);               // logical chunk stored at physical position pc
        const unsigned short *base = lside ? (im ? Ll : Lh) : (im ? Ul : Uh);
        gsrc[i] = base + (grow < lim ? grow : 0) * (long long)(lside ? ksL : ksU) + c * 8;   // image row strides (elements) >= Kp
        ldst[i] = (lside ? NIMG * UARR + im * LARR : im * UARR) + prow * RB;
    }
    // ---- consumer role: wave (wr, wc) owns the (MT * 32) x (NT * 32) block at (m0t + MT * 32 * wr, n0t + NT * 32 * wc) -----
    const int wr = wave % WM, wc = wave / WM;
    f16_t acc[NT][MT], accx[SPLIT ? NT : 1][SPLIT ? MT : 1];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 16; ++g) { acc[nt][mt][g] = 0.f; if (SPLIT) accx[nt][mt][g] = 0.f; }
    int uo[NT], lo_[MT], usw[NT], lsw[MT];     // byte offsets of this lane's rows inside an image's stage block, row swizzles
#pragma unroll
    for (int t = 0; t < NT; ++t) { const int tr = wc * NT * 32 + t * 32 + r; uo[t] = tr * RB; usw[t] = big_swz<SPLIT>(tr); }
#pragma unroll
    for (int t = 0; t < MT; ++t) { const int tr = wr * MT * 32 + t * 32 + r; lo_[t] = NIMG * UARR + tr * RB; lsw[t] = big_swz<SPLIT>(tr); }
    auto frag = [&](const unsigned char *blk, int rowoff, int sw, int ks) -> h8_t {
        const int c = ks * 2 + h;
        return *(const h8_t *)(blk + rowoff + (c ^ sw) * 16);
    };
    // Software-pipelined K loop.  Units: k-steps of 16 (one MFMA per accumulator tile); a stage holds KS of them.
    //   * the barrier at the top of stage i makes stage i + 1 visible to everyone and says everyone has finished reading stage
    //     i - 1, whose ring slot is then refilled;
    //   * the fragments of k-step j + 1 are read from LDS while the MFMAs of k-step j run (two fragment sets in registers), so
    //     no MFMA ever waits for an LDS read issued just in front of it -- the first MFMA after a barrier has had its operands
    //     since the previous stage;
    //   * the operand pieces of later stages are moved one at a time BETWEEN groups of MFMAs (the two waves of a SIMD run the
    //     same code in step: spread out, one wave's piece overlaps the other's MFMAs).
    // Two loaders.  REG = false (the product's): global -> LDS directly (global_load_lds_dwordx4), ring of NS = 4 stages, stage
    // i + 3 issued during stage i.  REG = true (probe library only): global -> registers (global_load_dwordx4) -> LDS
    // (ds_write_b128), two stages in registers (stage i + 4 requested during stage i, stage i + 2 written during stage i), ring
    // of NS = 3 stages (96 KB of LDS instead of 128, + 8 LPS registers).  Built because the wave-level counters show the waves
    // stalled at issue for 59 % of their cycles at an MFMA busy of 0.41 and a DMA piece is known to hold its wave for 60-100
    // cycles; measured 702 against 726 TFLOP/s (m = n = 28672, K = 1024, same box): the DMA pieces are not what stalls them.
    const int nst = dbg == 2 ? 0 : Kp / (16 * KS);
    struct Frags { h8_t a[NT], b[MT], al[SPLIT ? NT : 1], bl[SPLIT ? MT : 1]; };
    Frags F0, F1;
    auto load_step = [&](int st_i, int ks, Frags &F) {
        if (no_frag && (st_i | ks) != 0) return;
        const unsigned char *st = ring + (st_i % NS) * STAGE;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            F.a[t] = frag(st, uo[t], usw[t], ks);
            if (SPLIT) F.al[t] = frag(st + UARR, uo[t], usw[t], ks);
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            F.b[t] = frag(st, lo_[t], lsw[t], ks);
            if (SPLIT) F.bl[t] = frag(st + LARR, lo_[t], lsw[t], ks);
        }
    };
    u4_t R0[REG ? LPS : 1], R1[REG ? LPS : 1];   // REG: the two stages in flight through registers
    auto gload_piece = [&](u4_t (&Rg)[REG ? LPS : 1], int s, int i) {
        Rg[REG ? i : 0] = *(const u4_t *)(gsrc[i] + s * 16 * KS);
    };
    auto lwrite_piece = [&](const u4_t (&Rg)[REG ? LPS : 1], int s, int i) {
        *(u4_t *)(ring + (s % NS) * STAGE + ldst[i] + lane * 16) = Rg[REG ? i : 0];
    };
    auto dma_piece = [&](int s, int i) {   // piece i (of LPS) of stage s, global -> LDS
        if (no_dma) return;
        unsigned char *st = ring + (s % NS) * STAGE;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc[i] + s * 16 * KS),
                                         (__attribute__((address_space(3))) void *)(st + ldst[i]), 16, 0, 0);
    };
    // the piece work of stage si: DMA: piece of stage si + 3.  REG: write the piece of stage si + 2 from the stage's register
    // set, then request the piece of stage si + 4 into the same registers.
    auto piece_work = [&](u4_t (&Rg)[REG ? LPS : 1], int si, int i, bool tail) {
        if (!REG) { if (!tail || si + NS - 1 < nst) dma_piece(si + NS - 1, i); return; }
        if (!tail || si + 2 < nst) lwrite_piece(Rg, si + 2, i);
        if (!tail || si + 4 < nst) gload_piece(Rg, si + 4, i);
    };
    // One k-step: its first MFMA (which waits for the step's own fragments -- the only LDS reads then outstanding, issued a
    // whole k-step earlier), then the LDS reads of the NEXT k-step's fragments into the other register set, then the remaining
    // MFMAs with the stage's pieces dealt between them (every PER_PIECE MFMAs one piece).
    constexpr int MPS = (SPLIT ? 3 : 1) * NT * MT;           // MFMAs per k-step
    constexpr int PPS = (LPS + KS - 1) / KS;                 // pieces per k-step
    constexpr int PER_PIECE = MPS / (PPS > 0 ? PPS : 1);
    auto kstep = [&](const Frags &F, Frags &Fnext, int nst_i, int nks, u4_t (&Rg)[REG ? LPS : 1], int si, int p0, bool tail) {
        int done = 0, piece = p0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[nt], F.b[mt], acc[nt][mt], 0, 0, 0);
                if (nt == 0 && mt == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (nst_i >= 0) load_step(nst_i, nks, Fnext);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (SPLIT) {
                    accx[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[nt], F.bl[mt], accx[nt][mt], 0, 0, 0);
                    accx[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.al[nt], F.b[mt], accx[nt][mt], 0, 0, 0);
                }
                done += SPLIT ? 3 : 1;
                if (piece < p0 + PPS && piece < LPS && done >= (piece - p0 + 1) * PER_PIECE) {
                    piece_work(Rg, si, piece, tail);
                    ++piece;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    };
    // top of stage i: stage i + 1 must be in LDS for everyone.  DMA: it has landed once at most the pieces of stage i + 2
    // (issued during stage i - 1) are outstanding.  REG: this wave's ds_writes of it (issued during stage i - 1) have completed.
    // A bare s_barrier (no fence: __syncthreads would wait for EVERY outstanding piece): each wave has waited for its own
    // pieces, the barrier makes that collective; LDS is coherent within the workgroup.
    // ---- the wave's block of C (needed before the K loop ends when its first batch is requested early) -----------------------
    const long long m0 = m0t + wr * MT * 32, n0 = n0t + wc * NT * 32;
    const long long mrem = m - m0, nrem = n - n0;
    const bool wave_in = mrem > 0 && nrem > 0;   // wave-uniform
    const long long ncl = nrem < NT * 32 ? nrem : NT * 32, mcl = mrem < MT * 32 ? mrem : MT * 32;
    constexpr unsigned ES = C32 ? 4u : 8u;    // bytes per element of the updated block
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((char *)Cv + (wave_in ? (m0 + n0 * ldc) * (long long)ES : 0)), 0, wave_in ? (int)(((ncl - 1) * ldc + mcl) * ES) : 0, 0x00020000);
    const unsigned ldc8 = (unsigned)ldc * ES;
    constexpr bool PF = EPI == 2 && C32 && !SPLIT && !REG;   // first batch requested inside the K loop
    constexpr int NPF = NT * 16;                               // its loads
    float cfA[C32 ? NT : 1][16];
    auto c_voff = [&](int mt) -> unsigned {
        return (mt * 32 + r < mrem) ? (unsigned)(mt * 32 + r) * ES + (unsigned)(4 * h) * ldc8 : 0x80000000u;
    };
    auto c_load32 = [&](float (&cf)[C32 ? NT : 1][16], int mt) {
        const unsigned voff = c_voff(mt);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const unsigned soff = (unsigned)(nt * 32 + (g & 3) + 8 * (g >> 2)) * ldc8;
                cf[C32 ? nt : 0][g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (int)voff, (int)soff, C_AUX));
            }
    };
    // top of stage i: stage i + 1 must be in LDS for everyone.  DMA: it has landed once at most the pieces of the stages after it
    // (issued during stages i - 1 ...) are outstanding -- and, with the early C batch, its NPF loads, which are younger than every
    // piece.  REG: this wave's ds_writes of it (issued during stage i - 1) have completed.
    // mode: 0 = later pieces stay in flight, 1 = nothing but (possibly) the C batch is younger than what must have landed,
    //       2 = nothing to wait for (every piece has landed before), 3 = wait for everything
    auto top_of_stage = [&](int mode) {
        if (REG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if (mode == 0 && NS > 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * LPS) : "memory");
        else if (mode == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPF) : "memory");
        else if (mode != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // The early C batch goes out right after the top of stage PFS: NS = 4: nst - 3 (the pieces of the last stage, nst - 1, are
    // then in flight and older: the top of stage nst - 2 waits for all but the NPF youngest operations); NS = 3: nst - 2 (the
    // top of that stage has waited for every piece).  The tops after it wait for nothing: every piece has landed.
    const int PFS = PF ? (NS > 3 ? nst - 3 : nst - 2) : -1;
    // one stage: KS k-steps; the fragment sets alternate per k-step (KS = 2: F0, F1 inside a stage; KS = 1: by stage parity)
    auto stage = [&](int i, bool odd, u4_t (&Rg)[REG ? LPS : 1], bool tail) {
        if (i > 0) {
            if (PF && tail && i > PFS) top_of_stage((NS > 3 && i == PFS + 1) ? 1 : 2);
            else top_of_stage(i + NS - 2 < nst ? 0 : 3);
        }
        if constexpr (PF) { if (tail && i == PFS) { c_load32(cfA, 0); __builtin_amdgcn_sched_barrier(0); } }
        if (KS == 2) {
            kstep(F0, F1, i, 1, Rg, i, 0, tail);
            kstep(F1, F0, i + 1 < nst ? i + 1 : -1, 0, Rg, i, PPS, tail);
        } else if (!odd) kstep(F0, F1, i + 1 < nst ? i + 1 : -1, 0, Rg, i, 0, tail);
        else kstep(F1, F0, i + 1 < nst ? i + 1 : -1, 0, Rg, i, 0, tail);
    };
    if (REG) {   // stages 0, 1 into LDS, stages 2, 3 requested
#pragma unroll
        for (int i = 0; i < LPS; ++i) { gload_piece(R0, 0, i); if (1 < nst) gload_piece(R1, 1, i); }
#pragma unroll
        for (int i = 0; i < LPS; ++i) { lwrite_piece(R0, 0, i); if (1 < nst) lwrite_piece(R1, 1, i); }
#pragma unroll
        for (int i = 0; i < LPS; ++i) { if (2 < nst) gload_piece(R0, 2, i); if (3 < nst) gload_piece(R1, 3, i); }
    } else {
        for (int s2 = 0; s2 < NS - 1 && s2 < nst; ++s2)
#pragma unroll
            for (int i = 0; i < LPS; ++i) dma_piece(s2, i);
    }
    top_of_stage(NS - 2 < nst ? 0 : 3);
#ifdef MPF_PROBE   // stamps of the K loop: shader cycles and the constant 100-MHz clock (stamp buffer in place of the unused Ul)
    unsigned long long st_c0 = 0, st_r0 = 0;
    const bool stamp = dbg_ != 0 && !SPLIT && Ul != nullptr && tid == 0;
    if (stamp) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    load_step(0, 0, F0);
    // trips of two stages (nst is even): register sets and fragment sets alternate statically.  Main part: no edge tests.
    int i = 0;
    const int imain = (nst > 4 ? nst - 4 : 0) & ~1;
    for (; i < imain; i += 2) { stage(i, false, R0, false); stage(i + 1, true, R1, false); }
    for (; i < nst; i += 2) { stage(i, false, R0, true); stage(i + 1, true, R1, true); }
#ifdef MPF_PROBE
    if (stamp) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *sb = (unsigned long long *)Ul;
        atomicAdd(sb + 0, c1 - st_c0); atomicAdd(sb + 1, r1 - st_r0); atomicAdd(sb + 2, 1ull);
    }
#endif
    // ---- epilogue: the wave's block through registers, one MFMA tile-row (NT tiles) per batch -----------------------------
    if (!wave_in) return; // wave-uniform, after the last barrier
    if (dbg == 1) {   // keep the accumulators alive without the C stream
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#if defined(__HIP_DEVICE_COMPILE__)   // (a "v" constraint makes the HOST pass drop the kernel's stub without a diagnostic)
                asm volatile("" ::"v"(acc[nt][mt]));
#endif
            }
        return;
    }
    if constexpr (EPI >= 1 && C32) {
        // Pipelined: the loads of up to three batches are in flight (the fragment registers are free now), the subtraction
        // happens in the staging set -- free again as soon as its stores have been issued: a store needs no waiting for.
        // Order per batch b: [loads b + 2 issued] consume b, store b.
        float cfB[C32 ? NT : 1][16], cfC[C32 ? NT : 1][16];
        auto consume = [&](float (&cf)[C32 ? NT : 1][16], int mt) {
            const unsigned voff = c_voff(mt);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    float pf = acc[nt][mt][g];
                    if (SPLIT) pf += accx[SPLIT ? nt : 0][SPLIT ? mt : 0][g] * (float)(1.0 / SPLIT_SCALE);
                    cf[C32 ? nt : 0][g] -= pf;   // (in place in the accumulator hipcc 7.2 folds the 16 elements of a tile into one)
                }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const unsigned soff = (unsigned)(nt * 32 + (g & 3) + 8 * (g >> 2)) * ldc8;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cf[C32 ? nt : 0][g]), rc, (int)voff, (int)soff, C_AUX);
                }
        };
        if (!PF) c_load32(cfA, 0);
        else {   // the early batch has landed (so has every operand piece): say so BEFORE the next loads go out, or the compiler's
                 // counter model waits for everything in flight at the first use of the batch
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 16; ++g) asm volatile("" : "+v"(cfA[C32 ? nt : 0][g]));
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MT > 1) c_load32(cfB, 1);
        if (MT > 2) c_load32(cfC, 2);
        __builtin_amdgcn_sched_barrier(0);
        consume(cfA, 0);
        if (MT > 3) { c_load32(cfA, 3); __builtin_amdgcn_sched_barrier(0); }
        if (MT > 1) consume(cfB, 1);
        if (MT > 2) consume(cfC, 2);
        if (MT > 3) consume(cfA, 3);
        static_assert(MT <= 4, "four batches");
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const unsigned voff = c_voff(mt);
        float cf[C32 ? NT : 1][16];
        double cv[C32 ? 1 : NT][16];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const unsigned soff = (unsigned)(nt * 32 + (g & 3) + 8 * (g >> 2)) * ldc8;
                if (C32) cf[C32 ? nt : 0][g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (int)voff, (int)soff, C_AUX));
                else cv[C32 ? 0 : nt][g] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rc, (int)voff, (int)soff, C_AUX));
            }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const unsigned soff = (unsigned)(nt * 32 + (g & 3) + 8 * (g >> 2)) * ldc8;
                if (C32) {
                    float pf = acc[nt][mt][g];
                    if (SPLIT) pf += accx[SPLIT ? nt : 0][SPLIT ? mt : 0][g] * (float)(1.0 / SPLIT_SCALE);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cf[C32 ? nt : 0][g] - pf), rc, (int)voff, (int)soff, C_AUX);
                } else {
                    double p = (double)acc[nt][mt][g];
                    if (SPLIT) p += (double)accx[SPLIT ? nt : 0][SPLIT ? mt : 0][g] * (1.0 / SPLIT_SCALE);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, cv[C32 ? 0 : nt][g] - p), rc, (int)voff, (int)soff, C_AUX);
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
// image at c->h_U + elem_off, row stride kstride elements (0: the padded K itself)
int launch_cvt_u12(mpf_ctx *c, const double *B, int64_t ldb, int K, int64_t n, int split, int64_t elem_off, int kstride) {
    if (n <= 0 || K <= 0) return 0;
    const int Kp = (K + 63) & ~63;
    if (!c->h_U) { c->err = "fp16 operand image not allocated"; return -1; }
    unsigned short *Uh = c->h_U + elem_off, *Ul = Uh + c->h_rows * c->h_kmax;
    long long cb = n < 4096 ? n : 4096;
    // a block of a wider image: only its own K columns are written (the image's padding is zeroed by its owner)
    cvt_u12_kernel<<<(int)cb, 256, 0, c->stream>>>(B, ldb, K, kstride ? K : Kp, n, Uh, split ? Ul : nullptr, kstride ? kstride : Kp);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
static int hgemm_minus_any(mpf_ctx *c, int64_t m, int64_t n, int K, const double *B, int64_t ldb, void *C, int64_t ldc, bool c32,
                           int split, int img, int64_t elem_off) {
    if (m <= 0 || n <= 0 || K <= 0) return 0;
    int rc = launch_cvt_u12(c, B, ldb, K, n, split, 0, 0);
    if (!rc) rc = launch_hgemm_images(c, m, n, K, C, ldc, c32, split, img, elem_off);
    return rc;
}
template <bool SPLIT, bool C32, int MT, int NT, int WM, int WN, int NS, bool REG, int WPE = 1, int EPI = 0>
static int launch_big(mpf_ctx *c, int slot, int64_t m, int64_t n, int Kp, const HgemmImages &im, void *C, int64_t ldc) {
    constexpr int TM = WM * MT * 32, TN = WN * NT * 32;
    constexpr int LDS = NS * (SPLIT ? 2 : 1) * (TM + TN) * (SPLIT ? 32 : 64);
    auto *kern = hgemm_big_kernel<SPLIT, C32, MT, NT, WM, WN, NS, REG, WPE, EPI>;
    // the dynamic-LDS attribute belongs to the device: set once per (kernel, context) -- a context is tied to one device
    const unsigned bit = 1u << (slot + (C32 ? 8 : 0));
    if (!(c->attr_big & bit)) {
        MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        c->attr_big |= bit;
    }
    const long long bm = (m + TM - 1) / TM, bn = (n + TN - 1) / TN;
#ifdef MPF_PROBE
    const int dbg = c->tune.hgemm_dbg;
    const unsigned short *Ul = (dbg && !SPLIT) ? (const unsigned short *)c->ws->hp_stamps : im.Ul;   // K-loop stamps (sums; cleared by the reader)
#else
    const int dbg = 0;
    const unsigned short *Ul = im.Ul;
#endif
    kern<<<(int)(bm * bn), 512, LDS, c->stream>>>(m, n, Kp, im.Lh, im.Uh, im.Ll, Ul, C, ldc, (int)bm, (int)bn, im.ksL, im.ksU, dbg);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

// L image: buffer img at elem_off, row stride ksL (0: padded K); U image: c->h_U + u_off, row stride ksU (0: padded K)
int launch_hgemm_images(mpf_ctx *c, int64_t m, int64_t n, int K, void *C, int64_t ldc, bool c32, int split, int img, int64_t elem_off,
                        int64_t u_off, int ksL, int ksU) {
    if (m <= 0 || n <= 0 || K <= 0) return 0;
    unsigned short *Lh = l_image(c, img);
    if (!Lh || !c->h_U) { c->err = "fp16 operand image not allocated"; return -1; }
    HgemmImages im;
    const int64_t lo = c->h_rows * c->h_kmax;
    im.Lh = Lh + elem_off; im.Ll = im.Lh + lo; im.Uh = c->h_U + u_off; im.Ul = im.Uh + lo; im.ksL = ksL; im.ksU = ksU;
    return launch_hgemm_ptrs(c, m, n, K, im, C, ldc, c32, split);
}
// The same with the roles of the two operand images exchanged: the kernel computes D[n][m] tiles with D's row index on C's
// column, so C^T (n x m, "column-major" with leading dimension = the ROW stride of a row-major C) -= Uimg * Limg^T is the same
// kernel with (m, n) and the images swapped.  This is how the row-major fp32 working copy is updated.
int launch_hgemm_images_rowmajor(mpf_ctx *c, int64_t m, int64_t n, int K, float *Crm, int64_t ldrow, int split, int img, int64_t elem_off,
                                 int64_t u_off, int ksL, int ksU) {
    if (m <= 0 || n <= 0 || K <= 0) return 0;
    unsigned short *Lh = l_image(c, img);
    if (!Lh || !c->h_U) { c->err = "fp16 operand image not allocated"; return -1; }
    const int Kp = (K + 63) & ~63;
    HgemmImages im;
    const int64_t lo = c->h_rows * c->h_kmax;
    im.Uh = Lh + elem_off; im.Ul = im.Uh + lo; im.Lh = c->h_U + u_off; im.Ll = im.Lh + lo; im.ksU = ksL ? ksL : Kp; im.ksL = ksU ? ksU : Kp;
    return launch_hgemm_ptrs(c, n, m, K, im, Crm, ldrow, true, split);
}
int launch_hgemm_ptrs(mpf_ctx *c, int64_t m, int64_t n, int K, const HgemmImages &im, void *C, int64_t ldc, bool c32, int split) {
    if (m <= 0 || n <= 0 || K <= 0) return 0;
    // the kernel addresses a wave's block of C with 32-bit byte offsets from the block's base (63 * ldc * 8 < 2^31)
    if (ldc > (1ll << 21)) { c->err = "hgemm: leading dimension too large for 32-bit tile offsets"; return -1; }
    if (((m + 127) / 128) * ((n + 127) / 128) > 0x7FFFFFFFll) { c->err = "hgemm: too many tiles"; return -1; }
    const int Kp = (K + 63) & ~63;
    const unsigned short *Lh = im.Lh, *Ll = im.Ll, *Uh = im.Uh, *Ul = im.Ul;
    int ksL = im.ksL ? im.ksL : Kp, ksU = im.ksU ? im.ksU : Kp;
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
    // big shapes (the K = sb * nb updates of the two-level schedule): the eight-wave big-tile kernel
    if (c->tune.hgemm_big && m >= 1024 && n >= 1024 && Kp >= 256) {
        const HgemmImages im2 = {Lh, Ll, Uh, Ul, ksL, ksU};
        // tile choice (option hgemm_big_tile): 0 = 256 x 256 (plain) -- the fastest kernel alone, but its workgroup (432 of the
        // SIMD's 512 registers per lane, 128 KB of LDS) leaves no room on its CU for any kernel of the panel chain or the inner
        // lane: beside a running update those wait for workgroups to retire; 1 = 128 x 256 (64 x 64 per wave: ~230 registers
        // per lane for its two waves, 96 KB): a TRSM, an fp64-panel or a small-update workgroup fits beside it.
#ifdef MPF_PROBE   // probe library only: operands through registers (ring of 3 stages) instead of LDS-DMA -- measured 3 % slower
        if (c->tune.hgemm_big_reg) {
            if (split) return c32 ? launch_big<true, true, 2, 2, 4, 2, 3, true>(c, 0, m, n, Kp, im2, C, ldc) : launch_big<true, false, 2, 2, 4, 2, 3, true>(c, 0, m, n, Kp, im2, C, ldc);
            return c32 ? launch_big<false, true, 4, 2, 2, 4, 3, true>(c, 1, m, n, Kp, im2, C, ldc) : launch_big<false, false, 4, 2, 2, 4, 3, true>(c, 1, m, n, Kp, im2, C, ldc);
        }
#endif
        const int tile = c->tune.hgemm_big_tile;
        // plain operands: the 16x16x32 family (hgemm16.hip) unless switched off
        if (!split && c->tune.hgemm_mfma16) return launch_hgemm16_big(c, m, n, Kp, im2, C, ldc, c32);
        // The fp32 copy with plain operands has two kernels.  hgemm_pp_kernel (hgemm_pp.hip: persistent workgroups, ping-pong wave
        // groups) is the faster one ALONE (815 against 802 TFLOP/s at K = 1024, 965 against 938 at K = 2048), but its 256 workgroups
        // keep every CU for the whole launch: inside a factorization the pivot kernel of the chain -- which needs whole CUs -- then
        // waits for a launch of ~0.7 ms to end instead of for a tile of ~30 us to retire, and the fp16 mode at N = 32768 takes 160.0 ms
        // instead of 146.6 although the update launches themselves are faster (525 against 440 TFLOP/s in the schedule;
        // gpurun_out/r04_r_tilemodes.log).  hgemm_big_tile = 0 (default): hgemm_pp_kernel for stand-alone calls (the step operator
        // mpf_hgemm_minus_f32), hgemm_big_kernel with the pipelined C stream inside factorizations; 5 = hgemm_pp_kernel everywhere,
        // 4 = hgemm_big_kernel everywhere, 3 = round 3's kernel (serial C batches).
        if (((tile == 0 && c->hgemm_standalone) || tile == 5) && c32 && !split && (((uintptr_t)C | (uintptr_t)(ldc * 4)) & 15) == 0)
            return launch_hgemm_pp(c, m, n, Kp, im2, (float *)C, ldc);
        // (EPI: the fp32 copy's C tile pipelined; an fp64 tile keeps the serial form -- two of its batches do not fit the registers)
        if (split) return c32 ? (tile == 3 ? launch_big<true, true, 2, 2, 4, 2, 4, false>(c, 6, m, n, Kp, im2, C, ldc)
                                           : launch_big<true, true, 2, 2, 4, 2, 4, false, 1, 1>(c, 2, m, n, Kp, im2, C, ldc))
                              : launch_big<true, false, 2, 2, 4, 2, 4, false>(c, 2, m, n, Kp, im2, C, ldc);
        if (tile == 1)
            return c32 ? launch_big<false, true, 2, 2, 2, 4, 4, false>(c, 3, m, n, Kp, im2, C, ldc) : launch_big<false, false, 2, 2, 2, 4, 4, false>(c, 3, m, n, Kp, im2, C, ldc);
        // 2 = the 128 x 256 tile with TWO workgroups per CU (ring of three 24-KB stages, <= 128 registers per lane): one
        // workgroup's C tile streams while the other's K loop runs
        if (tile == 2)   // (an fp64 C tile does not fit 128 registers beside the accumulators: one workgroup per CU)
            return c32 ? launch_big<false, true, 2, 2, 2, 4, 3, false, 4>(c, 4, m, n, Kp, im2, C, ldc) : launch_big<false, false, 2, 2, 2, 4, 4, false>(c, 3, m, n, Kp, im2, C, ldc);
        if (tile == 3)   // round 3's kernel (serial C batches): A/B reference
            return c32 ? launch_big<false, true, 4, 2, 2, 4, 4, false>(c, 7, m, n, Kp, im2, C, ldc) : launch_big<false, false, 4, 2, 2, 4, 4, false>(c, 5, m, n, Kp, im2, C, ldc);
        return c32 ? launch_big<false, true, 4, 2, 2, 4, 4, false, 1, 2>(c, 5, m, n, Kp, im2, C, ldc) : launch_big<false, false, 4, 2, 2, 4, 4, false>(c, 5, m, n, Kp, im2, C, ldc);
    }
    if (!split && c->tune.hgemm_mfma16) { const HgemmImages im2 = {Lh, Ll, Uh, Ul, ksL, ksU}; return launch_hgemm16_ring(c, m, n, Kp, im2, C, ldc, c32); }
    if (split) {
        if (c32) hgemm_ring_kernel<true, true><<<g, 256, pad_split, c->stream>>>(m, n, Kp, Lh, Uh, Ll, Ul, C, ldc, (int)tm, (int)tn, ksL, ksU);
        else hgemm_ring_kernel<true, false><<<g, 256, pad_split, c->stream>>>(m, n, Kp, Lh, Uh, Ll, Ul, C, ldc, (int)tm, (int)tn, ksL, ksU);
    } else {
        if (c32) hgemm_ring_kernel<false, true><<<g, 256, pad_plain, c->stream>>>(m, n, Kp, Lh, Uh, nullptr, nullptr, C, ldc, (int)tm, (int)tn, ksL, ksU);
        else hgemm_ring_kernel<false, false><<<g, 256, pad_plain, c->stream>>>(m, n, Kp, Lh, Uh, nullptr, nullptr, C, ldc, (int)tm, (int)tn, ksL, ksU);
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
#ifdef MPF_PROBE
// probe library: the update kernel ALONE on the images the last mpf_hgemm_minus_f32 call left in the context
extern "C" int mpf_debug_hgemm_again(mpf_ctx *c, int64_t m, int64_t n, int32_t k, float *d_C, int64_t ldc, int32_t split) {
    if (!c || !d_C) return -1;
    c->hgemm_standalone = true;
    const int rc = launch_hgemm_images(c, m, n, k, d_C, ldc, true, split, 0, 0);
    c->hgemm_standalone = false;
    return rc;
}
#endif
int launch_hgemm_minus(mpf_ctx *c, int64_t m, int64_t n, int K, const double *B, int64_t ldb, double *C, int64_t ldc,
                       int split, int img, int64_t elem_off) {
    return hgemm_minus_any(c, m, n, K, B, ldb, C, ldc, false, split, img, elem_off);
}
// the same update on the fp32 working copy of the trailing matrix (two-level schedule of the fp16 modes)
int launch_hgemm_minus_w32(mpf_ctx *c, int64_t m, int64_t n, int K, const double *B, int64_t ldb, float *C, int64_t ldc,
                           int split, int img, int64_t elem_off) {
    return hgemm_minus_any(c, m, n, K, B, ldb, C, ldc, true, split, img, elem_off);
}

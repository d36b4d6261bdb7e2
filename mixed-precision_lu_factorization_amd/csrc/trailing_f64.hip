// fp64 trailing update for gfx950: replaces the cublasDtrsm call (reference MPF.cu:215-225) and the
// cublasDgemm call (MPF.cu:230-239) -- where two thirds of N^3 flops live.
//
// dgemm_minus: C[m x n] -= A[m x K] * B[K x n], all column-major.
//   * v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per instruction).  Measured on MI355X (tools/mfma_pat_probe.py,
//     tools/mfma4_probe.py, profiles/r02_mfma_f64_issue.txt): v_mfma_f64_16x16x4_f64 runs at its full rate only while
//     consecutive MFMAs use the SAME accumulator (72 cycles issue to issue; 138 when the accumulator changes: 36 TFLOP/s chip-wide
//     with independent accumulators at any occupancy, which is what capped the round-1 kernel at 0.64-0.74 of peak), whereas the
//     4x4x4 form issues every 16.5 cycles on independent accumulators from ONE wave per SIMD (72-75 TFLOP/s chip-wide).
//     Operand lanes: A lane 16k+4b+i = A_b[i][k], B lane 16k+4b+j = B_b[k][j], D lane 16i+4b+j = D_b[i][j] -- i.e. with the
//     16 x 4 fragments of the 16x16x4 form one instruction yields the four DIAGONAL 4x4 blocks of the 16 x 16 product; the other
//     twelve come from the same A fragment against the B fragment rotated by 4, 8, 12 entries (read from LDS at rotated lane
//     addresses).  Per element the result is the chain c = fma(a_k, b_k, c), k ascending (3200 / 3200 bit matches against an
//     exact fma chain), the same as the 16x16x4 form: contract C5 is unchanged and so is every bit of the output.
//   * The MFMA's "A" index (i, 4b+i) is mapped to C's COLUMN and its "B" index to C's ROW, so the lanes of a 16-lane group
//     cover 4 columns x 4 consecutive rows (32-byte runs; the four rotations of a tile complete the 128-byte lines).
//   * 128 x 128 tile per 256-thread workgroup (4 waves, 64 x 64 each = 16 accumulator tiles,
//     128 VGPRs), K stepped 16 at a time through a double-buffered LDS stage (73.7 KB => two
//     workgroups per CU, one hides the C prologue/epilogue of the other).
//   * LDS images are padded for conflict-free ds_read_b64 fragments: A-tile [k][m] stride 144
//     doubles (lanes 16..31 land 128 B further in the bank row), B-tile [n][k] stride 18.
//   * Per element the update is the chain c = fma(-a_k, b_k, c), k ascending (contract C5): the
//     oracle reproduces it bit for bit, which is what keeps later fp16 pivots identical.
//   * blockIdx -> tile mapping is XCD-aware (bijective remap: the 8 XCDs each get a contiguous
//     run of tiles, tiles of a run share the same B column panel in that XCD's L2).
#include "mpf_internal.h"

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));

// 8-byte load through a buffer descriptor: address = base (scalar registers) + voff (one VGPR) + soff (scalar).
// The staging loads of the GEMM differ only in their scalar part, so sixteen of them need TWO address VGPRs in
// total instead of sixteen 64-bit VGPR addresses -- which is what lets all of them be in flight at once
// (with VGPR addresses the compiler recycled address registers as load destinations and serialised the loads).
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const u2_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)soff, 0);
    return __builtin_bit_cast(double, v);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7FFFFFFF, 0x00020000);
}

// rotate a double inside each 16-lane row (DPP row_ror: lane i takes lane (i - n) & 15; CTRL = 0x120 + n)
template <int CTRL>
__device__ __forceinline__ double dpp_row_ror(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

#ifndef MPF_DGEMM_C_NT
#define MPF_DGEMM_C_NT 1
#endif
// the C tile is read once and written once per launch: streamed with the non-temporal policy so that it does not evict the
// operand tiles, which every workgroup of a tile row / column re-reads, from the XCD's L2
__device__ __forceinline__ double c_load(const void *p) {
#if MPF_DGEMM_C_NT
    return __builtin_nontemporal_load((const double *)p);
#else
    return *(const double *)p;
#endif
}
__device__ __forceinline__ void c_store(void *p, double v) {
#if MPF_DGEMM_C_NT
    __builtin_nontemporal_store(v, (double *)p);
#else
    *(double *)p = v;
#endif
}

constexpr int GT = 128;   // tile edge
constexpr int GBK = 16;   // K per stage.  (8 was tried so that two GEMM workgroups and a pivot workgroup of the look-ahead
                          // chain fit on one CU: the extra barriers cost the GEMM 16 %, more than the sharing it avoids.)
constexpr int GSA = 144;  // LDS stride of the A image [k][m]
constexpr int GSB = GBK + 2; // LDS stride of the B image [n][k] (18 / 10: conflict-free ds_read_b64 fragments)
constexpr int G_LDS_DOUBLES = 2 * GBK * GSA + 2 * GT * GSB;
constexpr int G_EPT = GBK * GT / 256; // staged elements per thread and operand

// ---- 16x16x4 form of the tile (round-1 kernel; kept for A/B measurement: MPF_GEMM_MF=0 classic order, 1 accumulator runs) ----
template <bool EDGE, int ORDER, bool STAMP = false>
__device__ __forceinline__ void dgemm_tile16(long long m, long long n, int K, const double *__restrict__ A, long long lda,
                                           const double *__restrict__ B, long long ldb, double *__restrict__ C,
                                           long long ldc, long long m0, long long n0, double *As, double *Bs,
                                             unsigned long long *stamps = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0}, tlast = 0, rt0 = 0;
    if (STAMP) { tlast = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
#define G_STAMP(i) do { if (STAMP) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); seg[i] += tn_ - tlast; tlast = tn_; } } while (0)
    const int wm = wave & 1, wn = wave >> 1;
    const int lj = lane & 15, lk = lane >> 4;
    const int mrem = (int)((m - m0) < GT ? (m - m0) : GT), nrem = (int)((n - n0) < GT ? (n - n0) : GT);

    // ---- accumulators <- C tile: register rr of tile (nt, mt) is C[.. + lj, .. + lk + 4 rr] --------
    char *Cb = (char *)(C + m0 + n0 * ldc);
    const int crow = wm * 64 + lj, ccol = wn * 64 + lk;
    const unsigned ldc8 = (unsigned)ldc * 8u;
    d4_t acc[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int col = ccol + nt * 16 + 4 * rr;
            const unsigned coff = (unsigned)col * ldc8 + (unsigned)crow * 8u;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (!EDGE || (crow + mt * 16 < mrem && col < nrem))
                    acc[nt][mt][rr] = c_load(Cb + coff + mt * 128);
                else
                    acc[nt][mt][rr] = 0.0;
            }
        }

    // ---- staging: thread loads G_EPT + G_EPT doubles per K stage ----------------------------------------
    constexpr int NSTEP = 256 / GBK;           // B image: columns covered by one pass of the 256 threads
    const int mA = tid & 127, kA0 = tid >> 7;  // A image element i: (k = kA0 + 2i, m = mA)
    const int kB = tid & (GBK - 1), nB0 = tid / GBK; // B image element i: (n = nB0 + NSTEP*i, k = kB)
    const unsigned lda8 = (unsigned)lda * 8u, ldb8 = (unsigned)ldb * 8u;
    const unsigned offA0 = (unsigned)mA * 8u + (unsigned)kA0 * lda8;
    const unsigned offB0 = (unsigned)kB * 8u + (unsigned)nB0 * ldb8;
    double ra[G_EPT], rb[G_EPT];
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(A + m0), rB = make_rsrc(B + n0 * ldb);
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < G_EPT; ++i) {
            if (!EDGE || (mA < mrem && k0 + kA0 + 2 * i < K)) ra[i] = buf_load_f64(rA, offA0, (unsigned)(k0 + 2 * i) * lda8);
            else ra[i] = 0.0;
            if (!EDGE || (nB0 + NSTEP * i < nrem && k0 + kB < K)) rb[i] = buf_load_f64(rB, offB0, (unsigned)k0 * 8u + (unsigned)(NSTEP * i) * ldb8);
            else rb[i] = 0.0;
        }
    };
    auto sstore = [&](int buf) {
        double *as = As + buf * GBK * GSA + kA0 * GSA + mA;
        double *bs = Bs + buf * GT * GSB + nB0 * GSB + kB;
#pragma unroll
        for (int i = 0; i < G_EPT; ++i) {
            as[2 * i * GSA] = -ra[i]; // negate here, not at the load: the loads must not be waited for before the MFMAs
            bs[NSTEP * i * GSB] = rb[i];
        }
    };

    const int nK = (K + GBK - 1) / GBK;
    gload(0);
    sstore(0);
    __syncthreads();
    G_STAMP(0);
    // Pin the C loads as COMPLETE before the K loop.  Otherwise the compiler leaves a few of them in flight into the
    // loop and guards the MFMAs that consume them with s_waitcnt vmcnt(3..0) -- in the shared loop body, i.e. in EVERY
    // iteration, where those waits also drain the sixteen staging loads issued at the top of the same iteration.
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) asm volatile("" : "+v"(acc[nt][mt]));
    for (int it = 0; it < nK; ++it) {
        const int buf = it & 1;
        if (it + 1 < nK) gload((it + 1) * GBK);
        const double *as = As + buf * GBK * GSA + wm * 64 + lj;
        const double *bs = Bs + buf * GT * GSB + (wn * 64 + lj) * GSB;
        if (ORDER == 0) {
#pragma unroll
            for (int kk = 0; kk < GBK; kk += 4) {
                double af[4], bf[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) af[nt] = bs[nt * 16 * GSB + kk + lk];          // B[k][n]
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) bf[mt] = as[(kk + lk) * GSA + mt * 16];        // -A[m][k]
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[nt], bf[mt], acc[nt][mt], 0, 0, 0);
            }
        } else {
            // every accumulator takes the stage's k-steps back to back: the 16x16x4 form only runs at its full rate while
            // consecutive MFMAs write the same accumulator (same per-element order: k ascending)
            double af[GBK / 4][4], bf[GBK / 4][4];
#pragma unroll
            for (int kq = 0; kq < GBK / 4; ++kq) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) af[kq][nt] = bs[nt * 16 * GSB + 4 * kq + lk];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) bf[kq][mt] = as[(4 * kq + lk) * GSA + mt * 16];
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int kq = 0; kq < GBK / 4; ++kq)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kq][nt], bf[kq][mt], acc[nt][mt], 0, 0, 0);
        }
        if (STAMP) asm volatile("s_nop 0" ::: "memory");
        G_STAMP(1);
        if (it + 1 < nK) sstore(buf ^ 1);
        G_STAMP(2);
        __syncthreads();
        G_STAMP(3);
    }

    // ---- C tile <- accumulators ----------------------------------------------------------------------
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int col = ccol + nt * 16 + 4 * rr;
            const unsigned coff = (unsigned)col * ldc8 + (unsigned)crow * 8u;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                if (!EDGE || (crow + mt * 16 < mrem && col < nrem)) c_store(Cb + coff + mt * 128, acc[nt][mt][rr]);
        }
    if (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G_STAMP(4);
        if (stamps && tid == 0) { for (int i = 0; i < 5; ++i) stamps[i] = seg[i]; stamps[5] = __builtin_amdgcn_s_memrealtime() - rt0; }
    }
#undef G_STAMP
}

// One 128 x 128 tile.  EDGE = false: the tile is interior and K is a multiple of GBK (no guards).
// All global addresses are a wave-uniform 64-bit base plus a 32-bit per-lane byte offset.
template <bool EDGE, int ROT, bool STAMP = false>
__device__ __forceinline__ void dgemm_tile(long long m, long long n, int K, const double *__restrict__ A, long long lda,
                                           const double *__restrict__ B, long long ldb, double *__restrict__ C,
                                           long long ldc, long long m0, long long n0, double *As, double *Bs,
                                           unsigned long long *stamps = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0}, tlast = 0, rt0 = 0;
    if (STAMP) { tlast = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
#define G_STAMP(i) do { if (STAMP) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); seg[i] += tn_ - tlast; tlast = tn_; } } while (0)
    const int wm = wave & 1, wn = wave >> 1;
    const int lj = lane & 15, lk = lane >> 4;
    const int mrem = (int)((m - m0) < GT ? (m - m0) : GT), nrem = (int)((n - n0) < GT ? (n - n0) : GT);

    // ---- accumulators <- C tile.  Register s of tile (nt, mt) in lane (lj, lk) is the element
    //      row  mt*16 + ((lj + 4 s) & 15),  column  nt*16 + 4 (lj >> 2) + lk   of the wave's 64 x 64 block --------
    char *Cb = (char *)(C + m0 + n0 * ldc);
    const unsigned ldc8 = (unsigned)ldc * 8u;
    const int ccolL = wn * 64 + 4 * (lj >> 2) + lk;
    int crowL[4];
    unsigned coffL[4];
#pragma unroll
    for (int sr = 0; sr < 4; ++sr) {
        crowL[sr] = wm * 64 + ((lj + 4 * sr) & 15);
        coffL[sr] = (unsigned)ccolL * ldc8 + (unsigned)crowL[sr] * 8u;
    }
    double acc[4][4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int sr = 0; sr < 4; ++sr) {
                if (!EDGE || (crowL[sr] + mt * 16 < mrem && ccolL + nt * 16 < nrem))
                    acc[nt][mt][sr] = c_load(Cb + coffL[sr] + (unsigned)(nt * 16) * ldc8 + mt * 128);
                else
                    acc[nt][mt][sr] = 0.0;
            }

    // ---- staging: thread loads G_EPT + G_EPT doubles per K stage ----------------------------------------
    constexpr int NSTEP = 256 / GBK;           // B image: columns covered by one pass of the 256 threads
    const int mA = tid & 127, kA0 = tid >> 7;  // A image element i: (k = kA0 + 2i, m = mA)
    const int kB = tid & (GBK - 1), nB0 = tid / GBK; // B image element i: (n = nB0 + NSTEP*i, k = kB)
    const unsigned lda8 = (unsigned)lda * 8u, ldb8 = (unsigned)ldb * 8u;
    const unsigned offA0 = (unsigned)mA * 8u + (unsigned)kA0 * lda8;
    const unsigned offB0 = (unsigned)kB * 8u + (unsigned)nB0 * ldb8;
    double ra[G_EPT], rb[G_EPT];
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(A + m0), rB = make_rsrc(B + n0 * ldb);
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < G_EPT; ++i) {
            if (!EDGE || (mA < mrem && k0 + kA0 + 2 * i < K)) ra[i] = buf_load_f64(rA, offA0, (unsigned)(k0 + 2 * i) * lda8);
            else ra[i] = 0.0;
            if (!EDGE || (nB0 + NSTEP * i < nrem && k0 + kB < K)) rb[i] = buf_load_f64(rB, offB0, (unsigned)k0 * 8u + (unsigned)(NSTEP * i) * ldb8);
            else rb[i] = 0.0;
        }
    };
    auto sstore = [&](int buf) {
        double *as = As + buf * GBK * GSA + kA0 * GSA + mA;
        double *bs = Bs + buf * GT * GSB + nB0 * GSB + kB;
#pragma unroll
        for (int i = 0; i < G_EPT; ++i) {
            as[2 * i * GSA] = -ra[i]; // negate here, not at the load: the loads must not be waited for before the MFMAs
            bs[NSTEP * i * GSB] = rb[i];
        }
    };

    const int nK = (K + GBK - 1) / GBK;
    gload(0);
    sstore(0);
    __syncthreads();
    G_STAMP(0); // prologue: C tile + first stage
    // Pin the C loads as COMPLETE before the K loop.  Otherwise the compiler leaves a few of them in flight into the
    // loop and guards the MFMAs that consume them with s_waitcnt vmcnt(3..0) -- in the shared loop body, i.e. in EVERY
    // iteration, where those waits also drain the sixteen staging loads issued at the top of the same iteration.
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int sr = 0; sr < 4; ++sr) asm volatile("" : "+v"(acc[nt][mt][sr]));
    int rot[4];
#pragma unroll
    for (int sr = 0; sr < 4; ++sr) rot[sr] = (lj + 4 * sr) & 15;
    for (int it = 0; it < nK; ++it) {
        const int buf = it & 1;
        if (it + 1 < nK) gload((it + 1) * GBK);
        const double *as = As + buf * GBK * GSA + wm * 64;
        const double *bs = Bs + buf * GT * GSB + (wn * 64 + lj) * GSB;
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 4) {
            double af[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) af[nt] = bs[nt * 16 * GSB + kk + lk];          // B[k][n], n = .. + lj
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                double bq[4];
                if (ROT == 0) {
#pragma unroll
                    for (int sr = 0; sr < 4; ++sr) bq[sr] = as[(kk + lk) * GSA + mt * 16 + rot[sr]]; // -A[m][k], m rotated by 4 sr
                } else {
                    // one LDS read, three rotations inside the 16-lane rows on the VALU (DPP row_ror): lane f <- lane (f + 4 sr) & 15
                    bq[0] = as[(kk + lk) * GSA + mt * 16 + rot[0]];
                    bq[1] = dpp_row_ror<0x12C>(bq[0]);
                    bq[2] = dpp_row_ror<0x128>(bq[0]);
                    bq[3] = dpp_row_ror<0x124>(bq[0]);
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int sr = 0; sr < 4; ++sr)
                        acc[nt][mt][sr] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[nt], bq[sr], acc[nt][mt][sr], 0, 0, 0);
            }
        }
        if (STAMP) asm volatile("s_nop 0" ::: "memory");
        G_STAMP(1); // MFMA phase (issue)
        if (it + 1 < nK) sstore(buf ^ 1);
        G_STAMP(2); // wait for the staging loads + LDS writes
        __syncthreads();
        G_STAMP(3); // barrier
    }

    // ---- C tile <- accumulators ----------------------------------------------------------------------
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int sr = 0; sr < 4; ++sr)
                if (!EDGE || (crowL[sr] + mt * 16 < mrem && ccolL + nt * 16 < nrem))
                    c_store(Cb + coffL[sr] + (unsigned)(nt * 16) * ldc8 + mt * 128, acc[nt][mt][sr]);
    if (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G_STAMP(4); // epilogue: C stores drained
        if (stamps && tid == 0) { for (int i = 0; i < 5; ++i) stamps[i] = seg[i]; stamps[5] = __builtin_amdgcn_s_memrealtime() - rt0; }
    }
#undef G_STAMP
}

// diagnostic build of the same tile: one workgroup per CU-slot as usual, block 0 / wave 0 leaves its segment cycle sums
__global__ __launch_bounds__(256, 2) void dgemm_minus_stamp_kernel(long long m, long long n, int K, const double *__restrict__ A,
                                                                   long long lda, const double *__restrict__ B, long long ldb,
                                                                   double *__restrict__ C, long long ldc, int tiles_m, int tiles_n,
                                                                   unsigned long long *stamps) {
    extern __shared__ __attribute__((aligned(16))) double g_lds[];
    double *As = g_lds;
    double *Bs = g_lds + 2 * GBK * GSA;
    const int bid = blockIdx.x;
    const int tm = bid % tiles_m, tn = bid / tiles_m;
    dgemm_tile<false, true>(m, n, K, A, lda, B, ldb, C, ldc, (long long)tm * GT, (long long)tn * GT, As, Bs, bid == 0 ? stamps : nullptr);
}

// ---- interior tile with direct global -> LDS staging (LDS-DMA, 16 bytes per lane) ---------------------------------------
// Measured with the stamped build (tools/gemm_stamp_probe.py): whatever the MFMA form, a K stage of the register-staged tile
// takes ~9 150 cycles for two workgroups per CU = 8.8 B/clk/CU of operand traffic -- the CU's vector-memory path with 8-byte
// loads, not the matrix pipe, bounds the kernel.  Here every staging instruction moves 1 KB per wave straight into LDS
// (global_load_lds_dwordx4): a quarter of the instructions, no staging registers, no ds_write pass, no negation pass (the
// product is negated by the MFMA's NEG modifier).  A image [k][m]: one instruction = one 1-KB k-row, rows padded to 144
// doubles as before.  B image [n][16 k] = 128-byte rows; the eight 16-byte chunks of a row are stored XOR-swizzled by
// (n >> 1) & 7 -- applied to the per-lane SOURCE address, the LDS side of the DMA is lane-linear -- which makes the
// ds_read_b64 fragments conflict-free without padding.  Needs 16-byte aligned operands (even lda / ldb, K % 16 == 0);
// everything else takes the register-staged tile.
constexpr int DSA = GBK * GSA * 8;          // bytes of the A image of a stage (18 432)
constexpr int DSB = GT * GBK * 8;           // bytes of the B image of a stage (16 384)
constexpr int DSTAGE = DSA + DSB;
__device__ __forceinline__ void dgemm_dma_tile(int K, const double *__restrict__ A, long long lda, const double *__restrict__ B,
                                               long long ldb, double *__restrict__ C, long long ldc, long long m0, long long n0,
                                               unsigned char *lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int lj = lane & 15, lk = lane >> 4;
    // ---- accumulators <- C tile (layout of dgemm_tile) ----
    char *Cb = (char *)(C + m0 + n0 * ldc);
    const unsigned ldc8 = (unsigned)ldc * 8u;
    const int ccolL = wn * 64 + 4 * (lj >> 2) + lk;
    unsigned coffL[4];
    int rot[4];
#pragma unroll
    for (int sr = 0; sr < 4; ++sr) {
        rot[sr] = (lj + 4 * sr) & 15;
        coffL[sr] = (unsigned)ccolL * ldc8 + (unsigned)(wm * 64 + rot[sr]) * 8u;
    }
    double acc[4][4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int sr = 0; sr < 4; ++sr)
                acc[nt][mt][sr] = c_load(Cb + coffL[sr] + (unsigned)(nt * 16) * ldc8 + mt * 128);
    // ---- loader role: wave w brings A rows k = 4w .. 4w+3 and B rows n = 32w .. 32w+31 of every stage ----
    const double *asrc = A + m0 + 2 * lane + (long long)(4 * wave) * lda;         // + k * lda per instruction, + k0 * lda per stage
    const double *bsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nrow = 32 * wave + 8 * j + (lane >> 3);
        const int q = (lane & 7) ^ ((nrow >> 1) & 7);                               // logical k pair stored at physical chunk lane & 7
        bsrc[j] = B + (n0 + nrow) * ldb + 2 * q;
    }
    auto issue = [&](int st, int buf) {
        unsigned char *sa = lds + buf * DSTAGE + (4 * wave) * (GSA * 8);
        unsigned char *sb = lds + buf * DSTAGE + DSA + (32 * wave) * (GBK * 8);
        const long long k0 = (long long)st * GBK;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(asrc + (k0 + j) * lda),
                                             (__attribute__((address_space(3))) void *)(sa + j * (GSA * 8)), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bsrc[j] + k0),
                                             (__attribute__((address_space(3))) void *)(sb + j * (8 * GBK * 8)), 16, 0, 0);
    };
    // ---- consumer role ----
    int boff[4][2]; // byte offset of this lane's B fragment of tile column nt for even / odd k pair parity (kk + lk) >> 1 ...
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int nrow = wn * 64 + nt * 16 + lj;
        boff[nt][0] = nrow * (GBK * 8) + (lk & 1) * 8;      // + (((kk + lk) >> 1) ^ sw) * 16 below
        boff[nt][1] = (nrow >> 1) & 7;
    }
    const int nK = K / GBK;
    issue(0, 0);
    // the C loads must be complete before the loop (see dgemm_tile)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int sr = 0; sr < 4; ++sr) asm volatile("" : "+v"(acc[nt][mt][sr]));
    for (int it = 0; it < nK; ++it) {
        const int buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                     // stage `it` has landed for everyone; everyone is done reading the other buffer
        if (it + 1 < nK) issue(it + 1, buf ^ 1);
        const unsigned char *sa = lds + buf * DSTAGE, *sb = sa + DSA;
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 4) {
            double af[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                af[nt] = *(const double *)(sb + boff[nt][0] + ((((kk + lk) >> 1) ^ boff[nt][1]) << 4));
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                double bq[4];
#pragma unroll
                for (int sr = 0; sr < 4; ++sr) bq[sr] = *(const double *)(sa + ((kk + lk) * GSA + wm * 64 + mt * 16 + rot[sr]) * 8);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int sr = 0; sr < 4; ++sr)
                        acc[nt][mt][sr] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[nt], bq[sr], acc[nt][mt][sr], 0, 0, 1); // NEG: c - a b
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int sr = 0; sr < 4; ++sr)
                c_store(Cb + coffL[sr] + (unsigned)(nt * 16) * ldc8 + mt * 128, acc[nt][mt][sr]);
}

// MF: 0 = 16x16x4 MFMAs, classic order; 1 = 16x16x4 with per-accumulator runs; 2 = 4x4x4 MFMAs, rotations read from LDS;
//     3 = 4x4x4 MFMAs, rotations by DPP.  All four produce identical bits (contract C5).
template <int MF, bool EDGE, bool STAMP = false>
__device__ __forceinline__ void dgemm_tile_any(long long m, long long n, int K, const double *__restrict__ A, long long lda,
                                               const double *__restrict__ B, long long ldb, double *__restrict__ C, long long ldc,
                                               long long m0, long long n0, double *As, double *Bs, unsigned long long *stamps = nullptr) {
    if (MF == 0) dgemm_tile16<EDGE, 0, STAMP>(m, n, K, A, lda, B, ldb, C, ldc, m0, n0, As, Bs, stamps);
    else if (MF == 1) dgemm_tile16<EDGE, 1, STAMP>(m, n, K, A, lda, B, ldb, C, ldc, m0, n0, As, Bs, stamps);
    else if (MF == 2) dgemm_tile<EDGE, 0, STAMP>(m, n, K, A, lda, B, ldb, C, ldc, m0, n0, As, Bs, stamps);
    else dgemm_tile<EDGE, 1, STAMP>(m, n, K, A, lda, B, ldb, C, ldc, m0, n0, As, Bs, stamps);
}

template <int MF>
__global__ __launch_bounds__(256, 2) void dgemm_minus_kernel(long long m, long long n, int K, const double *__restrict__ A,
                                                             long long lda, const double *__restrict__ B, long long ldb,
                                                             double *__restrict__ C, long long ldc, int tiles_m,
                                                             int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) double g_lds[];
    double *As = g_lds;                  // [2][GBK * GSA]
    double *Bs = g_lds + 2 * GBK * GSA;  // [2][GT * GSB]
    // XCD-aware bijective remap of the linear block id (8 XCDs, round-robin dispatch)
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    // tiles are walked in groups of 8 tile-columns, tile-column fastest: the ~64 workgroups an XCD runs at a time
    // form an 8 x 8 block of tiles that shares 8 A and 8 B operand tiles (4 MB = one XCD's L2)
    const int grp = lin / (tiles_m * 8);
    const int gw = (tiles_n - grp * 8) < 8 ? (tiles_n - grp * 8) : 8;
    const int idx = lin - grp * tiles_m * 8;
    const int tm = idx / gw, tn = grp * 8 + idx % gw;
    const long long m0 = (long long)tm * GT, n0 = (long long)tn * GT;
    const bool edge = (m0 + GT > m) || (n0 + GT > n) || (K % GBK != 0);
    if (MF == 4) {
        if (edge) dgemm_tile<true, 0>(m, n, K, A, lda, B, ldb, C, ldc, m0, n0, As, Bs);
        else dgemm_dma_tile(K, A, lda, B, ldb, C, ldc, m0, n0, (unsigned char *)g_lds);
        return;
    }
    if (edge) dgemm_tile_any<MF, true>(m, n, K, A, lda, B, ldb, C, ldc, m0, n0, As, Bs);
    else dgemm_tile_any<MF, false>(m, n, K, A, lda, B, ldb, C, ldc, m0, n0, As, Bs);
}

// diagnostic build: block 0 leaves its segment cycle sums (s_memtime) and the 100 MHz real-time ticks of the whole tile
template <int MF>
__global__ __launch_bounds__(256, 2) void dgemm_minus_stamp_kernel(long long m, long long n, int K, const double *__restrict__ A,
                                                                   long long lda, const double *__restrict__ B, long long ldb,
                                                                   double *__restrict__ C, long long ldc, int tiles_m, int tiles_n,
                                                                   unsigned long long *stamps) {
    extern __shared__ __attribute__((aligned(16))) double g_lds[];
    double *As = g_lds;
    double *Bs = g_lds + 2 * GBK * GSA;
    const int bid = blockIdx.x;
    const int tm = bid % tiles_m, tn = bid / tiles_m;
    dgemm_tile_any<MF, false, true>(m, n, K, A, lda, B, ldb, C, ldc, (long long)tm * GT, (long long)tn * GT, As, Bs,
                                    bid == (tiles_m * tiles_n) / 2 ? stamps : nullptr);
}

int launch_dgemm_minus(mpf_ctx *c, int64_t m, int64_t n, int k, const double *A, int64_t lda, const double *B,
                       int64_t ldb, double *C, int64_t ldc) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const long long tm = (m + GT - 1) / GT, tn = (n + GT - 1) / GT;
    if (tm * tn > 0x7FFFFFFFll) { c->err = "dgemm: too many tiles"; return -1; }
    // The kernel addresses its operands with 32-bit byte offsets from per-tile bases (buffer loads): the A image of one
    // launch spans K * lda * 8 bytes, a B tile 128 * ldb * 8 + K * 8, a C tile 128 * ldc * 8.  Leading dimensions are
    // bounded here and K is cut into chunks that keep every offset below 2^31 -- consecutive launches continue each
    // element's fma chain with k ascending, so chunking does not change a bit (contract C5).
    if (lda > (1ll << 27) || ldb > (1ll << 20) || ldc > (1ll << 20)) { c->err = "dgemm: leading dimension too large for 32-bit tile offsets"; return -1; }
    static int mf = -1, stampmode = 0;
    const size_t lds = G_LDS_DOUBLES * sizeof(double);
    if (mf < 0) {
        const char *e = getenv("MPF_GEMM_MF");
        mf = e ? atoi(e) : 0;
        if (mf < 0 || mf > 4) mf = 0;
        e = getenv("MPF_GEMM_STAMP");
        stampmode = (e && e[0] == '1') ? 1 : 0;
#define SETA(K_) MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)K_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
        SETA(dgemm_minus_kernel<0>); SETA(dgemm_minus_kernel<1>); SETA(dgemm_minus_kernel<2>); SETA(dgemm_minus_kernel<3>); SETA(dgemm_minus_kernel<4>);
        SETA(dgemm_minus_stamp_kernel<0>); SETA(dgemm_minus_stamp_kernel<1>); SETA(dgemm_minus_stamp_kernel<2>); SETA(dgemm_minus_stamp_kernel<3>);
#undef SETA
    }
    long long kmax = ((1ll << 31) - 1) / (lda * 8) - GBK;       // (k0 + 2 i) * lda * 8 < 2^31 for every staged row
    const long long kmax_b = ((1ll << 31) - 1 - 128 * ldb * 8) / 8 - GBK;
    if (kmax_b < kmax) kmax = kmax_b;
    kmax = kmax / GBK * GBK;
    if (kmax < GBK) { c->err = "dgemm: leading dimension too large for 32-bit tile offsets"; return -1; }
    const int g = (int)(tm * tn);
    if (stampmode && m % GT == 0 && n % GT == 0 && k % GBK == 0 && k <= kmax) {
#define LS(MF_) dgemm_minus_stamp_kernel<MF_><<<g, 256, lds, c->stream>>>(m, n, k, A, lda, B, ldb, C, ldc, (int)tm, (int)tn, c->ws->hp_stamps)
        if (mf == 0) LS(0); else if (mf == 1) LS(1); else if (mf == 3) LS(3); else LS(2);
#undef LS
        MPF_HIP_TRY(c, hipGetLastError());
        return 0;
    }
    for (long long k0 = 0; k0 < k; k0 += kmax) {
        const int kc = (int)((k - k0) < kmax ? (k - k0) : kmax);
#define LG(MF_) dgemm_minus_kernel<MF_><<<g, 256, lds, c->stream>>>(m, n, kc, A + k0 * lda, lda, B + k0, ldb, C, ldc, (int)tm, (int)tn)
        const bool aligned = (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && (lda & 1) == 0 && (ldb & 1) == 0 && kc % GBK == 0;
        if (mf == 4 && aligned) LG(4);
        else if (mf == 4) LG(2);
        else if (mf == 0) LG(0); else if (mf == 1) LG(1); else if (mf == 2) LG(2); else LG(3);
#undef LG
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// dtrsm_llnu: B[m x n] := L^-1 B, L unit lower triangular m x m (m <= 256 = panel width).  Contract C4.
//
// Entirely on v_mfma_f64_16x16x4_f64: a wave owns 16 columns and keeps ALL of their finished 16-row tiles
// of X in registers (16 tiles x 4 f64).  In the MFMA's C/D layout (lane = column, register r = rows
// (lane>>4)+4r) register kk of a finished tile IS the B operand of k-step kk of a later tile product, so
// the solve never moves X between lanes or through LDS.  For tile row bi:
//     R    = B_bi - sum_{bj<bi} L[bi,bj] X_bj        (bi x 4 MFMAs, A operand = -L from an LDS image)
//     X_bi = inv(L[bi,bi]) R                         (4 MFMAs, accumulator starts at 0)
// The 16 diagonal-tile inverses are rebuilt by every workgroup in LDS (256 threads = 16 tiles x 16 columns
// of forward substitution on the identity): identical arithmetic everywhere, no extra launch.
// ---------------------------------------------------------------------------------------------
constexpr int TR_T = 16;          // tile
constexpr int TR_MAXT = 16;       // up to 256 rows

__global__ __launch_bounds__(256, 2) void dtrsm_llnu_kernel(int m, long long n, const double *__restrict__ L, long long ldl,
                                                           double *B, long long ldb) {
    __shared__ __attribute__((aligned(16))) double Linv[TR_MAXT * 256]; // [tile][k][i]: A-operand order
    __shared__ __attribute__((aligned(16))) double Lrow[256 * TR_T];    // [k 0..16*bi)[i]: -L[16bi+i][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mt = (m + TR_T - 1) / TR_T;
    const int li = lane & 15, lk = lane >> 4;

    // ---- phase 0: diagonal tiles -> LDS (into Lrow as scratch), inverses -> Linv -------------------------
    for (int e = tid; e < mt * 256; e += 256) {
        const int t = e >> 8, j = (e >> 4) & 15, i = e & 15; // element (i, j) of tile t
        const int gi = t * TR_T + i, gj = t * TR_T + j;
        Lrow[e] = (i > j && gi < m && gj < m) ? L[gi + (long long)gj * ldl] : 0.0; // strictly lower part, [t][j][i]
    }
    __syncthreads();
    {
        const int t = tid >> 4, c = tid & 15;
        if (t < mt) {
            const double *lt = Lrow + t * 256;
            double x[TR_T];
#pragma unroll
            for (int i = 0; i < TR_T; ++i) x[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int j = 0; j < TR_T; ++j)
#pragma unroll
                for (int i = j + 1; i < TR_T; ++i) x[i] = __builtin_fma(-lt[j * 16 + i], x[j], x[i]);
#pragma unroll
            for (int i = 0; i < TR_T; ++i) Linv[t * 256 + c * 16 + i] = x[i]; // inv[i][c] at [k = c][i]
        }
    }

    const long long col = (long long)blockIdx.x * 64 + wave * 16 + li;
    const bool cok = col < n;
    double *bcol = B + (cok ? col : 0) * ldb;
    d4_t X[TR_MAXT];
#pragma unroll
    for (int bi = 0; bi < TR_MAXT; ++bi) {
        if (bi < mt) {
            __syncthreads(); // previous Lrow image (or the phase-0 scratch) is no longer read
            // -L row block bi: rows 16bi..16bi+15, columns 0..16bi-1, image [k][i]
            for (int e = tid; e < bi * 256; e += 256) {
                const int i = e & 15, k = e >> 4;
                const int gi = bi * TR_T + i;
                Lrow[e] = (gi < m) ? -L[gi + (long long)k * ldl] : 0.0;
            }
            __syncthreads();
            d4_t R;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bi * TR_T + lk + 4 * r;
                R[r] = (cok && row < m) ? bcol[row] : 0.0;
            }
#pragma unroll
            for (int bj = 0; bj < TR_MAXT; ++bj) {
                if (bj < bi) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const double a = Lrow[(bj * TR_T + 4 * kk + lk) * 16 + li];
                        R = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[bj][kk], R, 0, 0, 0);
                    }
                }
            }
            d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double a = Linv[bi * 256 + (4 * kk + lk) * 16 + li];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, R[kk], acc, 0, 0, 0);
            }
            X[bi] = acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bi * TR_T + lk + 4 * r;
                if (cok && row < m) bcol[row] = acc[r];
            }
        }
    }
}

int launch_dtrsm_llnu(mpf_ctx *c, int m, int64_t n, const double *L, int64_t ldl, double *B, int64_t ldb) {
    if (m <= 0 || n <= 0) return 0;
    const long long blocks = (n + 63) / 64;
    // More than 256 rows (panels wider than 256): blocked forward substitution over 256-row blocks.  Block i first loses
    // L[i, 0:i0] X[0:i0] through the GEMM (per element the fma chain k = 0 .. i0-1 ascending), then the kernel continues the
    // same chain inside the block -- exactly the operation sequence contract C4 defines for the whole m x m triangle.
    for (int i0 = 0; i0 < m; i0 += TR_T * TR_MAXT) {
        const int mb = (m - i0) < TR_T * TR_MAXT ? (m - i0) : TR_T * TR_MAXT;
        if (i0 > 0) {
            const int rc = launch_dgemm_minus(c, mb, n, i0, L + i0, ldl, B, ldb, B + i0, ldb);
            if (rc) return rc;
        }
        dtrsm_llnu_kernel<<<(int)blocks, 256, 0, c->stream>>>(mb, n, L + i0 + (long long)i0 * ldl, ldl, B + i0, ldb);
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

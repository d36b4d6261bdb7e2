// fp64 trailing update for gfx950: replaces the cublasDtrsm call (reference MPF.cu:215-225) and the
// cublasDgemm call (MPF.cu:230-239) -- where two thirds of N^3 flops live.
//
// dgemm_minus: C[m x n] -= A[m x K] * B[K x n], all column-major.
//   * v_mfma_f64_16x16x4_f64.  The MFMA's row index i is mapped to C's COLUMN and its column index j
//     to C's ROW (D'[n][m] = sum_k B[k][n] * (-A[m][k])), so that a lane's 16-lane group walks 16
//     consecutive rows of C: the accumulator loads/stores of the in-place update are 128-byte
//     contiguous runs in column-major HBM instead of 16 different columns.
//     Issue behaviour measured on MI355X (profiles/r02_mfma_f64_issue.txt): the instruction issues every 71.5 cycles while
//     consecutive MFMAs of a wave write the same accumulator and every 138 when the accumulator changes; two waves per
//     SIMD hide each other's changes, and the K loop below runs at 71.5 cycles per MFMA per SIMD -- the pipe's own rate
//     (28.6 flop/clk/SIMD, 69-74 TFLOP/s; the 4x4x4_4b form reaches the same ceiling and was measured no faster here).
//   * 128 x 128 tile per workgroup, K stepped 16 at a time through a double-buffered LDS stage (73.7 KB => two
//     workgroups per CU, one hides the C prologue / epilogue of the other).  Full tiles: 512 threads, eight waves of
//     64 x 32 (8 accumulator tiles, 122 VGPRs => FOUR waves per SIMD): while one workgroup of the CU loads or stores
//     its C tile the other one still has two waves on every SIMD, which is what the pipe needs to stay near its rate
//     (round 2: + 4 % at K = 256, + 8 % at K = 1024 = 67 TFLOP/s, 98 % of the register-only loop; the four-wave form
//     with 64 x 64 per wave left a single wave per SIMD in those phases).  Ragged edge strips: the guarded 256-thread
//     kernel with four waves of 64 x 64.  Same fma chain per element in both.
//   * The C tile is read and written exactly once: non-temporal loads / stores keep it from displacing the operand
//     panels in L2 (+ 2-5 %).
//   * LDS images are padded for conflict-free ds_read_b64 fragments: A-tile [k][m] stride 144
//     doubles (lanes 16..31 land 128 B further in the bank row), B-tile [n][k] stride 18.
//   * Per element the update is the chain c = fma(-a_k, b_k, c), k ascending (contract C5): the
//     oracle reproduces it bit for bit, which is what keeps later fp16 pivots identical.
//   * blockIdx -> tile mapping is XCD-aware (bijective remap: the 8 XCDs each get a contiguous
//     run of tiles, tiles of a run share the same B column panel in that XCD's L2).
#include "mpf_internal.h"

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
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

constexpr int GT = 128;   // tile edge
constexpr int GBK = 16;   // K per stage.  (8 was tried so that two GEMM workgroups and a pivot workgroup of the look-ahead
                          // chain fit on one CU: the extra barriers cost the GEMM 16 %, more than the sharing it avoids.)
constexpr int GSA = 144;  // LDS stride of the A image [k][m]
constexpr int GSB = GBK + 2; // LDS stride of the B image [n][k] (18 / 10: conflict-free ds_read_b64 fragments)
constexpr int G_LDS_DOUBLES = 2 * GBK * GSA + 2 * GT * GSB;
#ifndef DMA_NEG_BLGP
#define DMA_NEG_BLGP 2   // the operand the kernels feed the A tile through is the MFMA's B-side source
#endif


// One 128 x 128 tile.  EDGE = false: the tile is interior and K is a multiple of GBK (no guards).
// NT = 16-column accumulator tiles per wave: 4 -> four waves of 64 x 64 (256 threads), 2 -> eight waves of 64 x 32
// (512 threads, <= 128 VGPRs: four waves per SIMD with two workgroups per CU).
// All global addresses are a wave-uniform 64-bit base plus a 32-bit per-lane byte offset.
template <bool EDGE, int NT>
__device__ __forceinline__ void dgemm_tile(long long m, long long n, int K, const double *__restrict__ A, long long lda,
                                           const double *__restrict__ B, long long ldb, double *__restrict__ C,
                                           long long ldc, long long m0, long long n0, double *As, double *Bs) {
    constexpr int THREADS = 64 * 2 * (GT / (16 * NT));
    constexpr int EPT = GBK * GT / THREADS;    // staged elements per thread and operand
    constexpr int KSTEP = THREADS / GT;        // A image: k rows covered by one pass of the threads
    constexpr int NSTEP = THREADS / GBK;       // B image: columns covered by one pass of the threads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int lj = lane & 15, lk = lane >> 4;
    const int mrem = (int)((m - m0) < GT ? (m - m0) : GT), nrem = (int)((n - n0) < GT ? (n - n0) : GT);

    // ---- accumulators <- C tile: register rr of tile (nt, mt) is C[.. + lj, .. + lk + 4 rr] --------
    char *Cb = (char *)(C + m0 + n0 * ldc);
    const int crow = wm * 64 + lj, ccol = wn * 16 * NT + lk;
    const unsigned ldc8 = (unsigned)ldc * 8u;
    d4_t acc[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int col = ccol + nt * 16 + 4 * rr;
            const unsigned coff = (unsigned)col * ldc8 + (unsigned)crow * 8u;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (!EDGE || (crow + mt * 16 < mrem && col < nrem))
                    acc[nt][mt][rr] = __builtin_nontemporal_load((const double *)(Cb + coff + mt * 128));
                else
                    acc[nt][mt][rr] = 0.0;
            }
        }

    // ---- staging: thread loads EPT + EPT doubles per K stage ----------------------------------------
    const int mA = tid & (GT - 1), kA0 = tid / GT;   // A image element i: (k = kA0 + KSTEP i, m = mA)
    const int kB = tid & (GBK - 1), nB0 = tid / GBK; // B image element i: (n = nB0 + NSTEP i, k = kB)
    const unsigned lda8 = (unsigned)lda * 8u, ldb8 = (unsigned)ldb * 8u;
    const unsigned offA0 = (unsigned)mA * 8u + (unsigned)kA0 * lda8;
    const unsigned offB0 = (unsigned)kB * 8u + (unsigned)nB0 * ldb8;
    double ra[EPT], rb[EPT];
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(A + m0), rB = make_rsrc(B + n0 * ldb);
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            if (!EDGE || (mA < mrem && k0 + kA0 + KSTEP * i < K)) ra[i] = buf_load_f64(rA, offA0, (unsigned)(k0 + KSTEP * i) * lda8);
            else ra[i] = 0.0;
            if (!EDGE || (nB0 + NSTEP * i < nrem && k0 + kB < K)) rb[i] = buf_load_f64(rB, offB0, (unsigned)k0 * 8u + (unsigned)(NSTEP * i) * ldb8);
            else rb[i] = 0.0;
        }
    };
    auto sstore = [&](int buf) {
        double *as = As + buf * GBK * GSA + kA0 * GSA + mA;
        double *bs = Bs + buf * GT * GSB + nB0 * GSB + kB;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            as[KSTEP * i * GSA] = -ra[i]; // negate here, not at the load: the loads must not be waited for before the MFMAs
            bs[NSTEP * i * GSB] = rb[i];
        }
    };

    const int nK = (K + GBK - 1) / GBK;
    gload(0);
    sstore(0);
    __syncthreads();
    // Pin the C loads as COMPLETE before the K loop.  Otherwise the compiler leaves a few of them in flight into the
    // loop and guards the MFMAs that consume them with s_waitcnt vmcnt(3..0) -- in the shared loop body, i.e. in EVERY
    // iteration, where those waits also drain the staging loads issued at the top of the same iteration.
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) asm volatile("" : "+v"(acc[nt][mt]));
    for (int it = 0; it < nK; ++it) {
        const int buf = it & 1;
        if (it + 1 < nK) gload((it + 1) * GBK);
        const double *as = As + buf * GBK * GSA + wm * 64 + lj;
        const double *bs = Bs + buf * GT * GSB + (wn * 16 * NT + lj) * GSB;
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 4) {
            double af[NT], bf[4];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) af[nt] = bs[nt * 16 * GSB + kk + lk];          // B[k][n]
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) bf[mt] = as[(kk + lk) * GSA + mt * 16];        // -A[m][k]
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[nt], bf[mt], acc[nt][mt], 0, 0, 0);
        }
        if (it + 1 < nK) sstore(buf ^ 1);
        __syncthreads();
    }

    // ---- C tile <- accumulators ----------------------------------------------------------------------
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int col = ccol + nt * 16 + 4 * rr;
            const unsigned coff = (unsigned)col * ldc8 + (unsigned)crow * 8u;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                if (!EDGE || (crow + mt * 16 < mrem && col < nrem)) __builtin_nontemporal_store(acc[nt][mt][rr], (double *)(Cb + coff + mt * 128));
        }
}

// XCD-aware bijective remap of the linear block id (8 XCDs, round-robin dispatch); tiles are walked in groups of 8
// tile-columns, tile-column fastest: the ~64 workgroups an XCD runs at a time form an 8 x 8 block of tiles that shares
// 8 A and 8 B operand tiles (4 MB = one XCD's L2)
__device__ __forceinline__ void dgemm_tile_of_block(int tiles_m, int tiles_n, int &tm, int &tn) {
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int grp = lin / (tiles_m * 8);
    const int gw = (tiles_n - grp * 8) < 8 ? (tiles_n - grp * 8) : 8;
    const int idx = lin - grp * tiles_m * 8;
    tm = idx / gw; tn = grp * 8 + idx % gw;
}

template <int NT>
__device__ __forceinline__ void dgemm_body(long long m, long long n, int K, const double *__restrict__ A, long long lda,
                                           const double *__restrict__ B, long long ldb, double *__restrict__ C, long long ldc,
                                           int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) double g_lds[];
    double *As = g_lds;                  // [2][GBK * GSA]
    double *Bs = g_lds + 2 * GBK * GSA;  // [2][GT * GSB]
    int tm, tn;
    dgemm_tile_of_block(tiles_m, tiles_n, tm, tn);
    const long long m0 = (long long)tm * GT, n0 = (long long)tn * GT;
    const bool edge = (m0 + GT > m) || (n0 + GT > n) || (K % GBK != 0);
    if (edge) dgemm_tile<true, NT>(m, n, K, A, lda, B, ldb, C, ldc, m0, n0, As, Bs);
    else dgemm_tile<false, NT>(m, n, K, A, lda, B, ldb, C, ldc, m0, n0, As, Bs);
}

__global__ __launch_bounds__(256, 2) void dgemm_minus_kernel(long long m, long long n, int K, const double *__restrict__ A,
                                                             long long lda, const double *__restrict__ B, long long ldb,
                                                             double *__restrict__ C, long long ldc, int tiles_m,
                                                             int tiles_n) {
    dgemm_body<4>(m, n, K, A, lda, B, ldb, C, ldc, tiles_m, tiles_n);
}

// eight waves per workgroup: four waves per SIMD, so two of them keep the pipe at its rate while the other workgroup of
// the CU is in its C prologue / epilogue (a single wave cycling through its accumulators issues at half rate).
// Interior tiles only (m, n multiples of 128, K of 16): the guards of the edge path cost the one register too many.
__global__ __launch_bounds__(512, 2) void dgemm_minus_kernel8(long long m, long long n, int K, const double *__restrict__ A,
                                                              long long lda, const double *__restrict__ B, long long ldb,
                                                              double *__restrict__ C, long long ldc, int tiles_m,
                                                              int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) double g_lds[];
    int tm, tn;
    dgemm_tile_of_block(tiles_m, tiles_n, tm, tn);
    dgemm_tile<false, 2>(m, n, K, A, lda, B, ldb, C, ldc, (long long)tm * GT, (long long)tn * GT, g_lds, g_lds + 2 * GBK * GSA);
}

// ---- LDS-DMA form of the eight-wave tile ---------------------------------------------------------------------------
// The operand slices go global -> LDS directly (global_load_lds_dwordx4): no staging registers, no LDS-write pass, and the
// negation of A moves into the MFMA (BLGP bit 0 negates the A-side operand: exact, the fma chain per element is unchanged).
// A image [k][m]: one wave instruction = one k-row (128 doubles = 64 lanes x 16 B), rows padded to DSA doubles.
// B image [n][k]: BK doubles per column, 16-byte chunks XOR-swizzled by the column so that the 16 columns of a fragment read
// fall on different banks; one wave instruction = 64 / CH columns.  NS ring stages of BK k-steps each.
// RUN = k4-steps issued back to back on the same accumulator.  Measured (m = n = 28672, K = 256 / 1024): <16, 2, 1> 65.5 / 71.4
// TFLOP/s (register-staged form: 61.4 / 67.4); RUN = 2 the same; four stages of 8 (<8, 4, 1>) 64.0 / 69.8; 128 x 64 tiles with
// four waves and four workgroups per CU 60.7 / 69.0.  Only <16, 2, 1> is instantiated.
template <int BK, int NS, int RUN>
struct DmaCfg {
    static constexpr int DSA = GT + 16;              // A row stride (doubles)
    static constexpr int CH = BK / 2;                // 16-byte chunks per B column
    static constexpr int A_DBL = BK * DSA, B_DBL = GT * BK;
    static constexpr int STAGE_DBL = A_DBL + B_DBL;
    static constexpr int LDS_BYTES = NS * STAGE_DBL * 8;
    static constexpr int SW_SHIFT = BK == 16 ? 1 : 2; // rows per 256 bytes: 2 (BK = 16) or 4 (BK = 8)
};

template <int BK, int NS, int RUN>
__global__ __launch_bounds__(512, 2) void dgemm_minus_kernel8d(long long m, long long n, int K, const double *__restrict__ A,
                                                               long long lda, const double *__restrict__ B, long long ldb,
                                                               double *__restrict__ C, long long ldc, int tiles_m, int tiles_n) {
    using Cfg = DmaCfg<BK, NS, RUN>;
    extern __shared__ __attribute__((aligned(16))) double g_lds[];
    int tm, tn;
    dgemm_tile_of_block(tiles_m, tiles_n, tm, tn);
    const long long m0 = (long long)tm * GT, n0 = (long long)tn * GT;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int lj = lane & 15, lk = lane >> 4;

    // ---- accumulators <- C tile ------------------------------------------------------------------------------------
    // MFMA column index lj of accumulator tile mt is C row wm * 64 + (mt >> 1) * 32 + 2 * lj + (mt & 1): a lane owns two
    // adjacent rows per tile pair, so the C tile moves in 16-byte pieces (256 contiguous bytes per 16 lanes) and the A
    // fragments of a tile pair are one ds_read_b128.
    char *Cb = (char *)(C + m0 + n0 * ldc);
    const int crow = wm * 64 + 2 * lj, ccol = wn * 32 + lk;
    const unsigned ldc8 = (unsigned)ldc * 8u;
    d4_t acc[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const unsigned coff = (unsigned)(ccol + nt * 16 + 4 * rr) * ldc8 + (unsigned)crow * 8u;
#pragma unroll
            for (int mp = 0; mp < 2; ++mp) {
                const d2_t v = __builtin_nontemporal_load((const d2_t *)(Cb + coff + mp * 256));
                acc[nt][2 * mp][rr] = v[0];
                acc[nt][2 * mp + 1][rr] = v[1];
            }
        }

    // ---- loader role: per stage the wave issues AI + BI instructions -----------------------------------------------------
    constexpr int AI = BK / 8;                     // A rows per wave and stage (BK rows over 8 waves)
    constexpr int BCOLS = 64 / Cfg::CH;            // B columns per instruction
    constexpr int BI = GT / BCOLS / 8;             // B instructions per wave and stage
    const double *ga[AI], *gb[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) ga[i] = A + m0 + (long long)(wave * AI + i) * lda + lane * 2;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int col = (wave * BI + i) * BCOLS + lane / Cfg::CH, p = lane % Cfg::CH;
        const int q = p ^ ((col >> Cfg::SW_SHIFT) & (Cfg::CH - 1));   // logical chunk stored at physical position p
        gb[i] = B + (n0 + col) * ldb + q * 2;
    }
    auto issue = [&](int s) {
        double *st = g_lds + (s % NS) * Cfg::STAGE_DBL;
        const long long k0 = (long long)s * BK;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ga[i] + k0 * lda),
                                             (__attribute__((address_space(3))) void *)(st + (wave * AI + i) * Cfg::DSA), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BI; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gb[i] + k0),
                                             (__attribute__((address_space(3))) void *)(st + Cfg::A_DBL + (wave * BI + i) * BCOLS * BK), 16, 0, 0);
    };
    constexpr int LPS = AI + BI;                   // loads per wave and stage

    // ---- consumer role ------------------------------------------------------------------------------------------------
    // fragment addresses inside a stage (doubles): A: (kk + lk) * DSA + wm * 64 + mt * 16 + lj;  B: column nn = wn * 32 + nt * 16 + lj
    int boff[2], bsw[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int nn = wn * 32 + nt * 16 + lj;
        boff[nt] = nn * BK;
        bsw[nt] = (nn >> Cfg::SW_SHIFT) & (Cfg::CH - 1);
    }
    const int nst = K / BK;
    for (int s = 0; s < NS - 1 && s < nst; ++s) issue(s);
    // the C loads are older than every operand load: waiting for the first stage below completes them too
    for (int i = 0; i < nst; ++i) {
        if (i + NS - 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // stage i is in LDS for everyone; everyone is done with stage i - 1
        if (i + NS - 1 < nst) issue(i + NS - 1);
        const double *as = g_lds + (i % NS) * Cfg::STAGE_DBL + wm * 64 + 2 * lj;
        const double *bs = as - (wm * 64 + 2 * lj) + Cfg::A_DBL;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4 * RUN) {
            double af[RUN][2], bf[RUN][4];
#pragma unroll
            for (int r = 0; r < RUN; ++r) {
                const int k = kk + 4 * r + lk;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) af[r][nt] = bs[boff[nt] + (((k >> 1) ^ bsw[nt]) << 1) + (k & 1)];   // B[k][n]
#pragma unroll
                for (int mp = 0; mp < 2; ++mp) {                                                                      // A[m][k], two rows
                    const d2_t v = *(const d2_t *)(as + k * Cfg::DSA + mp * 32);
                    bf[r][2 * mp] = v[0];
                    bf[r][2 * mp + 1] = v[1];
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < RUN; ++r)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[r][nt], bf[r][mt], acc[nt][mt], 0, 0, DMA_NEG_BLGP);
        }
    }

    // ---- C tile <- accumulators ------------------------------------------------------------------------------------------
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const unsigned coff = (unsigned)(ccol + nt * 16 + 4 * rr) * ldc8 + (unsigned)crow * 8u;
#pragma unroll
            for (int mp = 0; mp < 2; ++mp) {
                const d2_t v = {acc[nt][2 * mp][rr], acc[nt][2 * mp + 1][rr]};
                __builtin_nontemporal_store(v, (d2_t *)(Cb + coff + mp * 256));
            }
        }
}

int launch_dgemm_minus(mpf_ctx *c, int64_t m, int64_t n, int k, const double *A, int64_t lda, const double *B,
                       int64_t ldb, double *C, int64_t ldc) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    if (((m + GT - 1) / GT) * ((n + GT - 1) / GT) > 0x7FFFFFFFll) { c->err = "dgemm: too many tiles"; return -1; }
    // The kernels address their operands with 32-bit byte offsets from per-tile bases (buffer loads): the A image of one
    // launch spans K * lda * 8 bytes, a B tile 128 * ldb * 8 + K * 8, a C tile 128 * ldc * 8.  Leading dimensions are
    // bounded here and K is cut into chunks that keep every offset below 2^31 -- consecutive launches continue each
    // element's fma chain with k ascending, so chunking does not change a bit (contract C5).
    if (lda > (1ll << 27) || ldb > (1ll << 20) || ldc > (1ll << 20)) { c->err = "dgemm: leading dimension too large for 32-bit tile offsets"; return -1; }
    size_t lds = G_LDS_DOUBLES * sizeof(double);
#ifdef MPF_PROBE
    const int w8 = c->tune.dgemm_w8;             // probe library: 0 = the four-wave kernel everywhere (A/B switch)
    lds += (size_t)c->tune.gemm_lds_pad;         // probe library: extra dynamic LDS to force fewer workgroups per CU
#else
    const int w8 = 1;
#endif
    const int dma = c->tune.dgemm_dma;           // 0: register-staged eight-wave kernel (A/B switch, same bits)
    if (!(c->attr_done & ATTR_DGEMM)) {          // per context => per device (mpf_create(&c, device) allows several)
        MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)dgemm_minus_kernel8d<16, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, DmaCfg<16, 2, 1>::LDS_BYTES));
        MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)dgemm_minus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)dgemm_minus_kernel8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->attr_done |= ATTR_DGEMM;
    }
    long long kmax = ((1ll << 31) - 1) / (lda * 8) - GBK;       // (k0 + i) * lda * 8 < 2^31 for every staged row
    const long long kmax_b = ((1ll << 31) - 1 - 128 * ldb * 8) / 8 - GBK;
    if (kmax_b < kmax) kmax = kmax_b;
    kmax = kmax / GBK * GBK;
    if (kmax < GBK) { c->err = "dgemm: leading dimension too large for 32-bit tile offsets"; return -1; }
    auto four = [&](int64_t mm, int64_t nn, int kc, const double *a, const double *b, double *cc) {
        const long long tm = (mm + GT - 1) / GT, tn = (nn + GT - 1) / GT;
        dgemm_minus_kernel<<<(int)(tm * tn), 256, lds, c->stream>>>(mm, nn, kc, a, lda, b, ldb, cc, ldc, (int)tm, (int)tn);
    };
    // full 128 x 128 tiles go to the eight-wave kernel, the ragged right / bottom strips to the guarded four-wave one
    const int64_t mi = w8 ? m / GT * GT : 0, ni = w8 ? n / GT * GT : 0;
    for (long long k0 = 0; k0 < k; k0 += kmax) {
        const int kc = (int)((k - k0) < kmax ? (k - k0) : kmax);
        const double *a = A + k0 * lda, *b = B + k0;
        const bool dma_ok = dma && ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && lda % 2 == 0 && ldb % 2 == 0;
        if (mi > 0 && ni > 0 && kc % GBK == 0) {
            const int g = (int)((mi / GT) * (ni / GT)), tmi = (int)(mi / GT), tni = (int)(ni / GT);
            if (dma_ok && dma == 1) dgemm_minus_kernel8d<16, 2, 1><<<g, 512, DmaCfg<16, 2, 1>::LDS_BYTES, c->stream>>>(mi, ni, kc, a, lda, b, ldb, C, ldc, tmi, tni);
            else
            dgemm_minus_kernel8<<<(int)((mi / GT) * (ni / GT)), 512, lds, c->stream>>>(mi, ni, kc, a, lda, b, ldb, C, ldc, (int)(mi / GT), (int)(ni / GT));
            if (m > mi) four(m - mi, n, kc, a + mi, b, C + mi);                    // bottom strip, all columns
            if (n > ni) four(mi, n - ni, kc, a, b + ni * ldb, C + ni * ldc);      // right strip above it
        } else {
            four(m, n, kc, a, b, C);
        }
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// dtrsm_llnu: B[m x n] := L^-1 B, L unit lower triangular m x m (m <= 256 = panel width).  Contract C4.
//
// Entirely on v_mfma_f64_16x16x4_f64, right-looking over the 16-row tile rows.  A workgroup owns CG groups of 16 columns; its
// four waves SPLIT THE TILE ROWS (wave w keeps the right-hand sides of tile rows w, w + 4, w + 8, w + 12 in registers, in the
// MFMA's C/D layout: lane = column, register r = rows (lane >> 4) + 4r).  Step bj:
//     owner wave (bj & 3):   X_bj = inv(L[bj,bj]) R_bj        (4 MFMAs per column group, accumulator starts at 0; stored to B)
//     X_bj goes through a lane-private LDS slot to the other waves: register kk of a finished tile IS the B operand of k-step kk
//     every wave:            R_bi = R_bi - L[bi,bj] X_bj       for its tile rows bi > bj (4 MFMAs each, nearest tile row first;
//                                                               the A operand comes straight from global memory, one step ahead,
//                                                               negated by the MFMA's operand modifier)
// Per element the operations and their order are those of the left-looking form (off-diagonal products k ascending, then the
// inverse applied k ascending from 0): identical bits.  What the split buys is the critical path: 16 x (8 dependent MFMAs + one
// LDS hand-off) instead of one wave's 544 dependent MFMAs and 30 barriers -- the next panel's strip (256 columns) took 43 us in
// the one-wave-per-16-columns form whatever its width.
// The 16 diagonal-tile inverses are rebuilt by every workgroup (each wave its own four, forward substitution on the identity
// in LDS): identical arithmetic everywhere, no extra launch.
// ---------------------------------------------------------------------------------------------
constexpr int TR_T = 16;          // tile
constexpr int TR_MAXT = 16;       // up to 256 rows
// 16 loaded values pass through one empty asm: the loads before it stay unconditional (the compiler would otherwise sink each
// one into the branch of the select that consumes it, with a wait of its own)
#define KEEP16(v)                                                                                                              \
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), \
                 "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]))

// Right-hand side element (row, col) at B[row * rs + col * cs]: (rs, cs) = (1, ldb) for the column-major matrix, (ld, 1) for a
// row-major working copy -- same arithmetic, only addresses differ.
template <int CG>
__global__ __launch_bounds__(256, 2) void dtrsm_llnu_kernel(int m, long long n, const double *__restrict__ L, long long ldl,
                                                                        double *B, long long rs, long long cs) {
    // wave w's four diagonal tiles, then their inverses (A-operand order), at sh[w * 1024 ...]; X slots [2][CG][64 lanes] behind
    __shared__ __attribute__((aligned(16))) double sh[4 * 4 * 256 + 2 * CG * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int mt = (m + TR_T - 1) / TR_T;
    const int li = lane & 15, lk = lane >> 4;

    long long coff[CG];
    bool cok[CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) {
        const long long col = ((long long)blockIdx.x * CG + g) * 16 + li;
        cok[g] = col < n;
        coff[g] = (cok[g] ? col : 0) * cs;
    }
    // the wave's right-hand sides go into registers up front: one memory latency for the kernel
    d4_t R[4][CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (w + 4 * (q >> 2)) * TR_T + lk + 4 * (q & 3);
            v[q] = B[(row < m ? row : m - 1) * rs + coff[g]];   // always a valid address ...
        }
        KEEP16(v);                                              // ... and no branch (with its own wait) around each load
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (w + 4 * (q >> 2)) * TR_T + lk + 4 * (q & 3);
            R[q >> 2][g][q & 3] = (row < m) ? v[q] : 0.0;
        }
    }
    // A operands (raw L; the MFMA negates) of step bj for the wave's tile rows, fetched one step ahead
    auto load_l = [&](int bj, double (&f)[4][4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int bi = w + 4 * t, gi = bi * TR_T + li, gk = bj * TR_T + 4 * kk + lk;
                f[t][kk] = (bi > bj && gi < m) ? L[gi + (long long)gk * ldl] : 0.0;
            }
    };
    double lf[4][4];
    if (mt > 1) load_l(0, lf);
    // ---- phase 0: the wave's diagonal tiles -> LDS, inverses by forward substitution on the identity -----------------------
    double *shw = sh + w * 1024;
    {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = lane + 64 * q, tt = e >> 8, j = (e >> 4) & 15, i = e & 15; // element (i, j) of the wave's tile tt
            const int gi = (w + 4 * tt) * TR_T + i, gj = (w + 4 * tt) * TR_T + j;
            v[q] = L[(gi < m ? gi : m - 1) + (long long)(gj < m ? gj : m - 1) * ldl];
        }
        KEEP16(v);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = lane + 64 * q, tt = e >> 8, j = (e >> 4) & 15, i = e & 15;
            const int gi = (w + 4 * tt) * TR_T + i, gj = (w + 4 * tt) * TR_T + j;
            shw[e] = (i > j && gi < m && gj < m) ? v[q] : 0.0; // strictly lower part, [tt][j][i]
        }
    }
    __syncthreads();
    {
        const int tt = lane >> 4, c = lane & 15;
        const double *lt = shw + tt * 256;
        double x[TR_T];
#pragma unroll
        for (int i = 0; i < TR_T; ++i) x[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < TR_T; ++j)
#pragma unroll
            for (int i = j + 1; i < TR_T; ++i) {
                x[i] = __builtin_fma(-lt[j * 16 + i], x[j], x[i]);
                if (i == TR_T - 1) __builtin_amdgcn_sched_barrier(0); // keep the LDS reads of later columns out of the registers
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TR_T; ++i) shw[tt * 256 + c * 16 + i] = x[i]; // inv[i][c] at [k = c][i]: A-operand order
    }
    __syncthreads();
    d4_t *slot = reinterpret_cast<d4_t *>(sh + 4 * 4 * 256);

#pragma unroll
    for (int bj = 0; bj < TR_MAXT; ++bj) {
        if (bj < mt) {
            const int to = bj >> 2;
            const bool owner = (w == (bj & 3));
            d4_t xg[CG];
            if (owner) {
                double inv[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) inv[kk] = shw[to * 256 + (4 * kk + lk) * 16 + li];
#pragma unroll
                for (int g = 0; g < CG; ++g) {
                    d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(inv[kk], R[to][g][kk], acc, 0, 0, 0);
                    xg[g] = acc;
                    R[to][g] = acc;   // the finished tile stays here until the store at the end
                }
                if (bj + 1 < mt) {
#pragma unroll
                    for (int g = 0; g < CG; ++g) slot[((bj & 1) * CG + g) * 64 + lane] = xg[g];
                }
            }
            if (bj + 1 < mt) {
                // X_bj is in its slot (the slot of step bj - 2 was read before the barrier of step bj - 1).  A bare barrier: the
                // fence of __syncthreads would also wait for the L fragments in flight, one L2 latency per step on the critical path
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (!owner) {
#pragma unroll
                    for (int g = 0; g < CG; ++g) xg[g] = slot[((bj & 1) * CG + g) * 64 + lane];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (w + 4 * t > bj) {
#pragma unroll
                        for (int g = 0; g < CG; ++g)
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk)
                                R[t][g] = __builtin_amdgcn_mfma_f64_16x16x4f64(lf[t][kk], xg[g][kk], R[t][g], 0, 0, 1 /* -L */);
                    }
                }
                if (bj + 2 < mt) load_l(bj + 1, lf); // in flight while the next owner applies its inverse
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < CG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (w + 4 * t) * TR_T + lk + 4 * r;
                if (cok[g] && row < m) B[row * rs + coff[g]] = R[t][g][r];
            }
}

// Narrow right-hand sides (the next panel's strip, the chain-bound end of a factorization) take one column group per workgroup:
// the launch is latency-bound and 16 x more workgroups cost nothing; wide ones take four, so that an L fragment fetched from L2
// feeds four MFMAs.
static void dtrsm_launch(mpf_ctx *c, int m, int64_t n, const double *L, int64_t ldl, double *B, int64_t rs, int64_t cs) {
    if (n <= 12288) dtrsm_llnu_kernel<1><<<(int)((n + 15) / 16), 256, 0, c->stream>>>(m, n, L, ldl, B, rs, cs);
    else dtrsm_llnu_kernel<4><<<(int)((n + 63) / 64), 256, 0, c->stream>>>(m, n, L, ldl, B, rs, cs);
}

int launch_dtrsm_llnu(mpf_ctx *c, int m, int64_t n, const double *L, int64_t ldl, double *B, int64_t ldb) {
    if (m <= 0 || n <= 0) return 0;
    // More than 256 rows (panels wider than 256): blocked forward substitution over 256-row blocks.  Block i first loses
    // L[i, 0:i0] X[0:i0] through the GEMM (per element the fma chain k = 0 .. i0-1 ascending), then the kernel continues the
    // same chain inside the block -- exactly the operation sequence contract C4 defines for the whole m x m triangle.
    for (int i0 = 0; i0 < m; i0 += TR_T * TR_MAXT) {
        const int mb = (m - i0) < TR_T * TR_MAXT ? (m - i0) : TR_T * TR_MAXT;
        if (i0 > 0) {
            const int rc = launch_dgemm_minus(c, mb, n, i0, L + i0, ldl, B, ldb, B + i0, ldb);
            if (rc) return rc;
        }
        dtrsm_launch(c, mb, n, L + i0 + (long long)i0 * ldl, ldl, B + i0, 1, ldb);
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
int launch_dtrsm_llnu_strided(mpf_ctx *c, int m, int64_t n, const double *L, int64_t ldl, double *B, int64_t rs, int64_t cs) {
    if (m <= 0 || n <= 0) return 0;
    if (m > TR_T * TR_MAXT) { c->err = "dtrsm (strided): at most 256 rows"; return -1; }
    dtrsm_launch(c, m, n, L, ldl, B, rs, cs);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

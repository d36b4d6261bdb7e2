// fp16-in / fp32-accumulate trailing update on v_mfma_f32_16x16x32_f16 (round 5; the reference's update is the cublasDgemm call
// at MPF.cu:230-239, contract C6).  Plain (un-split) operands only; the split-operand ("fp16x3") kernels stay in trailing_f16.hip.
//
//   C (fp32 working copy or the fp64 matrix, column-major m x n, leading dimension ldc)  -=  Lh[m][K] * Uh[n][K]^T
//
// Why the 16x16x32 shape: the same flops per cycle as 32x32x16, but the chip holds a higher clock under it (register-only loop
// 2.10 against 1.74 PFLOP/s); and a 16 x 16 accumulator tile leaves the mapping of MFMA columns to matrix rows free
// in steps of 16, which the C stream uses:
//   * operand rows are PERMUTED on their way into LDS (a global_load_lds piece takes a per-lane source address, so the
//     permutation costs nothing): LDS row 16 q + c of a group of 64 holds image row 4 c + q.  The MFMA of sub-tile q then
//     leaves lane c with matrix row 4 c + q, the four sub-tiles of a group give every lane FOUR CONSECUTIVE rows, and the fp32
//     block moves as dwordx4 accesses: 16 lanes x 16 B = 256-byte runs, four runs (columns) per access, 32 loads + 32 stores
//     per wave block instead of 128 + 128 dword accesses;
//   * fragment reads: lane (c = lane & 15, g = lane >> 4) reads row c of its sub-tile, 16-byte chunk g ^ swz(c) of the row's
//     64 bytes (32 k), swz(c) = (-(c >> 2)) & 3: each of the four 16-lane groups a ds_read_b128 is serviced in ({0-3, 12-15,
//     20-27}, {4-11, 16-19, 28-31}, + 32) then covers all 64 banks once (MI355X_MICROARCH.md, LDS).
// One k-step of the MFMA is a whole 32-k stage.  A wave's 128 x 64 block = 4 (U side) x 8 (L side) sub-tiles = 32 MFMAs per
// stage; the L fragments are single-buffered and re-read for the next stage right after their last use (pass j of a stage
// uses b[j] with all four U fragments), the U fragments double-buffered: 64 fragment registers beside 128 accumulators.
// Per output element: ONE fp32 accumulation chain over k ascending in steps of 32, one subtraction -- every kernel of this file
// produces the same bits for the same operands (the multi-rank runs rely on it).
#include "mpf_internal.h"
#include <cstdlib>

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));

#ifndef MPF_C_AUX
#define MPF_C_AUX 2
#endif

namespace {
constexpr int H_AUX = MPF_C_AUX;   // cache policy of the streamed C block (2 = nt: the operand images keep the L2)
constexpr int H_RB = 64;           // bytes an operand row contributes to a stage (32 k)

__device__ __forceinline__ int h_swz(int pos) { return (0 - (pos >> 2)) & 3; }
// LDS row position -> image row inside groups of 16 P rows: position 16 q + c holds row P c + q (q < P)
template <int P> __device__ __forceinline__ int h_perm(int pos) {
    if (P == 1) return pos;
    constexpr int G = 16 * P;
    const int w = pos & (G - 1);
    return (pos & ~(G - 1)) + (w & 15) * P + (w >> 4);
}
__device__ __forceinline__ f4_t mfma16(h8_t a, h8_t b, f4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// One operand piece, global -> LDS (64 lanes x 16 B = 1 KB at LDS byte address `lds`, lane l at + 16 l), source = base + the
// lane's 32-bit byte offset.  Written as inline asm ON PURPOSE: with the builtin (__builtin_amdgcn_global_load_lds) in a loop the
// compiler gives up counting LDS reads and puts s_waitcnt lgkmcnt(0) in front of every first use -- here that would wait for
// the fragment reads issued at the end of the previous stage; as asm the reads are counted (lgkmcnt(N)).  The pieces' own
// completion is waited for by hand (vmcnt) in top_of_stage.  M0 is used by nothing else in these kernels.
__device__ __forceinline__ void lds_dma16(const void *base, unsigned voff, unsigned lds) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory");
#endif
}
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)p; }
}  // namespace

// ---- big-K update: 256 x 256 tile, eight waves of 128 (L side) x 64 (U side), ring of four 32-KB stages -----------------------------
// C32: the updated block is the fp32 working copy (L rows permuted in groups of 64, dwordx4 C accesses); otherwise the fp64
// matrix (no permutation: lane c = row c of its sub-tile, 8-byte accesses in 128-byte runs).
// One tile per workgroup.  A ring of five stages with sixteen pieces of the C block staged global -> LDS into the ring slots the last
// stages free was built and measured too (round 5): the same rate within the boxes' spread as a PROBE build, 2-4 % slower as the
// product build (register allocation: two spilled registers) -- this simpler form stays.  A persistent form (one workgroup per CU walking its tiles, the next tile's first stages requested before
// the C stream) was built and measured in round 5 and is gone again: 732 against 883 TFLOP/s at K = 1024.  A workgroup that lives on
// has its C stores in its vmcnt counter, the stores of a tile take ~9 us to be acknowledged while HBM is saturated (stamped), and
// the next tile's first counted wait for operand pieces waits for them; a workgroup that ends does not.
// DBG (probe library only; the product instantiates 0): 1 = K loop only, 2 = C stream only, 3 = everything but the C stores, 5 = everything
// but the C loads (what each direction of the C stream costs beside the K loop), 4 = everything with in-kernel stamps,
// 12 = 4 + the epilogue stamp waits for the stores' acknowledgement.  A compile-time parameter: as a run-time flag the probe build's
// register allocation differed from the product's and its timings were not the product kernel's (round 5: 887 against 848 TFLOP/s).
template <bool C32, int DBG = 0>
__global__ __launch_bounds__(512, 1) void hgemm16_big_kernel(long long m, long long n, int Kp, const unsigned short *__restrict__ Lh,
                                                             const unsigned short *__restrict__ Uh, void *__restrict__ Cv, long long ldc,
                                                             int tiles_m, int tiles_n, int ksL, int ksU, unsigned long long *stamps) {
    constexpr int TM = 256, TN = 256, NS = 4, LPS = 4;
    constexpr int UARR = TN * H_RB, LARR = TM * H_RB, STAGE = UARR + LARR;
    constexpr int PL = C32 ? 4 : 1;
    constexpr int GW = 4;                      // tile-columns walked together: an XCD's run of tiles shares few operand rows
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    constexpr bool NOST = (DBG & 7) == 3, NOLD = (DBG & 7) == 5;
    constexpr int dbg = ((DBG & 7) == 4 || NOST || NOLD) ? 0 : (DBG & 7);
    constexpr bool STAMP = DBG != 0 && !NOST && !NOLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g4 = lane >> 4;
    // ---- the workgroup's tile: XCD x (= workgroup index mod 8) owns a contiguous chunk of the tile sequence, walked in groups of GW
    //      tile-columns ------------------------------------------------------------------------------------------------------------------
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int grp = lin / (tiles_m * GW);
    const int gw = (tiles_n - grp * GW) < GW ? (tiles_n - grp * GW) : GW;
    const int idx = lin - grp * tiles_m * GW;
    const long long m0t = (long long)(idx / gw) * TM, n0t = (long long)(grp * GW + idx % gw) * TN;

    // ---- loader: piece pi of a stage = 16 LDS rows of one side; wave w moves pieces w, w + 8 (U rows), w + 16, w + 24 (L rows) ------
    const int lr = lane >> 2, pc = lane & 3;   // row within a piece, physical chunk
    unsigned goff[LPS];                        // the lane's byte offset inside its image at k = 0 (< 4 GB: checked by the launcher)
    unsigned ldst[LPS];
    const unsigned ring0 = lds_addr(ring);
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const bool lside = i >= 2;
        const int prow = ((wave + 8 * i) & 15) * 16, trow = prow + lr;
        const int irow = lside ? h_perm<PL>(trow) : trow;
        const long long grow = (lside ? m0t : n0t) + irow, lim = lside ? m : n;
        const int cch = pc ^ h_swz(trow);      // logical chunk stored at physical position pc
        goff[i] = (unsigned)(grow < lim ? grow : 0) * (unsigned)((lside ? ksL : ksU) * 2) + (unsigned)(cch * 16);
        ldst[i] = ring0 + (lside ? UARR : 0) + prow * H_RB;
    }
    auto dma_piece = [&](int s, int i) {
        lds_dma16((const unsigned char *)(i >= 2 ? Lh : Uh) + (size_t)s * 64, goff[i], ldst[i] + (s & (NS - 1)) * STAGE);
    };
    // ---- consumer: wave (wr, wc) owns rows 128 wr .. (L side) x columns 64 wc .. (U side) of the tile --------------------------------
    const int wr = wave & 1, wc = wave >> 1;
    const int fo = c16 * H_RB + ((g4 ^ h_swz(c16)) << 4);
    const int ubase = wc * 64 * H_RB + fo, lbase = UARR + wr * 128 * H_RB + fo;
    f4_t acc[4][8];                            // [U sub-tile][L sub-tile]: register i of lane (c, g) = (U row 16 su + 4 g + i, L position 16 sl + c)
#pragma unroll
    for (int su = 0; su < 4; ++su)
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) acc[su][sl] = (f4_t){0.f, 0.f, 0.f, 0.f};
    h8_t aA[4], aB[4], b[8];
    const int nst = Kp / 32;                   // >= 8 (the launcher's condition on Kp)

    // ---- the wave's block of C ---------------------------------------------------------------------------------------------------------
    const long long m0 = m0t + wr * 128, n0 = n0t + wc * 64;
    const long long mrem = m - m0, nrem = n - n0;
    const bool wave_in = mrem > 0 && nrem > 0, c_full = mrem >= 128 && nrem >= 64;   // wave-uniform
    const long long ncl = nrem < 64 ? nrem : 64, mcl = mrem < 128 ? mrem : 128;
    constexpr unsigned ES = C32 ? 4u : 8u;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((char *)Cv + (wave_in ? (m0 + n0 * ldc) * (long long)ES : 0)), 0, wave_in ? (int)(((ncl - 1) * ldc + mcl) * ES) : 0, 0x00020000);
    const unsigned ldcb = __builtin_amdgcn_readfirstlane((unsigned)ldc * ES);   // (kept scalar: the accesses' column offsets are SGPR operands)
    // full fp32 block: batch su = the eight dwordx4 accesses (register i, row group G) of U sub-tile su
    const unsigned voff32 = (unsigned)(4 * c16) * 4u + (unsigned)(4 * g4) * ldcb;
    u4_t cfA[C32 ? 8 : 1];
    auto c_load32 = [&](u4_t (&cf)[C32 ? 8 : 1], int su) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int Gq = 0; Gq < 2; ++Gq) {
                if (NOLD) cf[C32 ? 2 * i + Gq : 0] = (u4_t){0u, 0u, 0u, 0u};
                else cf[C32 ? 2 * i + Gq : 0] = __builtin_amdgcn_raw_buffer_load_b128(rc, (int)(voff32 + 256u * Gq), (int)((unsigned)(16 * su + i) * ldcb), H_AUX);
            }
    };
    constexpr int NPF = 8;                     // loads of the early batch
    const bool PF = C32 && c_full && dbg == 0 && !NOLD; // the first batch is requested three stages before the K loop ends

    // top of stage i: stage i + 1 must be in LDS for everyone.  It has landed once at most the pieces of stage i + 2 (issued during
    // stage i - 1) are outstanding -- and, behind them, the early C batch.  A bare s_barrier: each wave has waited for its own
    // pieces, the barrier makes that collective and says everyone is done reading stage i - 1, whose slot is refilled next.
    // mode 0: stage i + 2's pieces stay in flight; 1: only the early C batch is younger than what must have landed; 2: nothing to
    // wait for; 3: wait for everything
    auto top_of_stage = [&](int mode) {
        if (mode == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else if (mode == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPF) : "memory");
        else if (mode == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    const int PFS = PF ? nst - 3 : -1;
    // One stage: pass j = the four MFMAs of L fragment j with the four U fragments; behind it b[j] (and, in the first four passes,
    // one U fragment) of the NEXT stage are read from LDS; every second pass one operand piece of stage i + 3 goes out.
    // top: 0 = the caller has done the top of this stage, 1 = main part (no edge tests), 2 = tail
    auto stage = [&](h8_t (&acur)[4], h8_t (&anxt)[4], int i, int top) {
        const bool tail = top == 2;
        if (top == 1) top_of_stage(0);
        else if (tail && PF && i > PFS) top_of_stage(i == PFS + 1 ? 1 : 2);
        else if (tail) top_of_stage(i + 2 < nst ? 0 : 3);
        if (C32) { if (tail && i == PFS) { c_load32(cfA, 0); __builtin_amdgcn_sched_barrier(0); } }
        const int ron = ((i + 1) & (NS - 1)) * STAGE;
        const bool nxt = !tail || i + 1 < nst, dma = !tail || i + NS - 1 < nst;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int su = 0; su < 4; ++su) acc[su][j] = mfma16(acur[su], b[j], acc[su][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (nxt) {
                b[j] = *(const h8_t *)(ring + ron + lbase + j * 16 * H_RB);
                if (j < 4) anxt[j] = *(const h8_t *)(ring + ron + ubase + j * 16 * H_RB);
            }
            if ((j & 1) && dma) dma_piece(i + NS - 1, j >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    unsigned long long st_c0 = 0, st_r0 = 0;
    const bool stamp = STAMP && stamps != nullptr && dbg != 2 && tid == 0;
    if (dbg != 2) {
        for (int s2 = 0; s2 < NS - 1; ++s2)
#pragma unroll
            for (int i = 0; i < LPS; ++i) dma_piece(s2, i);
        top_of_stage(0);                       // stages 0 and 1 are in LDS
        if (STAMP && stamp) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll
        for (int j = 0; j < 4; ++j) aA[j] = *(const h8_t *)(ring + ubase + j * 16 * H_RB);
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = *(const h8_t *)(ring + lbase + j * 16 * H_RB);
        // trips of two stages (nst is even: Kp is a multiple of 64): the U fragment sets alternate statically.  Main part: no edge tests.
        stage(aA, aB, 0, 0);
        stage(aB, aA, 1, 1);
        int i = 2;
        const int imain = (nst - 4) & ~1;
#pragma clang loop unroll(disable)
        for (; i < imain; i += 2) { stage(aA, aB, i, 1); stage(aB, aA, i + 1, 1); }
#pragma clang loop unroll(disable)
        for (; i < nst; i += 2) { stage(aA, aB, i, 2); stage(aB, aA, i + 1, 2); }
        if (STAMP && stamp) {
            atomicAdd(stamps + 0, __builtin_amdgcn_s_memtime() - st_c0); atomicAdd(stamps + 1, __builtin_amdgcn_s_memrealtime() - st_r0);
            atomicAdd(stamps + 2, 1ull);
            st_r0 = __builtin_amdgcn_s_memrealtime();
        }
    }
    // ---- epilogue ------------------------------------------------------------------------------------------------------------------------
    if (!wave_in) return;                      // wave-uniform, after the last barrier
    if (dbg == 1) {                            // keep the accumulators alive without the C stream
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int su = 0; su < 4; ++su)
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) asm volatile("" ::"v"(acc[su][sl]));
#endif
        return;
    }
    if (C32 && c_full) {
        // A batch is consumed in two steps: every subtraction (which waits for the batch's loads), then the stores; and no VALU
        // instruction writes a register a store has just been given.  On gfx950 a 16-byte buffer_store with an SGPR soffset whose data
        // registers the NEXT VALU instruction rewrites loses elements (dword 3 of lanes 12-15 of every 16: the tail of the store's
        // register read), and hipcc 7.2 inserts no wait state there (its hazard model exempts SGPR-soffset stores; found in round 4,
        // HISTORY appendix).  The first form of this epilogue computed each difference into the same four registers right behind the
        // previous store: sixteen wrong elements every few dozen tiles (tools/hgemm16_check.py shows the pattern).
        u4_t cfB[C32 ? 8 : 1], cfC[C32 ? 8 : 1];
        unsigned ldcs = ldcb;                  // (the stores' column offsets from a fresh scalar: re-using the early batch's offsets across the
#if defined(__HIP_DEVICE_COMPILE__)            //  end of the K loop, hipcc parked them in VGPRs and wrapped the stores in readfirstlane loops)
        asm volatile("" : "+s"(ldcs));
#endif
        auto sub_batch = [&](u4_t (&cf)[C32 ? 8 : 1], int su) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int Gq = 0; Gq < 2; ++Gq) {
                    // (whole-vector arithmetic: updating the loaded u4_t element by element through bit casts, hipcc 7.2 subtracted
                    // element 0's result from the other three accumulators -- seen in the ISA and in the results)
                    const f4_t cv = __builtin_bit_cast(f4_t, cf[C32 ? 2 * i + Gq : 0]);
                    const f4_t av = (f4_t){acc[su][4 * Gq][i], acc[su][4 * Gq + 1][i], acc[su][4 * Gq + 2][i], acc[su][4 * Gq + 3][i]};
                    cf[C32 ? 2 * i + Gq : 0] = __builtin_bit_cast(u4_t, cv - av);
                }
        };
        auto store_batch = [&](u4_t (&cf)[C32 ? 8 : 1], int su) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int Gq = 0; Gq < 2; ++Gq) {
                    if (NOST) {
#if defined(__HIP_DEVICE_COMPILE__)
                        asm volatile("" ::"v"(cf[C32 ? 2 * i + Gq : 0]));
#endif
                    } else __builtin_amdgcn_raw_buffer_store_b128(cf[C32 ? 2 * i + Gq : 0], rc, (int)(voff32 + 256u * Gq), (int)((unsigned)(16 * su + i) * ldcs), H_AUX);
                }
        };
        if (PF) {   // the early batch first: it has landed, nothing waits; then the other three, all their loads in flight together
            sub_batch(cfA, 0);
            __builtin_amdgcn_sched_barrier(0);
            store_batch(cfA, 0);
            __builtin_amdgcn_sched_barrier(0);
            c_load32(cfB, 1); c_load32(cfC, 2); c_load32(cfA, 3);
            __builtin_amdgcn_sched_barrier(0);
            sub_batch(cfB, 1); sub_batch(cfC, 2); sub_batch(cfA, 3);
            __builtin_amdgcn_sched_barrier(0);
            store_batch(cfB, 1); store_batch(cfC, 2); store_batch(cfA, 3);
        } else {
            c_load32(cfA, 0); c_load32(cfB, 1); c_load32(cfC, 2);
            __builtin_amdgcn_sched_barrier(0);
            sub_batch(cfA, 0); sub_batch(cfB, 1); sub_batch(cfC, 2);
            __builtin_amdgcn_sched_barrier(0);
            store_batch(cfA, 0); store_batch(cfB, 1); store_batch(cfC, 2);
            __builtin_amdgcn_sched_barrier(0);
            c_load32(cfA, 3);
            __builtin_amdgcn_sched_barrier(0);
            sub_batch(cfA, 3);
            __builtin_amdgcn_sched_barrier(0);
            store_batch(cfA, 3);
        }
    } else if (!C32 && c_full) {
        // fp64 block, no row permutation: register i of sub-tile (su, sl) = (column 16 su + 4 g + i, row 16 sl + c); a b64 access
        // covers four 128-byte runs.  Two batches of 16 accesses (one su, four sl each) per round: all loads, all subtractions, then
        // the stores (see above).
        const unsigned voff64 = (unsigned)c16 * 8u + (unsigned)(4 * g4) * ldcb;
        double cv[2][16];
        auto ld = [&](double (&d)[16], int bt) {
            const int su = bt >> 1, s0 = (bt & 1) * 4;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    d[4 * s + i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rc, (int)(voff64 + 128u * (s0 + s)), (int)((unsigned)(16 * su + i) * ldcb), H_AUX));
        };
        auto sub = [&](double (&d)[16], int bt) {
            const int su = bt >> 1, s0 = (bt & 1) * 4;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) d[4 * s + i] -= (double)acc[su][s0 + s][i];
        };
        auto stv = [&](double (&d)[16], int bt) {
            const int su = bt >> 1, s0 = (bt & 1) * 4;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, d[4 * s + i]), rc, (int)(voff64 + 128u * (s0 + s)),
                                                          (int)((unsigned)(16 * su + i) * ldcb), H_AUX);
        };
#pragma unroll
        for (int bt = 0; bt < 8; bt += 2) {
            ld(cv[0], bt); ld(cv[1], bt + 1);
            __builtin_amdgcn_sched_barrier(0);
            sub(cv[0], bt); sub(cv[1], bt + 1);
            __builtin_amdgcn_sched_barrier(0);
            stv(cv[0], bt); stv(cv[1], bt + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // ragged block: one sub-tile at a time, every access masked.  (The lane indices pass through an empty asm: otherwise hipcc
        // computes the masked offsets BEFORE the K loop -- they depend on nothing the loop changes -- and spills them.)
        int cq = c16, gq = g4;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(cq), "+v"(gq));
#endif
#pragma unroll
        for (int su = 0; su < 4; ++su)
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const int row = PL == 1 ? 16 * sl + cq : (sl >> 2) * 64 + 4 * cq + (sl & 3);
                unsigned off[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = 16 * su + 4 * gq + i;
                    off[i] = (row < mrem && col < nrem) ? (unsigned)row * ES + (unsigned)col * ldcb : 0x80000000u;
                }
                if (C32) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (int)off[i], 0, H_AUX));
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] -= acc[su][sl][i];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[i]), rc, (int)off[i], 0, H_AUX);
                } else {
                    double v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rc, (int)off[i], 0, H_AUX));
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] -= (double)acc[su][sl][i];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, v[i]), rc, (int)off[i], 0, H_AUX);
                }
            }
    }
    if (STAMP && stamp) {   // (DBG & 8: the stamp includes the acknowledgement of the tile's stores)
        __builtin_amdgcn_sched_barrier(0);
        if (DBG & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicAdd(stamps + 3, __builtin_amdgcn_s_memrealtime() - st_r0);
    }
}

template <bool C32, int DBG>
static int launch_big16(mpf_ctx *c, int64_t m, int64_t n, int Kp, const HgemmImages &im, void *C, int64_t ldc) {
    constexpr int LDS = 4 * (256 + 256) * H_RB;
    auto *kern = hgemm16_big_kernel<C32, DBG>;
    MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));   // (cheap; the probe build has several instantiations)
    const long long bm = (m + 255) / 256, bn = (n + 255) / 256;
    unsigned long long *stamps = DBG != 0 ? c->ws->hp_stamps : nullptr;
    kern<<<(int)(bm * bn), 512, LDS, c->stream>>>(m, n, Kp, im.Lh, im.Uh, C, ldc, (int)bm, (int)bn, im.ksL, im.ksU, stamps);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

// shapes the launcher sends here: m, n >= 1024, Kp a multiple of 64, >= 256; row strides of the images in im.ksL / im.ksU
int launch_hgemm16_big(mpf_ctx *c, int64_t m, int64_t n, int Kp, const HgemmImages &im, void *C, int64_t ldc, bool c32) {
    if (Kp < 256 || (Kp & 63)) { c->err = "hgemm16_big: K must be a multiple of 64, at least 256"; return -1; }
    if (((m + 255) / 256) * ((n + 255) / 256) > 0x7FFFFFFFll) { c->err = "hgemm16_big: too many tiles"; return -1; }
    // the loader addresses an image row with a 32-bit byte offset
    if ((m + 256) * (int64_t)im.ksL * 2 >= (1ll << 32) || (n + 256) * (int64_t)im.ksU * 2 >= (1ll << 32)) { c->err = "hgemm16_big: operand image beyond 4 GB"; return -1; }
#ifdef MPF_PROBE
    if (c32) switch (c->tune.hgemm_dbg) {
        case 1: return launch_big16<true, 1>(c, m, n, Kp, im, C, ldc);
        case 2: return launch_big16<true, 2>(c, m, n, Kp, im, C, ldc);
        case 3: return launch_big16<true, 3>(c, m, n, Kp, im, C, ldc);
        case 5: return launch_big16<true, 5>(c, m, n, Kp, im, C, ldc);
        case 4: return launch_big16<true, 4>(c, m, n, Kp, im, C, ldc);
        case 12: return launch_big16<true, 12>(c, m, n, Kp, im, C, ldc);
        default: break;
    }
#endif
    return c32 ? launch_big16<true, 0>(c, m, n, Kp, im, C, ldc) : launch_big16<false, 0>(c, m, n, Kp, im, C, ldc);
}

// ---- any shape: 128 x 128 tile, four waves of 64 x 64, ring of three 16-KB stages, up to three workgroups per CU -------------------------
// The kernel behind every plain-operand update the big-tile kernel does not take (m or n below 1024, K below 256).  Same fragment
// layout, same MFMA, same k order: the same bits.  Stage i is waited for at its own top (the other workgroups of the CU cover the
// wait); the C block of a wave is 16 dwordx4 accesses (fp32 copy) or 64 b64 accesses (fp64 matrix).
template <bool C32>
__global__ __launch_bounds__(256, 3) void hgemm16_ring_kernel(long long m, long long n, int Kp, const unsigned short *__restrict__ Lh,
                                                              const unsigned short *__restrict__ Uh, void *__restrict__ Cv, long long ldc,
                                                              int tiles_m, int tiles_n, int ksL, int ksU) {
    constexpr int NS = 3, LPS = 4;
    constexpr int UARR = 128 * H_RB, STAGE = 2 * UARR;
    constexpr int PL = C32 ? 4 : 1;
    __shared__ __attribute__((aligned(16))) unsigned char ring[NS * STAGE];
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int grp = lin / (tiles_m * 8);
    const int gw = (tiles_n - grp * 8) < 8 ? (tiles_n - grp * 8) : 8;
    const int idx = lin - grp * tiles_m * 8;
    const long long m0t = (long long)(idx / gw) * 128, n0t = (long long)(grp * 8 + idx % gw) * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g4 = lane >> 4;
    // loader: wave w moves U pieces w, w + 4 and L pieces w, w + 4 (16 LDS rows each)
    const int lr = lane >> 2, pc = lane & 3;
    unsigned goff[LPS], ldst[LPS];
    const unsigned ring0 = lds_addr(ring);
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const bool lside = i >= 2;
        const int prow = (wave + 4 * (i & 1)) * 16, trow = prow + lr;
        const int irow = lside ? h_perm<PL>(trow) : trow;
        const long long grow = (lside ? m0t : n0t) + irow, lim = lside ? m : n;
        const int cch = pc ^ h_swz(trow);
        goff[i] = (unsigned)(grow < lim ? grow : 0) * (unsigned)((lside ? ksL : ksU) * 2) + (unsigned)(cch * 16);
        ldst[i] = ring0 + (lside ? UARR : 0) + prow * H_RB;
    }
    auto issue = [&](int s, int slot) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) lds_dma16((const unsigned char *)(i >= 2 ? Lh : Uh) + (size_t)s * 64, goff[i], ldst[i] + slot * STAGE);
    };
    const int wr = wave & 1, wc = wave >> 1;
    const int fo = c16 * H_RB + ((g4 ^ h_swz(c16)) << 4);
    const int ubase = wc * 64 * H_RB + fo, lbase = UARR + wr * 64 * H_RB + fo;
    f4_t acc[4][4];
#pragma unroll
    for (int su = 0; su < 4; ++su)
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) acc[su][sl] = (f4_t){0.f, 0.f, 0.f, 0.f};
    const int nst = Kp / 32;
    issue(0, 0);
    if (1 < nst) issue(1, 1);
    int slot = 0;
    for (int i = 0; i < nst; ++i) {
        // stage i has landed once at most the pieces of stage i + 1 are outstanding
        if (i + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // everyone's part of stage i is in LDS; everyone is done with stage i - 1
        __builtin_amdgcn_sched_barrier(0);
        const int s2 = slot == 0 ? 2 : slot - 1;                         // slot of stage i + 2 = slot of stage i - 1
        if (i + 2 < nst) issue(i + 2, s2);
        const unsigned char *st = ring + slot * STAGE;
        h8_t a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = *(const h8_t *)(st + ubase + j * 16 * H_RB); b[j] = *(const h8_t *)(st + lbase + j * 16 * H_RB); }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
#pragma unroll
            for (int su = 0; su < 4; ++su) acc[su][sl] = mfma16(a[su], b[sl], acc[su][sl]);
        slot = slot == 2 ? 0 : slot + 1;
    }
    // ---- epilogue ------------------------------------------------------------------------------------------------------------------------
    const long long m0 = m0t + wr * 64, n0 = n0t + wc * 64;
    const long long mrem = m - m0, nrem = n - n0;
    if (mrem <= 0 || nrem <= 0) return;        // wave-uniform, after the last barrier
    const bool c_full = mrem >= 64 && nrem >= 64;
    const long long ncl = nrem < 64 ? nrem : 64, mcl = mrem < 64 ? mrem : 64;
    constexpr unsigned ES = C32 ? 4u : 8u;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)((char *)Cv + (m0 + n0 * ldc) * (long long)ES), 0, (int)(((ncl - 1) * ldc + mcl) * ES), 0x00020000);
    const unsigned ldcb = __builtin_amdgcn_readfirstlane((unsigned)ldc * ES);
    if (C32 && c_full) {
        // every load, every subtraction, then the stores (see hgemm16_big_kernel)
        const unsigned voff32 = (unsigned)(4 * c16) * 4u + (unsigned)(4 * g4) * ldcb;
        u4_t cf[C32 ? 16 : 1];
#pragma unroll
        for (int su = 0; su < 4; ++su)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                cf[C32 ? 4 * su + i : 0] = __builtin_amdgcn_raw_buffer_load_b128(rc, (int)voff32, (int)((unsigned)(16 * su + i) * ldcb), H_AUX);
#pragma unroll
        for (int su = 0; su < 4; ++su)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f4_t cv = __builtin_bit_cast(f4_t, cf[C32 ? 4 * su + i : 0]);
                const f4_t av = (f4_t){acc[su][0][i], acc[su][1][i], acc[su][2][i], acc[su][3][i]};
                cf[C32 ? 4 * su + i : 0] = __builtin_bit_cast(u4_t, cv - av);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int su = 0; su < 4; ++su)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(cf[C32 ? 4 * su + i : 0], rc, (int)voff32, (int)((unsigned)(16 * su + i) * ldcb), H_AUX);
    } else if (!C32 && c_full) {
        const unsigned voff64 = (unsigned)c16 * 8u + (unsigned)(4 * g4) * ldcb;
#pragma unroll
        for (int su = 0; su < 4; ++su) {       // one U sub-tile (16 accesses) per round
            double cv[16];
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    cv[4 * sl + i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rc, (int)(voff64 + 128u * sl), (int)((unsigned)(16 * su + i) * ldcb), H_AUX));
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int i = 0; i < 4; ++i) cv[4 * sl + i] -= (double)acc[su][sl][i];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, cv[4 * sl + i]), rc, (int)(voff64 + 128u * sl), (int)((unsigned)(16 * su + i) * ldcb), H_AUX);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int su = 0; su < 4; ++su)
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int row = PL == 1 ? 16 * sl + c16 : 4 * c16 + sl;
                unsigned off[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = 16 * su + 4 * g4 + i;
                    off[i] = (row < mrem && col < nrem) ? (unsigned)row * ES + (unsigned)col * ldcb : 0x80000000u;
                }
                if (C32) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (int)off[i], 0, H_AUX));
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] -= acc[su][sl][i];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[i]), rc, (int)off[i], 0, H_AUX);
                } else {
                    double v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rc, (int)off[i], 0, H_AUX));
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] -= (double)acc[su][sl][i];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, v[i]), rc, (int)off[i], 0, H_AUX);
                }
            }
    }
}

// any m, n > 0; Kp a multiple of 64 (the images are zero-padded to it)
int launch_hgemm16_ring(mpf_ctx *c, int64_t m, int64_t n, int Kp, const HgemmImages &im, void *C, int64_t ldc, bool c32) {
    if (Kp < 64 || (Kp & 63)) { c->err = "hgemm16_ring: K must be a multiple of 64"; return -1; }
    const long long tm = (m + 127) / 128, tn = (n + 127) / 128;
    if (tm * tn > 0x7FFFFFFFll) { c->err = "hgemm16_ring: too many tiles"; return -1; }
    if ((m + 128) * (int64_t)im.ksL * 2 >= (1ll << 32) || (n + 128) * (int64_t)im.ksU * 2 >= (1ll << 32)) { c->err = "hgemm16_ring: operand image beyond 4 GB"; return -1; }
    if (c32) hgemm16_ring_kernel<true><<<(int)(tm * tn), 256, 0, c->stream>>>(m, n, Kp, im.Lh, im.Uh, C, ldc, (int)tm, (int)tn, im.ksL, im.ksU);
    else hgemm16_ring_kernel<false><<<(int)(tm * tn), 256, 0, c->stream>>>(m, n, Kp, im.Lh, im.Uh, C, ldc, (int)tm, (int)tn, im.ksL, im.ksU);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

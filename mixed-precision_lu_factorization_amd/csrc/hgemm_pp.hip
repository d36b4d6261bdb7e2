// Big-K fp16-in / fp32-accumulate trailing update on the fp32 working copy, round 4 form ("ping-pong"): the kernel behind the
// K = sb * nb updates of the two-level schedule (reference: the cublasDgemm call at MPF.cu:230-239; contract C6).
//
//   C (fp32, column-major m x n, leading dimension ldc)  -=  Lh[m][K] * Uh[n][K]^T      (fp16 images, k contiguous)
//
// What round 3's hgemm_big_kernel lost (probe library's modes and in-kernel stamps, m = n = 28672, K = 1024):
//   * ~4 us per tile of workgroup turnaround (launch, first operand stages, teardown) with one workgroup per CU;
//   * ~11 us per tile of C stream: four serial batches of 32 loads, then 32 stores, per wave;
//   * a K loop at 1360-1420 cycles per 32-k stage where the matrix pipe needs 1040: both waves of a SIMD ran the same stream in
//     step (fragment reads, DMA pieces and MFMAs interleaved) and met at the same DMA instructions; a stage's MFMAs alone, with its
//     barrier, take 1180-1390 cycles in these structures.
// This kernel:
//   * PERSISTENT: one workgroup per CU walks its tiles; the first three operand stages of the next tile are requested before the
//     C stream of the current one starts (the ring is free by then), so a tile's K loop starts on resident operands;
//   * PING-PONG K loop: the waves of a SIMD belong to different groups (waves 0-3 / 4-7); in every barrier-delimited slot one group
//     issues the 16 MFMAs of a 32-k stage from registers (+ two of its operand pieces between them) while the other reads its 12
//     fragments of the next stage from LDS and requests its other two pieces (global_load_lds_dwordx4).  Ring of three 32-KB stages;
//   * the C block in batches of 32 dword accesses (two 128-byte runs each), three batches of loads out before the first is consumed.
// Measured alone (gpurun_out/r04_m_pp.log, same box, K = 512 / 1024 / 2048): 622 / 815 / 965 TFLOP/s against 617 / 802 / 938 for
// hgemm_big_kernel with the pipelined C stream and 563 / 764 / 918 for round 3's kernel; K loop 1340 cycles per stage.
// Tried here and dropped: the accumulator transposed so that C moves in 16-byte pieces (32-byte runs in 32 rows per access:
// 2.7 TB/s instead of 5.8); all four pieces in the fragment slot (that slot then takes 733 cycles against the MFMA slot's 590).
// Per output element: ONE fp32 MFMA accumulation chain (v_mfma_f32_32x32x16_f16) over k ascending, one subtraction -- the same
// numbers as hgemm_ring_kernel / hgemm_big_kernel produce (same operand order, same k order).
#include "mpf_internal.h"
#include <type_traits>

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f16_t __attribute__((ext_vector_type(16)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));

#ifndef MPF_C_AUX
#define MPF_C_AUX 2
#endif
namespace {
constexpr int PP_AUX = MPF_C_AUX;   // cache policy of the streamed C block (2 = nt: the operand images keep the L2)
constexpr int PP_TM = 256, PP_TN = 256;
constexpr int PP_NS = 3;            // ring depth
constexpr int PP_RB = 64;           // bytes an operand row contributes to a stage (32 k)
constexpr int PP_UARR = PP_TN * PP_RB, PP_LARR = PP_TM * PP_RB;
constexpr int PP_STAGE = PP_UARR + PP_LARR;          // 32 KB
constexpr int PP_LPS = 4;           // DMA pieces (16 rows x 64 B) per wave and stage: 32 pieces over 8 waves
constexpr int PP_LDS = PP_NS * PP_STAGE;

// chunk swizzle of the 64-byte LDS rows (same as hgemm_big_kernel: a ds_read_b128's four 16-lane groups each hit 64 banks)
__device__ __forceinline__ int pp_swz(int trow) {
    const int qd = (trow >> 2) & 7;
    return (qd ^ (qd >> 1)) & 3;
}
}  // namespace

__global__ __launch_bounds__(512, 1) void hgemm_pp_kernel(long long m, long long n, int Kp, const unsigned short *__restrict__ Lh,
                                                          const unsigned short *__restrict__ Uh, float *__restrict__ Cv, long long ldc,
                                                          int tiles_m, int tiles_n, int ksL, int ksU, int dbg_, unsigned long long *stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
#ifdef MPF_PROBE   // probe library: 1 = K loop only (no C stream), 2 = C stream only (no K loop).  (Round 4's ablations -- no operand
                   // DMA, no fragment reads, images read as [k / 32][row][32] -- are recorded in DESIGN 4.4; the switches cost the probe
                   // build registers it does not have: it spilled, and its timings stopped being the product kernel's.)
    const int dbg = dbg_ >= 3 ? 1 : dbg_;
#else
    constexpr int dbg = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                 // ping-pong group: waves w and w + 4 share a SIMD
    const int wr = grp, wc = wave & 3;         // the wave's 128 x 64 block of the tile: rows 128 wr (L side), columns 64 wc (U side)
    const int r = lane & 31, h = lane >> 5;
    const int nst = dbg == 2 ? 0 : Kp / 32;

    // ---- the workgroup's tiles: XCD x owns a contiguous chunk of the tile sequence, its workgroups walk it round robin ---------
    const int nt_all = tiles_m * tiles_n;
    const int bid = blockIdx.x, G = gridDim.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int gx = (G - xcd + 7) >> 3;                                   // workgroups with this xcd label
    const int q8 = nt_all >> 3, r8 = nt_all & 7;
    const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, chunkn = q8 + (xcd < r8 ? 1 : 0);
    constexpr int GW = 4;                      // tile-columns walked together

    // ---- per-lane constants of the loader and the fragment reads -----------------------------------------------------------------
    const int lr = lane >> 2, pc = lane & 3;   // row within a DMA piece, physical chunk
    // piece i of wave w is piece w + 8 i of the stage: i = 0, 1 -> U rows 16 w .. and 16 (w + 8) .., i = 2, 3 -> the same L rows
    int ldst[PP_LPS], prow_[PP_LPS];
#pragma unroll
    for (int i = 0; i < PP_LPS; ++i) {
        prow_[i] = (wave + 8 * (i & 1)) * 16;
        ldst[i] = (i >= 2 ? PP_UARR : 0) + prow_[i] * PP_RB;
    }
    // fragment reads: lane = tile row (mod 32); chunk (2 ks + h) ^ swizzle(row); tile t of a side at + t * 32 rows
    const int sw = pp_swz(r);                  // (tile row bases are multiples of 32: the swizzle depends on r only)
    int fo[2];                                 // byte offset of the lane's chunk inside a tile's 32 x 64-B block, per k-step
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fo[ks] = r * PP_RB + (((2 * ks + h) ^ sw) << 4);
    const int ubase = (wc * 64) * PP_RB, lbase = PP_UARR + (wr * 128) * PP_RB;

    h8_t Fu[2][2], Fl[2][4];                   // [k-step][tile]: the stage's fragments
    f16_t acc[4][2];                           // [L tile][U tile]; MFMA rows = U side, columns (lanes) = L side = C's contiguous index
    auto load_frags = [&](int ro) {            // fragments of the stage at ring byte offset ro (wave-uniform)
        const unsigned char *st = ring + ro;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int t = 0; t < 2; ++t) Fu[ks][t] = *(const h8_t *)(st + ubase + t * 32 * PP_RB + fo[ks]);
#pragma unroll
            for (int t = 0; t < 4; ++t) Fl[ks][t] = *(const h8_t *)(st + lbase + t * 32 * PP_RB + fo[ks]);
        }
    };
    unsigned goff[PP_LPS];                     // byte offset of this lane's source of each piece at k = 0 inside its image (< 4 GB)
    auto issue_piece = [&](int ro, int s, int i) {   // piece i of stage s into the ring at byte offset ro, global -> LDS
        const unsigned char *base = (const unsigned char *)(i >= 2 ? Lh : Uh);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)(goff[i] + (unsigned)s * 64u)),
                                         (__attribute__((address_space(3))) void *)(ring + ro + ldst[i]), 16, 0, 0);
    };
    // The 16 MFMAs of a stage from registers; the wave's pieces 2 and 3 of a later stage (ring offset dro, stage ds; ds < 0: none)
    // go out behind the 4th and the 12th MFMA: four pieces in the other slot made that slot the longer one (733 cycles against the
    // 590 the MFMAs need: gpurun_out/r04_h_pp_ablate.log).
    auto mfma_stage = [&](int dro, int ds) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fu[ks][nt], Fl[ks][mt], acc[mt][nt], 0, 0, 0);
                if (nt == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (ds >= 0) issue_piece(dro, ds, 2 + ks);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
    };

    auto set_sources = [&](long long m0t, long long n0t) {
#pragma unroll
        for (int i = 0; i < PP_LPS; ++i) {
            const int trow = prow_[i] + lr;
            const long long grow = (i >= 2 ? m0t : n0t) + trow, lim = i >= 2 ? m : n;
            const int c = pc ^ pp_swz(trow);   // logical chunk stored at physical position pc
            goff[i] = (unsigned)((grow < lim ? grow : 0) * (long long)(i >= 2 ? ksL : ksU) * 2 + c * 16);
        }
    };
    auto issue_stage = [&](int ro, int s) {    // the wave's four pieces of stage s
#pragma unroll
        for (int i = 0; i < PP_LPS; ++i) issue_piece(ro, s, i);
    };
    auto issue_half = [&](int ro, int s) {     // its pieces 0 and 1 (2 and 3 go out inside the MFMA section)
        issue_piece(ro, s, 0); issue_piece(ro, s, 1);
    };
    auto issue_first = [&]() {                 // stages 0 .. 2 of a tile
        if (0 < nst) issue_stage(0, 0);
        if (1 < nst) issue_stage(PP_STAGE, 1);
        if (2 < nst) issue_stage(2 * PP_STAGE, 2);
    };
    auto ring_next = [&](int ro) -> int { return ro == 2 * PP_STAGE ? 0 : ro + PP_STAGE; };
    auto tile_of = [&](int pos, long long &m0t, long long &n0t) {   // position in this XCD's chunk -> tile origin
        const int lin = chunk0 + pos;
        const int g4 = lin / (tiles_m * GW);
        const int gw = (tiles_n - g4 * GW) < GW ? (tiles_n - g4 * GW) : GW;
        const int idx = lin - g4 * tiles_m * GW;
        m0t = (long long)(idx / gw) * PP_TM; n0t = (long long)(g4 * GW + idx % gw) * PP_TN;
    };

    if (slot >= chunkn) return;                // (more workgroups than tiles)
    long long m0t, n0t;
    tile_of(slot, m0t, n0t);
    set_sources(m0t, n0t);
    // first tile: its first three stages
    issue_first();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MPF_PROBE
    unsigned long long st_c = 0, st_r = 0, st_n = 0;
#endif

#pragma clang loop unroll(disable)
    for (int pos = slot; pos < chunkn; pos += gx) {
        // ---- top of a tile: stages 0 .. 2 are in LDS as far as THIS wave's pieces go (first tile: waited for above; later tiles:
        //      requested before the previous tile's C loads, which have all been consumed); make that collective ----------------
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        {   // (an opaque zero: a literal one makes hipcc peel the first stage of every tile to fold it into the MFMA's C operand)
            float z = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(z));
#endif
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 16; ++g) acc[mt][nt][g] = z;
        }
        // ---- this wave's 128 x 64 block of C (its first batch is requested inside the K loop when the block is full) ------------
        const long long cm0 = m0t + wr * 128, cn0 = n0t + wc * 64;
        const long long mrem = m - cm0, nrem = n - cn0;
        const bool c_any = mrem > 0 && nrem > 0, c_full = mrem >= 128 && nrem >= 64;          // wave-uniform
        const long long ncl = nrem < 64 ? nrem : 64, mcl = mrem < 128 ? mrem : 128;
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(Cv + (c_any ? cm0 + cn0 * ldc : 0)), 0, c_any ? (int)(((ncl - 1) * ldc + mcl) * 4) : 0, 0x00020000);
        const unsigned ldc4 = (unsigned)ldc * 4u;
        const unsigned voff = (unsigned)r * 4u + (unsigned)(4 * h) * ldc4;
        constexpr bool PFC = false;   // (the early batch of 32 registers does not fit beside the K loop: that build spilled; hgemm_big_kernel has it)
        float cfA[2][16];
        auto c_load = [&](float (&cf)[2][16], int mt) {   // batch mt of a full block: per-lane offset + an immediate per mt + a scalar per (nt, g)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    cf[nt][g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rc, (int)(voff + (unsigned)(32 * mt) * 4u), (int)((unsigned)(32 * nt + (g & 3) + 8 * (g >> 2)) * ldc4), PP_AUX));
        };
        auto c_prefetch = [&]() { c_load(cfA, 0); };
#ifdef MPF_PROBE
        const unsigned long long c0 = (stamps && tid == 0) ? __builtin_amdgcn_s_memtime() : 0, r0 = (stamps && tid == 0) ? __builtin_amdgcn_s_memrealtime() : 0;
#endif
        // Slots are delimited by bare barriers; both groups execute the same number of them.  Stage i lives in ring slot i % 3.
        //   even slot 2i    : group 0 computes stage i; group 1 reads its fragments of stage i and requests stage i + 2
        //   odd slot 2i + 1 : group 1 computes stage i; group 0 reads its fragments of stage i + 1 and requests stage i + 3
        // At the end of an even slot every wave has waited for its own pieces of stage i + 1 (read in the next slot): they have
        // landed once at most the pieces of stage i + 2 are outstanding (stages 1 and 2 have landed before the tile started).
        // Requests: a wave's pieces 0, 1 of a stage go out in its fragment slot, pieces 2, 3 in its next MFMA slot:
        //   group 0: stage i + 3 in odd slot 2i + 1 (half) and even slot 2i + 2 (half)    [stage i's ring slot: free after slot 2i]
        //   group 1: stage i + 2 in even slot 2i (half) and odd slot 2i + 1 (half)         [stage i - 1's ring slot]
        // so at the end of even slot 2i a wave of group 0 may have all four pieces of stage i + 2 outstanding, one of group 1 two,
        // when its pieces of stage i + 1 have landed.  Three stages before the end the first batch of the wave's C block is
        // requested (32 loads, younger than every piece): the wait of stage nst - 2 then leaves those outstanding, and the last
        // one has nothing to wait for.
        auto wait_next = [&](int i, int left) {   // left = pieces of stage i + 2 this wave has requested by now
            if (i >= 2 && i + 1 < nst) {
                if (PFC && i == nst - 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                else if (i + 2 < nst) { if (left == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        };
        auto bar = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
        if (nst > 0) {
            int ro = 0;                            // ring offset of stage i
            if (grp == 0) {
                load_frags(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bar();
#pragma clang loop unroll(disable)
                for (int i = 0; i < nst; ++i) {
                    const int rn = ring_next(ro), rp = ring_next(rn);      // ring slots of stages i + 1, i + 2 (= i - 1)
                    mfma_stage(rp, (i >= 1 && i + 2 < nst) ? i + 2 : -1);  // (second half of stage i + 2, begun in slot 2i - 1)
                    __builtin_amdgcn_sched_barrier(0);
                    wait_next(i, 4);
                    bar();
                    if (PFC && i == nst - 3) { c_prefetch(); __builtin_amdgcn_sched_barrier(0); }
                    if (i + 1 < nst) load_frags(rn);
                    if (i + 3 < nst) issue_half(ro, i + 3);              // (stage i + 3 reuses stage i's slot)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    bar();
                    ro = rn;
                }
            } else {
                bar();
#pragma clang loop unroll(disable)
                for (int i = 0; i < nst; ++i) {
                    load_frags(ro);
                    const int rn = ring_next(ro), rp = ring_next(rn);
                    const bool req = i >= 1 && i + 2 < nst;
                    if (req) issue_half(rp, i + 2);                      // (stage i + 2 reuses stage i - 1's slot)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    wait_next(i, 2);
                    bar();
                    mfma_stage(rp, req ? i + 2 : -1);
                    if (PFC && i == nst - 3) { __builtin_amdgcn_sched_barrier(0); c_prefetch(); }   // (behind this slot's pieces)
                    bar();
                    ro = rn;
                }
            }
        }
#ifdef MPF_PROBE
        if (stamps && tid == 0) { st_c += __builtin_amdgcn_s_memtime() - c0; st_r += __builtin_amdgcn_s_memrealtime() - r0; st_n += 1; }
#endif
        // ---- the ring is free (its last reads were completed before the last barrier): request the next tile's first stages;
        //      before that, the early C batch is declared landed (it has been in flight for three stages: this wait is for it
        //      alone -- behind the next tile's requests the compiler's counter model would wait for those too) ------------------
        if (PFC) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 16; ++g) asm volatile("" : "+v"(cfA[nt][g]));
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool more = pos + gx < chunkn;
        if (more) {
            tile_of(pos + gx, m0t, n0t);
            set_sources(m0t, n0t);
            issue_first();
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- C stream of the wave's block ------------------------------------------------------------------------------------------
        // Element (32 mt + r, 32 nt + (g & 3) + 8 (g >> 2) + 4 h) of the block sits in register g of acc[mt][nt]: a dword access of the
        // wave covers two runs of 32 consecutive elements (128 bytes each).  One batch = one mt (32 loads, 32 stores); up to three
        // batches of loads are out before the first is consumed (a wave has at most 63 operations outstanding), a store is never
        // waited for.  (C as 16-byte pieces with the accumulator transposed -- 32-byte runs in 32 rows per access -- streamed at
        // 2.7 TB/s instead of 5.8: measured, gpurun_out/r04_g_pp.log.)
        if (dbg == 1) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) asm volatile("" ::"v"(acc[mt][nt]));
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no C loads behind the next tile's stages: wait for them here)
        } else if (!c_any) {                    // no block of C, hence no C loads behind the next tile's stages
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (c_full) {
            float cfB[2][16], cfC[2][16];
            auto c_consume = [&](float (&cf)[2][16], int mt) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 16; ++g) cf[nt][g] -= acc[mt][nt][g];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 16; ++g)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cf[nt][g]), rc, (int)(voff + (unsigned)(32 * mt) * 4u),
                                                              (int)((unsigned)(32 * nt + (g & 3) + 8 * (g >> 2)) * ldc4), PP_AUX);
            };
            if (!PFC) c_load(cfA, 0);
            c_load(cfB, 1); c_load(cfC, 2);
            __builtin_amdgcn_sched_barrier(0);
            c_consume(cfA, 0); c_load(cfA, 3);
            __builtin_amdgcn_sched_barrier(0);
            c_consume(cfB, 1);
            c_consume(cfC, 2);
            c_consume(cfA, 3);
        } else {
            // ragged block: one MFMA tile at a time, every access masked.  (The lane indices pass through an empty asm: otherwise
            // hipcc computes the 128 masked offsets BEFORE the K loop -- they depend on nothing the loop changes -- and spills them.)
            int hq = h, rq = r;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(hq), "+v"(rq));
#endif
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float cv[16];
                    const bool mok = 32 * mt + rq < mrem;
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int col = 32 * nt + (g & 3) + 8 * (g >> 2) + 4 * hq;
                        const unsigned off = (mok && col < nrem) ? (unsigned)(32 * mt + rq) * 4u + (unsigned)col * ldc4 : 0x80000000u;
                        cv[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (int)off, 0, PP_AUX));
                    }
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int col = 32 * nt + (g & 3) + 8 * (g >> 2) + 4 * hq;
                        const unsigned off = (mok && col < nrem) ? (unsigned)(32 * mt + rq) * 4u + (unsigned)col * ldc4 : 0x80000000u;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cv[g] - acc[mt][nt][g]), rc, (int)off, 0, PP_AUX);
                    }
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next tile's stages were requested BEFORE this tile's C loads, and every C load has been consumed: they have landed
        // (the counter is in order); only this tile's stores may still be outstanding, and nothing waits for them
    }
#ifdef MPF_PROBE
    if (stamps && tid == 0) { atomicAdd(stamps + 0, st_c); atomicAdd(stamps + 1, st_r); atomicAdd(stamps + 2, st_n); }
#endif
}

// C (fp32, column-major, ldc) -= Lh * Uh^T on ready images; shapes the launcher sends here: m, n >= 1024, Kp a multiple of 64 >= 256
int launch_hgemm_pp(mpf_ctx *c, int64_t m, int64_t n, int Kp, const HgemmImages &im, float *C, int64_t ldc) {
    if (Kp < 96 || (Kp & 31)) { c->err = "hgemm_pp: K must be a multiple of 32, at least 96"; return -1; }
    if (!(c->attr_done & ATTR_HGEMM_PP)) {
        MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)hgemm_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS));
        c->attr_done |= ATTR_HGEMM_PP;
    }
    const long long bm = (m + PP_TM - 1) / PP_TM, bn = (n + PP_TN - 1) / PP_TN;
    const long long tiles = bm * bn;
    // (the kernel's tile walk deals the tiles to eight XCD labels: with fewer than eight workgroups a label would own tiles nobody walks)
    if (tiles < 8) { c->err = "hgemm_pp: fewer than 8 tiles"; return -1; }
    const int cus = c->num_cus > 0 ? c->num_cus : 256;
    const int grid = (int)(tiles < cus ? tiles : cus);
    unsigned long long *stamps = nullptr;
    int dbg = 0;
#ifdef MPF_PROBE
    dbg = c->tune.hgemm_dbg;
    if (dbg) stamps = c->ws->hp_stamps;
#endif
    hgemm_pp_kernel<<<grid, 512, PP_LDS, c->stream>>>(m, n, Kp, im.Lh, im.Uh, C, ldc, (int)bm, (int)bn, im.ksL, im.ksU, dbg, stamps);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

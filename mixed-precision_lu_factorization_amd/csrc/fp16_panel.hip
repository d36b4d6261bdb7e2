// fp16 pivot panel for gfx950: replaces reference MPF.cu:108-159 (panel extract, double_to_fp16_block,
// HGETF2_kernel launch, pivot globalisation) and hgetf2_kernel.cu:15-120.
//
// Design (MI355X-first, not a translation of the cooperative-grid reference kernel):
//   * The panel never exists in HBM as fp16.  Each workgroup (512 threads) converts its HP_R = 256 rows
//     of the fp64 panel on load (contract C1) and keeps them as an LDS-resident slab: ROW-major by row
//     pair (two rows packed per dword), 260-dword stride, so that a lane's 4 columns are one ds_read_b128
//     / ds_write_b128 (conflict-free across the 16-lane groups) and one whole row pair is a single
//     wave-wide access.
//   * Rows are never moved.  A row swap j <-> p of the reference (hgetf2_kernel.cu:92-98) is pure
//     bookkeeping: every physical row carries its current logical position pos[]; the pivot search uses
//     pos to reproduce the reference's tie-break exactly (lowest 256-row block of t = pos - j, then the
//     smallest 8-bit bit-reversed lane index inside the block -- what the strict-'>' binary tree of
//     hgetf2_kernel.cu:47-56 and the serial block scan :73-78 amount to).
//   * The per-column critical path is kept minimal: after the pivot row of column j is known only the
//     multipliers and column j+1 are updated (128 threads), the next local candidate is found, ONE wave
//     brings that candidate's row pair up to date and publishes it; the rest of the rank-1 update runs
//     while the hand-off is in flight.
//   * One hand-off round per column, no grid barrier (the reference needs five): every workgroup
//     publishes {epoch | |a| | ~tiekey} as ONE 8-byte write-through granule plus its candidate row as
//     self-tagged 8-byte granules {tag32 | 2 x fp16}: no drain, no flag.  Every workgroup sweeps the
//     candidate granules, picks the same winner and reads the winner's row granules (sc1 loads),
//     re-reading until every tag matches.
//   * Arithmetic is contract C2: v_pk_mul_f16 then v_pk_add_f16 (never fma), division = IEEE fp32
//     quotient rounded once to fp16.
//   * On exit the kernel also leaves the list of rows that moved (src -> dst) for the fp64 row
//     interchange kernel: the sequential swap chain of LASWP_kernel (MPF.cu:47-57) is already resolved
//     by the position bookkeeping.
#include "mpf_internal.h"
#include <mutex>
#include "fp16_device.h"

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));

// one rank-1 step on a packed row pair: x - m*u with separately rounded product and difference
__device__ __forceinline__ unsigned pk_elim(unsigned xw, h2_t m2, unsigned u2w) {
    const h2_t x = __builtin_bit_cast(h2_t, xw), u2 = __builtin_bit_cast(h2_t, u2w);
    const h2_t t = m2 * u2; // v_pk_mul_f16   (hgetf2_kernel.cu:113, no fma)
    const h2_t y = x - t;   // v_pk_add_f16 neg
    return __builtin_bit_cast(unsigned, y);
}

// max over the 64 lanes of a wave, returned in every lane.  Within each 16-lane row: DPP butterflies
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) on the VALU; across the four rows: readlane.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_max_step(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = dpp_max_step<0xB1>(v);  // quad_perm [1,0,3,2]
    v = dpp_max_step<0x4E>(v);  // quad_perm [2,3,0,1]
    v = dpp_max_step<0x141>(v); // row_half_mirror
    v = dpp_max_step<0x140>(v); // row_mirror
    unsigned r = 0;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)v, row * 16);
        r = w > r ? w : r;
    }
    return r;
}

// Search key of a row for the pivot of column j, 32 bits: {|a|:15 | ~tie:17} -- the maximum key is the reference's pivot (the
// largest |a|, ties to the smallest tie order; t = pos - j < 65536 rows => tie_key(t) < 2^17).  Never 0 for a real row; 0 = none.
// It is also the payload of the cross-workgroup candidate granule {tag:31 << 32 | key}, tag = launch sequence << 9 | epoch (the
// row granules' tag): a granule from an earlier launch or column can never match the tag, so the array is never cleared between
// launches (round 2 zeroed it with a memset node in front of every pivot kernel: 258 fills on the latency chain).
__device__ __forceinline__ unsigned row_key32(unsigned hb, int p, int j) {
    return ((hb & 0x7FFFu) << 17) | (0x1FFFFu - tie_key((unsigned)(p - j)));
}
__device__ __forceinline__ int key32_pos(unsigned key, int j) { return j + (int)tie_key(0x1FFFFu - (key & 0x1FFFFu)); }

struct HpArgs {
    const double *A64; long long lda;     // fp64 source panel (or null)
    unsigned short *P16; long long ld16;  // fp16 source panel, factored in place (or null)
    unsigned short *out16; long long ldo; // optional factored fp16 output (rows physically swapped)
    int rows, cols, ipiv_offset;
    int *ipiv;
    MpfWorkspace *ws;
    unsigned tag_base;                    // (launch sequence << 9): row-granule tag = tag_base | epoch
    MovedList *moved;                     // if set: leave the moved-row list here (global rows = ipiv_offset + ...)
    int acq_fence;                        // 1: agent-scope acquire after the poll (debug aid)
    unsigned spin_limit;                  // every cross-workgroup spin gives up after this many polls (never hangs)
    unsigned seq;                         // launch sequence number (progress word)
    int local_G;                          // single-XCD form: slabs of the panel (the grid is larger: workgroups of the other XCDs leave at once)
    unsigned long long *signal;           // option gate_wait_value: the progress word once more, in signal memory a stream can wait on
};

// pivot of column j is final: published device-wide (write-through), and every 32 columns the progress word behind it --
// the fp64 panel of the same chain follows on another stream, 32 columns behind (launch_hgetf2_gate)
__device__ __forceinline__ void hp_publish_pivot(const HpArgs &a, int j, int p) {
    __hip_atomic_store(&a.ipiv[j], p + 1 + a.ipiv_offset, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // hgetf2_kernel.cu:80-81 + MPF.cu:152
    if (((j + 1) & 31) == 0 || j + 1 == a.cols)
        __hip_atomic_store(&a.ws->hp_progress, ((unsigned long long)a.seq << 32) | (unsigned)(j + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (a.signal && (((j + 1) & 31) == 0 || j + 1 == a.cols))
        __hip_atomic_store(a.signal, ((unsigned long long)a.seq << 32) | (unsigned)(j + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

constexpr int HP_RS = HP_MAXCOLS + 4;     // dword stride of one row pair in the slab (260)
// LDS carve (bytes) for R rows per workgroup
constexpr int HP_OFF_WRED = 0;                          // 2 x u64
constexpr int HP_OFF_MISC = 16;                         // 12 ints: [1] pivot position [2] aborted [4],[5] per-wave candidate row
constexpr int HP_OFF_POS = 64;                          // R ints
template <int R> struct HpCarve {
    static constexpr int PAIRS = R / 2;
    static constexpr int OFF_UROW = HP_OFF_POS + R * 4;         // 2 x 256 dwords of (u,u)
    static constexpr int OFF_MBUF = OFF_UROW + 2 * 256 * 4;     // PAIRS dwords (m_lo, m_hi)
    static constexpr int OFF_MASK = OFF_MBUF + PAIRS * 4;       // PAIRS dwords (row-active half masks)
    static constexpr int OFF_SLAB = OFF_MASK + PAIRS * 4;       // PAIRS x HP_RS dwords
    static constexpr int LDS_BYTES = OFF_SLAB + PAIRS * HP_RS * 4;
    static_assert(OFF_SLAB % 16 == 0 && OFF_UROW % 16 == 0, "b128 LDS accesses need 16-byte alignment");
};

// an 8-byte store that stays in the XCD's L2 (no write-through): the single-XCD form's hand-off
__device__ __forceinline__ void hp_store_plain(unsigned long long *p, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
#endif
}

// R = rows per workgroup: 256 (137 KB LDS: a CU to itself) or 128 (70 KB: fits beside one 74-KB GEMM
// workgroup, which is what lets the pivot chain of panel k+1 run under the trailing update of panel k)
// STAMP = true is a diagnostic build of the same kernel: wave 0 of every workgroup timestamps the segments of each column
// step with s_memtime and workgroup 0 leaves the per-segment cycle sums in ws->hp_stamps (read through mpf_microbench 70..).
// LOCAL = true (round 5, option hp_local_xcd): the panel's G <= CUs / 8 slabs are taken by workgroups of ONE XCD -- the first workgroup to
// arrive names its XCD, workgroups of that XCD take the slabs in the order they arrive, everyone else leaves at once (the grid is
// 8 x (G + 2): dispatch deals workgroups round the XCDs).  Inside one XCD the L2 is the point of coherence: keys and row granules go
// out as PLAIN stores (237 ns to the other workgroup's sc1 poll against 387 ns for a write-through store across XCDs,
// profiles/r05_xcd_flag_probe.log); a plain store is never seen on another XCD, so the roles are only given to that XCD's own.
// Fewer than G arrivals (a dispatch that is not round-robin) end in the bounded waits giving up (-4), like any missing workgroup.
template <int R, bool STAMP = false, bool LOCAL = false>
__global__ __launch_bounds__(HP_T) void hgetf2_lds_kernel(HpArgs a) {
    constexpr int HP_PAIRS = R / 2;            // row pairs per workgroup
    constexpr int NW1 = HP_PAIRS / 64;         // waves that run the critical part (one row pair per lane)
    constexpr int NCG = HP_T / HP_PAIRS;       // column groups of the deferred update
    constexpr int HP_OFF_UROW = HpCarve<R>::OFF_UROW, HP_OFF_MBUF = HpCarve<R>::OFF_MBUF;
    constexpr int HP_OFF_MASK = HpCarve<R>::OFF_MASK, HP_OFF_SLAB = HpCarve<R>::OFF_SLAB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned *wred = (unsigned *)(smem_raw + HP_OFF_WRED);
    int *misc = (int *)(smem_raw + HP_OFF_MISC); // [0] candidate row, [1] pivot position, [2] aborted
    int *pos = (int *)(smem_raw + HP_OFF_POS);
    unsigned *urow2 = (unsigned *)(smem_raw + HP_OFF_UROW);
    unsigned *mbuf = (unsigned *)(smem_raw + HP_OFF_MBUF);
    unsigned *maskbuf = (unsigned *)(smem_raw + HP_OFF_MASK);
    unsigned *slab = (unsigned *)(smem_raw + HP_OFF_SLAB);

    // this kernel is a latency chain that usually shares its CU with trailing-update workgroups: win arbitration
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int g = blockIdx.x, G = gridDim.x;
    if (LOCAL) {
        int *role = (int *)(smem_raw + HP_OFF_MISC) + 8;
        if (tid == 0) {
            const unsigned x = (unsigned)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u;   // HW_REG_XCC_ID[3:0]
            const unsigned long long sq = (unsigned long long)a.seq << 32;
            unsigned long long cur = __hip_atomic_load(&a.ws->hp_xcd_target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while ((cur >> 32) != a.seq) {
                if (__hip_atomic_compare_exchange_strong(&a.ws->hp_xcd_target, &cur, sq | (x + 1), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { cur = sq | (x + 1); break; }
            }
            int r = -1;
            if ((unsigned)(cur & 0xFFFFFFFFull) == x + 1) {
                unsigned long long c2 = __hip_atomic_load(&a.ws->hp_xcd_roles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (;;) {
                    const unsigned long long nv = (c2 >> 32) == a.seq ? c2 + 1 : (sq | 1ull);
                    if (__hip_atomic_compare_exchange_strong(&a.ws->hp_xcd_roles, &c2, nv, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { r = (int)(nv & 0xFFFFFFFFull) - 1; break; }
                }
                if (r >= a.local_G) r = -1;
            }
            *role = r;
        }
        __syncthreads();
        g = *role;
        G = a.local_G;
        if (g < 0) return;                     // (workgroup-uniform)
        __syncthreads();
    }
    const int rows = a.rows, cols = a.cols;
    const int tp = tid % HP_PAIRS, cg = tid / HP_PAIRS; // row pair / column group
    const long long row0 = (long long)g * R;

    // ---- load + convert the slab (MPF.cu:108-121 fused) -------------------------------------------
    {
        const long long ra = row0 + 2 * tp, rb = ra + 1;
        for (int c = cg; c < cols; c += NCG) {
            unsigned lo = 0, hi = 0;
            if (a.A64) {
                if (ra < rows) lo = double_to_fp16_bits(a.A64[ra + (long long)c * a.lda]);
                if (rb < rows) hi = double_to_fp16_bits(a.A64[rb + (long long)c * a.lda]);
            } else {
                if (ra < rows) lo = a.P16[ra + (long long)c * a.ld16];
                if (rb < rows) hi = a.P16[rb + (long long)c * a.ld16];
            }
            slab[tp * HP_RS + c] = lo | (hi << 16);
        }
        if (tid < R) { const long long r = row0 + tid; pos[tid] = r < rows ? (int)r : -1; }
        if (tid == 0) { misc[2] = 0; misc[0] = -1; misc[4] = -1; misc[5] = -1; }
    }
    __syncthreads();


    // ---- prologue: candidates of column 0 ------------------------------------------------------------
    unsigned gmax = 0;
    {
        unsigned k0 = 0, k1 = 0;
        if (tid < HP_PAIRS) {
            const unsigned w = slab[tp * HP_RS + 0];
            const int pa = pos[2 * tp], pb = pos[2 * tp + 1];
            if (pa >= 0) k0 = row_key32(w & 0xFFFFu, pa, 0);
            if (pb >= 0) k1 = row_key32(w >> 16, pb, 0);
            const unsigned km = k0 > k1 ? k0 : k1;
            const unsigned wm = wave_max_u32(km);
            if (lane == 0) { wred[wave] = wm; if (wm == 0) misc[4 + wave] = -1; }
            if (km == wm && wm != 0) misc[4 + wave] = (k0 == wm) ? 2 * tp : 2 * tp + 1; // keys are unique
        }
        __syncthreads();
        gmax = (NW1 == 1 || wred[0] > wred[1]) ? wred[0] : wred[1];
    }

    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, first_hits = 0;
    unsigned first_best = 0;
    if (STAMP) tlast = __builtin_amdgcn_s_memtime();
#define HP_STAMP(i) do { if (STAMP && wave == 0) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); seg[i] += tn_ - tlast; tlast = tn_; } } while (0)
    int prev_p = -1; // pivot position of step j-1
    for (int j = 0; j < cols; ++j) {
        const int cr = gmax == 0 ? -1 : ((NW1 == 1 || wred[0] >= wred[1]) ? misc[4] : misc[5]); // candidate row for column j
        const int par = j & 1;
        unsigned *ucur = urow2 + par * 256;        // pivot row of step j   (written this iteration)
        const unsigned *uprev = urow2 + (par ^ 1) * 256; // pivot row of step j-1 (read by the deferred update)
        const unsigned epoch = (unsigned)(j + 1);
        const unsigned tag = a.tag_base | epoch;

        // ---- wave 0: bring the candidate's row pair up to date (step j-1, columns >= j+1), publish ------
        if (wave == 0) {
            // The candidate KEY is known since the end of the previous step: it goes out first, so that its trip to the other
            // workgroups runs under the ~1000 cycles the row below needs; the row's granules carry their own tags and are
            // only read after the sweep of the keys has completed.
            if (G > 1 && lane == 0)
                // (an atomic exchange, result unused, instead of a write-through store: the read-modify-write is carried out at once
                //  where the other XCDs' polls look, a store waits its turn in the write path -- 2.40 -> 2.33 us per column at 32768
                //  rows, 2.15 -> 2.08 at 4096; the 128 row granules per column are better off as plain stores: 2.40 as atomics)
                { if (LOCAL) hp_store_plain(&a.ws->cand[par][g], ((unsigned long long)tag << 32) | gmax);
                  else (void)__hip_atomic_exchange(&a.ws->cand[par][g], ((unsigned long long)tag << 32) | gmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            u4_t xv = (u4_t){0u, 0u, 0u, 0u};
            if (cr >= 0) {
                const int tpc = cr >> 1;
                xv = *(const u4_t *)(slab + tpc * HP_RS + 4 * lane);
                if (j > 0) {
                    const unsigned mw = mbuf[tpc], rmask = maskbuf[tpc];
                    const h2_t m2 = __builtin_bit_cast(h2_t, mw);
                    const u4_t uv = *(const u4_t *)(uprev + 4 * lane);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 4 * lane + e;
                        const unsigned y = pk_elim(xv[e], m2, uv[e]);
                        const unsigned keep = (c >= j + 1) ? rmask : 0u;
                        xv[e] = (y & keep) | (xv[e] & ~keep);
                    }
                    *(u4_t *)(slab + tpc * HP_RS + 4 * lane) = xv;
                }
            }
            // the row itself: halves of row cr in columns 4*lane .. 4*lane+3
            unsigned h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (cr & 1) ? (xv[e] >> 16) : (xv[e] & 0xFFFFu);
            if (G > 1) {
                u4_t gr;
                gr[0] = h[0] | (h[1] << 16); gr[1] = tag; gr[2] = h[2] | (h[3] << 16); gr[3] = tag;
                // two self-tagged 8-byte granules per lane, write-through, no drain and no flag
                unsigned long long *dst = &a.ws->rowbuf[par][g][2 * lane];
                if (LOCAL) {
                    hp_store_plain(dst, ((unsigned long long)gr[1] << 32) | gr[0]);
                    hp_store_plain(dst + 1, ((unsigned long long)gr[3] << 32) | gr[2]);
                } else {
                    __hip_atomic_store(dst, ((unsigned long long)gr[1] << 32) | gr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 1, ((unsigned long long)gr[3] << 32) | gr[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                // single workgroup: the local winner is the pivot row
                u4_t uu;
#pragma unroll
                for (int e = 0; e < 4; ++e) uu[e] = h[e] | (h[e] << 16);
                *(u4_t *)(ucur + 4 * lane) = uu;
                if (lane == 0) {
                    const int p = key32_pos(gmax, j);
                    misc[1] = p;
                    hp_publish_pivot(a, j, p);
                }
            }
        }

        HP_STAMP(0); // candidate row brought up to date + published
        // ---- everyone: deferred rank-1 update of step j-1 on columns >= j+1 (candidate pair excluded) ----
        // With more than one workgroup the first column group (waves 0 and 1 for R = 256) stays out of it: wave 0 goes straight
        // to the hand-off, so that its polls run beside the update instead of behind its own share of it, and the other
        // NCG - 1 groups cover the columns between them (a wave's share is bound by its own LDS round trips, not by the
        // SIMD it shares: giving the orphaned columns to two waves only, or items round robin to seven, measured slower).
        // Round 4: which wave takes which rows is chosen by SIMD.  Waves sit on SIMD (wave mod 4); with 256 rows and several
        // workgroups wave 0 is out (hand-off), and "row pairs 0-63: waves 2, 4, 6 / 64-127: waves 3, 5, 7" put two full shares on
        // SIMDs 2 and 3 and one on SIMDs 0 and 1 -- the update is bound by VALU issue and LDS, the step waited for SIMDs 2 and 3.
        // Now row pairs 0-63 go to waves 2, 5, 7 (a third of the quads each) and 64-127 to waves 1, 3, 4, 6 (a quarter each, wave 1
        // being idle at this point anyway): 16, 37, 37, 37 quads per SIMD instead of 21, 21, 43, 43.
        const bool remap = G > 1 && R == 256;
        int ngrp = (G > 1 && NCG > 1) ? NCG - 1 : NCG;
        int cgx = (G > 1 && NCG > 1) ? cg - 1 : cg;
        int tpd = tp;                              // the row pair this thread carries through the deferred update
        if (remap) {
            const int hf = (0x5A >> wave) & 1;     // waves 1, 3, 4, 6 -> row pairs 64 .. 127; waves 2, 5, 7 -> 0 .. 63
            const int rk = (int)((0x23121000u >> (4 * wave)) & 15u);   // rank of the wave within its half: waves 1 .. 7 -> 0, 0, 1, 2, 1, 3, 2
            tpd = 64 * hf + lane; ngrp = hf ? 4 : 3; cgx = wave == 0 ? -1 : rk;
        }
        if (j > 0 && j + 1 < cols && cgx >= 0) {
            const int tp = tpd;
            const unsigned rmask = maskbuf[tp];
            if (rmask != 0 && tp != (cr >> 1)) {
                const h2_t m2 = __builtin_bit_cast(h2_t, mbuf[tp]);
                const int q0 = (j + 1) >> 2, q1 = (cols - 1) >> 2;
                int q = q0 + (((cgx - q0) % ngrp) + ngrp) % ngrp;
                unsigned *row = slab + tp * HP_RS;
                // only the quad that holds column j+1 has columns to leave alone: it is peeled off, the others carry no column test
                if (q == q0) {
                    u4_t xv = *(const u4_t *)(row + 4 * q);
                    const u4_t uv = *(const u4_t *)(uprev + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned y = pk_elim(xv[e], m2, uv[e]);
                        const unsigned keep = (4 * q + e >= j + 1) ? rmask : 0u;
                        xv[e] = (y & keep) | (xv[e] & ~keep);
                    }
                    *(u4_t *)(row + 4 * q) = xv;
                    q += ngrp;
                }
                // four quads per LDS round trip: all reads, then the arithmetic, then the writes (written in this order: the
                // compiler keeps a write in front of the next quad's reads, which it cannot tell apart)
                for (; q + 3 * ngrp <= q1; q += 4 * ngrp) {
                    u4_t xv[4], uv[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        xv[b] = *(const u4_t *)(row + 4 * (q + b * ngrp));
                        uv[b] = *(const u4_t *)(uprev + 4 * (q + b * ngrp));
                    }
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned y = pk_elim(xv[b][e], m2, uv[b][e]);
                            xv[b][e] = (y & rmask) | (xv[b][e] & ~rmask);
                        }
#pragma unroll
                    for (int b = 0; b < 4; ++b) *(u4_t *)(row + 4 * (q + b * ngrp)) = xv[b];
                }
                for (; q <= q1; q += ngrp) {
                    u4_t xv = *(const u4_t *)(row + 4 * q);
                    const u4_t uv = *(const u4_t *)(uprev + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned y = pk_elim(xv[e], m2, uv[e]);
                        xv[e] = (y & rmask) | (xv[e] & ~rmask);
                    }
                    *(u4_t *)(row + 4 * q) = xv;
                }
            }
        }

        HP_STAMP(1); // this wave's share of the deferred update
        // ---- wave 0: sweep all candidate granules, pick the winner, fetch its row ----------------------
        if (G > 1 && wave == 0) {
            unsigned wkey = 0;                 // the winner's key
            const bool aborted = misc[2] != 0;
            u4_t gr = (u4_t){0u, 0u, 0u, 0u};
            int gw = 0;
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
                unsigned bestk = 0;
                int besti = 0;
                // all candidate loads of a poll go out together (a lane beyond G re-reads the last candidate: a branch per lane
                // would put a memory round trip between two loads; which loads exist is a wave-uniform question)
                unsigned long long xs[HP_MAXG / 64];
#pragma unroll
                for (int i = 0; i < HP_MAXG / 64; ++i) {
                    const int idx = lane + 64 * i;
                    xs[i] = 64 * i < G ? __hip_atomic_load(&a.ws->cand[par][idx < G ? idx : G - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                }
#pragma unroll
                for (int i = 0; i < HP_MAXG / 64; ++i) {
                    if (64 * i < G) {
                        const int idx = lane + 64 * i;
                        const unsigned long long x = xs[i];
                        ok &= (unsigned)(x >> 32) == tag;
                        if ((unsigned)x > bestk || i == 0) { bestk = (unsigned)x; besti = idx < G ? idx : G - 1; }
                    }
                }
                if (STAMP) {   // diagnostic build: polls per column, keys present at the first poll, first poll's best = the winner?
                    seg[6] += 1;
                    if (spins == 0) {
                        int have = 0;
                        unsigned pk = 0;
#pragma unroll
                        for (int i = 0; i < HP_MAXG / 64; ++i)
                            if (64 * i < G) {
                                const bool here = lane + 64 * i < G && (unsigned)(xs[i] >> 32) == tag;
                                have += __builtin_popcountll(__ballot(here));
                                if (here && (unsigned)xs[i] > pk) pk = (unsigned)xs[i];
                            }
                        seg[7] += (unsigned long long)have;
                        first_best = wave_max_u32(pk);
                    }
                }
                if (__all(ok)) {
                    // keys of real rows are unique: the lowest lane that holds the maximum names the winner's workgroup
                    wkey = wave_max_u32(bestk);
                    if (STAMP && first_best == wkey) first_hits += 1;
                    const unsigned long long owners = __ballot(bestk == wkey);
                    gw = __builtin_amdgcn_readlane(besti, (int)__builtin_ctzll(owners));
                    const unsigned long long *src = &a.ws->rowbuf[par][gw][2 * lane];
                    const unsigned long long g0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long g1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gr[0] = (unsigned)g0; gr[1] = (unsigned)(g0 >> 32); gr[2] = (unsigned)g1; gr[3] = (unsigned)(g1 >> 32);
                    if (__all(gr[1] == tag && gr[3] == tag)) break;
                }
                if (aborted) break;
                if (spins + 1 >= a.spin_limit) { // give up: flag it, never hang
                    if (lane == 0) { atomicAdd(&a.ws->hp_timeouts, 1); misc[2] = 1; }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (a.acq_fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            int p = key32_pos(wkey, j);
            // after a give-up the sweep may hold stale candidates: whatever happens next is garbage (the launch is reported
            // as failed, -4), but every row index derived from p must stay inside the panel
            if (p < j || p >= rows) p = j;
            u4_t uu;
            uu[0] = (gr[0] & 0xFFFFu) * 0x10001u; uu[1] = (gr[0] >> 16) * 0x10001u;
            uu[2] = (gr[2] & 0xFFFFu) * 0x10001u; uu[3] = (gr[2] >> 16) * 0x10001u;
            *(u4_t *)(ucur + 4 * lane) = uu;
            if (lane == 0) {
                misc[1] = p;
                if (g == 0) hp_publish_pivot(a, j, p);
            }
        }
        HP_STAMP(2); // sweep + winner's row (the cross-workgroup wait)
        __syncthreads(); // (3) pivot row of step j in LDS, deferred update of step j-1 complete
        HP_STAMP(3); // barrier: the slowest wave's deferred update

        // ---- critical part of step j: positions, multipliers, column j+1, next local candidates -------
        const int piv_pos = misc[1];
        unsigned k0 = 0, k1 = 0;
        if (tid < HP_PAIRS) {
            // everything this part reads from LDS is requested up front, in ONE round trip (left to the compiler the column's two
            // elements and the pivot row's come in three dependent ones: behind the position logic, behind the division)
            const int jn = j + 1 < cols ? j + 1 : j;
            const unsigned dw = slab[tp * HP_RS + j], xw0 = slab[tp * HP_RS + jn];
            const unsigned uj = ucur[j], ujn = ucur[jn];
            int pa = pos[2 * tp], pb = pos[2 * tp + 1];
            if (pa == piv_pos) pa = j; else if (pa == j) pa = piv_pos;   // hgetf2_kernel.cu:92-98 as bookkeeping
            if (pb == piv_pos) pb = j; else if (pb == j) pb = piv_pos;
            pos[2 * tp] = pa; pos[2 * tp + 1] = pb;
            const unsigned rmask = (pa > j ? 0x0000FFFFu : 0u) | (pb > j ? 0xFFFF0000u : 0u);
            maskbuf[tp] = rmask;
            if (rmask) {
                const _Float16 ujj = bits_h(uj & 0xFFFFu);
                h2_t m2;
                m2.x = hdiv_ieee(bits_h(dw & 0xFFFFu), ujj);   // :108
                m2.y = hdiv_ieee(bits_h(dw >> 16), ujj);
                const unsigned mw = __builtin_bit_cast(unsigned, m2);
                mbuf[tp] = mw;
                slab[tp * HP_RS + j] = (mw & rmask) | (dw & ~rmask); // :109
                if (j + 1 < cols) {
                    const unsigned xw = xw0;
                    const unsigned y = pk_elim(xw, m2, ujn);
                    const unsigned nw = (y & rmask) | (xw & ~rmask);
                    slab[tp * HP_RS + j + 1] = nw;
                    if (pa > j) k0 = row_key32(nw & 0xFFFFu, pa, j + 1);
                    if (pb > j) k1 = row_key32(nw >> 16, pb, j + 1);
                }
            }
            const unsigned km = k0 > k1 ? k0 : k1;
            const unsigned wm = wave_max_u32(km);
            if (lane == 0) { wred[wave] = wm; if (wm == 0) misc[4 + wave] = -1; }
            if (km == wm && wm != 0) misc[4 + wave] = (k0 == wm) ? 2 * tp : 2 * tp + 1;
        }
        prev_p = piv_pos;
        HP_STAMP(4); // critical part: positions, multipliers, column j+1, next local candidate
        __syncthreads(); // (1) next candidate known to everyone
        gmax = (NW1 == 1 || wred[0] > wred[1]) ? wred[0] : wred[1];
        HP_STAMP(5); // barrier + candidate read-back
    }
    (void)prev_p;
    if (STAMP && g == 0 && tid == 0)
    { for (int i = 0; i < 6; ++i) a.ws->hp_stamps[i] = seg[i];
      a.ws->hp_stamps[6] = (seg[6] << 32) | first_hits; a.ws->hp_stamps[7] = seg[7]; }
#undef HP_STAMP

    // ---- outputs: moved-row list for the fp64 interchange, optional factored fp16 panel -------------------
    if (a.moved && tid < R) {
        const int p = pos[tid];
        const long long r = row0 + tid;
        if (p >= 0 && p != (int)r) {
            const int i = atomicAdd(&a.moved->n, 1); // the list's counter is zeroed by the host before the launch
            if (i < LASWP_MAXMOVED) {
                a.moved->src[i] = a.ipiv_offset + (int)r;
                a.moved->dst[i] = a.ipiv_offset + p;
            }
        }
    }
    unsigned short *out = a.out16 ? a.out16 : a.P16;
    const long long ldo = a.out16 ? a.ldo : a.ld16;
    if (out) {
        const int pa = pos[2 * tp], pb = pos[2 * tp + 1];
        for (int c = cg; c < cols; c += NCG) {
            const unsigned w = slab[tp * HP_RS + c];
            if (pa >= 0) out[pa + (long long)c * ldo] = (unsigned short)(w & 0xFFFFu);
            if (pb >= 0) out[pb + (long long)c * ldo] = (unsigned short)(w >> 16);
        }
    }
}


// ---- the same kernel with a column WINDOW in LDS ------------------------------------------------------------------------------
// A 256-row workgroup of hgetf2_lds_kernel holds all 256 columns in LDS (137 KB): it has its CU to itself, and a 30 000-row
// panel holds 118 of the 256 CUs for the whole pivot chain while the trailing update runs on the rest.  Nothing on the
// per-column critical path ever touches more than columns j and j+1; everything to the right only receives the deferred
// rank-1 update.  So only a WINDOW of HW_W = 136 columns lives in LDS -- [0, 136) until column HW_JS = 120, [120, 256) after
// it -- and the columns right of the first window wait in REGISTERS (10 quads = 40 packed dwords per thread of column groups
// 1..3, statically indexed), taking the same deferred updates in the same order.  At column 120 they are written into the
// slab, over the factored columns nobody reads again.  76 KB of LDS per workgroup: two pivot workgroups, or one and a
// 68-KB update workgroup, share a CU.  The cross-workgroup protocol, the per-element operation order and therefore every
// bit of the result are those of hgetf2_lds_kernel (tests/test_gpu_steps.py compares the two and the oracle).
// Not covered (launch_hgetf2 keeps the full-slab kernel for them): an fp16 source panel and the factored fp16 output --
// the factored columns left of the window are gone by the end.
constexpr int HW_W = 136;                 // columns in the LDS window
constexpr int HW_RS = HW_W + 4;           // dword stride of one row pair (140 = 12 mod 64: b128 accesses of 16 row pairs conflict-free)
constexpr int HW_JS = 120;                // the window moves when this column is reached
constexpr int HW_RQ = 10;                 // register quads per thread: 3 groups x 10 quads x 4 = columns 136 .. 255
static_assert(HW_W + 3 * HW_RQ * 4 == HP_MAXCOLS && HW_JS + HW_W == HP_MAXCOLS && HW_JS % 4 == 0 && HW_W % 4 == 0, "window layout");
struct HwCarve {
    static constexpr int PAIRS = HP_R / 2;
    static constexpr int OFF_UROW = HP_OFF_POS + HP_R * 4;
    static constexpr int OFF_MBUF = OFF_UROW + 2 * 256 * 4;
    static constexpr int OFF_MASK = OFF_MBUF + PAIRS * 4;
    static constexpr int OFF_SLAB = OFF_MASK + PAIRS * 4;
    static constexpr int LDS_BYTES = OFF_SLAB + PAIRS * HW_RS * 4;   // 75 840
    static_assert(OFF_SLAB % 16 == 0 && OFF_UROW % 16 == 0 && (HW_RS * 4) % 16 == 0, "b128 LDS accesses need 16-byte alignment");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

__global__ __launch_bounds__(HP_T, 4) void hgetf2_win_kernel(HpArgs a) {   // 4 waves per SIMD: two of these workgroups on a CU
    constexpr int R = HP_R, HP_PAIRS = R / 2, NCG = HP_T / HP_PAIRS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned *wred = (unsigned *)(smem_raw + HP_OFF_WRED);
    int *misc = (int *)(smem_raw + HP_OFF_MISC);
    int *pos = (int *)(smem_raw + HP_OFF_POS);
    unsigned *urow2 = (unsigned *)(smem_raw + HwCarve::OFF_UROW);
    unsigned *mbuf = (unsigned *)(smem_raw + HwCarve::OFF_MBUF);
    unsigned *maskbuf = (unsigned *)(smem_raw + HwCarve::OFF_MASK);
    unsigned *slab = (unsigned *)(smem_raw + HwCarve::OFF_SLAB);

    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, G = gridDim.x;
    const int rows = a.rows, cols = a.cols;
    const int tp = tid % HP_PAIRS, cg = tid / HP_PAIRS;
    const long long row0 = (long long)g * R;
    const bool regcols = cols > HW_W;          // some columns start in registers
    const int rg = cg - 1;                     // register column group 0..2 (column group 0 holds none: its first wave runs the hand-off)
    unsigned rc[4 * HW_RQ];                    // row pair tp, columns HW_W + 4 * (3 * s + rg) + e

    // ---- load + convert (MPF.cu:108-121 fused) ---------------------------------------------------------------------------
    {
        // every load is unconditional (row and column clamped into the panel, the value dropped afterwards): a branch per
        // element would put a wait behind each of them
        const long long ra = row0 + 2 * tp, rb = ra + 1;
        const bool va = ra < rows, vb = rb < rows;
        const double *pa = a.A64 + (va ? ra : rows - 1), *pb = a.A64 + (vb ? rb : rows - 1);
        const int wcols = cols < HW_W ? cols : HW_W;
#pragma unroll 4
        for (int c = cg; c < wcols; c += NCG) {
            const unsigned lo = double_to_fp16_bits(pa[(long long)c * a.lda]), hi = double_to_fp16_bits(pb[(long long)c * a.lda]);
            slab[tp * HW_RS + c] = (va ? lo : 0u) | ((vb ? hi : 0u) << 16);
        }
#pragma unroll
        for (int i = 0; i < 4 * HW_RQ; ++i) rc[i] = 0u;
        if (regcols && rg >= 0) {
#pragma unroll
            for (int s = 0; s < HW_RQ; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = HW_W + 4 * (3 * s + rg) + e;
                    const long long off = (long long)(c < cols ? c : cols - 1) * a.lda;
                    const unsigned lo = double_to_fp16_bits(pa[off]), hi = double_to_fp16_bits(pb[off]);
                    rc[4 * s + e] = c < cols ? ((va ? lo : 0u) | ((vb ? hi : 0u) << 16)) : 0u;
                }
        }
        if (tid < R) { const long long r = row0 + tid; pos[tid] = r < rows ? (int)r : -1; }
        if (tid == 0) { misc[2] = 0; misc[0] = -1; misc[4] = -1; misc[5] = -1; }
    }
    __syncthreads();


    unsigned gmax = 0;
    {
        unsigned k0 = 0, k1 = 0;
        if (tid < HP_PAIRS) {
            const unsigned w = slab[tp * HW_RS + 0];
            const int pa = pos[2 * tp], pb = pos[2 * tp + 1];
            if (pa >= 0) k0 = row_key32(w & 0xFFFFu, pa, 0);
            if (pb >= 0) k1 = row_key32(w >> 16, pb, 0);
            const unsigned km = k0 > k1 ? k0 : k1;
            const unsigned wm = wave_max_u32(km);
            if (lane == 0) { wred[wave] = wm; if (wm == 0) misc[4 + wave] = -1; }
            if (km == wm && wm != 0) misc[4 + wave] = (k0 == wm) ? 2 * tp : 2 * tp + 1;
        }
        __syncthreads();
        gmax = wred[0] > wred[1] ? wred[0] : wred[1];
    }

    int wb = 0;   // first column of the LDS window
    for (int j = 0; j < cols; ++j) {
        const int cr = gmax == 0 ? -1 : (wred[0] >= wred[1] ? misc[4] : misc[5]);
        const int par = j & 1;
        unsigned *ucur = urow2 + par * 256;
        const unsigned *uprev = urow2 + (par ^ 1) * 256;
        const unsigned tag = a.tag_base | (unsigned)(j + 1);
        const bool in_regs = regcols && wb == 0 && rg >= 0;   // this thread still holds columns in registers

        // ---- the candidate's row pair: brought up to date (step j-1, columns >= j+1) and published.  Wave 0 does the
        //      window's columns; the three threads that hold the pair's register columns do those.
        if (wave == 0) {
            if (G > 1 && lane == 0)
                (void)__hip_atomic_exchange(&a.ws->cand[par][g], ((unsigned long long)tag << 32) | gmax,
                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (see hgetf2_lds_kernel)
            if (4 * lane < HW_W) {
                const int c0 = wb + 4 * lane;
                u4_t xv = (u4_t){0u, 0u, 0u, 0u};
                if (cr >= 0) {
                    const int tpc = cr >> 1;
                    xv = *(const u4_t *)(slab + tpc * HW_RS + 4 * lane);
                    if (j > 0) {
                        const unsigned mw = mbuf[tpc], rmask = maskbuf[tpc];
                        const h2_t m2 = __builtin_bit_cast(h2_t, mw);
                        const u4_t uv = *(const u4_t *)(uprev + c0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned y = pk_elim(xv[e], m2, uv[e]);
                            const unsigned keep = (c0 + e >= j + 1) ? rmask : 0u;
                            xv[e] = (y & keep) | (xv[e] & ~keep);
                        }
                        *(u4_t *)(slab + tpc * HW_RS + 4 * lane) = xv;
                    }
                }
                unsigned h[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (cr & 1) ? (xv[e] >> 16) : (xv[e] & 0xFFFFu);
                if (G > 1) {
                    unsigned long long *dst = &a.ws->rowbuf[par][g][c0 >> 1];
                    __hip_atomic_store(dst, ((unsigned long long)tag << 32) | (h[0] | (h[1] << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 1, ((unsigned long long)tag << 32) | (h[2] | (h[3] << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    u4_t uu;
#pragma unroll
                    for (int e = 0; e < 4; ++e) uu[e] = h[e] | (h[e] << 16);
                    *(u4_t *)(ucur + c0) = uu;
                }
            }
            if (G == 1 && lane == 0) {
                const int p = key32_pos(gmax, j);
                misc[1] = p;
                hp_publish_pivot(a, j, p);
            }
        }
        // (register quads right of the panel hold zeros and take part like the others: no per-quad branch, so that the ten
        //  LDS reads of the pivot row are in flight together)
        if (in_regs && cr >= 0 && tp == (cr >> 1)) {
            const unsigned rmask = j > 0 ? maskbuf[tp] : 0u;
            const h2_t m2 = __builtin_bit_cast(h2_t, j > 0 ? mbuf[tp] : 0u);
            const unsigned *up = uprev + HW_W + 4 * rg;
#pragma unroll
            for (int s = 0; s < HW_RQ; ++s) {
                const u4_t uv = *(const u4_t *)(up + 12 * s);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned x = rc[4 * s + e];
                    const unsigned y = pk_elim(x, m2, uv[e]);          // every register column is right of j+1
                    rc[4 * s + e] = (y & rmask) | (x & ~rmask);
                }
            }
            const unsigned sh = (cr & 1) ? 16u : 0u;
            if (G > 1) {
                unsigned long long *dst = &a.ws->rowbuf[par][g][(HW_W + 4 * rg) >> 1];
#pragma unroll
                for (int s = 0; s < HW_RQ; ++s) {
                    const unsigned h0 = (rc[4 * s] >> sh) & 0xFFFFu, h1 = (rc[4 * s + 1] >> sh) & 0xFFFFu;
                    const unsigned h2 = (rc[4 * s + 2] >> sh) & 0xFFFFu, h3 = (rc[4 * s + 3] >> sh) & 0xFFFFu;
                    __hip_atomic_store(dst + 6 * s, ((unsigned long long)tag << 32) | (h0 | (h1 << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 6 * s + 1, ((unsigned long long)tag << 32) | (h2 | (h3 << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                unsigned *ud = ucur + HW_W + 4 * rg;
#pragma unroll
                for (int s = 0; s < HW_RQ; ++s) {
                    u4_t uu;
#pragma unroll
                    for (int e = 0; e < 4; ++e) uu[e] = ((rc[4 * s + e] >> sh) & 0xFFFFu) * 0x10001u;
                    *(u4_t *)(ud + 12 * s) = uu;
                }
            }
        }

        // ---- everyone: deferred rank-1 update of step j-1 on columns >= j+1 (candidate pair excluded) --------------------
        const int ngrp = G > 1 ? NCG - 1 : NCG;
        const int cgx = G > 1 ? cg - 1 : cg;
        if (j > 0 && j + 1 < cols) {
            const unsigned rmask = maskbuf[tp];
            if (rmask != 0 && tp != (cr >> 1)) {
                const h2_t m2 = __builtin_bit_cast(h2_t, mbuf[tp]);
                if (cgx >= 0) {
                    const int wend = (cols < wb + HW_W ? cols : wb + HW_W);
                    const int q0 = (j + 1) >> 2, q1 = (wend - 1) >> 2;
                    int q = q0 + (((cgx - q0) % ngrp) + ngrp) % ngrp;
                    unsigned *row = slab + tp * HW_RS - wb;
                    if (q == q0) {                 // the quad that holds column j+1 (hgetf2_lds_kernel: peeled, then batches of four)
                        u4_t xv = *(const u4_t *)(row + 4 * q);
                        const u4_t uv = *(const u4_t *)(uprev + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned y = pk_elim(xv[e], m2, uv[e]);
                            const unsigned keep = (4 * q + e >= j + 1) ? rmask : 0u;
                            xv[e] = (y & keep) | (xv[e] & ~keep);
                        }
                        *(u4_t *)(row + 4 * q) = xv;
                        q += ngrp;
                    }
                    for (; q + 3 * ngrp <= q1; q += 4 * ngrp) {
                        u4_t xv[4], uv[4];
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            xv[b] = *(const u4_t *)(row + 4 * (q + b * ngrp));
                            uv[b] = *(const u4_t *)(uprev + 4 * (q + b * ngrp));
                        }
#pragma unroll
                        for (int b = 0; b < 4; ++b)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned y = pk_elim(xv[b][e], m2, uv[b][e]);
                                xv[b][e] = (y & rmask) | (xv[b][e] & ~rmask);
                            }
#pragma unroll
                        for (int b = 0; b < 4; ++b) *(u4_t *)(row + 4 * (q + b * ngrp)) = xv[b];
                    }
                    for (; q <= q1; q += ngrp) {
                        u4_t xv = *(const u4_t *)(row + 4 * q);
                        const u4_t uv = *(const u4_t *)(uprev + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned y = pk_elim(xv[e], m2, uv[e]);
                            xv[e] = (y & rmask) | (xv[e] & ~rmask);
                        }
                        *(u4_t *)(row + 4 * q) = xv;
                    }
                }
                if (in_regs) {
                    const unsigned *up = uprev + HW_W + 4 * rg;
#pragma unroll
                    for (int s = 0; s < HW_RQ; ++s) {
                        const u4_t uv = *(const u4_t *)(up + 12 * s);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned x = rc[4 * s + e];
                            const unsigned y = pk_elim(x, m2, uv[e]);
                            rc[4 * s + e] = (y & rmask) | (x & ~rmask);
                        }
                    }
                }
            }
        }

        // ---- wave 0: sweep all candidate granules, pick the winner, fetch its row ------------------------------------------
        if (G > 1 && wave == 0) {
            unsigned wkey = 0;                 // the winner's key
            const bool aborted = misc[2] != 0;
            u4_t gr = (u4_t){0u, 0u, 0u, 0u};
            int gw = 0;
            // only columns j .. cols-1 of the pivot row are ever read, and only those are sure to have been published
            const bool need = 4 * lane + 3 >= j && 4 * lane < cols;
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
                unsigned bestk = 0;
                int besti = 0;
                // all candidate loads of a poll go out together (a lane beyond G re-reads the last candidate: a branch per lane
                // would put a memory round trip between two loads; which loads exist is a wave-uniform question)
                unsigned long long xs[HP_MAXG / 64];
#pragma unroll
                for (int i = 0; i < HP_MAXG / 64; ++i) {
                    const int idx = lane + 64 * i;
                    xs[i] = 64 * i < G ? __hip_atomic_load(&a.ws->cand[par][idx < G ? idx : G - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                }
#pragma unroll
                for (int i = 0; i < HP_MAXG / 64; ++i) {
                    if (64 * i < G) {
                        const int idx = lane + 64 * i;
                        const unsigned long long x = xs[i];
                        ok &= (unsigned)(x >> 32) == tag;
                        if ((unsigned)x > bestk || i == 0) { bestk = (unsigned)x; besti = idx < G ? idx : G - 1; }
                    }
                }
                if (__all(ok)) {
                    // keys of real rows are unique: the lowest lane that holds the maximum names the winner's workgroup
                    wkey = wave_max_u32(bestk);
                    const unsigned long long owners = __ballot(bestk == wkey);
                    gw = __builtin_amdgcn_readlane(besti, (int)__builtin_ctzll(owners));
                    const unsigned long long *src = &a.ws->rowbuf[par][gw][2 * lane];
                    const unsigned long long g0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long g1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gr[0] = (unsigned)g0; gr[1] = (unsigned)(g0 >> 32); gr[2] = (unsigned)g1; gr[3] = (unsigned)(g1 >> 32);
                    if (__all(!need || (gr[1] == tag && gr[3] == tag))) break;
                }
                if (aborted) break;
                if (spins + 1 >= a.spin_limit) {
                    if (lane == 0) { atomicAdd(&a.ws->hp_timeouts, 1); misc[2] = 1; }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (a.acq_fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            int p = key32_pos(wkey, j);
            if (p < j || p >= rows) p = j;
            u4_t uu;
            uu[0] = (gr[0] & 0xFFFFu) * 0x10001u; uu[1] = (gr[0] >> 16) * 0x10001u;
            uu[2] = (gr[2] & 0xFFFFu) * 0x10001u; uu[3] = (gr[2] >> 16) * 0x10001u;
            *(u4_t *)(ucur + 4 * lane) = uu;
            if (lane == 0) {
                misc[1] = p;
                if (g == 0) hp_publish_pivot(a, j, p);
            }
        }
        __syncthreads(); // (3) pivot row of step j in LDS, deferred update of step j-1 complete

        // ---- once per panel: the window moves to [HW_JS, HW_JS + HW_W) ------------------------------------------------------
        if (regcols && j == HW_JS) {
            const u4_t t = *(const u4_t *)(slab + tp * HW_RS + HW_JS + 4 * cg);  // columns 120..135: one quad per thread
            __syncthreads();
            *(u4_t *)(slab + tp * HW_RS + 4 * cg) = t;
            if (rg >= 0) {
#pragma unroll
                for (int s = 0; s < HW_RQ; ++s) {
                    const int c0 = HW_W + 4 * (3 * s + rg);
                    *(u4_t *)(slab + tp * HW_RS + (c0 - HW_JS)) = (u4_t){rc[4 * s], rc[4 * s + 1], rc[4 * s + 2], rc[4 * s + 3]};
                }
            }
            __syncthreads();
            wb = HW_JS;
        }

        // ---- critical part of step j: positions, multipliers, column j+1, next local candidates ---------------------------
        const int piv_pos = misc[1];
        unsigned k0 = 0, k1 = 0;
        if (tid < HP_PAIRS) {
            unsigned *row = slab + tp * HW_RS - wb;
            const int jn = j + 1 < cols ? j + 1 : j;            // (one LDS round trip for the whole part: see hgetf2_lds_kernel)
            const unsigned dw = row[j], xw0 = row[jn];
            const unsigned uj = ucur[j], ujn = ucur[jn];
            int pa = pos[2 * tp], pb = pos[2 * tp + 1];
            if (pa == piv_pos) pa = j; else if (pa == j) pa = piv_pos;   // hgetf2_kernel.cu:92-98 as bookkeeping
            if (pb == piv_pos) pb = j; else if (pb == j) pb = piv_pos;
            pos[2 * tp] = pa; pos[2 * tp + 1] = pb;
            const unsigned rmask = (pa > j ? 0x0000FFFFu : 0u) | (pb > j ? 0xFFFF0000u : 0u);
            maskbuf[tp] = rmask;
            if (rmask) {
                const _Float16 ujj = bits_h(uj & 0xFFFFu);
                h2_t m2;
                m2.x = hdiv_ieee(bits_h(dw & 0xFFFFu), ujj);   // :108
                m2.y = hdiv_ieee(bits_h(dw >> 16), ujj);
                const unsigned mw = __builtin_bit_cast(unsigned, m2);
                mbuf[tp] = mw;
                row[j] = (mw & rmask) | (dw & ~rmask); // :109
                if (j + 1 < cols) {
                    const unsigned xw = xw0;
                    const unsigned y = pk_elim(xw, m2, ujn);
                    const unsigned nw = (y & rmask) | (xw & ~rmask);
                    row[j + 1] = nw;
                    if (pa > j) k0 = row_key32(nw & 0xFFFFu, pa, j + 1);
                    if (pb > j) k1 = row_key32(nw >> 16, pb, j + 1);
                }
            }
            const unsigned km = k0 > k1 ? k0 : k1;
            const unsigned wm = wave_max_u32(km);
            if (lane == 0) { wred[wave] = wm; if (wm == 0) misc[4 + wave] = -1; }
            if (km == wm && wm != 0) misc[4 + wave] = (k0 == wm) ? 2 * tp : 2 * tp + 1;
        }
        __syncthreads(); // (1) next candidate known to everyone
        gmax = wred[0] > wred[1] ? wred[0] : wred[1];
    }

    if (a.moved && tid < R) {
        const int p = pos[tid];
        const long long r = row0 + tid;
        if (p >= 0 && p != (int)r) {
            const int i = atomicAdd(&a.moved->n, 1);
            if (i < LASWP_MAXMOVED) {
                a.moved->src[i] = a.ipiv_offset + (int)r;
                a.moved->dst[i] = a.ipiv_offset + p;
            }
        }
    }
}


// ---- gate: lets work on another stream follow the pivot kernel while it runs -------------------------------------------------
__global__ void hgetf2_gate_kernel(const unsigned long long *progress, int *timeouts, unsigned seq, unsigned target,
                                   unsigned long long max_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned long long v = __hip_atomic_load(progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == seq && (unsigned)v >= target) break;
        // never hang: once anything has given up (the factorization is reported as failed, -4) nothing waits any more
        if (__hip_atomic_load(timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        // A gate that expires lets work through that relies on pivots which are NOT final: that is a failure of the whole
        // factorization, flagged exactly like a give-up inside the pivot kernel (every later gate then leaves at once and
        // mpf_factor_dev / mpf_factor_dist return -4), never a silent pass.
        if (__builtin_amdgcn_s_memrealtime() - t0 > max_ticks) { atomicAdd(timeouts, 1); break; }
        __builtin_amdgcn_s_sleep(8);
    }
}
int launch_hgetf2_gate(mpf_ctx *c, int target) {
    hgetf2_gate_kernel<<<1, 64, 0, c->stream>>>(&c->ws->hp_progress, &c->ws->hp_timeouts, c->hp_seq, (unsigned)target,
                                                (unsigned long long)c->tune.hp_gate_ticks /* 100 MHz clock: default 2 s */);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

// ---- element-wise helpers (MPF.cu:20-25 and the contract's division) ---------------------------
__global__ void double_to_fp16_kernel(const double *in, unsigned short *out, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = double_to_fp16_bits(in[i]);
}
__global__ void hdiv_kernel(const unsigned short *a, const unsigned short *b, unsigned short *q, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) q[i] = h_bits(hdiv_ieee(bits_h(a[i]), bits_h(b[i])));
}

int launch_double_to_fp16(mpf_ctx *c, const double *in, uint16_t *out, int64_t n) {
    if (n <= 0) return 0;
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    double_to_fp16_kernel<<<blocks, 256, 0, c->stream>>>(in, out, n);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}
int launch_hdiv(mpf_ctx *c, const uint16_t *a, const uint16_t *b, uint16_t *q, int64_t n) {
    if (n <= 0) return 0;
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hdiv_kernel<<<blocks, 256, 0, c->stream>>>(a, b, q, n);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

// The LDS kernel's workgroups hand pivot candidates to each other inside one launch, so ALL of them must be resident at
// once (the reference gets that guarantee from its cooperative launch, MPF.cu:129-133).  A workgroup takes a CU's whole LDS
// (137 KB), so the bound is one per CU -- asked of the runtime, not assumed.  Shapes beyond it, and devices that cannot
// hold the grid, take the generic path (fp16_panel_generic.hip), which never spins.
static int hp_setup(mpf_ctx *c) {
    if (c->hp_resident_per_cu >= 0 && (c->attr_done & ATTR_HP)) return 0;
    MPF_HIP_TRY(c, hipSetDevice(c->device));   // function attributes belong to the device they were set on
    MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)hgetf2_lds_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)hgetf2_lds_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifdef MPF_PROBE
    MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)hgetf2_lds_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
    MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)hgetf2_win_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)hgetf2_lds_kernel<256, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)hgetf2_lds_kernel<256>, HP_T, HpCarve<256>::LDS_BYTES) != hipSuccess)
        per_cu = 0;
    c->hp_resident_per_cu = per_cu;
    int win_per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&win_per_cu, (const void *)hgetf2_win_kernel, HP_T, HwCarve::LDS_BYTES) != hipSuccess)
        win_per_cu = 0;
    c->hp_win_per_cu = win_per_cu;
    // What a workgroup of the gated interchange kernel leaves of its CU (the room is derived, not a measured
    // constant).  A CU has 160 KB of LDS and 512 registers per SIMD lane; a 512-thread pivot workgroup puts two waves on every SIMD,
    // a 256-thread waiter one.  LDS is allocated in CONTIGUOUS blocks: a waiter that arrived behind a 128-KB update workgroup
    // keeps its 12 KB in the middle of the CU's LDS after that workgroup has retired, and the holes on either side are what a pivot
    // workgroup can get -- in the worst case two holes of (160 KB - waiter) / 2.  By sizes alone a 137-KB full-slab workgroup fits
    // beside a 12-KB waiter; by holes it does not, and neither do 76-KB window workgroups: that is round 4's measured "192 + 64
    // run, 208 + 64 never" (DESIGN 4.1), now the result of this computation.
    c->hp_full_beside_waiter = c->hp_win_beside_waiter = 0;
    {
        hipFuncAttributes ff, fw;
        int wl = 0, wv = 0, wt = 0;
        if (laswp_gated_footprint(&wl, &wv, &wt) == 0 && hipFuncGetAttributes(&ff, (const void *)hgetf2_lds_kernel<256>) == hipSuccess &&
            hipFuncGetAttributes(&fw, (const void *)hgetf2_win_kernel) == hipSuccess) {
            auto gran = [](int r) { return (r + 7) / 8 * 8; };
            auto fits = [&](int n, int lds_one, int regs_one) {
                const long long hole = (160 * 1024 - (long long)wl) / 2;                       // worst case: the waiter sits in the middle
                const long long by_holes = lds_one > 0 ? 2 * (hole / lds_one) : n;
                const int regs = n * (HP_T / 256) * gran(regs_one) + (wt / 256) * gran(wv);   // per SIMD lane
                return n <= by_holes && regs <= 512 && n * (HP_T / 64) + wt / 64 <= 32;
            };
            for (int nfit = per_cu; nfit >= 1; --nfit) if (fits(nfit, HpCarve<256>::LDS_BYTES + (int)ff.sharedSizeBytes, ff.numRegs)) { c->hp_full_beside_waiter = nfit; break; }
            for (int nfit = win_per_cu; nfit >= 1; --nfit) if (fits(nfit, HwCarve::LDS_BYTES + (int)fw.sharedSizeBytes, fw.numRegs)) { c->hp_win_beside_waiter = nfit; break; }
        }
    }
    c->attr_done |= ATTR_HP;
    return 0;
}
// workgroups of each form that can be resident beside `waiters` waiting workgroups (one per CU: the dispatcher spreads them)
static void hp_capacity(mpf_ctx *c, int waiters, long long &full, long long &win) {
    const int cus = c->num_cus > 0 ? c->num_cus : 0;
    const int w = waiters < 0 ? 0 : (waiters > cus ? cus : waiters);
    full = (long long)c->hp_resident_per_cu * (cus - w) + (long long)c->hp_full_beside_waiter * w;
    win = (long long)c->hp_win_per_cu * (cus - w) + (long long)c->hp_win_beside_waiter * w;
    if (full > HP_MAXG) full = HP_MAXG;
    if (win > HP_MAXG) win = HP_MAXG;
}
// 0 = full slab, 1 = column window, -1 = neither form has room.  need_full: the caller wants the fp16 panel (only the full-slab form has it)
static int hp_choose_form(mpf_ctx *c, int rows, bool need_full, int waiters, int prefer_window_rows) {
    long long cap_full = 0, cap_win = 0;
    hp_capacity(c, waiters, cap_full, cap_win);
    const long long G = ((long long)rows + HP_R - 1) / HP_R;
    const bool full_ok = G <= cap_full, win_ok = !need_full && G <= cap_win;
    const int opt = c->tune.hp_window;      // 0 never, n >= 1 from n rows on, -1 automatic
    const bool want_win = opt == 0 ? false : (opt > 0 ? rows >= opt : (prefer_window_rows > 0 && rows >= prefer_window_rows));
    if (want_win && win_ok) return 1;
    if (full_ok) return 0;
    if (win_ok && opt != 0) return 1;
    return -1;
}
bool hgetf2_fits_beside(mpf_ctx *c, int rows, int cols, int waiters) {
    if (hp_setup(c) != 0 || cols > HP_MAXCOLS) return false;
    return hp_choose_form(c, rows, false, waiters, 0) >= 0;
}
long long hgetf2_capacity_rows(mpf_ctx *c, int waiters, int form) {   // form: 0 = either, 1 = full slab, 2 = column window
    if (hp_setup(c) != 0) return 0;
    long long cap_full = 0, cap_win = 0;
    hp_capacity(c, waiters, cap_full, cap_win);
    if (c->tune.hp_window == 0) cap_win = 0;
    if (form == 1) return cap_full * HP_R;
    if (form == 2) return cap_win * HP_R;
    return (cap_full > cap_win ? cap_full : cap_win) * HP_R;
}
int hp_query_residency(mpf_ctx *c) { return hp_setup(c); }   // fills c->hp_resident_per_cu (mpf_factor_dist: the ranks agree on it)
bool hgetf2_lds_eligible(mpf_ctx *c, int rows, int cols) {
    if (hp_setup(c) != 0) return false;
    if (cols > HP_MAXCOLS) return false;
    const long long resident = (long long)c->hp_resident_per_cu * (c->num_cus > 0 ? c->num_cus : 0);
    const long long G = ((long long)rows + HP_R - 1) / HP_R;
    return G <= HP_MAXG && G <= resident;
}

int launch_hgetf2(mpf_ctx *c, const double *A64, int64_t lda, uint16_t *P16, int64_t ld16, int rows, int cols,
                  int ipiv_offset, int *d_ipiv, uint16_t *out16, int64_t ldo, MovedList *moved, int waiters, int prefer_window_rows) {
    if (rows < 1 || cols < 1 || cols > rows) { c->err = "hgetf2: need 1 <= cols <= rows"; return -1; }
    if (!hgetf2_lds_eligible(c, rows, cols)) { c->err = "hgetf2: shape not covered by the LDS-resident kernel (caller must take the generic path)"; return -1; }
    // 256 rows per workgroup (137 KB of LDS: the workgroup has its CU to itself).  Measured against 128-row slabs under the fp64
    // mode's running update, where they share CUs with update workgroups: a hand-off chain on CUs of its own keeps its idle latency
    // (round 2: the look-ahead chain took 266 ms instead of 392 ms per factorization; round 5: 444 against 477 ms at N = 32768).
    // Round 5, option hp_half_slabs: in the fp16 modes' schedules and the step operator (callers that pass prefer_window_rows = 0:
    // nothing update-bound beside the chain) 256-column panels of at most 16384 rows DO take 128-row slabs -- a workgroup's share of
    // the per-column update of the slab, the larger part of a column step at this width, halves: 1.77 against 2.14 us per column at
    // 8192 x 256 alone, - 4 % on the fp16 mode at N = 8192 and 16384 (profiles/r05_hp_r128.log); narrower panels lose with it.
    int R = 256;
    if (c->tune.hp_half_slabs && prefer_window_rows == 0 && cols > 128 && rows > 256 && rows <= c->tune.hp_half_slabs_rows &&
        (rows + 127) / 128 <= (c->num_cus > 0 ? c->num_cus - (waiters > 0 ? waiters : 0) : 0)) R = 128;
#ifdef MPF_PROBE
    if (c->tune.hp_r256_upto != (1 << 30)) R = (rows <= c->tune.hp_r256_upto || rows > 128 * HP_MAXG) ? 256 : 128;
#endif
    const int G = (rows + R - 1) / R;
    if (G > HP_MAXG || (c->num_cus > 0 && G > c->num_cus)) { c->err = "hgetf2: more workgroups than CUs"; return -1; }
    // full slab (137 KB of LDS, a CU per workgroup; the only form that keeps the fp16 panel) or column window (76 KB, two per CU):
    // whichever has room beside the workgroups that will wait for this launch -- every pivot workgroup must be resident at once
    const int form = R == 256 ? hp_choose_form(c, rows, !A64 || P16 || out16, waiters, prefer_window_rows) : 0;
    if (form < 0) { c->err = "hgetf2: the panel's workgroups do not all fit on the device beside the kernels that wait for it (panel too tall for this form / option hp_window = 0)"; return -1; }
    // Hand-off granules (candidates and rows) carry the launch sequence number in their tags: nothing is cleared between
    // launches.  The moved-row counters of a factorization's per-panel lists are zeroed once, at its start.
    HpArgs a;
    a.A64 = A64; a.lda = lda; a.P16 = P16; a.ld16 = ld16; a.out16 = out16; a.ldo = ldo;
    a.rows = rows; a.cols = cols; a.ipiv_offset = ipiv_offset; a.ipiv = d_ipiv; a.ws = c->ws;
    c->hp_seq = (c->hp_seq + 1) & 0x3FFFFFu;
    if (c->hp_seq == 0) c->hp_seq = 1;             // tag 0 is what a never-written granule holds
    a.tag_base = c->hp_seq << 9;
    a.seq = c->hp_seq;
    a.signal = (c->tune.gate_wait_value && c->hp_signal) ? c->hp_signal : nullptr;
    a.moved = moved;
    const bool own_list = moved && c->lists && moved >= c->lists && moved < c->lists + c->lists_cap;
    if (moved && !own_list) MPF_HIP_TRY(c, hipMemsetAsync(&moved->n, 0, sizeof(int), c->stream));
    // single-XCD form: the slabs + the waiters' share of one XCD's CUs must fit that XCD
    const int per_xcd = c->num_cus / 8;
    // (hp_local_xcd = 1: only for callers without an update-bound schedule beside them -- the fp16 modes' schedules and the step operator, which pass
    //  prefer_window_rows = 0; under the fp64 mode's running update the form's larger grid waits for CUs on every XCD and loses: N = 8192, nb = 128: + 3 %)
    const bool local = (c->tune.hp_local_xcd == 2 || (c->tune.hp_local_xcd == 1 && prefer_window_rows == 0)) && form == 0 && R == 256 && G > 1 && per_xcd >= 8 && G + (waiters + 7) / 8 <= per_xcd && c->hp_resident_per_cu >= 1
#ifdef MPF_PROBE
                       && !c->tune.hp_stamp
#endif
        ;
    a.local_G = local ? G : 0;
    a.acq_fence = c->tune.hp_acq_fence;
    a.spin_limit = (unsigned)c->tune.hp_spin_limit;
    // Two pivot kernels at once on one device (two contexts of this process) could each hold part of the CUs and starve each
    // other's hand-offs until the bounded waits give up: every launch is ordered behind the previous one on the same device.
    // (Another PROCESS on the GPU is not covered: MPF_SAFE_PIVOTS=1 / pivot_path = 1 is the setting for that.)
    static std::mutex hp_mu;
    static hipEvent_t hp_last[64] = {nullptr};
    {
        std::lock_guard<std::mutex> lk(hp_mu);
        const int dv = c->device >= 0 && c->device < 64 ? c->device : 0;
        if (!hp_last[dv]) MPF_HIP_TRY(c, hipEventCreateWithFlags(&hp_last[dv], hipEventDisableTiming));
        else MPF_HIP_TRY(c, hipStreamWaitEvent(c->stream, hp_last[dv], 0));
#ifdef MPF_PROBE
        if (c->tune.hp_stamp && R == 256) hgetf2_lds_kernel<256, true><<<G, HP_T, HpCarve<256>::LDS_BYTES, c->stream>>>(a);
        else
#endif
        if (R == 128) hgetf2_lds_kernel<128><<<G, HP_T, HpCarve<128>::LDS_BYTES, c->stream>>>(a);
        else
        // the column-window form (76 KB of LDS: shares its CU) covers what the factorization chain asks for: an fp64 source,
        // no fp16 copy of the factored panel
        if (form == 1) hgetf2_win_kernel<<<G, HP_T, HwCarve::LDS_BYTES, c->stream>>>(a);
        else if (local) hgetf2_lds_kernel<256, false, true><<<8 * (G + 2), HP_T, HpCarve<256>::LDS_BYTES, c->stream>>>(a);
        else
        hgetf2_lds_kernel<256><<<G, HP_T, HpCarve<256>::LDS_BYTES, c->stream>>>(a);
        MPF_HIP_TRY(c, hipGetLastError());
        MPF_HIP_TRY(c, hipEventRecord(hp_last[dv], c->stream));
    }
    return 0;
}

// fp16 pivot panel for gfx950: replaces reference MPF.cu:108-159 (panel extract, double_to_fp16_block,
// HGETF2_kernel launch, pivot globalisation) and hgetf2_kernel.cu:15-120.
//
// Design (MI355X-first, not a translation of the cooperative-grid reference kernel):
//   * The panel never exists in HBM as fp16.  Each workgroup converts its HP_R = 256 rows of the fp64
//     panel on load (contract C1) and keeps them as an LDS-resident slab, column-major, two rows packed
//     per dword, column stride 129 dwords (odd => conflict-free ds_read/ds_write_b32 across a wave).
//   * Rows are never moved.  A row swap j <-> p of the reference (hgetf2_kernel.cu:92-98) is pure
//     bookkeeping: every physical row carries its current logical position pos[]; the pivot search uses
//     pos to reproduce the reference's tie-break exactly (lowest 256-row block of t = pos - j, then the
//     smallest 8-bit bit-reversed lane index inside the block -- what the strict-'>' binary tree of
//     hgetf2_kernel.cu:47-56 and the serial block scan :73-78 amount to).
//   * One hand-off round per column: each workgroup publishes {epoch | |a| | ~tiekey} as ONE 8-byte
//     write-through granule plus its candidate row (512 B, write-through, drained before the granule);
//     every workgroup sweeps all granules, picks the same global winner and reads the winner's row
//     with sc1 loads.  No grid barrier (the reference needs five per column).
//   * Arithmetic is contract C2: v_pk_mul_f16 then v_pk_add_f16 (never fma), division = IEEE fp32
//     quotient rounded once to fp16.
#include "mpf_internal.h"

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned short h_bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ _Float16 bits_h(unsigned b) { return __builtin_bit_cast(_Float16, (unsigned short)b); }

// fp16_utils.h:15-23 double_to_fp16 (contract C1)
__device__ __forceinline__ unsigned short double_to_fp16_bits(double x) {
    float xf = (float)x;
    const float FP16_MAX = 65504.0f;
    const float FP16_MIN_POS = 6.10352e-05f;
    if (xf > FP16_MAX) xf = FP16_MAX;
    else if (xf < -FP16_MAX) xf = -FP16_MAX;
    if (xf > -FP16_MIN_POS && xf < FP16_MIN_POS) xf = 0.0f;
    return h_bits((_Float16)xf);
}

// IEEE quotient of two fp16 values rounded once to fp16 (the '/' of hgetf2_kernel.cu:108).  The fp32
// operands are hidden from the optimiser so the division stays a correctly rounded fp32 division
// (24 >= 2*11+2 bits: rounding its result to fp16 equals rounding the exact quotient).
__device__ __forceinline__ _Float16 hdiv_ieee(_Float16 a, _Float16 b) {
    float fa = (float)a, fb = (float)b;
    asm volatile("" : "+v"(fa), "+v"(fb));
    return (_Float16)(fa / fb);
}

__device__ __forceinline__ unsigned bitrev8(unsigned x) { return __brev(x) >> 24; }
// order in which equal maxima are preferred (smaller wins); an involution on t
__device__ __forceinline__ unsigned tie_key(unsigned t) { return (t & ~255u) | bitrev8(t & 255u); }

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long w = __shfl_xor(v, o);
        v = w > v ? w : v;
    }
    return v;
}

struct HpArgs {
    const double *A64; long long lda;     // fp64 source panel (or null)
    unsigned short *P16; long long ld16;  // fp16 source panel, factored in place (or null)
    unsigned short *out16; long long ldo; // optional factored fp16 output (rows physically swapped)
    int rows, cols, ipiv_offset;
    int *ipiv;
    MpfWorkspace *ws;
    int acq_fence;                        // 1: agent-scope acquire after the poll (debug aid)
};

constexpr unsigned HP_SPIN_LIMIT = 1u << 21;

__global__ __launch_bounds__(HP_T) void hgetf2_lds_kernel(HpArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long *wred = (unsigned long long *)smem_raw;          // 4 x u64
    int *misc = (int *)(smem_raw + 32);                                 // 8 ints
    int *pos = (int *)(smem_raw + 64);                                  // HP_R ints
    unsigned *urow = (unsigned *)(smem_raw + 64 + HP_R * 4);            // 128 dwords
    unsigned *slab = (unsigned *)(smem_raw + 64 + HP_R * 4 + 512);      // cols x HP_RPD dwords

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, G = gridDim.x;
    const int rows = a.rows, cols = a.cols;
    const int tp = tid & 127, cg = tid >> 7; // row pair / column group of the slab work split
    const long long row0 = (long long)g * HP_R;

    // ---- load + convert the slab (MPF.cu:108-121 fused) -----------------------------------
    {
        const long long ra = row0 + 2 * tp, rb = ra + 1;
        for (int c = cg; c < cols; c += 2) {
            unsigned lo = 0, hi = 0;
            if (a.A64) {
                if (ra < rows) lo = double_to_fp16_bits(a.A64[ra + (long long)c * a.lda]);
                if (rb < rows) hi = double_to_fp16_bits(a.A64[rb + (long long)c * a.lda]);
            } else {
                if (ra < rows) lo = a.P16[ra + (long long)c * a.ld16];
                if (rb < rows) hi = a.P16[rb + (long long)c * a.ld16];
            }
            slab[c * HP_RPD + tp] = lo | (hi << 16);
        }
        const long long r = row0 + tid;
        pos[tid] = r < rows ? (int)r : -1;
        if (tid == 0) misc[2] = 0; // aborted flag
    }
    __syncthreads();

    int prev_p = -1, prev_j = -1; // swap of the previous step, applied to pos[] by the row's owner
    for (int j = 0; j < cols; ++j) {
        // deferred pos update of step j-1 (owner thread only; readers are behind the barriers below)
        int mypos = pos[tid];
        if (prev_j >= 0) {
            if (mypos == prev_p) mypos = prev_j;
            else if (mypos == prev_j) mypos = prev_p;
            pos[tid] = mypos;
        }
        // ---- phase A: local argmax of |a[:, j]| over logical positions >= j  (:32-62) ----------
        unsigned long long key = 0;
        if (mypos >= j) {
            unsigned w = slab[j * HP_RPD + (tid >> 1)];
            unsigned hb = (tid & 1) ? (w >> 16) : (w & 0xFFFFu);
            key = ((unsigned long long)(hb & 0x7FFFu) << 32) | (0xFFFFFFFFu - tie_key((unsigned)(mypos - j)));
        }
        unsigned long long wmax = wave_max_u64(key);
        if (lane == 0) wred[wave] = wmax;
        if (tid == 0) misc[0] = -1;
        __syncthreads();
        unsigned long long gmax = wred[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) gmax = wred[i] > gmax ? wred[i] : gmax;
        if (key != 0 && key == gmax) misc[0] = tid; // keys are unique (pos is unique)
        __syncthreads();
        const int cr = misc[0]; // local candidate row or -1

        int piv_pos;
        if (G > 1) {
            const int par = j & 1;
            const unsigned epoch = (unsigned)(j + 1);
            // ---- phase B: publish candidate row, drain, then the granule -------------------
            if (tid < 128) {
                unsigned v = 0;
                if (cr >= 0) {
                    const int c0 = 2 * tid, c1 = c0 + 1;
                    unsigned h0 = 0, h1 = 0;
                    if (c0 < cols) { unsigned w = slab[c0 * HP_RPD + (cr >> 1)]; h0 = (cr & 1) ? (w >> 16) : (w & 0xFFFFu); }
                    if (c1 < cols) { unsigned w = slab[c1 * HP_RPD + (cr >> 1)]; h1 = (cr & 1) ? (w >> 16) : (w & 0xFFFFu); }
                    v = h0 | (h1 << 16);
                }
                __hip_atomic_store(&a.ws->rowbuf[par][g][tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store(&a.ws->cand[par][g], ((unsigned long long)epoch << 48) | gmax,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ---- phase C: wave 0 sweeps all granules, picks the winner, fetches its row ----
            if (wave == 0) {
                unsigned long long best = 0;
                const bool aborted = misc[2] != 0;
                for (unsigned spins = 0;; ++spins) {
                    bool ok = true;
                    best = 0;
#pragma unroll
                    for (int i = 0; i < HP_MAXG / 64; ++i) {
                        const int idx = lane + 64 * i;
                        if (idx < G) {
                            unsigned long long x =
                                __hip_atomic_load(&a.ws->cand[par][idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok &= (unsigned)(x >> 48) == epoch;
                            unsigned long long comb = ((x & 0xFFFFFFFFFFFFull) << 8) | (unsigned)idx;
                            best = comb > best ? comb : best;
                        }
                    }
                    if (__all(ok) || aborted) break;
                    if (spins > HP_SPIN_LIMIT) { // give up: flag it, never hang
                        if (lane == 0) { atomicAdd(&a.ws->flags[0], 1); misc[2] = 1; }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                if (a.acq_fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                best = wave_max_u64(best);
                const int gw = (int)(best & 255u);
                const unsigned low = (unsigned)((best >> 8) & 0xFFFFFFFFu);
                const int p = j + (int)tie_key(0xFFFFFFFFu - low);
                urow[lane] = __hip_atomic_load(&a.ws->rowbuf[par][gw][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                urow[lane + 64] = __hip_atomic_load(&a.ws->rowbuf[par][gw][lane + 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) {
                    misc[1] = p;
                    if (g == 0) a.ipiv[j] = p + 1 + a.ipiv_offset; // hgetf2_kernel.cu:80-81 + MPF.cu:152
                }
            }
            __syncthreads();
            piv_pos = misc[1];
        } else {
            // single workgroup: the local winner is the global one
            const unsigned low = (unsigned)(gmax & 0xFFFFFFFFu);
            piv_pos = j + (int)tie_key(0xFFFFFFFFu - low);
            if (tid < 128) {
                const int c0 = 2 * tid, c1 = c0 + 1;
                unsigned h0 = 0, h1 = 0;
                if (c0 < cols) { unsigned w = slab[c0 * HP_RPD + (cr >> 1)]; h0 = (cr & 1) ? (w >> 16) : (w & 0xFFFFu); }
                if (c1 < cols) { unsigned w = slab[c1 * HP_RPD + (cr >> 1)]; h1 = (cr & 1) ? (w >> 16) : (w & 0xFFFFu); }
                urow[tid] = h0 | (h1 << 16);
            }
            if (tid == 0) a.ipiv[j] = piv_pos + 1 + a.ipiv_offset;
            __syncthreads();
        }

        // ---- phase D: elimination on rows whose position after the swap is > j  (:104-115) ----
        {
            int pa = pos[2 * tp], pb = pos[2 * tp + 1]; // positions before this step's swap
            if (pa == piv_pos) pa = j; else if (pa == j) pa = piv_pos;
            if (pb == piv_pos) pb = j; else if (pb == j) pb = piv_pos;
            const unsigned mask = (pa > j ? 0x0000FFFFu : 0u) | (pb > j ? 0xFFFF0000u : 0u);
            if (mask) {
                const unsigned ujw = urow[j >> 1];
                const _Float16 ujj = bits_h((j & 1) ? (ujw >> 16) : (ujw & 0xFFFFu));
                const unsigned dw = slab[j * HP_RPD + tp];
                h2_t m2;
                m2.x = hdiv_ieee(bits_h(dw & 0xFFFFu), ujj);
                m2.y = hdiv_ieee(bits_h(dw >> 16), ujj);
                if (cg == 0) {
                    const unsigned mw = (unsigned)h_bits(m2.x) | ((unsigned)h_bits(m2.y) << 16);
                    slab[j * HP_RPD + tp] = (mw & mask) | (dw & ~mask); // :109
                }
                for (int c = j + 1 + cg; c < cols; c += 2) {
                    const unsigned uw = urow[c >> 1];
                    const _Float16 u = bits_h((c & 1) ? (uw >> 16) : (uw & 0xFFFFu));
                    h2_t u2; u2.x = u; u2.y = u;
                    const unsigned xw = slab[c * HP_RPD + tp];
                    const h2_t x = __builtin_bit_cast(h2_t, xw);
                    const h2_t t = m2 * u2;      // rounded product   (:113, no fma)
                    const h2_t y = x - t;        // rounded difference
                    const unsigned yw = __builtin_bit_cast(unsigned, y);
                    slab[c * HP_RPD + tp] = (yw & mask) | (xw & ~mask);
                }
            }
        }
        prev_p = piv_pos; prev_j = j;
        __syncthreads();
    }

    // final pos update, then optional output of the factored panel with rows where the reference
    // leaves them (row r of the slab ends at logical position pos[r])
    {
        int mypos = pos[tid];
        if (prev_j >= 0) {
            if (mypos == prev_p) mypos = prev_j;
            else if (mypos == prev_j) mypos = prev_p;
            pos[tid] = mypos;
        }
    }
    unsigned short *out = a.out16 ? a.out16 : a.P16;
    const long long ldo = a.out16 ? a.ldo : a.ld16;
    if (out) {
        __syncthreads();
        const int pa = pos[2 * tp], pb = pos[2 * tp + 1];
        for (int c = cg; c < cols; c += 2) {
            const unsigned w = slab[c * HP_RPD + tp];
            if (pa >= 0) out[pa + (long long)c * ldo] = (unsigned short)(w & 0xFFFFu);
            if (pb >= 0) out[pb + (long long)c * ldo] = (unsigned short)(w >> 16);
        }
    }
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

int launch_hgetf2(mpf_ctx *c, const double *A64, int64_t lda, uint16_t *P16, int64_t ld16, int rows, int cols,
                  int ipiv_offset, int *d_ipiv, uint16_t *out16, int64_t ldo) {
    if (rows < 1 || cols < 1 || cols > rows) { c->err = "hgetf2: need 1 <= cols <= rows"; return -1; }
    if (cols > HP_MAXCOLS) { c->err = "hgetf2: panel width > 256 is not supported"; return -1; }
    const int G = (rows + HP_R - 1) / HP_R;
    if (G > HP_MAXG || (c->num_cus > 0 && G > c->num_cus)) {
        c->err = "hgetf2: panel has more rows than the LDS-resident design covers (256 rows x #CUs)";
        return -1;
    }
    static bool attr_set = false;
    const size_t lds = 64 + HP_R * 4 + 512 + (size_t)cols * HP_RPD * 4;
    if (!attr_set) {
        MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)hgetf2_lds_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (G > 1) MPF_HIP_TRY(c, hipMemsetAsync(c->ws, 0, HP_SYNC_BYTES, c->stream));
    HpArgs a;
    a.A64 = A64; a.lda = lda; a.P16 = P16; a.ld16 = ld16; a.out16 = out16; a.ldo = ldo;
    a.rows = rows; a.cols = cols; a.ipiv_offset = ipiv_offset; a.ipiv = d_ipiv; a.ws = c->ws;
    static int fence = -1;
    if (fence < 0) { const char *e = getenv("MPF_HP_ACQ_FENCE"); fence = (e && e[0] == '1') ? 1 : 0; }
    a.acq_fence = fence;
    hgetf2_lds_kernel<<<G, HP_T, lds, c->stream>>>(a);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

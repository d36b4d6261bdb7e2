// Generic fp16 pivot panel: the same reference semantics as fp16_panel.hip (MPF.cu:108-159, hgetf2_kernel.cu:15-120,
// numeric contract C1 / C2) for every shape the LDS-resident kernel does not cover:
//   * panels wider than 256 columns (MPF() takes any r, MPF.cu:66,100-102),
//   * panels taller than 256 rows x #CUs (the reference allows 1024 blocks = 262 144 rows, hgetf2_kernel.cu:6; here any height),
//   * devices where the LDS kernel's workgroups cannot all be resident (shared / partitioned GPUs), or when the caller asks
//     for a pivot path that never spins on other workgroups (mpf_opts.pivot_path = 1, MPF_SAFE_PIVOTS=1).
// The fp16 panel lives in HBM (as in the reference) and every column step is TWO ordinary launches (round 3; four before:
// search, pick + swap, scale, rank-1 update) -- so the only inter-workgroup synchronisation is the kernel boundary: nothing
// spins, nothing needs co-residency:
//   search j   one thread per candidate row: the multiplier of column j-1 is stored (the scale step of column j-1), the
//              row's key for column j goes into the block maximum; block 0 also commits the two rows step j-1 exchanged;
//   update j   every workgroup picks the winner from the block maxima itself, then applies the rank-1 update to its rows x
//              64 columns.  The two rows of the interchange are NOT written in place (other workgroups of the same launch
//              still read the old pivot row and the old row j): their new contents go to two side rows, which the next
//              search launch reads where it needs them and copies into the panel.
// It is the slow, always-valid path (the reference itself pays five grid barriers per column); per-element arithmetic and the
// pivot tie-break are bit-identical to the LDS kernel and the oracle.  Option generic_fused = 0 keeps the four-launch form.
#include "mpf_internal.h"
#include "fp16_device.h"

constexpr int GP_T = 256;      // threads per workgroup == the reference's block size, which defines the tie-break blocks
constexpr int GP_CH = 64;      // columns per workgroup of the rank-1 update

// MPF.cu:108-121: strided fp64 panel -> packed fp16 panel
__global__ __launch_bounds__(GP_T) void gp_convert_kernel(const double *__restrict__ A, long long lda, unsigned short *__restrict__ P,
                                                         long long ldp, int rows) {
    const long long r = (long long)blockIdx.x * GP_T + threadIdx.x;
    const long long c = blockIdx.y;
    if (r < rows) P[r + c * ldp] = double_to_fp16_bits(A[r + c * lda]);
}

__device__ __forceinline__ unsigned long long gp_block_max(unsigned long long key, unsigned long long *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)key, o), hi = (unsigned)__shfl_xor((int)(unsigned)(key >> 32), o);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        key = w > key ? w : key;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = key;
    __syncthreads();
    unsigned long long m = red[0];
#pragma unroll
    for (int i = 1; i < GP_T / 64; ++i) m = red[i] > m ? red[i] : m;
    return m;
}

// hgetf2_kernel.cu:32-62: candidate of each 256-row block of t = row - j
__global__ __launch_bounds__(GP_T) void gp_search_kernel(const unsigned short *__restrict__ P, long long ldp, int rows, int j,
                                                        unsigned long long *__restrict__ cand) {
    __shared__ unsigned long long red[GP_T / 64];
    const long long t = (long long)blockIdx.x * GP_T + threadIdx.x;
    const long long r = t + j;
    unsigned long long key = 0;
    if (r < rows) key = pivot_key(P[r + (long long)j * ldp], (unsigned)t);
    const unsigned long long m = gp_block_max(key, red);
    if (threadIdx.x == 0) cand[blockIdx.x] = m;
}

// hgetf2_kernel.cu:68-98 (+ MPF.cu:145-155): winner over the blocks, pivot index out, rows j <-> p swapped in ALL columns
__global__ __launch_bounds__(GP_T) void gp_pickswap_kernel(unsigned short *P, long long ldp, int rows, int cols, int j,
                                                          const unsigned long long *__restrict__ cand, int nblocks, int *ipiv,
                                                          int ipiv_offset) {
    __shared__ unsigned long long red[GP_T / 64];
    unsigned long long key = 0;
    for (int b = threadIdx.x; b < nblocks; b += GP_T) key = cand[b] > key ? cand[b] : key;
    const unsigned long long m = gp_block_max(key, red);
    int p = j + (int)tie_key(0xFFFFFFFFu - (unsigned)(m & 0xFFFFFFFFu));
    if (m == 0 || p < j || p >= rows) p = j; // no candidate at all cannot happen for j < rows; keeps every access in range regardless
    if (threadIdx.x == 0) ipiv[j] = p + 1 + ipiv_offset;
    if (p != j)
        for (int c = threadIdx.x; c < cols; c += GP_T) {
            unsigned short *col = P + (long long)c * ldp;
            const unsigned short a = col[j];
            col[j] = col[p];
            col[p] = a;
        }
}

// hgetf2_kernel.cu:104-109: multipliers of column j
__global__ __launch_bounds__(GP_T) void gp_scale_kernel(unsigned short *P, long long ldp, int rows, int j) {
    const long long r = (long long)blockIdx.x * GP_T + threadIdx.x + j + 1;
    if (r >= rows) return;
    unsigned short *cj = P + (long long)j * ldp;
    cj[r] = h_bits(hdiv_ieee(bits_h(cj[r]), bits_h(cj[j])));
}

// hgetf2_kernel.cu:112-114: a[row][k] -= m * a[j][k], product and difference rounded separately (no fma: contract C2)
__global__ __launch_bounds__(GP_T) void gp_update_kernel(unsigned short *P, long long ldp, int rows, int cols, int j) {
    __shared__ unsigned short u[GP_CH];
    const int k0 = j + 1 + blockIdx.y * GP_CH;
    if (threadIdx.x < GP_CH) u[threadIdx.x] = (k0 + threadIdx.x < cols) ? P[j + (long long)(k0 + threadIdx.x) * ldp] : (unsigned short)0;
    __syncthreads();
    const long long r = (long long)blockIdx.x * GP_T + threadIdx.x + j + 1;
    if (r >= rows) return;
    const _Float16 m = bits_h(P[r + (long long)j * ldp]);
    const int kn = (cols - k0) < GP_CH ? (cols - k0) : GP_CH;
    for (int kk = 0; kk < kn; ++kk) {
        unsigned short *x = P + r + (long long)(k0 + kk) * ldp;
        const _Float16 t = m * bits_h(u[kk]); // v_mul_f16 (file is built with -ffp-contract=off)
        *x = h_bits(bits_h(*x) - t);           // v_sub_f16
    }
}

// ---- two launches per column ---------------------------------------------------------------------------------------------
// side rows: top[c] = new content of row j (the pivot row), piv[c] = new content of the row the pivot came from (old row j
// after elimination; column j holds its multiplier); pinfo[0] = that row's index (== j: no interchange).
__global__ __launch_bounds__(GP_T) void gp_search2_kernel(unsigned short *P, long long ldp, int rows, int cols, int j,
                                                         unsigned long long *__restrict__ cand, const unsigned short *__restrict__ top,
                                                         const unsigned short *__restrict__ piv, const int *__restrict__ pinfo, int has_prev) {
    __shared__ unsigned long long red[GP_T / 64];
    const long long t = (long long)blockIdx.x * GP_T + threadIdx.x;
    const long long r = t + j;
    const int pp = has_prev ? pinfo[0] : -1;                 // the row step j-1 took its pivot from
    const bool swapped = has_prev && pp != j - 1;
    if (has_prev && r < rows && !(swapped && r == pp)) {     // hgetf2_kernel.cu:104-109 for column j-1 (row pp: in its side row)
        unsigned short *cj = P + (long long)(j - 1) * ldp;
        cj[r] = h_bits(hdiv_ieee(bits_h(cj[r]), bits_h(top[j - 1])));
    }
    unsigned long long key = 0;
    if (j < cols && r < rows) key = pivot_key((swapped && r == pp) ? piv[j] : P[r + (long long)j * ldp], (unsigned)t);
    const unsigned long long m = gp_block_max(key, red);
    if (threadIdx.x == 0 && j < cols) cand[blockIdx.x] = m;
    if (blockIdx.x == 0 && has_prev)                         // the interchange of step j-1 lands in the panel (hgetf2_kernel.cu:92-98)
        for (int c = threadIdx.x; c < cols; c += GP_T) {
            P[(j - 1) + (long long)c * ldp] = top[c];
            if (swapped) P[pp + (long long)c * ldp] = piv[c];
        }
}

__global__ __launch_bounds__(GP_T) void gp_update2_kernel(unsigned short *P, long long ldp, int rows, int cols, int j,
                                                         const unsigned long long *__restrict__ cand, int nblocks, int *ipiv, int ipiv_offset,
                                                         unsigned short *__restrict__ top, unsigned short *__restrict__ piv, int *pinfo) {
    __shared__ unsigned long long red[GP_T / 64];
    __shared__ unsigned short u[GP_CH], jrow[GP_CH];
    unsigned long long key = 0;
    for (int b = threadIdx.x; b < nblocks; b += GP_T) key = cand[b] > key ? cand[b] : key;
    const unsigned long long m = gp_block_max(key, red);     // hgetf2_kernel.cu:68-81: every workgroup finds the same winner
    int p = j + (int)tie_key(0xFFFFFFFFu - (unsigned)(m & 0xFFFFFFFFu));
    if (m == 0 || p < j || p >= rows) p = j;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { ipiv[j] = p + 1 + ipiv_offset; pinfo[0] = p; }
    const int c0 = blockIdx.y * GP_CH;
    if (c0 + GP_CH <= j + 1 && blockIdx.x != 0) return;     // columns <= j: only the interchange (first row block) has work there
    if (threadIdx.x < GP_CH) {
        const int k = c0 + threadIdx.x;
        u[threadIdx.x] = k < cols ? P[p + (long long)k * ldp] : (unsigned short)0;       // old pivot row
        jrow[threadIdx.x] = k < cols ? P[j + (long long)k * ldp] : (unsigned short)0;    // old row j
    }
    __syncthreads();
    const _Float16 ujj = bits_h(P[p + (long long)j * ldp]);
    if (blockIdx.x == 0 && threadIdx.x < GP_CH && c0 + threadIdx.x < cols) {   // the two rows of the interchange -> side rows
        const int k = c0 + threadIdx.x;
        top[k] = u[threadIdx.x];
        if (p != j) {
            const _Float16 mp = hdiv_ieee(bits_h(P[j + (long long)j * ldp]), ujj);
            unsigned short v = jrow[threadIdx.x];
            if (k == j) v = h_bits(mp);
            else if (k > j) { const _Float16 t = mp * bits_h(u[threadIdx.x]); v = h_bits(bits_h(v) - t); }
            piv[k] = v;
        }
    }
    const long long r = (long long)blockIdx.x * GP_T + threadIdx.x + j + 1;
    if (r >= rows || r == p) return;
    const _Float16 mr = hdiv_ieee(bits_h(P[r + (long long)j * ldp]), ujj);   // column j itself is stored by the next search launch
    // 16 columns at a time: all loads of a batch in flight before the first store (the element-by-element form waits a
    // memory round trip per column: the same pointer is read and written)
#pragma unroll
    for (int kb = 0; kb < GP_CH; kb += 16) {
        unsigned short xv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = c0 + kb + i;
            xv[i] = (k > j && k < cols) ? P[r + (long long)k * ldp] : (unsigned short)0;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = c0 + kb + i;
            if (k > j && k < cols) {
                const _Float16 t = mr * bits_h(u[kb + i]);                    // v_mul_f16 (file is built with -ffp-contract=off)
                P[r + (long long)k * ldp] = h_bits(bits_h(xv[i]) - t);        // v_sub_f16
            }
        }
    }
}

// LASWP_kernel as the reference has it (MPF.cu:42-59): one thread per column walks the panel's swaps in order.  No plan, no
// list, any number of swaps -- the interchange of the generic path.
__global__ __launch_bounds__(GP_T) void laswp_seq_kernel(double *A, long long lda, long long ncols, int k, int cols,
                                                        const int *ipiv, long long nrows) {
    const long long col = (long long)blockIdx.x * GP_T + threadIdx.x;
    if (col >= ncols) return;
    double *a = A + col * lda;
    for (int pc = 0; pc < cols; ++pc) {
        // (device-scope load: in the pipelined chain the pivots come from a pivot kernel that is still running)
        const long long cur = (long long)k + pc, piv = (long long)__hip_atomic_load(&ipiv[pc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1;
        if (piv != cur && piv >= 0 && piv < nrows) { const double t = a[cur]; a[cur] = a[piv]; a[piv] = t; }
    }
}

int launch_laswp_seq(mpf_ctx *c, double *A, int64_t lda, int64_t ncols, int k, int cols, const int *d_ipiv, int64_t nrows) {
    if (cols < 1 || ncols < 1) return 0;
    laswp_seq_kernel<<<(unsigned)((ncols + GP_T - 1) / GP_T), GP_T, 0, c->stream>>>(A, lda, ncols, k, cols, d_ipiv, nrows);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

int launch_hgetf2_generic(mpf_ctx *c, const double *A64, int64_t lda, uint16_t *P16, int64_t ld16, int rows, int cols,
                          int ipiv_offset, int *d_ipiv, uint16_t *out16, int64_t ldo) {
    if (rows < 1 || cols < 1 || cols > rows) { c->err = "hgetf2: need 1 <= cols <= rows"; return -1; }
    unsigned short *W;
    long long ldw;
    if (P16) { W = P16; ldw = ld16; }
    else if (out16) { W = out16; ldw = ldo; }
    else {
        const size_t need = (size_t)rows * (size_t)cols;
        if (need > c->g16_cap) {
            if (c->g16) hipFree(c->g16);
            c->g16 = nullptr; c->g16_cap = 0;
            MPF_HIP_TRY(c, hipMalloc((void **)&c->g16, need * sizeof(unsigned short)));
            c->g16_cap = need;
        }
        W = c->g16; ldw = rows;
    }
    const int nblk0 = (rows + GP_T - 1) / GP_T;
    // block maxima + the two side rows of the interchange (cols fp16 each) + the pivot row index, in one allocation
    const int need_u64 = nblk0 + (2 * cols + 2) / 4 + 4;
    if (need_u64 > c->gcand_cap) {
        if (c->gcand) hipFree(c->gcand);
        c->gcand = nullptr; c->gcand_cap = 0;
        MPF_HIP_TRY(c, hipMalloc((void **)&c->gcand, (size_t)need_u64 * sizeof(unsigned long long)));
        c->gcand_cap = need_u64;
    }
    if (A64) {
        dim3 g((unsigned)nblk0, (unsigned)cols);
        gp_convert_kernel<<<g, GP_T, 0, c->stream>>>(A64, lda, W, ldw, rows);
    }
    if (c->tune.generic_fused) {
        unsigned short *top = (unsigned short *)(c->gcand + nblk0), *piv = top + cols;
        int *pinfo = (int *)(top + 2 * (size_t)cols);   // 4-byte aligned: top is 8-byte aligned, 2 * cols fp16 values in front
        for (int j = 0; j <= cols; ++j) {
            const int nblk = (rows - j + GP_T - 1) / GP_T > 0 ? (rows - j + GP_T - 1) / GP_T : 1;
            gp_search2_kernel<<<nblk, GP_T, 0, c->stream>>>(W, ldw, rows, cols, j, c->gcand, top, piv, pinfo, j > 0 ? 1 : 0);
            if (j == cols) break;
            const int gb = (rows - j - 1 + GP_T - 1) / GP_T > 0 ? (rows - j - 1 + GP_T - 1) / GP_T : 1;
            dim3 g((unsigned)gb, (unsigned)((cols + GP_CH - 1) / GP_CH));
            gp_update2_kernel<<<g, GP_T, 0, c->stream>>>(W, ldw, rows, cols, j, c->gcand, nblk, d_ipiv, ipiv_offset, top, piv, pinfo);
        }
        MPF_HIP_TRY(c, hipGetLastError());
        return 0;
    }
    for (int j = 0; j < cols; ++j) {
        const int nblk = (rows - j + GP_T - 1) / GP_T;
        gp_search_kernel<<<nblk, GP_T, 0, c->stream>>>(W, ldw, rows, j, c->gcand);
        gp_pickswap_kernel<<<1, GP_T, 0, c->stream>>>(W, ldw, rows, cols, j, c->gcand, nblk, d_ipiv, ipiv_offset);
        const int below = rows - j - 1;
        if (below > 0) {
            const int gb = (below + GP_T - 1) / GP_T;
            gp_scale_kernel<<<gb, GP_T, 0, c->stream>>>(W, ldw, rows, j);
            const int right = cols - j - 1;
            if (right > 0) {
                dim3 g((unsigned)gb, (unsigned)((right + GP_CH - 1) / GP_CH));
                gp_update_kernel<<<g, GP_T, 0, c->stream>>>(W, ldw, rows, cols, j);
            }
        }
    }
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

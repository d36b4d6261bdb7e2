// Row interchanges on the fp64 matrix: replaces LASWP_kernel (reference MPF.cu:42-59, launch :162).
//
// The reference gives every column to one thread that walks the `cols` swaps one after the other
// (a dependent chain of uncoalesced 8-byte accesses).  Here the chain is resolved once per panel:
//   laswp_plan   -- one workgroup turns the sequential swap list into a net permutation of the
//                   <= 2*cols rows it touches: a list (src -> dst) of rows that really move.
//   laswp_apply  -- every column reads all its moving elements into registers, waits for them, and
//                   writes them to their destinations: two independent passes instead of `cols`
//                   dependent round trips.  A workgroup handles LASWP_CPB columns per pass.
#include "mpf_internal.h"

constexpr int LASWP_CPB = 8; // columns per workgroup pass

__global__ __launch_bounds__(256) void laswp_plan_kernel(const int *ipiv, int k, int cols, MpfWorkspace *ws) {
    // slots 0..cols-1 <-> rows k..k+cols-1; a pivot row beyond the panel's top block gets slot
    // cols + (index of its first occurrence in ipiv)
    __shared__ int slot[HP_MAXCOLS];
    __shared__ int rowof[2 * HP_MAXCOLS];
    __shared__ int content[2 * HP_MAXCOLS];
    __shared__ int used[2 * HP_MAXCOLS];
    __shared__ int count;
    const int t = threadIdx.x;
    for (int s = t; s < 2 * cols; s += 256) { content[s] = s; used[s] = s < cols; rowof[s] = s < cols ? k + s : -1; }
    if (t == 0) count = 0;
    __syncthreads();
    if (t < cols) {
        const int p = ipiv[t] - 1; // 0-based global row, MPF.cu:49
        int s;
        if (p < k + cols) s = p - k;
        else {
            int first = t;
            for (int i = 0; i < t; ++i)
                if (ipiv[i] - 1 == p) { first = i; break; }
            s = cols + first;
            if (first == t) { rowof[s] = p; used[s] = 1; }
        }
        slot[t] = s;
    }
    __syncthreads();
    if (t == 0) { // the sequential part: cols swaps on a 2*cols-entry label array (MPF.cu:47-57)
        for (int j = 0; j < cols; ++j) {
            const int s = slot[j];
            if (s != j) { const int tmp = content[j]; content[j] = content[s]; content[s] = tmp; }
        }
    }
    __syncthreads();
    for (int s = t; s < 2 * cols; s += 256)
        if (used[s] && content[s] != s) {
            const int i = atomicAdd(&count, 1);
            ws->laswp_src[i] = rowof[content[s]];
            ws->laswp_dst[i] = rowof[s];
        }
    __syncthreads();
    if (t == 0) ws->laswp_n = count;
}

__global__ __launch_bounds__(256) void laswp_apply_kernel(double *A, long long lda, long long ncols,
                                                         const MpfWorkspace *ws, int from_pivot_kernel) {
    int n = from_pivot_kernel ? ws->flags[1] : ws->laswp_n;
    if (n > LASWP_MAXMOVED) n = LASWP_MAXMOVED;
    if (n == 0) return;
    const int t = threadIdx.x;
    const int i0 = t, i1 = t + 256;
    const int s0 = i0 < n ? ws->laswp_src[i0] : -1, d0 = i0 < n ? ws->laswp_dst[i0] : -1;
    const int s1 = i1 < n ? ws->laswp_src[i1] : -1, d1 = i1 < n ? ws->laswp_dst[i1] : -1;
    for (long long cb = (long long)blockIdx.x * LASWP_CPB; cb < ncols; cb += (long long)gridDim.x * LASWP_CPB) {
        double v0[LASWP_CPB], v1[LASWP_CPB];
#pragma unroll
        for (int c = 0; c < LASWP_CPB; ++c) {
            const long long col = cb + c;
            v0[c] = (s0 >= 0 && col < ncols) ? A[s0 + col * lda] : 0.0;
            v1[c] = (s1 >= 0 && col < ncols) ? A[s1 + col * lda] : 0.0;
        }
        // all gathers of this pass must have RETURNED before any scatter of the pass is issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int c = 0; c < LASWP_CPB; ++c) {
            const long long col = cb + c;
            if (d0 >= 0 && col < ncols) A[d0 + col * lda] = v0[c];
            if (d1 >= 0 && col < ncols) A[d1 + col * lda] = v1[c];
        }
        // columns of the next pass are disjoint from this one: no barrier needed here
    }
}

int launch_laswp(mpf_ctx *c, double *A, int64_t lda, int64_t ncols, int k, int cols, const int *d_ipiv) {
    if (cols < 1 || ncols < 1) return 0;
    if (cols > HP_MAXCOLS) { c->err = "laswp: more than 256 swaps per call"; return -1; }
    laswp_plan_kernel<<<1, 256, 0, c->stream>>>(d_ipiv, k, cols, c->ws);
    MPF_HIP_TRY(c, hipGetLastError());
    long long blocks = (ncols + LASWP_CPB - 1) / LASWP_CPB;
    if (blocks > 8192) blocks = 8192;
    laswp_apply_kernel<<<(int)blocks, 256, 0, c->stream>>>(A, lda, ncols, c->ws, 0);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

int launch_laswp_from_list(mpf_ctx *c, double *A, int64_t lda, int64_t ncols) {
    if (ncols < 1) return 0;
    long long blocks = (ncols + LASWP_CPB - 1) / LASWP_CPB;
    if (blocks > 8192) blocks = 8192;
    laswp_apply_kernel<<<(int)blocks, 256, 0, c->stream>>>(A, lda, ncols, c->ws, 1);
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

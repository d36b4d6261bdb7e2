// The reference generator's stream, produced on the device (reference matrix_generator.cpp:55-80 as benchmark.cpp:192-194
// reads it): element t of the N x N matrix, t = col * N + row (tokens are stored linearly and the buffer is interpreted
// column-major, benchmark.cpp:19), is (rand() % 100) / 10.0 where rand() is glibc's default generator, never seeded
// (seed 1), after `skip` earlier draws (`matgen f N (N-2) lin` emits a 2 x 2 matrix first: skip = 4).
//
// glibc's TYPE_3 rand() is the additive-feedback recurrence o[k] = o[k-3] + o[k-31] (mod 2^32), output o[k] >> 1, seeded by
// a 31-word LCG table and 310 discarded outputs.  A linear recurrence can be entered anywhere: with
// P(t) = t^31 - t^28 - 1 and t^J mod P = sum_i c_i t^i, o[k+J] = sum_i c_i o[k+i].  The host computes the 31-word state in
// front of every matrix COLUMN with polynomial arithmetic (31 x 31 words per column), one thread then runs the recurrence
// down its column with the ring held in registers.  No host-side N^2 work, no PCIe transfer of the matrix.
#include "mpf_internal.h"
#include <cstring>

namespace {
constexpr int RD = 31;

// coefficients of a(t) * b(t) mod P(t), arithmetic mod 2^32
void poly_mulmod(const uint32_t *a, const uint32_t *b, uint32_t *out) {
    uint32_t w[2 * RD - 1];
    memset(w, 0, sizeof w);
    for (int i = 0; i < RD; ++i) {
        if (!a[i]) continue;
        for (int j = 0; j < RD; ++j) w[i + j] += a[i] * b[j];
    }
    for (int d = 2 * RD - 2; d >= RD; --d) { // t^d = t^(d-3) + t^(d-31)
        w[d - 3] += w[d];
        w[d - RD] += w[d];
    }
    memcpy(out, w, RD * sizeof(uint32_t));
}
void poly_tpow(uint64_t e, uint32_t *out) { // t^e mod P
    uint32_t result[RD] = {1}, base[RD] = {0, 1}, tmp[RD];
    while (e) {
        if (e & 1) { poly_mulmod(result, base, tmp); memcpy(result, tmp, sizeof tmp); }
        poly_mulmod(base, base, tmp); memcpy(base, tmp, sizeof tmp);
        e >>= 1;
    }
    memcpy(out, result, sizeof result);
}
// raw 32-bit words o[0 .. n) of the never-seeded generator, discards included (glibc random_r.c: srandom_r + random_r)
void raw_stream_head(uint32_t *o, int n) {
    int32_t r[RD];
    r[0] = 1;
    for (int i = 1; i < RD; ++i) {
        const int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
        int64_t w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        r[i] = (int32_t)w;
    }
    uint32_t ring[RD];
    for (int i = 0; i < RD; ++i) ring[i] = (uint32_t)r[i];
    int f = 3, b = 0;
    for (int k = 0; k < n; ++k) {
        ring[f] += ring[b];
        o[k] = ring[f];
        f = (f + 1) % RD; b = (b + 1) % RD;
    }
}
} // namespace

// thread = one matrix column; ring position u holds o[k - 31] when output k = u (mod 31) is due
__global__ __launch_bounds__(64) void matgen_cols_kernel(double *A, long long lda, long long n, long long ncols,
                                                        const uint32_t *__restrict__ states) {
    const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
    if (c >= ncols) return;
    uint32_t s[RD];
#pragma unroll
    for (int u = 0; u < RD; ++u) s[u] = states[c * RD + u];
    double *col = A + c * lda;
    long long r = 0;
    for (; r + RD <= n; r += RD) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
            s[u] += s[(u + RD - 3) % RD];
            col[r + u] = (double)((s[u] >> 1) % 100u) / 10.0; // matrix_generator.cpp:66
        }
    }
#pragma unroll
    for (int u = 0; u < RD; ++u) {
        if (r + u < n) {
            s[u] += s[(u + RD - 3) % RD];
            col[r + u] = (double)((s[u] >> 1) % 100u) / 10.0;
        }
    }
}

extern "C" int mpf_matgen_cols_dev(mpf_ctx *c, double *d_A, int64_t lda, int64_t N, int64_t skip, int64_t col0, int64_t ncols) {
    if (!c || !d_A) return -1;
    if (N <= 0 || ncols <= 0) return 0;
    if (lda < N || skip < 0 || col0 < 0 || col0 + ncols > N) { c->err = "matgen: bad arguments"; return -1; }
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    uint32_t head[2 * RD];
    raw_stream_head(head, 2 * RD);
    // rand() call number j (0-based, after the 310 discards of srandom) is raw word 310 + j; column col starts at call
    // skip + col * N, and the ring in front of it is raw words [g - 31, g), g = 310 + skip + col * N
    std::vector<uint32_t> states((size_t)ncols * RD);
    uint32_t p[RD], step[RD], tmp[RD];
    poly_tpow((uint64_t)(310 - RD) + (uint64_t)skip + (uint64_t)col0 * (uint64_t)N, p);
    poly_tpow((uint64_t)N, step);
    for (int64_t cc = 0; cc < ncols; ++cc) {
        uint32_t *st = &states[(size_t)cc * RD];
        for (int i = 0; i < RD; ++i) {
            uint32_t acc = 0;
            for (int j = 0; j < RD; ++j) acc += p[j] * head[i + j];
            st[i] = acc;
        }
        poly_mulmod(p, step, tmp);
        memcpy(p, tmp, sizeof tmp);
    }
    uint32_t *d_states = nullptr;
    MPF_HIP_TRY(c, hipMalloc((void **)&d_states, states.size() * sizeof(uint32_t)));
    hipError_t e = hipMemcpyAsync(d_states, states.data(), states.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        matgen_cols_kernel<<<(unsigned)((ncols + 63) / 64), 64, 0, c->stream>>>(d_A, lda, N, ncols, d_states);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream); // `states` (host) and d_states must outlive the copy / kernel
    hipFree(d_states);
    if (e != hipSuccess) { c->err = std::string("matgen: ") + hipGetErrorString(e); return -2; }
    return 0;
}

// Host-only helper (no GPU needed): the 31 raw 32-bit words in front of rand() call number `call` -- what the kernel's
// ring is seeded with.  Lets the jump-ahead arithmetic be checked on a machine without a GPU.
extern "C" int mpf_matgen_state(int64_t call, uint32_t *out31) {
    if (call < 0 || !out31) return -1;
    uint32_t head[2 * RD], p[RD];
    raw_stream_head(head, 2 * RD);
    poly_tpow((uint64_t)(310 - RD) + (uint64_t)call, p);
    for (int i = 0; i < RD; ++i) {
        uint32_t acc = 0;
        for (int j = 0; j < RD; ++j) acc += p[j] * head[i + j];
        out31[i] = acc;
    }
    return 0;
}

extern "C" int mpf_matgen_dev(mpf_ctx *c, double *d_A, int64_t lda, int64_t N, int64_t skip) {
    return mpf_matgen_cols_dev(c, d_A, lda, N, skip, 0, N);
}

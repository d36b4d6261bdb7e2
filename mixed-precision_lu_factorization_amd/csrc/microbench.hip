// On-box peak probes ("re-measure on the box with a stream-copy and an MFMA issue-rate
// microbench and print both"): what the chip sustains for the two resources the hot path is priced
// against -- f64 / f16 MFMA issue rate at the clock the chip holds under load, and HBM stream bandwidth.
#include "mpf_internal.h"

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

// stamps[0..1]: shader-clock cycles and 100 MHz real-time ticks spent in the MFMA loop of block 0 / wave 0
__global__ __launch_bounds__(256) void mfma_f64_clock_kernel(int iters, double *sink, double seed, unsigned long long *stamps) {
    d4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double a = seed + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
    if (s == 12345.678) sink[0] = s;
}

__global__ __launch_bounds__(256) void mfma_f64_rate_kernel(int iters, double *sink, double seed) {
    d4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double a = seed + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s; // keep the chain live
}

__global__ __launch_bounds__(256) void mfma_f16_rate_kernel(int iters, float *sink, float seed) {
    f16v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    h8_t a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + 0.01f * (threadIdx.x % 7) + j); b[j] = (_Float16)(0.5f - 0.01f * (threadIdx.x % 5) * j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.678f) sink[0] = s;
}


// Structure probes of the fp16 update's K loop: T threads per workgroup (256 = one wave per SIMD, 512 = two), one workgroup per
// CU (dynamic LDS), NACC independent accumulators issued round robin, BAR > 0: one bare s_barrier every BAR MFMAs per wave,
// 16x16x32 or 32x32x16 shape; operands pseudo-random in registers.  Stamps: cycles and 100-MHz ticks of wave 0's loop.
template <int T, int BAR, bool S16>
__global__ __launch_bounds__(T) void mfma_f16_struct_kernel(int iters, float *sink, unsigned seed, unsigned long long *stamps) {
    extern __shared__ unsigned char pad_[];
    typedef float f4v __attribute__((ext_vector_type(4)));
    constexpr int NACC = 8;
    f16v acc[NACC]; f4v acc4[4 * NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc[i][j] = 0.f; acc4[4 * i + j / 4][j % 4] = 0.f; }
    h8_t a[2], b[4];
    unsigned x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9176u;
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = x * 1664525u + 1013904223u;
            const _Float16 v = (_Float16)(((int)(x >> 20) - 2048) * (1.0f / 2048.0f));
            if (q < 2) a[q][j] = v; else b[q - 2][j] = v;
        }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {   // 16 MFMA-equivalents (32x32x16) per trip
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (S16) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)   // two 16x16x32 = the flops of one 32x32x16
                        acc4[4 * i + 2 * (rep) + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 1], b[(i >> 1) & 3], acc4[4 * i + 2 * rep + u], 0, 0, 0);
                } else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], b[(i >> 1) & 3], acc[i], 0, 0, 0);
            }
            if (BAR == 8) __builtin_amdgcn_s_barrier();
        }
        if (BAR == 16) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { atomicAdd(stamps, c1 - c0); atomicAdd(stamps + 1, r1 - r0); atomicAdd(stamps + 2, 1ull); }
    float sm = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) sm += S16 ? acc4[4 * i + j / 4][j % 4] : acc[i][j];
    if (sm == 12345.678f) sink[0] = sm;
}

// HBM stream copy: U 16-byte loads in flight per lane before the first store, non-temporal both ways (round 3's form -- one load
// in flight per lane in a grid-stride loop -- read 4.8 TB/s where MI355X_MICROARCH.md measures 6.29 for this kind of copy).
// which = 2 runs <8> on 16 blocks per CU; which = 400 + 10 u + b sweeps U = {4, 8, 16}[u] and blocks per CU = {8, 16, 32, 64}[b]
// (tools/hbm_copy_probe.py).
typedef double d2v_t __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void stream_copy_kernel(const double2 *__restrict__ in_, double2 *__restrict__ out_, long long n2) {
    const d2v_t *in = (const d2v_t *)in_;
    d2v_t *out = (d2v_t *)out_;
    long long i = (long long)blockIdx.x * (256 * U) + threadIdx.x;
    const long long stride = (long long)gridDim.x * (256 * U);
    for (; i + (U - 1) * 256 < n2; i += stride) {
        d2v_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&in[i + u * 256]) : in[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], &out[i + u * 256]); else out[i + u * 256] = v[u]; }
    }
    for (; i < n2; i += 256) out[i] = in[i];   // (a ragged tail of the last sweep)
}

// MFMAs separated by PAD s_nop instructions / independent LDS reads: does a less dense stream issue MORE per second?
template <int PAD>
__global__ __launch_bounds__(256) void mfma_f64_pad_kernel(int iters, double *sink, double seed) {
    __shared__ double lds[2048];
    d4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = seed + i;
    __syncthreads();
    double a = seed + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            if (PAD == 1) { a = lds[(threadIdx.x + 16 * i + it) & 2047]; }
            if (PAD == 2) { a = lds[(threadIdx.x + 16 * i + it) & 2047]; b = lds[(threadIdx.x * 3 + 16 * i + it) & 2047]; }
            if (PAD == 3) asm volatile("s_nop 7");
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_f64_var_kernel(int iters, double *sink, double seed, unsigned long long *stamps) {
    d4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = seed + threadIdx.x * 1e-3 + i; b[i] = 1.0 - threadIdx.x * 1e-3 * (i + 1); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = t1 - t0;
    if (s == 12345.678) sink[0] = s;
}

// Issue-pattern scan of the f64 matrix instructions.  PAT 0: v_mfma_f64_16x16x4_f64, NACC accumulators, every MFMA with
// its own A and B registers.  PAT 1: v_mfma_f64_4x4x4_4b_f64 (four 4x4x4 blocks, 512 flop), same register pattern.
// PAT 2: 16x16x4 with the GEMM's operand reuse (A of tile row i & 3, B of tile column i >> 2).  PAT 3: 16x16x4, ONE
// accumulator (dependent chain: the instruction's latency).  stamps[0] = cycles (s_memtime) of block 0 / wave 0.
template <int PAT, int NACC>
__global__ __launch_bounds__(256) void mfma_f64_pat_kernel(int iters, double *sink, double seed, unsigned long long *stamps) {
    double a[NACC], b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { a[i] = seed + threadIdx.x * 1e-3 + i * 0.37; b[i] = 1.0 - threadIdx.x * 1e-3 * (i + 1); }
    double s = 0;
    unsigned long long t0, t1;
    if (PAT == 1) {
        double acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[i], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i];
        t1 = __builtin_amdgcn_s_memtime();
    } else {
        d4_t acc[PAT == 3 ? 1 : NACC];
#pragma unroll
        for (int i = 0; i < (PAT == 3 ? 1 : NACC); ++i) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (PAT == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[i], 0, 0, 0);
                if (PAT == 2) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
                if (PAT == 3) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[0], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < (PAT == 3 ? 1 : NACC); ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = t1 - t0;
    if (s == 12345.678) sink[0] = s;
}

// one v_mfma_f64_4x4x4_4b_f64 on caller-supplied per-lane operands (layout / arithmetic exploration; tools/mfma4_probe.py)
template <int CBSZ, int ABID, int BLGP>
__global__ __launch_bounds__(64) void mfma4_one_kernel(const double *a, const double *b, const double *cin, double *d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], cin[l], CBSZ, ABID, BLGP);
}
extern "C" int mpf_debug_mfma4(mpf_ctx *c, const double *a, const double *b, const double *cin, int variant, double *out) {
    if (!c || !a || !b || !cin || !out) return -1;
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    double *d = nullptr;
    MPF_HIP_TRY(c, hipMalloc((void **)&d, 4 * 64 * sizeof(double)));
    hipMemcpyAsync(d, a, 512, hipMemcpyHostToDevice, c->stream);
    hipMemcpyAsync(d + 64, b, 512, hipMemcpyHostToDevice, c->stream);
    hipMemcpyAsync(d + 128, cin, 512, hipMemcpyHostToDevice, c->stream);
#define M4(CB, AB, BL) mfma4_one_kernel<CB, AB, BL><<<1, 64, 0, c->stream>>>(d, d + 64, d + 128, d + 192)
    switch (variant) {
    case 0: M4(0, 0, 0); break;
    case 1: M4(2, 0, 0); break;
    case 2: M4(2, 1, 0); break;
    case 3: M4(2, 2, 0); break;
    case 4: M4(2, 3, 0); break;
    case 5: M4(0, 0, 1); break;
    case 6: M4(0, 0, 2); break;
    case 7: M4(0, 0, 4); break;
    case 8: M4(1, 0, 0); break;
    case 9: M4(1, 1, 0); break;
    default: hipFree(d); c->err = "mfma4: unknown variant"; return -1;
    }
#undef M4
    hipMemcpyAsync(out, d + 192, 512, hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) { c->err = hipGetErrorString(e); return -2; }
    return 0;
}


// Samples the shader clock while something else runs: one wave counts s_memtime ticks over `ticks10ns` ticks of the
// constant 100 MHz clock (s_memrealtime).  out[0] = shader cycles, out[1] = 10-ns ticks.
__global__ void clock_sampler_kernel(unsigned long long ticks10ns, unsigned long long *out) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks10ns) { __builtin_amdgcn_s_sleep(32); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[0] = t1 - t0; out[1] = r1 - r0;
}


// Where does the dispatcher put the workgroups of a GEMM-shaped launch?  out[2 b] = HW_ID, out[2 b + 1] = XCC_ID of block b.
__global__ __launch_bounds__(512, 2) void hwid_kernel(unsigned *out, int hold_ticks) {
    extern __shared__ double hw_lds[];
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
        hw_lds[0] = 1.0;
    }
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - r0 < (unsigned long long)hold_ticks) __builtin_amdgcn_s_sleep(16);
}


// Cross-XCD visibility: workgroups 0 and 1 (round-robin dispatch puts them on two XCDs) play ping-pong through two 8-byte flags,
// `rounds` times; the time per one-way trip = write of the flag (method WM) until the other side's poll (relaxed agent-scope
// loads, one in flight) has seen it.  WM: 0 write-through store (what the pivot kernel's row granules use), 1 atomic exchange
// (its keys), 2 release store (L2 write-back in front), 3 atomic add, 4 atomic max.  Result: 100 MHz ticks.
template <int WM>
__global__ __launch_bounds__(64) void xcd_pingpong_kernel(unsigned long long *flags, int rounds, unsigned long long *out) {
    if (blockIdx.x > 1 || threadIdx.x != 0) return;
    unsigned long long *mine = flags + 16 * blockIdx.x, *theirs = flags + 16 * (1 - blockIdx.x);   // separate 128-byte lines
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 1; k <= rounds; ++k) {
        {
            if (blockIdx.x == 0) {
                // write k, then wait for the echo
                const unsigned long long v = (unsigned long long)k;
                if (WM == 0) __hip_atomic_store(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (WM == 1) (void)__hip_atomic_exchange(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (WM == 2) __hip_atomic_store(mine, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                else if (WM == 3) (void)__hip_atomic_fetch_add(mine, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else (void)__hip_atomic_fetch_max(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spins = 0; spins < (1 << 22); ++spins)
                    if (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= v) break;
            } else {
                const unsigned long long v = (unsigned long long)k;
                for (int spins = 0; spins < (1 << 22); ++spins)
                    if (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= v) break;
                if (WM == 0) __hip_atomic_store(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (WM == 1) (void)__hip_atomic_exchange(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (WM == 2) __hip_atomic_store(mine, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                else if (WM == 3) (void)__hip_atomic_fetch_add(mine, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else (void)__hip_atomic_fetch_max(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (blockIdx.x == 0) out[0] = __builtin_amdgcn_s_memrealtime() - t0;
}

// Same ping-pong between workgroups 0 and `partner` with the accesses spelled out (round 5, the question behind a single-XCD pivot
// kernel for short panels): round-robin dispatch puts workgroup 8 on workgroup 0's XCD, where the L2 is the point of coherence and a
// trip need not go out to memory.  LM (poll): 0 = sc1 load (agent scope: what the kernels use), 1 = sc0 load (misses the CU's L1, may hit
// the XCD's L2), 2 = sc0 sc1.  SM (flag write): 0 = sc1 store (write-through), 1 = plain store (stays in the L2), 2 = sc0 sc1 store.
// out[0] = 100-MHz ticks for `rounds` round trips, out[1] = rounds completed (a poll that never sees its value gives up).
template <int LM, int SM>
__global__ __launch_bounds__(64) void xcd_pingpong_asm_kernel(unsigned long long *flags, int rounds, int partner, unsigned long long *out) {
    if ((blockIdx.x != 0 && blockIdx.x != (unsigned)partner) || threadIdx.x != 0) return;
    const int me = blockIdx.x == 0 ? 0 : 1;
    unsigned long long *mine = flags + 16 * me, *theirs = flags + 16 * (1 - me);
    auto ld = [&](const unsigned long long *p) {
        unsigned long long v;
        if (LM == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        else if (LM == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    };
    auto st = [&](unsigned long long *p, unsigned long long v) {
        if (SM == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        else if (SM == 1) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
        else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int done = 0;
    for (int k = 1; k <= rounds; ++k) {
        const unsigned long long v = (unsigned long long)k;
        bool seen = false;
        if (me == 0) {
            st(mine, v);
            for (int spins = 0; spins < (1 << 16) && !seen; ++spins) seen = ld(theirs) >= v;
        } else {
            for (int spins = 0; spins < (1 << 16) && !seen; ++spins) seen = ld(theirs) >= v;
            st(mine, v);
        }
        if (!seen) break;
        done = k;
    }
    if (me == 0) { out[0] = __builtin_amdgcn_s_memrealtime() - t0; out[1] = (unsigned long long)done; }
}

// ---- C-stream probe (round 5): the big fp16 update's epilogue alone, on a persistent grid of one 512-thread workgroup per CU
// walking 256 x 256 tiles of an fp32 matrix; each wave owns a 128 x 64 block held as the 32x32 MFMA accumulator layout
// (a dword access = two runs of 128 bytes).  MODE 0: the block through registers in four batches of 32 loads / 32 stores
// (three batches of loads in flight); MODE 1: 128 returnless buffer_atomic_add_f32 (the memory side does the read-modify-
// write, the wave moves on); MODE 2: nothing (the MFMA spin alone).  After its C work every wave issues `spin` MFMAs
// (32x32x16) on eight accumulators: the K loop of the NEXT tile, which the atomic form is meant to overlap with.
// stamps[0..2]: cycles wave 0 of workgroup 0 spent issuing its C work, 100-MHz ticks of the same, tiles.
template <int MODE>
__global__ __launch_bounds__(512, 1) void cstream_probe_kernel(float *C, long long ldc, int tiles_m, int tiles_n, int spin, unsigned seed,
                                                              float *sink, unsigned long long *stamps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    f16v acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    h8_t a[2], b[4];
    unsigned x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9176u;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { x = x * 1664525u + 1013904223u; a[i][j] = (_Float16)((float)((x >> 9) & 0x3FF) * (1.f / 1024.f) - 0.5f); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { x = x * 1664525u + 1013904223u; b[i][j] = (_Float16)((float)((x >> 9) & 0x3FF) * (1.f / 1024.f) - 0.5f); }
    const int nt_all = tiles_m * tiles_n;
    const unsigned ldc4 = (unsigned)ldc * 4u;
    unsigned long long sc = 0, sr = 0, sn = 0;
    for (int t = blockIdx.x; t < nt_all; t += gridDim.x) {
        const long long m0 = (long long)(t % tiles_m) * 256 + wr * 128, n0 = (long long)(t / tiles_m) * 256 + wc * 64;
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(C + m0 + n0 * ldc), 0, (int)((63 * ldc + 128) * 4), 0x00020000);
        const unsigned voff = (unsigned)r * 4u + (unsigned)(4 * h) * ldc4;
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        if (MODE == 0) {
            float cf[3][2][16];
            auto ld = [&](float (&d)[2][16], int mt) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 16; ++g)
                        d[nt][g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (int)(voff + 128u * mt), (int)((unsigned)(32 * nt + (g & 3) + 8 * (g >> 2)) * ldc4), 2));
            };
            auto stv = [&](float (&d)[2][16], int mt) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 16; ++g)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d[nt][g] - acc[2 * mt + nt][g]), rc, (int)(voff + 128u * mt),
                                                              (int)((unsigned)(32 * nt + (g & 3) + 8 * (g >> 2)) * ldc4), 2);
            };
            ld(cf[0], 0); ld(cf[1], 1); ld(cf[2], 2);
            __builtin_amdgcn_sched_barrier(0);
            stv(cf[0], 0); ld(cf[0], 3);
            __builtin_amdgcn_sched_barrier(0);
            stv(cf[1], 1); stv(cf[2], 2); stv(cf[0], 3);
        } else if (MODE == 1) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 16; ++g)
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(-acc[2 * mt + nt][g], rc, (int)(voff + 128u * mt),
                                                                         (int)((unsigned)(32 * nt + (g & 3) + 8 * (g >> 2)) * ldc4), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tid == 0 && blockIdx.x == 0) { sc += __builtin_amdgcn_s_memtime() - c0; sr += __builtin_amdgcn_s_memrealtime() - r0; sn += 1; }
        for (int it = 0; it < spin; it += 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], b[i >> 1], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = acc[i][j] * 1e-30f + 1.0f;   // keep the values tame, the chain live
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.678f) sink[0] = s;
    if (tid == 0 && blockIdx.x == 0) { stamps[0] = sc; stamps[1] = sr; stamps[2] = sn; }
}

extern "C" int mpf_microbench(mpf_ctx *c, int which, double *result) {
    if (!c || !result) return -1;
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    if (which == 0 || which == 1) {
        void *sink = nullptr;
        MPF_HIP_TRY(c, hipMalloc(&sink, 64));
        const int blocks = c->num_cus * 2, iters = which == 0 ? 20000 : 40000;
        for (int rep = 0; rep < 2; ++rep) { // rep 0 warms the clocks
            hipEventRecord(e0, c->stream);
            if (which == 0) mfma_f64_rate_kernel<<<blocks, 256, 0, c->stream>>>(iters, (double *)sink, 0.5);
            else mfma_f16_rate_kernel<<<blocks, 256, 0, c->stream>>>(iters, (float *)sink, 0.5f);
            hipEventRecord(e1, c->stream);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double waves = (double)blocks * 4;
        const double flop = which == 0 ? waves * iters * 8.0 * (16 * 16 * 4 * 2) : waves * iters * 4.0 * (32.0 * 32 * 16 * 2);
        *result = flop / (ms * 1e-3) / 1e12; // TFLOP/s
        hipFree(sink);
    } else if (which == 2 || (which >= 400 && which < 440)) {
        const size_t bytes = (size_t)2 << 30; // 2 GiB in, 2 GiB out: far beyond the 256 MiB Infinity Cache
        void *a = nullptr, *b = nullptr;
        MPF_HIP_TRY(c, hipMalloc(&a, bytes));
        if (hipMalloc(&b, bytes) != hipSuccess) { hipFree(a); c->err = "microbench: hipMalloc failed"; return -2; }
        hipMemsetAsync(a, 1, bytes, c->stream);
        const int ui = which == 2 ? 1 : ((which - 400) / 10) % 4, bi = which == 2 ? 1 : (which - 400) % 10;
        const int bpc = bi == 0 ? 8 : bi == 1 ? 16 : bi == 2 ? 32 : 64;
        const int grid = c->num_cus * bpc;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, c->stream);
            if (ui == 0) stream_copy_kernel<4, true><<<grid, 256, 0, c->stream>>>((const double2 *)a, (double2 *)b, (long long)(bytes / 16));
            else if (ui == 1) stream_copy_kernel<8, true><<<grid, 256, 0, c->stream>>>((const double2 *)a, (double2 *)b, (long long)(bytes / 16));
            else if (ui == 2) stream_copy_kernel<16, true><<<grid, 256, 0, c->stream>>>((const double2 *)a, (double2 *)b, (long long)(bytes / 16));
            else stream_copy_kernel<8, false><<<grid, 256, 0, c->stream>>>((const double2 *)a, (double2 *)b, (long long)(bytes / 16));
            hipEventRecord(e1, c->stream);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        *result = 2.0 * bytes / (ms * 1e-3) / 1e12; // TB/s read+write
        hipFree(a); hipFree(b);
    } else if (which == 3 || which == 4) {
        // f64 MFMA loop with clock stamps: result = cycles per MFMA (which == 3) or sustained shader clock in GHz (4)
        void *sink = nullptr; unsigned long long *st = nullptr;
        MPF_HIP_TRY(c, hipMalloc(&sink, 64));
        MPF_HIP_TRY(c, hipMalloc((void **)&st, 16));
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep)
            mfma_f64_clock_kernel<<<c->num_cus * 2, 256, 0, c->stream>>>(iters, (double *)sink, 0.5, st);
        unsigned long long h[2] = {0, 0};
        MPF_HIP_TRY(c, hipMemcpyAsync(h, st, 16, hipMemcpyDeviceToHost, c->stream));
        MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
        *result = which == 3 ? (double)h[0] / ((double)iters * 8.0) : (double)h[0] / ((double)h[1] * 10.0) ; // ticks are 10 ns
        hipFree(sink); hipFree(st);
    } else if (which >= 300 && which < 303) {
        // sustained shader clock (GHz) while the library's fp64 update (300) / fp16 update (301) / nothing (302) runs,
        // sampled by one wave on the side stream over the middle 20 ms
        if (!c->pstream) { c->err = "no side stream"; return -1; }
        const long long mm = 16384; const int kk = 256;
        double *A = nullptr, *B = nullptr, *C = nullptr; unsigned long long *st = nullptr;
        MPF_HIP_TRY(c, hipMalloc((void **)&A, mm * kk * 8)); MPF_HIP_TRY(c, hipMalloc((void **)&B, mm * kk * 8));
        MPF_HIP_TRY(c, hipMalloc((void **)&C, mm * mm * 8)); MPF_HIP_TRY(c, hipMalloc((void **)&st, 16));
        hipMemsetAsync(A, 0, mm * kk * 8, c->stream); hipMemsetAsync(B, 0, mm * kk * 8, c->stream); hipMemsetAsync(C, 0, mm * mm * 8, c->stream);
        hipStreamSynchronize(c->stream);
        hipEventRecord(e0, c->stream);
        for (int rep = 0; rep < 24; ++rep) {
            if (which == 300) launch_dgemm_minus(c, mm, mm, kk, A, mm, B, kk, C, mm);
            else if (which == 301) mpf_hgemm_minus(c, mm, mm, kk, A, mm, B, kk, C, mm, 0);
            if (rep == 2) clock_sampler_kernel<<<1, 64, 0, c->pstream>>>(2000000ull, st);
        }
        if (which == 302) clock_sampler_kernel<<<1, 64, 0, c->pstream>>>(2000000ull, st);
        hipEventRecord(e1, c->stream);
        hipStreamSynchronize(c->pstream); hipStreamSynchronize(c->stream);
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2] = {0, 1};
        MPF_HIP_TRY(c, hipMemcpy(h, st, 16, hipMemcpyDeviceToHost));
        *result = (double)h[0] / ((double)h[1] * 10.0);
        if (getenv("MPF_VERBOSE")) fprintf(stderr, "microbench %d: 24 launches %.2f ms, sampler %llu cycles / %llu ticks\n", which, ms, h[0], h[1]);
        hipFree(A); hipFree(B); hipFree(C); hipFree(st);
    } else if (which == 310) {
        // dispatch map of a 512-thread / 73.7 KB-LDS launch (the fp64 update's shape): prints block -> XCC, SE, CU, TG, wave slot
        unsigned *d = nullptr; const int nb = c->num_cus * 2 + 64;
        MPF_HIP_TRY(c, hipMalloc((void **)&d, nb * 8));
        MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)hwid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
        hwid_kernel<<<nb, 512, 73728, c->stream>>>(d, 2000);
        std::vector<unsigned> h(nb * 2);
        MPF_HIP_TRY(c, hipMemcpyAsync(h.data(), d, nb * 8, hipMemcpyDeviceToHost, c->stream));
        MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int b = 0; b < nb; ++b) {
            const unsigned hw = h[2 * b], xc = h[2 * b + 1];
            fprintf(stderr, "HWID block %d xcc %u se %u sh %u cu %u tg %u simd %u wave %u raw %08x\n", b, xc & 15u, (hw >> 13) & 7u, (hw >> 12) & 1u,
                    (hw >> 8) & 15u, (hw >> 16) & 15u, (hw >> 4) & 3u, hw & 15u, hw);
        }
        *result = nb;
        hipFree(d);
    } else if (which >= 50 && which < 54) {
        void *sink = nullptr;
        MPF_HIP_TRY(c, hipMalloc(&sink, 64));
        const int iters = 10000, v = which - 50;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, c->stream);
            if (v == 0) mfma_f64_pad_kernel<0><<<c->num_cus * 2, 256, 0, c->stream>>>(iters, (double *)sink, 0.5);
            else if (v == 1) mfma_f64_pad_kernel<1><<<c->num_cus * 2, 256, 0, c->stream>>>(iters, (double *)sink, 0.5);
            else if (v == 2) mfma_f64_pad_kernel<2><<<c->num_cus * 2, 256, 0, c->stream>>>(iters, (double *)sink, 0.5);
            else mfma_f64_pad_kernel<3><<<c->num_cus * 2, 256, 0, c->stream>>>(iters, (double *)sink, 0.5);
            hipEventRecord(e1, c->stream);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        *result = (double)c->num_cus * 2 * 4 * iters * 8 * 2048.0 / (ms * 1e-3) / 1e12;
        hipFree(sink);
    } else if (which >= 200 && which < 260) {
        // whole-launch TFLOP/s by HIP events: which = 200 + 10 * pattern + w, pattern as in 100.., w = workgroups of 4 waves per CU
        void *sink = nullptr; unsigned long long *st = nullptr;
        MPF_HIP_TRY(c, hipMalloc(&sink, 64));
        MPF_HIP_TRY(c, hipMalloc((void **)&st, 16));
        const int pat = (which - 200) / 10, w = (which - 200) % 10, iters = 4000;
        const int grid = c->num_cus * (w < 1 ? 1 : w);
        int nacc = 16;
        double flop_per = 2048.0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, c->stream);
            if (pat == 0) mfma_f64_pat_kernel<0, 16><<<grid, 256, 0, c->stream>>>(iters, (double *)sink, 0.5, st);
            else if (pat == 1) { mfma_f64_pat_kernel<1, 16><<<grid, 256, 0, c->stream>>>(iters, (double *)sink, 0.5, st); flop_per = 512.0; }
            else if (pat == 2) mfma_f64_pat_kernel<2, 16><<<grid, 256, 0, c->stream>>>(iters, (double *)sink, 0.5, st);
            else if (pat == 3) mfma_f64_pat_kernel<3, 16><<<grid, 256, 0, c->stream>>>(iters, (double *)sink, 0.5, st);
            else { mfma_f64_pat_kernel<1, 8><<<grid, 256, 0, c->stream>>>(iters, (double *)sink, 0.5, st); nacc = 8; flop_per = 512.0; }
            hipEventRecord(e1, c->stream);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        *result = (double)grid * 4 * iters * nacc * flop_per / (ms * 1e-3) / 1e12;
        hipFree(sink); hipFree(st);
    } else if (which >= 100 && which < 200) {
        // which = 100 + 10 * pattern + config: pattern 0..4 (0: 16x16x4 distinct operands x16, 1: 4x4x4_4b x16, 2: 16x16x4 GEMM
        // reuse x16, 3: 16x16x4 dependent chain, 4: 4x4x4_4b x8), config 0: one wave alone, 1: one wave per SIMD on every CU,
        // 2: two waves per SIMD, 3: four waves per SIMD.  Result: cycles per MFMA as wave 0 of block 0 sees them.
        void *sink = nullptr; unsigned long long *st = nullptr;
        MPF_HIP_TRY(c, hipMalloc(&sink, 64));
        MPF_HIP_TRY(c, hipMalloc((void **)&st, 16));
        const int pat = (which - 100) / 10, cfg = (which - 100) % 10, iters = 4000;
        const int grid = cfg == 0 ? 1 : c->num_cus * (cfg == 1 ? 1 : (cfg == 2 ? 2 : 4)), blk = cfg == 0 ? 64 : 256;
        int nacc = 16;
        for (int rep = 0; rep < 2; ++rep) {
            if (pat == 0) mfma_f64_pat_kernel<0, 16><<<grid, blk, 0, c->stream>>>(iters, (double *)sink, 0.5, st);
            else if (pat == 1) mfma_f64_pat_kernel<1, 16><<<grid, blk, 0, c->stream>>>(iters, (double *)sink, 0.5, st);
            else if (pat == 2) mfma_f64_pat_kernel<2, 16><<<grid, blk, 0, c->stream>>>(iters, (double *)sink, 0.5, st);
            else if (pat == 3) mfma_f64_pat_kernel<3, 16><<<grid, blk, 0, c->stream>>>(iters, (double *)sink, 0.5, st);
            else { mfma_f64_pat_kernel<1, 8><<<grid, blk, 0, c->stream>>>(iters, (double *)sink, 0.5, st); nacc = 8; }
        }
        unsigned long long h[2] = {0, 0};
        MPF_HIP_TRY(c, hipMemcpyAsync(h, st, 16, hipMemcpyDeviceToHost, c->stream));
        MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
        *result = (double)h[0] / ((double)iters * nacc);
        hipFree(sink); hipFree(st);
    } else if (which >= 70 && which < 78) {
        // segment cycle sums left by the last diagnostic launch of the pivot kernel (MPF_HP_STAMP=1)
        unsigned long long v = 0;
        MPF_HIP_TRY(c, hipMemcpy(&v, &c->ws->hp_stamps[which - 70], sizeof v, hipMemcpyDeviceToHost));
        *result = (double)v;
    } else if (which % 100 >= 90 && which % 100 < 96 && which < 400) {
        // fp16 MFMA structure probes.  which % 100: 90 one wave per SIMD, no barrier; 91 two waves per SIMD, no barrier; 92 two
        // waves, a bare s_barrier every 16 MFMAs per wave; 93 every 8; 94 / 95: the 16x16x32 shape (two per 32x32x16), two waves,
        // barrier every 16 equivalents / none.  which / 100: 0 = cycles per 32x32x16-equivalent MFMA per SIMD, 1 = sustained
        // shader clock (GHz), 2 = TFLOP/s by the event clock.
        void *sink = nullptr; unsigned long long *st = nullptr;
        MPF_HIP_TRY(c, hipMalloc(&sink, 64));
        MPF_HIP_TRY(c, hipMalloc((void **)&st, 32));
        const int v = which % 100, iters = 4000, blocks = c->num_cus;
        const size_t lds = 96 * 1024;   // one workgroup per CU
        unsigned long long h[3] = {0, 0, 0};
        float ms = 0;
#define STRUCT_RUN(T_, B_, S_) do { auto *kf = mfma_f16_struct_kernel<T_, B_, S_>; \
            MPF_HIP_TRY(c, hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            for (int rep = 0; rep < 2; ++rep) { MPF_HIP_TRY(c, hipMemsetAsync(st, 0, 32, c->stream)); hipEventRecord(e0, c->stream); \
                kf<<<blocks, T_, lds, c->stream>>>(iters, (float *)sink, 17u + rep, st); hipEventRecord(e1, c->stream); } \
            hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); } while (0)
        const int T = v == 90 ? 256 : 512;
        if (v == 90) STRUCT_RUN(256, 0, false);
        else if (v == 91) STRUCT_RUN(512, 0, false);
        else if (v == 92) STRUCT_RUN(512, 16, false);
        else if (v == 93) STRUCT_RUN(512, 8, false);
        else if (v == 94) STRUCT_RUN(512, 16, true);
        else STRUCT_RUN(512, 0, true);
#undef STRUCT_RUN
        MPF_HIP_TRY(c, hipMemcpy(h, st, 24, hipMemcpyDeviceToHost));
        const double wps = T / 256.0, mf = (double)iters * 16.0;   // MFMA-equivalents per wave
        if (which / 100 == 0) *result = h[2] ? (double)h[0] / h[2] / (mf * wps) : 0;
        else if (which / 100 == 1) *result = h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0;
        else *result = (double)blocks * (T / 64) * mf * 32768.0 / (ms * 1e-3) / 1e12;
        hipFree(sink); hipFree(st);
    } else if (which >= 500 && which < 505) {
        // one-way cross-XCD flag latency in ns (ping-pong of workgroups 0 and 1, 2000 rounds): write method which - 500
        unsigned long long *fl = nullptr, *out = nullptr;
        MPF_HIP_TRY(c, hipMalloc((void **)&fl, 512)); MPF_HIP_TRY(c, hipMalloc((void **)&out, 16));
        const int rounds = 2000;
        for (int rep = 0; rep < 2; ++rep) {
            MPF_HIP_TRY(c, hipMemsetAsync(fl, 0, 512, c->stream));
            switch (which - 500) {
                case 0: xcd_pingpong_kernel<0><<<2, 64, 0, c->stream>>>(fl, rounds, out); break;
                case 1: xcd_pingpong_kernel<1><<<2, 64, 0, c->stream>>>(fl, rounds, out); break;
                case 2: xcd_pingpong_kernel<2><<<2, 64, 0, c->stream>>>(fl, rounds, out); break;
                case 3: xcd_pingpong_kernel<3><<<2, 64, 0, c->stream>>>(fl, rounds, out); break;
                default: xcd_pingpong_kernel<4><<<2, 64, 0, c->stream>>>(fl, rounds, out); break;
            }
        }
        unsigned long long h = 0;
        MPF_HIP_TRY(c, hipMemcpyAsync(&h, out, 8, hipMemcpyDeviceToHost, c->stream));
        MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
        *result = (double)h * 10.0 / (2.0 * rounds);
        hipFree(fl); hipFree(out);
    } else if (which >= 510 && which < 600) {
        // 510 + 30 * same_xcd + 10 * LM + SM: one-way flag latency in ns between workgroups 0 and 1 (other XCD) or 0 and 8 (same XCD); a negative
        // result: that many rounds completed before a poll gave up (the write never became visible to that kind of load)
        const int w = which - 510, same = w / 30, lm = (w % 30) / 10, sm = w % 10;
        if (lm > 2 || sm > 2 || same > 1) { c->err = "microbench 510+: bad method"; return -1; }
        unsigned long long *fl = nullptr, *out = nullptr;
        MPF_HIP_TRY(c, hipMalloc((void **)&fl, 512)); MPF_HIP_TRY(c, hipMalloc((void **)&out, 16));
        const int rounds = 2000, partner = same ? 8 : 1;
        for (int rep = 0; rep < 2; ++rep) {
            MPF_HIP_TRY(c, hipMemsetAsync(fl, 0, 512, c->stream));
#define PP(L, S) xcd_pingpong_asm_kernel<L, S><<<16, 64, 0, c->stream>>>(fl, rounds, partner, out)
            switch (lm * 3 + sm) {
                case 0: PP(0, 0); break; case 1: PP(0, 1); break; case 2: PP(0, 2); break;
                case 3: PP(1, 0); break; case 4: PP(1, 1); break; case 5: PP(1, 2); break;
                case 6: PP(2, 0); break; case 7: PP(2, 1); break; default: PP(2, 2); break;
            }
#undef PP
        }
        unsigned long long h[2] = {0, 0};
        MPF_HIP_TRY(c, hipMemcpyAsync(h, out, 16, hipMemcpyDeviceToHost, c->stream));
        MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
        *result = (int)h[1] == rounds ? (double)h[0] * 10.0 / (2.0 * rounds) : -(double)h[1];
        hipFree(fl); hipFree(out);
    } else if (which >= 600 && which < 640) {
        // C-stream probe: mode = (which - 600) / 10 (0 registers, 1 atomics, 2 none), spin = 0 / 256 / 512 / 1024 MFMAs per wave and tile
        const int mode = (which - 600) / 10, sp = (which - 600) % 10;
        const int spin = sp == 0 ? 0 : sp == 1 ? 256 : sp == 2 ? 512 : 1024;
        const long long mm = 28672;
        float *Cm = nullptr, *sink = nullptr; unsigned long long *st = nullptr;
        MPF_HIP_TRY(c, hipMalloc((void **)&Cm, mm * mm * 4));
        MPF_HIP_TRY(c, hipMalloc((void **)&sink, 64));
        MPF_HIP_TRY(c, hipMalloc((void **)&st, 32));
        hipMemsetAsync(Cm, 0, mm * mm * 4, c->stream);
        const int tm = (int)(mm / 256);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, c->stream);
            if (mode == 0) cstream_probe_kernel<0><<<c->num_cus, 512, 0, c->stream>>>(Cm, mm, tm, tm, spin, 7u + rep, sink, st);
            else if (mode == 1) cstream_probe_kernel<1><<<c->num_cus, 512, 0, c->stream>>>(Cm, mm, tm, tm, spin, 7u + rep, sink, st);
            else cstream_probe_kernel<2><<<c->num_cus, 512, 0, c->stream>>>(Cm, mm, tm, tm, spin, 7u + rep, sink, st);
            hipEventRecord(e1, c->stream);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        unsigned long long hs[3] = {0, 0, 1};
        MPF_HIP_TRY(c, hipMemcpy(hs, st, 24, hipMemcpyDeviceToHost));
        fprintf(stderr, "cstream probe mode %d spin %d: %.3f ms per launch (C traffic %.0f GB/s, MFMA %.0f TFLOP/s); wave 0: %.0f cycles = %.2f us of C work per tile\n",
                mode, spin, best, mode == 2 ? 0.0 : 8.0 * mm * mm / best / 1e6, 2.0 * 32 * 32 * 16 * spin * 8.0 * tm * tm / best / 1e9,
                (double)hs[0] / (double)(hs[2] ? hs[2] : 1), (double)hs[1] / (double)(hs[2] ? hs[2] : 1) / 100.0);
        *result = best;
        hipFree(Cm); hipFree(sink); hipFree(st);
    } else if (which == 78) {   // clear the stamp sums (the fp16 update's K-loop stamps accumulate)
        MPF_HIP_TRY(c, hipMemset(c->ws->hp_stamps, 0, sizeof c->ws->hp_stamps));
        *result = 0;
    } else if (which >= 60 && which < 64) {
        // cycles per v_mfma_f64_16x16x4_f64 as one wave sees them (s_memtime around 10000 x 16 independent MFMAs on
        // distinct operand registers): 60 = ONE wave alone on the chip, 61 = one workgroup (one wave per SIMD of one CU),
        // 62 = one wave per SIMD on every CU, 63 = two waves per SIMD on every CU
        void *sink = nullptr; unsigned long long *st = nullptr;
        MPF_HIP_TRY(c, hipMalloc(&sink, 64));
        MPF_HIP_TRY(c, hipMalloc((void **)&st, 16));
        const int iters = 10000, v = which - 60;
        const int grid = v <= 1 ? 1 : c->num_cus * (v - 1), blk = v == 0 ? 64 : 256;
        for (int rep = 0; rep < 2; ++rep) mfma_f64_var_kernel<16><<<grid, blk, 0, c->stream>>>(iters, (double *)sink, 0.5, st);
        unsigned long long h[2] = {0, 0};
        MPF_HIP_TRY(c, hipMemcpyAsync(h, st, 16, hipMemcpyDeviceToHost, c->stream));
        MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
        *result = (double)h[0] / ((double)iters * 16.0);
        hipFree(sink); hipFree(st);
    } else if (which >= 10 && which < 40) {
        // f64 MFMA issue-interval scan: which = 10*w + v, w = workgroups (of 4 waves) per CU in {1,2,3}, v: 0 -> 4 acc, 1 -> 8, 2 -> 16
        void *sink = nullptr; unsigned long long *st = nullptr;
        MPF_HIP_TRY(c, hipMalloc(&sink, 64));
        MPF_HIP_TRY(c, hipMalloc((void **)&st, 16));
        const int w = which / 10, v = which % 10, iters = 10000;
        const int nacc = v == 0 ? 4 : (v == 1 ? 8 : 16);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, c->stream);
            if (v == 0) mfma_f64_var_kernel<4><<<c->num_cus * w, 256, 0, c->stream>>>(iters, (double *)sink, 0.5, st);
            else if (v == 1) mfma_f64_var_kernel<8><<<c->num_cus * w, 256, 0, c->stream>>>(iters, (double *)sink, 0.5, st);
            else mfma_f64_var_kernel<16><<<c->num_cus * w, 256, 0, c->stream>>>(iters, (double *)sink, 0.5, st);
            hipEventRecord(e1, c->stream);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        *result = (double)c->num_cus * w * 4 * iters * nacc * 2048.0 / (ms * 1e-3) / 1e12; // TFLOP/s, whole launch
        hipFree(sink); hipFree(st);
    } else {
        c->err = "microbench: unknown probe"; return -1;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 0;
}


// One gate launch with no pivot kernel behind it (tests: a gate that gives up must flag the context's time-out counter, which
// is what makes mpf_factor_dev / mpf_factor_dist return -4).  Returns the counter after the gate has run; resets it to 0.
extern "C" int mpf_debug_gate(mpf_ctx *c, int target) {
    if (!c) return -1;
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    MPF_HIP_TRY(c, hipMemsetAsync(&c->ws->hp_timeouts, 0, sizeof(int), c->stream));
    c->hp_seq = (c->hp_seq + 1) & 0x3FFFFFu;   // a sequence number no pivot kernel has published progress for
    // the gate as the chains use it: folded into the interchange launch it guards (laswp.hip).  An expired gate must apply nothing:
    // the two "pivots" below would swap rows 0 and 1 of the scratch column
    double *col = nullptr;
    int *piv = nullptr;
    MPF_HIP_TRY(c, hipMalloc((void **)&col, 8 * sizeof(double)));
    MPF_HIP_TRY(c, hipMalloc((void **)&piv, 2 * sizeof(int)));
    const double h[8] = {10, 11, 12, 13, 14, 15, 16, 17};
    const int hp[2] = {2, 2};
    MPF_HIP_TRY(c, hipMemcpyAsync(col, h, sizeof h, hipMemcpyHostToDevice, c->stream));
    MPF_HIP_TRY(c, hipMemcpyAsync(piv, hp, sizeof hp, hipMemcpyHostToDevice, c->stream));
    int rc = launch_laswp_block_gated(c, col, 8, 1, 0, 2, piv, 8, target);
    double back[8];
    if (!rc) MPF_HIP_TRY(c, hipMemcpyAsync(back, col, sizeof back, hipMemcpyDeviceToHost, c->stream));
    if (!rc) rc = launch_hgetf2_gate(c, target);   // and the stand-alone gate kernel (leaves at once: the counter is set)
    MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
    hipFree(col); hipFree(piv);
    if (rc) return rc;
    if (back[0] != 10 || back[1] != 11) { c->err = "debug gate: an expired gate applied its interchange"; return -1; }
    int flags = 0;
    MPF_HIP_TRY(c, hipMemcpyAsync(&flags, &c->ws->hp_timeouts, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
    MPF_HIP_TRY(c, hipMemsetAsync(&c->ws->hp_timeouts, 0, sizeof(int), c->stream));
    return flags;
}

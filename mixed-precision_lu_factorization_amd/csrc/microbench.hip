// On-box peak probes ("re-measure on the box with a stream-copy and an MFMA issue-rate
// microbench and print both"): what the chip sustains for the two resources the hot path is priced
// against -- f64 / f16 MFMA issue rate at the clock the chip holds under load, and HBM stream bandwidth.
#include "mpf_internal.h"

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

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

__global__ __launch_bounds__(256) void stream_copy_kernel(const double2 *__restrict__ in, double2 *__restrict__ out, long long n2) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n2; i += stride) out[i] = in[i];
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
    } else if (which == 2) {
        const size_t bytes = (size_t)2 << 30; // 2 GiB in, 2 GiB out: far beyond the 256 MiB Infinity Cache
        void *a = nullptr, *b = nullptr;
        MPF_HIP_TRY(c, hipMalloc(&a, bytes));
        if (hipMalloc(&b, bytes) != hipSuccess) { hipFree(a); c->err = "microbench: hipMalloc failed"; return -2; }
        hipMemsetAsync(a, 1, bytes, c->stream);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, c->stream);
            stream_copy_kernel<<<c->num_cus * 8, 256, 0, c->stream>>>((const double2 *)a, (double2 *)b, (long long)(bytes / 16));
            hipEventRecord(e1, c->stream);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        *result = 2.0 * bytes / (ms * 1e-3) / 1e12; // TB/s read+write
        hipFree(a); hipFree(b);
    } else {
        c->err = "microbench: unknown probe"; return -1;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 0;
}

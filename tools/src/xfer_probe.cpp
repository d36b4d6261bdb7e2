// Probe for the host entry point's transfers: what pinning the caller's pageable matrix costs, and what
// strided (2-KB runs) device -> host copies of finished block rows reach.  build: hipcc -O2 -o gpurun_out/xfer_probe tools/src/xfer_probe.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const long long N = argc > 1 ? atoll(argv[1]) : 32768, nb = 256;
    const size_t bytes = (size_t)N * N * 8;
    double *h = (double *)aligned_alloc(4096, bytes);
    for (size_t i = 0; i < (size_t)N * N; i += 512) h[i] = 1.0;   // touch every page
    double *d; CK(hipMalloc(&d, bytes));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); double t1 = now(); CK(hipStreamSynchronize(s)); double t2 = now();
        printf("pageable H2D whole: call %.1f ms, done %.1f ms = %.1f GB/s\n", t1 - t0, t2 - t0, bytes / (t2 - t0) / 1e6);
        t0 = now(); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s)); t1 = now(); CK(hipStreamSynchronize(s)); t2 = now();
        printf("pageable D2H whole: call %.1f ms, done %.1f ms = %.1f GB/s\n", t1 - t0, t2 - t0, bytes / (t2 - t0) / 1e6);
    }
    // strided block rows into pageable memory: rows [k, k + nb) of columns [k + nb, N)
    for (long long k : {0ll, 8192ll, 16384ll}) {
        const size_t w = nb * 8, hgt = N - k - nb;
        double t0 = now(); CK(hipMemcpy2DAsync(h + (k + nb) * N + k, N * 8, d + (k + nb) * N + k, N * 8, w, hgt, hipMemcpyDeviceToHost, s)); double t1 = now();
        CK(hipStreamSynchronize(s)); double t2 = now();
        printf("pageable D2H block row k=%lld (%zu x 2 KB): call %.2f ms, done %.2f ms = %.1f GB/s\n", k, hgt, t1 - t0, t2 - t0, w * hgt / (t2 - t0) / 1e6);
    }
    double t0 = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); double t1 = now();
    printf("hipHostRegister of %.1f GB: %.1f ms\n", bytes / 1e9, t1 - t0);
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); t1 = now(); CK(hipStreamSynchronize(s)); double t2 = now();
        printf("registered H2D whole: call %.1f ms, done %.1f ms = %.1f GB/s\n", t1 - t0, t2 - t0, bytes / (t2 - t0) / 1e6);
        t0 = now(); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s)); t1 = now(); CK(hipStreamSynchronize(s)); t2 = now();
        printf("registered D2H whole: call %.1f ms, done %.1f ms = %.1f GB/s\n", t1 - t0, t2 - t0, bytes / (t2 - t0) / 1e6);
    }
    for (long long k : {0ll, 8192ll, 16384ll, 28672ll}) {
        const size_t w = nb * 8, hgt = N - k - nb;
        t0 = now(); CK(hipMemcpy2DAsync(h + (k + nb) * N + k, N * 8, d + (k + nb) * N + k, N * 8, w, hgt, hipMemcpyDeviceToHost, s)); t1 = now();
        CK(hipStreamSynchronize(s)); double t2 = now();
        printf("registered D2H block row k=%lld (%zu x 2 KB): call %.2f ms, done %.2f ms = %.1f GB/s\n", k, hgt, t1 - t0, t2 - t0, w * hgt / (t2 - t0) / 1e6);
    }
    // column slabs of 64 MB H2D from registered memory (what a chunked upload would issue)
    { t0 = now(); const long long cs = 256; for (long long j = 0; j < N; j += cs) CK(hipMemcpyAsync(d + j * N, h + j * N, (size_t)cs * N * 8, hipMemcpyHostToDevice, s));
      t1 = now(); CK(hipStreamSynchronize(s)); double t2 = now();
      printf("registered H2D in %lld slabs: calls %.1f ms, done %.1f ms = %.1f GB/s\n", N / cs, t1 - t0, t2 - t0, bytes / (t2 - t0) / 1e6); }
    t0 = now(); CK(hipHostUnregister(h)); t1 = now();
    printf("hipHostUnregister: %.1f ms\n", t1 - t0);
    // pageable, column slabs (does the call block?)
    { t0 = now(); const long long cs = 1024; for (long long j = 0; j < N; j += cs) CK(hipMemcpyAsync(d + j * N, h + j * N, (size_t)cs * N * 8, hipMemcpyHostToDevice, s));
      t1 = now(); CK(hipStreamSynchronize(s)); double t2 = now();
      printf("pageable H2D in %lld slabs: calls %.1f ms, done %.1f ms = %.1f GB/s\n", N / cs, t1 - t0, t2 - t0, bytes / (t2 - t0) / 1e6); }
    return 0;
}

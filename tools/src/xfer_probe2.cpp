// Second transfer probe (rowsink.hip): what a device -> host copy into PAGEABLE memory waits for when another stream is busy, and the
// rate of a block-row copy (contiguous staging -> N runs of 2 KB) by destination kind.  hipcc -O2 -o /tmp/xfer_probe2 tools/src/xfer_probe2.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin_kernel(long long cycles, int *out) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} if (out) out[0] = 1; }
int main() {
    const long long N = 32768, nb = 256;
    const size_t bytes = (size_t)N * N * 8;
    double *h = (double *)aligned_alloc(4096, bytes);
    for (size_t i = 0; i < (size_t)N * N; i += 512) h[i] = 1.0;
    double *d, *stage; int *flag;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&stage, N * nb * 8)); CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
    hipStream_t sa, sb, sc;
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreate(&sa)); CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
    // ---- 1: pageable 4-byte D2H on stream sb while stream sa runs a 300-ms kernel ----------------------------------------------------
    for (int mode = 0; mode < 12; ++mode) {
        int *pinned; CK(hipHostMalloc(&pinned, 64));
        spin_kernel<<<1, 64, 0, sa>>>(300ll * 100000, nullptr);   // ~300 ms at 100 MHz clock64? (s_memtime counts at 100 MHz)
        double t0 = now();
        std::thread th([&] {
            CK(hipSetDevice(0));
            int v = 7;
            double a = now();
            const int mode0 = mode; const int mode = mode0 & 3;
            if (mode == 0) { CK(hipMemcpyAsync(&v, flag, 4, hipMemcpyDeviceToHost, sb)); CK(hipStreamSynchronize(sb)); }
            else if (mode == 1) { CK(hipMemcpyAsync(pinned, flag, 4, hipMemcpyDeviceToHost, sb)); CK(hipStreamSynchronize(sb)); }
            else if (mode == 2) { CK(hipMemcpyAsync(&v, flag, 4, hipMemcpyDeviceToHost, sc)); CK(hipStreamSynchronize(sc)); }
            else { CK(hipMemcpy2DAsync(h + 1024, N * 8, stage, nb * 8, nb * 8, N, hipMemcpyDeviceToHost, sb)); CK(hipStreamSynchronize(sb)); }
            printf("mode %d%s (%s): copy on another stream took %.2f ms while a kernel runs elsewhere\n", mode, mode0 >= 8 ? " + launching thread polling hipStreamQuery" : mode0 >= 4 ? " + launching thread in hipStreamSynchronize" : "",
                   mode == 0 ? "4 B to pageable, high-priority stream" : mode == 1 ? "4 B to pinned, high-priority stream" : mode == 2 ? "4 B to pageable, normal stream" : "16.8-MB block row to pageable", now() - a);
        });
        if (mode >= 8) { while (hipStreamQuery(sa) == hipErrorNotReady) std::this_thread::sleep_for(std::chrono::microseconds(100)); }
        else if (mode >= 4) CK(hipStreamSynchronize(sa));   // the launching thread waits INSIDE the runtime while the other thread copies
        th.join();
        CK(hipStreamSynchronize(sa));
        printf("   (the kernel itself: %.1f ms)\n", now() - t0);
        CK(hipHostFree(pinned));
    }
    // ---- 2: block-row copies on an idle device -------------------------------------------------------------------------------------------
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipMemcpy2DAsync(h + 4096 + rep * 256, N * 8, stage, nb * 8, nb * 8, N, hipMemcpyDeviceToHost, sb)); CK(hipStreamSynchronize(sb));
        printf("pageable, contiguous staging -> 32768 runs of 2 KB: %.2f ms = %.1f GB/s\n", now() - t0, N * nb * 8 / (now() - t0) / 1e6);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipMemcpy2DAsync(h + 8192 + rep * 256, N * 8, d + 8192, N * 8, nb * 8, N, hipMemcpyDeviceToHost, sb)); CK(hipStreamSynchronize(sb));
        printf("pageable, strided device rows -> 32768 runs of 2 KB: %.2f ms = %.1f GB/s\n", now() - t0, N * nb * 8 / (now() - t0) / 1e6);
    }
    for (int rep = 0; rep < 3; ++rep) {
        const long long hgt = 24320;
        double t0 = now(); CK(hipMemcpy2DAsync(h + (8192 + 256) * N + 8192, N * 8, d + (8192 + 256) * N + 8192, N * 8, nb * 8, hgt, hipMemcpyDeviceToHost, sb)); CK(hipStreamSynchronize(sb));
        printf("pageable, strided device rows -> 24320 runs of 2 KB (first probe's shape): %.2f ms = %.1f GB/s\n", now() - t0, hgt * nb * 8 / (now() - t0) / 1e6);
    }
    // a pinned bounce buffer + CPU scatter (what a staging ring would do)
    double *bounce; CK(hipHostMalloc(&bounce, N * nb * 8));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipMemcpyAsync(bounce, stage, N * nb * 8, hipMemcpyDeviceToHost, sb)); CK(hipStreamSynchronize(sb)); double t1 = now();
        for (long long c = 0; c < N; ++c) __builtin_memcpy(h + c * N + 12288, bounce + c * nb, nb * 8);
        printf("pinned bounce: copy %.2f ms (%.1f GB/s) + one thread's scatter %.2f ms\n", t1 - t0, N * nb * 8 / (t1 - t0) / 1e6, now() - t1);
    }
    return 0;
}

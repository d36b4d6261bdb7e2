/*
 * mpf_probe.h -- entry points that exist ONLY in libmpf_probe.so (tools/, bench.py's on-box peak measurements).
 *
 * libmpf_probe.so is libmpf_amd.so's sources compiled with -DMPF_PROBE: the whole C ABI of include/mpf_c.h plus the
 * register-only / stream-copy microbenchmarks, the cycle-stamped build of the pivot kernel (the 128-row build is a product kernel since round 5), the four-wave
 * A/B switch and the LDS-padding knobs of the fp64 update (options hp_stamp, hp_r256_upto, dgemm_w8, gemm_lds_pad).
 * None of that is reachable from libmpf_amd.so, which a product run loads.
 */
#ifndef MPF_PROBE_H
#define MPF_PROBE_H
#include "mpf_c.h"
#ifdef __cplusplus
extern "C" {
#endif
/* On-box peak probes: which = 0 f64-MFMA issue rate [TFLOP/s], 1 f16-MFMA issue rate [TFLOP/s], 2 HBM stream
 * copy read+write [TB/s]; other values: diagnostics used by tools/ (see csrc/microbench.hip).  Synchronous. */
int mpf_microbench(mpf_ctx *ctx, int which, double *result);
/* One gate launch (the kernel that lets the fp64 panel follow the pivot kernel) with no pivot kernel behind it: returns the
 * context's time-out counter after the gate has run (1 = the gate gave up and flagged it) and resets the counter. */
int mpf_debug_gate(mpf_ctx *ctx, int target);
/* The big fp16 update kernel ALONE (no operand conversion) on the fp16 images the last mpf_hgemm_minus_f32 call of this
 * context left behind: C (fp32, column-major, ldc) -= images.  Option hgemm_dbg: 1 = K loop only, 2 = C stream only. */
int mpf_debug_hgemm_again(mpf_ctx *ctx, int64_t m, int64_t n, int32_t k, float *d_C, int64_t ldc, int32_t split);
#ifdef __cplusplus
}
#endif
#endif

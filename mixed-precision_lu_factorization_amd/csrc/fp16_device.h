// Device-side fp16 helpers of the numeric contract (C1, C2), shared by the LDS-resident pivot kernel (fp16_panel.hip) and
// the generic global-memory one (fp16_panel_generic.hip).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ unsigned short h_bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ _Float16 bits_h(unsigned b) { return __builtin_bit_cast(_Float16, (unsigned short)b); }

// fp16_utils.h:15-23 double_to_fp16 (contract C1)
__device__ __forceinline__ unsigned short double_to_fp16_bits(double x) {
    float xf = (float)x;
    // The rounded value is hidden from the optimiser: left alone it turns the clamps into fp64 compares with a BRANCH around
    // the conversion, and a branch per element puts a full memory round trip behind every load of a panel (the pivot
    // kernels convert 128 elements per thread).  As selects on the fp32 value the loads stay in flight together.
    asm("" : "+v"(xf));
    const float FP16_MAX = 65504.0f;
    const float FP16_MIN_POS = 6.10352e-05f;
    xf = xf > FP16_MAX ? FP16_MAX : (xf < -FP16_MAX ? -FP16_MAX : xf);
    xf = (xf > -FP16_MIN_POS && xf < FP16_MIN_POS) ? 0.0f : xf;
    return h_bits((_Float16)xf);
}

// IEEE quotient of two fp16 values rounded once to fp16 (the '/' of hgetf2_kernel.cu:108).  The fp32
// operands are hidden from the optimiser so the division stays a correctly rounded fp32 division
// (24 >= 2*11+2 bits: rounding its result to fp16 equals rounding the exact quotient).
// The quotient itself: reciprocal, one Newton step on the quotient (exact residual through fma), special cases through
// v_div_fixup_f32 -- half the dependent instructions of the compiler's IEEE fp32 division (no range scaling: operands that come
// from fp16 cannot overflow or underflow fp32).  For fp16 operands the result is within 2^-24 of the exact quotient (exact when that
// is representable), and a quotient of two 11-bit significands is either a 12-bit rounding boundary itself or at least 2^-23 away
// from every one: rounding it to fp16 gives the correctly rounded fp16 quotient.  All 2^32 operand pairs are compared with the
// oracle's division (tests/test_gpu_steps.py::test_hdiv_ieee_all_pairs).
__device__ __forceinline__ _Float16 hdiv_ieee(_Float16 a, _Float16 b) {
    const float fa = (float)a, fb = (float)b;
    const float r = __builtin_amdgcn_rcpf(fb);
    float q = fa * r;
    const float e = __builtin_fmaf(-fb, q, fa);
    q = __builtin_fmaf(e, r, q);
    q = __builtin_amdgcn_div_fixupf(q, fb, fa);
    return (_Float16)q;
}

__device__ __forceinline__ unsigned bitrev8(unsigned x) { return __brev(x) >> 24; }
// order in which equal maxima are preferred (smaller wins); an involution on t = row - j: lowest 256-row block first
// (serial strict-'>' block scan, hgetf2_kernel.cu:73-78), then the smallest bit-reversed lane (strict-'>' tree, :47-56)
__device__ __forceinline__ unsigned tie_key(unsigned t) { return (t & ~255u) | bitrev8(t & 255u); }
// search key of a candidate: |a| in the high word, inverted tie order below; the maximum key is the reference's pivot
__device__ __forceinline__ unsigned long long pivot_key(unsigned hbits, unsigned t) {
    return ((unsigned long long)(hbits & 0x7FFFu) << 32) | (0xFFFFFFFFu - tie_key(t));
}

#pragma once
// Drop-in replacement for the reference's only public header (reference MPF.h:3).
// Same C++ linkage, same argument meaning, same in/out contract:
//   h_A   host pointer, N x N fp64, column-major, lda = N; overwritten with L\U
//   N     matrix order
//   r     panel width
//   IPIV  host int[N], pre-initialised by the caller to i+1 (reference benchmark.cpp:215-217);
//         on return LAPACK-style 1-based sequential swaps; IPIV[N-1] is left untouched when the
//         last panel is 1 x 1 (reference MPF.cu:104).
// Errors are printed, never returned (reference MPF.cu:72-75,134-138): with no HIP device the call
// prints to stderr and returns with both buffers untouched.
void MPF(double *h_A, int N, int r, int *IPIV);

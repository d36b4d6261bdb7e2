/*
 * mpf_c.h -- C ABI of the MI355X-native MPF hot path (libmpf_amd.so).
 *
 * Plain pointers and sizes only; no torch / C++ types.  Every entry point cites the piece of
 * the reference (paths relative to the reference repo) it replaces.  All `d_` pointers are
 * device (HBM) pointers; all matrices are fp64 column-major.  Calls are asynchronous on the
 * context's HIP stream unless stated otherwise.  Return value: 0 on success, < 0 on error
 * (mpf_last_error() gives the text), > 0 LAPACK-style "first zero pivot" where documented.
 */
#ifndef MPF_C_H
#define MPF_C_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpf_ctx mpf_ctx;

enum { MPF_TRAIL_FP64 = 0, MPF_TRAIL_FP16 = 1, MPF_TRAIL_FP16X3 = 2 };

typedef struct mpf_opts {
    int32_t trailing;    /* MPF_TRAIL_FP64: reference arithmetic (MPF.cu:215-239 in fp64).
                            MPF_TRAIL_FP16: fp16-in / fp32-accumulate MFMA trailing update.
                            MPF_TRAIL_FP16X3: the same with operands split hi + 2^-11 lo (three fp16 MFMA
                            products, ~22-bit operands): fp32-class factors at the same HBM-bound cost.
                            Both fp16 modes keep the matrix right of the current super-panel in an fp32 working
                            copy owned by the context (4 N^2 bytes of device memory, allocated at the first such
                            call; option fp16_work32 = 0 updates the fp64 matrix in place instead); panels, TRSMs and
                            the factors returned in d_A are fp64. */
    int32_t verbose;     /* 1: per-panel line on stdout like MPF.cu:137 */
    int32_t fused_panel; /* 0: separate fp64 mul/sub in the no-pivot panel (contract C3); 1: FMA */
    int32_t sync_timing; /* 1: no look-ahead, synchronise after every phase and fill the per-phase timers */
    int32_t no_lookahead;/* 1: single-stream schedule (panel k+1 only after the whole update k)   */
    int32_t superpanel;    /* 0: default (fp64: 1 = one-level loop; fp16 modes: 4); n > 1: n panels per super-panel, one
                              K = n * nb update of the matrix right of it (two-level schedule).  In the fp64 mode every
                              element keeps its fma chain, so the result does not depend on this value. */
    int32_t pivot_path;    /* 0: automatic -- the LDS-resident fp16 pivot kernel (its workgroups hand candidates to each other
                              inside one launch and must all be resident: one per CU) wherever the panel fits it, the generic
                              global-memory path otherwise (panels wider than 256 columns or taller than 256 rows x #CUs).
                              1: generic path and generic schedule always -- no kernel ever waits for another workgroup; for GPUs
                              shared with other processes (also option safe_pivots / MPF_SAFE_PIVOTS=1).  Results do not depend on this value. */
    int32_t reserved;
} mpf_opts;

typedef struct mpf_stats {
    double ms_total;  /* device time of the last mpf_factor_dev (hipEvents)        */
    double ms_h2d, ms_d2h; /* only mpf_factor_host (ms_d2h: wall clock after the factorization's end) */
    /* per-phase device time.  sync_timing=1: each phase alone.  Look-ahead schedule: HIP-event pairs
     * around the launches as they ran (ms_hpanel = whole panel chain on the side stream, ms_dpanel = 0;
     * ms_gemm = sum over the gemm_launches trailing-update kernel launches, concurrent panel work included; conversions of
     * operands / working-copy windows are booked under ms_cvt). */
    double ms_hpanel, ms_laswp, ms_dpanel, ms_trsm, ms_gemm;
    int64_t n;
    int32_t nb, panels;
    int32_t hpanel_timeouts; /* spin give-ups inside the fp16 pivot kernel (must be 0) */
    int32_t info;            /* first zero pivot (1-based) or 0                        */
    int32_t gemm_launches;   /* number of dgemm launches behind ms_gemm                */
    int32_t lookahead;       /* 1 if the look-ahead schedule ran                       */
    int32_t superpanel;      /* panels per super-panel the schedule really used (1 = one-level loop) */
    int32_t pivot_path;      /* 0: LDS-resident pivot kernel on every panel; 1: some panel took the generic (global-memory) one */
    double gemm_flops;       /* flops of the launches timed under ms_gemm (2 m n k each; fp16x3: counted once, not 3x) */
    double gemm_bytes;       /* algorithmic HBM bytes of the same launches: 16 per updated fp64 element + operand reads */
    /* fp16 trailing modes, two-level schedule: the K = sb * nb update launches ALONE (the fp16 MFMA kernel, no conversion, no
     * inside-super-panel update): HIP-event time, flops (2 m n K), algorithmic HBM bytes (C read + write at 4 or 8 bytes per
     * element + the fp16 operand images once) and number of launches -- what bench.py's mxp.roofline is made of. */
    double ms_gemm_big, gemm_big_flops, gemm_big_bytes;
    double ms_cvt;           /* operand-image conversions and fp32 <-> fp64 window conversions (not part of ms_gemm) */
    double ms_blockrow;      /* U block-row of the super-panels (part of ms_trsm) */
    int32_t gemm_big_launches;
    int32_t host_rows_streamed; /* mpf_factor_host: block rows (panels) that went to the caller's matrix WHILE the factorization ran (0: the
                                   matrix went back in one piece afterwards, as MPF.cu:245-247 does; ms_d2h is then that copy, otherwise
                                   what was left of the way home after the last kernel) */
    int32_t host_late_segments; /* mpf_factor_host: column segments of the matrix that went UP while the factorization had started on the
                                   first part (0: the whole matrix first, as MPF.cu:82; ms_h2d is then the whole upload, otherwise the first part's) */
    int32_t reserved;
} mpf_stats;

/* ---- lifetime -------------------------------------------------------------------------- */
/* Replaces the per-call cudaSetDevice/cudaMalloc/cublasCreate block, reference MPF.cu:69-97. */
int mpf_create(mpf_ctx **out, int device);
int mpf_destroy(mpf_ctx *ctx); /* reference MPF.cu:250-255 */
/* Use a caller-owned hipStream_t (e.g. torch's current stream) instead of the context's own. */
int mpf_set_stream(mpf_ctx *ctx, void *hip_stream);
int mpf_synchronize(mpf_ctx *ctx);
const char *mpf_last_error(mpf_ctx *ctx);
int mpf_get_stats(mpf_ctx *ctx, mpf_stats *out);
/* Per-context behaviour switches (schedule and kernel choices; results never depend on them unless stated).  A context takes
 * its defaults from the environment once, at mpf_create (variable MPF_<NAME>, upper case; e.g. MPF_SUPERPANEL for
 * "superpanel_fp16", see csrc/mpf_internal.h MpfTuning for the list); afterwards only these calls change them, so contexts
 * on different host threads are independent.  Names: safe_pivots, chain_pipeline, chain_pipeline_below, fp16_work32,
 * superpanel_fp16, superpanel_fp64, no_lookahead, verbose, timeline, hp_spin_limit, hp_gate_ticks, hp_acq_fence, hgemm_pad,
 * hgemm_split_pad, hgemm_big, hgemm_big_tile, dgemm_dma, generic_fused, fp64_rowmajor, fp64_rowmajor_min_n, dist_instalments, dist_instalment_min_bytes, lazy_gather, dpanel_fused_form, trsm_laswp_fused.  mpf_option_name enumerates them (returns the count). */
/* Rows of the tallest panel the LDS-resident pivot kernel takes -- all its workgroups must be resident at once -- beside `waiters`
 * workgroups of kernels that wait for its progress (0: alone; a negative value -w: beside the pipelined chain's gated interchange
 * kernel on a panel of w columns).  Derived from the kernels' LDS / register footprints and the occupancy API (csrc/fp16_panel.hip):
 * taller panels run unpipelined, or on the generic path.  form: 0 = the better of the two forms, 1 = the full-slab form (a CU per
 * 256 rows), 2 = the column-window form (two workgroups per CU). */
int64_t mpf_hgetf2_capacity_rows(mpf_ctx *ctx, int32_t waiters, int32_t form);
int mpf_set_option(mpf_ctx *ctx, const char *name, int64_t value);
int mpf_get_option(mpf_ctx *ctx, const char *name, int64_t *value);
int mpf_option_name(int32_t index, char *buf, int64_t buflen);
/* HIP analogue of the reference's capability probe, check_cooperative_groups.cu:4-48.
 * Writes a human-readable report into buf; returns the number of HIP devices or < 0. */
int mpf_device_report(char *buf, int64_t buflen);


/* ---- whole path ------------------------------------------------------------------------ */
/* The body of the reference's MPF() (MPF.cu:66-256) on HOST buffers: H2D, factor, D2H.
 * ipiv_host follows MPF.h:3 semantics (caller pre-initialises to identity).  The device copy of the matrix (N x N doubles) and of
 * the pivots stays in the context and only grows -- the reference allocates and frees it inside every call, MPF.cu:80-94,250-255 --
 * next to the fp64 mode's row-major working copy (another N x N doubles, mpf_factor_dev): a context that has factored an N x N host
 * matrix holds 2 x 8 N^2 bytes until mpf_trim or mpf_destroy.
 * Round 5: in the fp64 mode's look-ahead schedules (what MPF() runs) the transfers overlap the factorization -- from N = 4096 on
 * finished block rows of the factors go to A_host while it still runs (options host_sink, host_sink_min_n), from N = 16384 on only
 * the first quarter of the matrix goes up before the first panel (host_late_parts, host_first_pct, host_late_min_n); same bits
 * (csrc/rowsink.hip).  The call then starts short-lived host threads of its own, keeps ~330 MB of pinned memory and two
 * more N x N device buffers (staging of the block rows; the matrix as uploaded, from which the call repeats itself on the generic
 * pivot path if a pivot kernel gives up (-4) after rows have left).  mpf_stats.host_rows_streamed / host_late_segments say what ran.
 * On a negative return: with host_sink = 0 nothing has been copied back; otherwise A_host may hold finished block rows beside
 * untouched ones. */
int mpf_factor_host(mpf_ctx *ctx, double *A_host, int64_t N, int32_t nb, int32_t *ipiv_host,
                    const mpf_opts *opts);
/* Gives the context's large cached buffers back to the device (host-path copies, row-major / fp32 working copies); they are
 * allocated again on demand. */
int mpf_trim(mpf_ctx *ctx);
/* The panel loop MPF.cu:100-242 on a DEVICE-resident matrix (lda >= N).  d_ipiv: N int32,
 * entries for a skipped 1x1 tail are left untouched (MPF.cu:104).  Synchronises at the end.
 * Any panel width 1 <= nb <= 65535 (the tuned schedules cover nb <= 256 and N <= 256 x #CUs; the rest runs the generic
 * schedule -- same results, see mpf_opts.pivot_path).
 * Returns -4 when the LDS pivot kernel's bounded inter-workgroup wait gave up (only possible when something else holds
 * CUs for seconds, e.g. another process on the same GPU): d_A and d_ipiv are then INVALID (partially factored with
 * unusable pivots; nothing outside them was touched); call again on a fresh copy with pivot_path = 1. */
int mpf_factor_dev(mpf_ctx *ctx, double *d_A, int64_t lda, int64_t N, int32_t nb, int32_t *d_ipiv,
                   const mpf_opts *opts);

/* ---- step operators (each replaces one reference kernel / library call) ------------------ */
/* double_to_fp16_block, MPF.cu:20-25 (+ fp16_utils.h:15-23): out[i] = double_to_fp16(in[i]). */
int mpf_double_to_fp16(mpf_ctx *ctx, const double *d_in, uint16_t *d_out, int64_t n);
/* Element-wise fp16 division with the contract's IEEE semantics (the '/' of
 * hgetf2_kernel.cu:108); exposed so the division can be tested exhaustively. */
int mpf_hdiv(mpf_ctx *ctx, const uint16_t *d_a, const uint16_t *d_b, uint16_t *d_q, int64_t n);
/* Steps 1.1-3.2 of MPF.cu (:108-159) fused: read the fp64 panel d_A[0:rows, 0:cols] (leading
 * dimension lda), convert with double_to_fp16, run the fp16 partial-pivot LU of
 * HGETF2_kernel (hgetf2_kernel.cu:15-120) with the panel resident in LDS, and write
 * d_ipiv[j] = panel-local pivot + ipiv_offset (1-based; MPF.cu:152 uses ipiv_offset = k).
 * d_panel16_out (optional, may be NULL): receives the factored fp16 panel, rows x cols, ld =
 * rows, rows physically swapped as the reference leaves them -- for parity tests. */
int mpf_hgetf2_pivots(mpf_ctx *ctx, const double *d_A, int64_t lda, int32_t rows, int32_t cols,
                      int32_t ipiv_offset, int32_t *d_ipiv, uint16_t *d_panel16_out);
/* HGETF2_kernel itself (hgetf2_kernel.cu:15): fp16 panel in, factored in place, 1-based
 * panel-local pivots out. */
int mpf_hgetf2(mpf_ctx *ctx, uint16_t *d_panel16, int64_t ld, int32_t rows, int32_t cols,
               int32_t *d_ipiv_panel);
/* LASWP_kernel, MPF.cu:42-59: apply `cols` sequential swaps (row k+pc <-> d_ipiv_global[pc]-1)
 * to ncols columns of d_A. */
int mpf_laswp(mpf_ctx *ctx, double *d_A, int64_t lda, int64_t ncols, int32_t k, int32_t cols,
              const int32_t *d_ipiv_global);
/* dgetf2_native_npv, dgetf2_native_npv.cu:11-36, in place with leading dimension ld (no packed
 * copy: replaces the extract / write-back memcpy loops MPF.cu:168-200 too). */
int mpf_dgetf2_npv(mpf_ctx *ctx, double *d_P, int64_t ld, int32_t rows, int32_t cols, int32_t fused);
/* cublasDtrsm(LEFT, LOWER, N, UNIT, m, n, 1.0, L, ldl, B, ldb), call site MPF.cu:215-225. */
int mpf_dtrsm_llnu(mpf_ctx *ctx, int32_t m, int64_t n, const double *d_L, int64_t ldl, double *d_B,
                   int64_t ldb);
/* cublasDgemm(N, N, m, n, k, -1.0, A, lda, B, ldb, 1.0, C, ldc), call site MPF.cu:230-239. */
int mpf_dgemm_minus(mpf_ctx *ctx, int64_t m, int64_t n, int32_t k, const double *d_A, int64_t lda,
                    const double *d_B, int64_t ldb, double *d_C, int64_t ldc);

/* Build-added speed mode of the same update (BASELINE north_star): C -= fp16(A) * fp16(B) with
 * v_mfma_f32_32x32x16_f16, fp32 accumulation over k, one fp64 subtraction per element. */
int mpf_hgemm_minus(mpf_ctx *ctx, int64_t m, int64_t n, int32_t k, const double *d_A, int64_t lda,
                    const double *d_B, int64_t ldb, double *d_C, int64_t ldc, int32_t split /* 0: fp16, 1: fp16x3 */);

/* The same update on an fp32 matrix: C (float, column-major) -= fp16(A) * fp16(B).  This is the operation the two-level
 * schedule of the fp16 trailing modes runs on its fp32 working copy of the trailing matrix (8 instead of 16 bytes of HBM per
 * updated element); exposed as a step operator so that it can be tested against the oracle's tolerance formula. */
int mpf_hgemm_minus_f32(mpf_ctx *ctx, int64_t m, int64_t n, int32_t k, const double *d_A, int64_t lda,
                        const double *d_B, int64_t ldb, float *d_C, int64_t ldc, int32_t split);

/* The fp32 working copy itself (row-major: element (i, j) at d_W[i * ldw + j]; the fp64 matrix is column-major), as step
 * operators: conversion of a rows x cols window in both directions, and LASWP_kernel's interchange (MPF.cu:42-59: `cols`
 * sequential swaps row k + pc <-> d_ipiv_global[pc] - 1) on ncols columns of the copy.  mpf_w32_laswp needs scratch of
 * 8 * N * max(cols, 256) bytes, which mpf_factor_dev allocates; stand-alone calls allocate it themselves. */
int mpf_w32_from_f64(mpf_ctx *ctx, const double *d_A, int64_t lda, float *d_W, int64_t ldw, int64_t rows, int64_t cols);
int mpf_w32_to_f64(mpf_ctx *ctx, const float *d_W, int64_t ldw, double *d_A, int64_t lda, int64_t rows, int64_t cols);
int mpf_w32_laswp(mpf_ctx *ctx, float *d_W, int64_t ldw, int64_t ncols, int32_t k, int32_t cols, const int32_t *d_ipiv_global);

/* ---- the reference generator's stream on the device (matrix_generator.cpp:55-80 as benchmark.cpp:192-194 reads it) ----
 * d_A[col * lda + row] = (rand() % 100) / 10.0 for t = col * N + row = 0 .. N^2-1 in order, rand() = glibc's default
 * generator, never seeded, after `skip` earlier draws (`matgen f N (N-2) lin` emits a 2 x 2 first: skip = 4).  Bit-identical
 * to the reference binary's file as benchmark.cpp parses it; no host-side N^2 work.  Synchronous.
 * _cols_: only columns [col0, col0 + ncols) of that N x N matrix, written to d_A's columns 0 .. ncols-1 (1-D block-column
 * layouts generate their own blocks).  mpf_matgen_state: host-only check of the jump-ahead (31 raw words in front of
 * rand() call number `call`). */
int mpf_matgen_dev(mpf_ctx *ctx, double *d_A, int64_t lda, int64_t N, int64_t skip);
int mpf_matgen_cols_dev(mpf_ctx *ctx, double *d_A, int64_t lda, int64_t N, int64_t skip, int64_t col0, int64_t ncols);
int mpf_matgen_state(int64_t call, uint32_t *out31);

/* ---- the reference's acceptance test at scale (benchmark.cpp:106-144: get_LU, L * U, row_permute, |A - P L U| <= 1e-10) ----
 * The reference multiplies L * U on the host with CBLAS (benchmark.cpp:77-82): 7e13 flops at N = 32768.  Here P^T A - L U is
 * formed on the device with the library's fp64 MFMA GEMM.  max_abs_err is the reference's criterion (compare with 1e-10),
 * fro_rel_err = ||A - P L U||_F / ||A||_F.  Needs 3 N^2 doubles of device scratch.  _host: host buffers (ld = N), device 0. */
int mpf_check_plu_dev(mpf_ctx *ctx, const double *d_A, int64_t lda, const double *d_LU, int64_t ldlu, const int32_t *d_ipiv,
                      int64_t N, double *max_abs_err, double *fro_rel_err);
int mpf_check_plu_host(const double *A, const double *LU, const int32_t *ipiv, int64_t N, double *max_abs_err, double *fro_rel_err);

/* ---- build-added solve (no reference counterpart; BASELINE north_star) -------------------- */
typedef struct mpf_ir_stats {
    int32_t iterations;   /* correction steps taken */
    int32_t converged;
    double rel_residual;  /* ||b - A x||_2 / ||b||_2 at exit */
    double history[32];   /* residual after step i (history[0] = after the first solve) */
    double ms_total;
    int32_t stalled;      /* 1: stopped early because the residual stopped shrinking (ratio > 0.7 twice in a row) */
    int32_t reserved;
} mpf_ir_stats;
/* Solve A x = b with the factors produced by mpf_factor_dev and fp64 iterative refinement:
 * x0 = U^-1 L^-1 P b; repeat r = b - A x (fp64), x += U^-1 L^-1 P r until
 * ||r||/||b|| <= tol or max_iter corrections.  d_A is the ORIGINAL matrix, d_LU the factors. */
int mpf_solve_ir(mpf_ctx *ctx, const double *d_A, int64_t lda, const double *d_LU, int64_t ldlu,
                 const int32_t *d_ipiv, int64_t N, const double *d_b, double *d_x, int32_t max_iter,
                 double tol, mpf_ir_stats *stats);

/* The same for nrhs right-hand sides: d_B / d_X are N x nrhs column-major (ldb, ldx >= N); stats = array of nrhs entries (or NULL).
 * The factors' diagonal-block inverses and the pivot gather index are prepared once. */
int mpf_solve_ir_nrhs(mpf_ctx *ctx, const double *d_A, int64_t lda, const double *d_LU, int64_t ldlu, const int32_t *d_ipiv,
                      int64_t N, int32_t nrhs, const double *d_B, int64_t ldb, double *d_X, int64_t ldx, int32_t max_iter,
                      double tol, mpf_ir_stats *stats);

/* GMRES-IR (Carson & Higham): refinement whose correction equation A d = r is solved by GMRES (restart `restart`, <= 100)
 * preconditioned with the factors, everything in fp64.  Converges where plain refinement does not contract -- the
 * generator's own matrices with MPF_TRAIL_FP16 factors -- at the price of one factor solve + one matrix-vector product per
 * inner iteration.  history[i] = ||b - A x|| / ||b|| before outer step i. */
typedef struct mpf_gmres_stats {
    int32_t outer_iterations, inner_iterations, converged;
    int32_t budget_expired;   /* 1: stopped by the wall-clock limit mpf_gesv gives it (the estimated time of an fp64 refactorization) */
    double rel_residual;
    double history[32];
    double ms_total;
} mpf_gmres_stats;
int mpf_solve_gmres_ir(mpf_ctx *ctx, const double *d_A, int64_t lda, const double *d_LU, int64_t ldlu, const int32_t *d_ipiv,
                       int64_t N, const double *d_b, double *d_x, int32_t max_outer, int32_t restart, double tol,
                       mpf_gmres_stats *stats);

/* Solve A x = b end to end with the fastest path that reaches the tolerance: (1) factor a copy of A in the fp16
 * trailing mode and refine in fp64; (2) if the refinement stalls or diverges (ill-conditioned input: kappa * 2^-11
 * is not << 1), factor again with the fp64 trailing update (the reference arithmetic) and solve with that.
 * d_A is preserved; d_work is an N x N fp64 scratch (ld = N) that holds the factors on return; d_ipiv N int32. */
typedef struct mpf_gesv_stats {
    int32_t path;            /* 1: fp16 trailing + refinement, 2: fp64 fallback, 3: fp16 trailing + GMRES-IR */
    int32_t info;
    double ms_factor_fp16, ms_ir_fp16, ms_factor_fp64, ms_ir_fp64, ms_total;
    mpf_ir_stats ir_fp16, ir_final;
    double gmres_budget_ms;  /* try_fp16 = 3: the wall-clock limit GMRES-IR ran under (0: it did not run) */
    int32_t gmres_budget_expired, reserved;
} mpf_gesv_stats;
int mpf_gesv(mpf_ctx *ctx, const double *d_A, int64_t lda, int64_t N, int32_t nb, double *d_work, int32_t *d_ipiv,
             const double *d_b, double *d_x, int32_t max_iter, double tol,
             int32_t try_fp16 /* 0: fp64 only, 1: fp16, 2: fp16x3, 3: fp16 and, if plain refinement stalls, GMRES-IR on the same factors --
                                  for at most the time an fp64 refactorization is estimated to take (from this context's last measured
                                  fp64 factorization rate, option gesv_fp64_tflops to override), then the fp64 path;
                                  GMRES-IR with the caller's own limits: mpf_solve_gmres_ir */,
             mpf_gesv_stats *stats);

/* ---- multi-GPU (build extension; the reference is single-device, MPF.cu:77) ------------------------------------
 * One process per GPU.  1-D block-cyclic columns: global column block b (nb columns) lives on rank b % world as local block
 * b / world; d_Aloc is the rank's N x (local columns) column-major matrix (ldloc >= N), d_ipiv the full pivot vector (N int32,
 * identity-initialised like MPF.h:3 wants it), replicated on every rank on return.  Per panel ONE exchange: the owner
 * broadcasts the factored panel + pivots + moved-row list; everything else is local.  Results are bit-identical to
 * mpf_factor_dev.  Panel width <= 256.
 * The exchange goes through callbacks so any transport can carry it (both are called on every rank, in the same order,
 * with a HIP stream the transfer must be ordered on; return 0 on success):
 *   bcast(user, d_buf, bytes, root, stream)        d_buf is the message on the root and the landing buffer elsewhere
 *   allreduce(user, d_buf, count, stream)          in-place sum of `count` doubles (refinement residual only)
 * NULL callbacks select the context's RCCL communicator (mpf_rccl_init: ncclBroadcast / ncclAllReduce over xGMI). */
typedef int (*mpf_bcast_fn)(void *user, void *d_buf, int64_t bytes, int32_t root, void *hip_stream);
typedef int (*mpf_allreduce_fn)(void *user, double *d_buf, int64_t count, void *hip_stream);
typedef struct mpf_dist {
    int32_t rank, world;
    mpf_bcast_fn bcast;
    mpf_allreduce_fn allreduce;
    void *user;
} mpf_dist;
/* RCCL communicator owned by the context (librccl is resolved with dlopen at the first call: single-GPU users never load it).
 * mpf_rccl_unique_id: 128 bytes to be produced on one rank and handed to all (e.g. through torch.distributed / MPI). */
int mpf_rccl_unique_id(void *out128);
int mpf_rccl_init(mpf_ctx *ctx, const void *id128, int32_t rank, int32_t world);
int mpf_rccl_destroy(mpf_ctx *ctx);
int mpf_rccl_version(void); /* ncclGetVersion(), or < 0 when librccl cannot be loaded */
/* What the context's communicator is (ncclCommCount / ncclCommUserRank: what RCCL itself says) and what this library has sent over
 * it since mpf_rccl_init; link_type / link_hops / peer_access: hipExtGetLinkTypeAndHopCount and hipDeviceCanAccessPeer from the
 * context's device to each visible device (index = device ordinal, up to 16). */
typedef struct mpf_rccl_info_t {
    int32_t version, has_comm, comm_count, comm_rank, has_p2p, device, visible_devices, reserved;
    int64_t bcast_calls, bcast_bytes, allreduce_calls, p2p_calls, p2p_bytes;
    int32_t link_type[16], link_hops[16], peer_access[16];
} mpf_rccl_info_t;
int mpf_rccl_info(mpf_ctx *ctx, mpf_rccl_info_t *out);
/* `reps` broadcasts of `bytes` from `root` on the communicator, timed on the context's stream: ms per broadcast (every rank calls it) */
int mpf_rccl_bcast_probe(mpf_ctx *ctx, int64_t bytes, int32_t root, int32_t reps, double *ms_per_bcast);
int mpf_rccl_selftest(mpf_ctx *ctx); /* one small broadcast + all-reduce on the communicator (every rank calls it) */
/* The panel loop MPF.cu:100-242 over the block-cyclic layout (look-ahead schedule: the owner of panel k+1 updates that block
 * first, runs its chain and posts the broadcast on a side stream under everybody's update k).  Returns this rank's info. */
int mpf_factor_dist(mpf_ctx *ctx, double *d_Aloc, int64_t ldloc, int64_t N, int32_t nb, int32_t *d_ipiv, const mpf_dist *dist,
                    const mpf_opts *opts);
/* Optional point-to-point transport next to the caller's own broadcast / all-reduce callbacks (with NULL callbacks in mpf_dist the
 * context's RCCL communicator supplies ncclSend / ncclRecv by itself):
 *   p2p(user, d_buf, bytes, peer, send, stream)    send != 0: d_buf goes to rank `peer`; send == 0: bytes from `peer` land in d_buf
 * Every send has exactly one matching receive, issued in the same order on both ranks. */
typedef int (*mpf_p2p_fn)(void *user, void *d_buf, int64_t bytes, int32_t peer, int32_t send, void *hip_stream);
int mpf_dist_set_p2p(mpf_ctx *ctx, mpf_p2p_fn fn, void *user);
/* mpf_solve_ir over the same layout: d_Aloc = the rank's columns of the ORIGINAL matrix, d_LUloc = of the factors; d_b and
 * d_x (N each) replicated.  Residual = local GEMV + all-reduce.  The triangular solves walk the column blocks; with a point-to-point
 * transport the running vector travels from owner to owner (2 (N / nb - 1) sends + one all-reduce per factor solve), otherwise the
 * owner applies a block to the replicated vector and broadcasts it on (2 N / nb broadcasts).  nb must be a multiple of 64. */
int mpf_solve_ir_dist(mpf_ctx *ctx, const double *d_Aloc, int64_t lda, const double *d_LUloc, int64_t ldlu, const int32_t *d_ipiv,
                      int64_t N, int32_t nb, const double *d_b, double *d_x, int32_t max_iter, double tol, const mpf_dist *dist,
                      mpf_ir_stats *stats);

#ifdef __cplusplus
}
#endif
#endif

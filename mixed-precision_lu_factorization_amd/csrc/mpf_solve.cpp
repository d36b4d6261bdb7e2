// Build-added solve on the factors of mpf_factor_dev (no reference counterpart: MPF.h:3 declares the factorization only;
// BASELINE north_star asks for the refinement sweep and "IR iterations to ||r||/||b|| < 1e-12"):
//   mpf_solve_ir          x0 = U^-1 L^-1 P b, then classical refinement with fp64 residuals
//   mpf_solve_ir_nrhs     the same for several right-hand sides (factors' diagonal-block inverses prepared once)
//   mpf_solve_gmres_ir    GMRES-IR (Carson & Higham): the correction equation A d = r is solved by GMRES preconditioned
//                         with the low-precision factors, in fp64 -- for inputs where plain refinement does not contract
//                         (kappa(A) times the factors' error is not << 1: the generator's own matrices in the fp16 mode)
#include "mpf_internal.h"
#include <chrono>
#include <cmath>
#include <cstring>

namespace {
// what every solve on one set of factors shares: the pivot sequence as a gather index and the inverted diagonal blocks
int solve_setup(mpf_ctx *c, const double *d_LU, int64_t ldlu, const int32_t *d_ipiv, int64_t N) {
    int rc = mpf_ensure_solve_buf(c, N);
    if (rc) return rc;
    // row permutation as a gather index: perm = P applied to identity (reverse of benchmark.cpp:84-95)
    std::vector<int32_t> ip((size_t)N), perm((size_t)N);
    MPF_HIP_TRY(c, hipMemcpyAsync(ip.data(), d_ipiv, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < N; ++i) perm[(size_t)i] = (int32_t)i;
    for (int64_t i = 0; i < N; ++i) {
        const int64_t p = (int64_t)ip[(size_t)i] - 1;
        if (p < 0 || p >= N) { c->err = "solve: ipiv entry out of range"; return -1; }
        if (p != i) std::swap(perm[(size_t)i], perm[(size_t)p]);
    }
    MPF_HIP_TRY(c, hipMemcpyAsync(c->perm_buf, perm.data(), (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    MPF_HIP_TRY(c, hipStreamSynchronize(c->stream)); // `perm` is a host temporary
    MPF_HIP_TRY(c, hipMemsetAsync(&c->ws->flags[0], 0, sizeof(int), c->stream));
    return launch_trsv_prepare(c, d_LU, ldlu, N);
}
// after the stream has been synchronised: did a bounded wait of the triangular solves give up?  (Then the vectors are not to be trusted.)
int solve_check_waits(mpf_ctx *c) {
    int flags = 0;
    MPF_HIP_TRY(c, hipMemcpy(&flags, &c->ws->flags[0], sizeof(int), hipMemcpyDeviceToHost));
    if (flags) { c->err = "solve: a step's wait for its neighbours timed out (GPU shared with another job?)"; return -4; }
    return 0;
}
int lu_solve(mpf_ctx *c, const double *d_LU, int64_t ldlu, int64_t N, const double *rhs, double *out) { // out = U^-1 L^-1 P rhs
    int e = launch_gather_rows(c, rhs, c->perm_buf, out, N);
    if (!e) e = launch_trsv_lower_unit(c, d_LU, ldlu, out, N);
    if (!e) e = launch_trsv_upper(c, d_LU, ldlu, out, N);
    return e;
}
int read_scalar(mpf_ctx *c, const double *d, double &out) {
    double h = 0;
    MPF_HIP_TRY(c, hipMemcpyAsync(&h, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
    out = h;
    return 0;
}
int norm2(mpf_ctx *c, const double *v, int64_t N, double &out) {
    double *scal = c->solve_buf + 4 * c->solve_n;
    int e = launch_norm2(c, v, N, scal);
    if (!e) e = read_scalar(c, scal, out);
    out = std::sqrt(out);
    return e;
}
int dot(mpf_ctx *c, const double *x, const double *y, int64_t N, double &out) {
    double *scal = c->solve_buf + 4 * c->solve_n;
    int e = launch_dot(c, x, y, N, scal);
    if (!e) e = read_scalar(c, scal, out);
    return e;
}
// classical refinement on prepared factors
int ir_core(mpf_ctx *c, const double *d_A, int64_t lda, const double *d_LU, int64_t ldlu, int64_t N, const double *d_b, double *d_x,
            int32_t max_iter, double tol, mpf_ir_stats &st) {
    const int64_t S = c->solve_n;
    double *r = c->solve_buf, *d = c->solve_buf + S;
    double nb2 = 0;
    int rc = norm2(c, d_b, N, nb2);
    if (rc) return rc;
    if (nb2 == 0) nb2 = 1;
    rc = lu_solve(c, d_LU, ldlu, N, d_b, d_x);
    if (rc) return rc;
    for (int it = 0;; ++it) {
        rc = launch_residual(c, d_A, lda, d_x, d_b, r, N);
        if (rc) return rc;
        double nr = 0;
        rc = norm2(c, r, N, nr);
        if (rc) return rc;
        st.rel_residual = nr / nb2;
        st.history[it] = st.rel_residual;
        st.iterations = it;
        if (st.rel_residual <= tol) { st.converged = 1; break; }
        if (it >= max_iter || !(st.rel_residual == st.rel_residual)) break;
        if (it >= 2 && st.history[it] > 0.7 * st.history[it - 1] && st.history[it - 1] > 0.7 * st.history[it - 2]) {
            st.stalled = 1; // plain refinement is not contracting: kappa(A) is too large for these factors
            break;
        }
        rc = lu_solve(c, d_LU, ldlu, N, r, d);
        if (rc) return rc;
        rc = launch_axpy(c, 1.0, d, d_x, N);
        if (rc) return rc;
    }
    return 0;
}
} // namespace

extern "C" {

int mpf_solve_ir(mpf_ctx *c, const double *d_A, int64_t lda, const double *d_LU, int64_t ldlu, const int32_t *d_ipiv,
                 int64_t N, const double *d_b, double *d_x, int32_t max_iter, double tol, mpf_ir_stats *stats) {
    return mpf_solve_ir_nrhs(c, d_A, lda, d_LU, ldlu, d_ipiv, N, 1, d_b, N, d_x, N, max_iter, tol, stats);
}

int mpf_solve_ir_nrhs(mpf_ctx *c, const double *d_A, int64_t lda, const double *d_LU, int64_t ldlu, const int32_t *d_ipiv,
                      int64_t N, int32_t nrhs, const double *d_B, int64_t ldb, double *d_X, int64_t ldx, int32_t max_iter,
                      double tol, mpf_ir_stats *stats) {
    if (!c || !d_A || !d_LU || !d_ipiv || !d_B || !d_X) return -1;
    if (N <= 0 || nrhs < 0) { c->err = "solve: N must be positive, nrhs >= 0"; return -1; }
    if (nrhs > 1 && (ldb < N || ldx < N)) { c->err = "solve: ldb / ldx < N"; return -1; }
    if (max_iter > 31) max_iter = 31;
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    hipEventRecord(c->ev0, c->stream);
    int rc = solve_setup(c, d_LU, ldlu, d_ipiv, N);
    if (rc) return rc;
    for (int j = 0; j < nrhs; ++j) {
        mpf_ir_stats st{};
        hipEvent_t e0 = nullptr;
        if (j > 0) { hipEventCreate(&e0); hipEventRecord(e0, c->stream); }
        rc = ir_core(c, d_A, lda, d_LU, ldlu, N, d_B + (int64_t)j * ldb, d_X + (int64_t)j * ldx, max_iter, tol, st);
        if (rc) { if (e0) hipEventDestroy(e0); return rc; }
        hipEventRecord(c->ev1, c->stream);
        MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
        float ms = 0;
        hipEventElapsedTime(&ms, j > 0 ? e0 : c->ev0, c->ev1); // the first right-hand side carries the set-up
        if (e0) hipEventDestroy(e0);
        st.ms_total = ms;
        if (stats) stats[j] = st;
    }
    return solve_check_waits(c);
}

int mpf_solve_gmres_ir(mpf_ctx *c, const double *d_A, int64_t lda, const double *d_LU, int64_t ldlu, const int32_t *d_ipiv,
                       int64_t N, const double *d_b, double *d_x, int32_t max_outer, int32_t restart, double tol,
                       mpf_gmres_stats *stats) {
    if (!c || !d_A || !d_LU || !d_ipiv || !d_b || !d_x) return -1;
    if (N <= 0) { c->err = "gmres: N must be positive"; return -1; }
    if (restart < 1) restart = 30;
    if (restart > 100) restart = 100;
    if (max_outer < 1) max_outer = 1;
    if (max_outer > 31) max_outer = 31;
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    hipEventRecord(c->ev0, c->stream);
    int rc = solve_setup(c, d_LU, ldlu, d_ipiv, N);
    if (rc) return rc;
    const int m = restart;
    const size_t vneed = (size_t)(m + 1) * (size_t)N;
    if (vneed > c->krylov_cap) {
        if (c->krylov) hipFree(c->krylov);
        c->krylov = nullptr; c->krylov_cap = 0;
        MPF_HIP_TRY(c, hipMalloc((void **)&c->krylov, vneed * sizeof(double)));
        c->krylov_cap = vneed;
    }
    double *V = c->krylov;
    const int64_t S = c->solve_n;
    double *r = c->solve_buf, *w = c->solve_buf + S;
    mpf_gmres_stats st{};
    double nb2 = 0;
    rc = norm2(c, d_b, N, nb2);
    if (rc) return rc;
    if (nb2 == 0) nb2 = 1;
    rc = lu_solve(c, d_LU, ldlu, N, d_b, d_x);
    if (rc) return rc;
    std::vector<double> H((size_t)(m + 1) * m), cs(m), sn(m), g(m + 1), y(m);
    // mpf_gesv gives GMRES-IR the time an fp64 refactorization would take (c->gmres_budget_ms; 0: no limit): every inner
    // iteration ends with a host read-back, so the wall clock is the measure
    const auto t_start = std::chrono::steady_clock::now();
    bool out_of_time = false;
    for (int outer = 0;; ++outer) {
        rc = launch_residual(c, d_A, lda, d_x, d_b, r, N);              // r = b - A x in fp64
        if (rc) return rc;
        double nr = 0;
        rc = norm2(c, r, N, nr);
        if (rc) return rc;
        st.rel_residual = nr / nb2;
        st.history[outer] = st.rel_residual;
        st.outer_iterations = outer;
        if (st.rel_residual <= tol) { st.converged = 1; break; }
        if (outer >= max_outer || out_of_time || !(st.rel_residual == st.rel_residual)) break;
        // ---- GMRES on M^-1 A d = M^-1 r, M = P^T L U (the factors), modified Gram-Schmidt, Givens rotations on the host ----
        rc = lu_solve(c, d_LU, ldlu, N, r, V);                           // z0 = M^-1 r
        if (rc) return rc;
        double beta = 0;
        rc = norm2(c, V, N, beta);
        if (rc) return rc;
        if (beta == 0 || !(beta == beta)) break;
        rc = launch_scal(c, 1.0 / beta, V, N);
        if (rc) return rc;
        std::fill(g.begin(), g.end(), 0.0);
        g[0] = beta;
        // inner tolerance: reduce the preconditioned residual far enough for this outer step to reach the target
        const double inner_tol = std::max(1e-14, std::min(1e-2, 0.1 * tol / st.rel_residual));
        int k = 0;
        for (; k < m; ++k) {
            double *vk = V + (int64_t)k * N, *vn = V + (int64_t)(k + 1) * N;
            rc = launch_residual(c, d_A, lda, vk, nullptr, w, N);        // w = -A v_k
            if (!rc) rc = launch_scal(c, -1.0, w, N);
            if (!rc) rc = lu_solve(c, d_LU, ldlu, N, w, vn);             // v_{k+1} = M^-1 A v_k
            if (rc) return rc;
            for (int i = 0; i <= k; ++i) {
                double h = 0;
                rc = dot(c, vn, V + (int64_t)i * N, N, h);
                if (!rc) rc = launch_axpy(c, -h, V + (int64_t)i * N, vn, N);
                if (rc) return rc;
                H[(size_t)i * m + k] = h;
            }
            double hn = 0;
            rc = norm2(c, vn, N, hn);
            if (rc) return rc;
            H[(size_t)(k + 1) * m + k] = hn;
            if (hn > 0) { rc = launch_scal(c, 1.0 / hn, vn, N); if (rc) return rc; }
            for (int i = 0; i < k; ++i) {                                // earlier rotations on the new column
                const double t = cs[i] * H[(size_t)i * m + k] + sn[i] * H[(size_t)(i + 1) * m + k];
                H[(size_t)(i + 1) * m + k] = -sn[i] * H[(size_t)i * m + k] + cs[i] * H[(size_t)(i + 1) * m + k];
                H[(size_t)i * m + k] = t;
            }
            const double a = H[(size_t)k * m + k], b2 = H[(size_t)(k + 1) * m + k], den = std::hypot(a, b2);
            cs[k] = den > 0 ? a / den : 1.0;
            sn[k] = den > 0 ? b2 / den : 0.0;
            H[(size_t)k * m + k] = den;
            H[(size_t)(k + 1) * m + k] = 0;
            g[k + 1] = -sn[k] * g[k];
            g[k] = cs[k] * g[k];
            st.inner_iterations++;
            if (c->gmres_budget_ms > 0 &&
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count() > c->gmres_budget_ms) { out_of_time = true; st.budget_expired = 1; }
            if (std::fabs(g[k + 1]) <= inner_tol * beta || hn == 0 || out_of_time) { ++k; break; }
        }
        for (int i = k - 1; i >= 0; --i) {                               // back substitution
            double s2 = g[i];
            for (int j = i + 1; j < k; ++j) s2 -= H[(size_t)i * m + j] * y[j];
            y[i] = s2 / H[(size_t)i * m + i];
        }
        for (int i = 0; i < k; ++i) { rc = launch_axpy(c, y[i], V + (int64_t)i * N, d_x, N); if (rc) return rc; } // x += V y
    }
    hipEventRecord(c->ev1, c->stream);
    MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    st.ms_total = ms;
    if (stats) *stats = st;
    return solve_check_waits(c);
}

} // extern "C"

/* Sanitizer driver of the CPU side (the reference's analogue is cuda-gdb memcheck, .vscode/launch.json:28).
 * Test infrastructure only: built by `make -C oracle san` with -fsanitize=address,undefined together with mpf_oracle.c and run
 * by tests/test_sanitizers.py.  Exercises every exported oracle entry point on small generator matrices, including ragged
 * panels, a 1x1 tail, sparsity (zero pivots) and the all-pairs division check on one divisor block. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

uint16_t orc_f32_to_f16_hw(float f);
void orc_double_to_fp16_block(const double *in, uint16_t *out, int64_t n);
void orc_fp16_to_double_block(const uint16_t *in, double *out, int64_t n);
void orc_hdiv_block(const uint16_t *a, const uint16_t *b, uint16_t *q, int64_t n);
int orc_hgetf2(uint16_t *panel, int64_t ld, int rows, int cols, int *ipiv_panel);
int orc_panel_pivots(const double *A, int64_t lda, int rows, int cols, int *ipiv_panel);
void orc_laswp(double *A, int64_t lda, int64_t ncols, int k, int cols, const int *ipiv_panel);
void orc_dgetf2_npv(int m, int n, double *panel, int64_t ld, int fused);
void orc_dtrsm_llnu(int m, int64_t n, const double *L, int64_t ldl, double *B, int64_t ldb);
int orc_mpf(double *A, int N, int r, int *IPIV, int fused_panel);
double orc_check_plu(const double *A, const double *LU, const int *ipiv, int n, double *fro);
void orc_rand_stream(unsigned seed, int *out, int n);
int orc_matgen(double *out, int want_n, int step, int func_exp, double sparsity);
void orc_matgen_skip(double *out, int n, int64_t skip);
void orc_lu_solve(const double *LU, const int *ipiv, int n, double *x);
double orc_residual(const double *A, const double *x, const double *b, int n, double *r);

int main(void) {
    int bad = 0;
    const int sizes[] = {1, 2, 3, 5, 8, 31, 32, 33, 64, 100, 129, 257};
    const int widths[] = {1, 3, 32, 64, 300};
    for (unsigned si = 0; si < sizeof sizes / sizeof sizes[0]; ++si)
        for (unsigned wi = 0; wi < sizeof widths / sizeof widths[0]; ++wi) {
            const int n = sizes[si], r = widths[wi];
            double *A = malloc(sizeof(double) * n * n), *LU = malloc(sizeof(double) * n * n);
            int *ip = malloc(sizeof(int) * n);
            orc_matgen_skip(A, n, 4 + si);
            memcpy(LU, A, sizeof(double) * n * n);
            for (int i = 0; i < n; ++i) ip[i] = i + 1;                 /* benchmark.cpp:215-217 */
            orc_mpf(LU, n, r, ip, (int)(wi & 1));
            double fro = 0;
            const double mx = orc_check_plu(A, LU, ip, n, &fro);
            if (!(mx <= 1e-10)) { printf("n=%d r=%d max|A-PLU|=%g\n", n, r, mx); bad++; }
            double *x = malloc(sizeof(double) * n), *b = malloc(sizeof(double) * n), *res = malloc(sizeof(double) * n);
            for (int i = 0; i < n; ++i) { b[i] = 0; for (int j = 0; j < n; ++j) b[i] += A[i + (size_t)j * n]; x[i] = b[i]; }
            orc_lu_solve(LU, ip, n, x);
            (void)orc_residual(A, x, b, n, res);
            free(x); free(b); free(res); free(A); free(LU); free(ip);
        }
    { /* generator with sparsity (zero pivots, inf / nan in the fp16 panel) through every step operator */
        const int n = 48;
        double *A = malloc(sizeof(double) * n * n);
        if (orc_matgen(A, n, 46, 0, 0.6) != 0) { printf("matgen failed\n"); bad++; }
        uint16_t *P = malloc(sizeof(uint16_t) * n * 16);
        double *back = malloc(sizeof(double) * n * 16);
        orc_double_to_fp16_block(A, P, (int64_t)n * 16);
        orc_fp16_to_double_block(P, back, (int64_t)n * 16);
        int piv[16];
        orc_hgetf2(P, n, n, 16, piv);
        orc_panel_pivots(A, n, n, 16, piv);
        for (int j = 0; j < 16; ++j) if (piv[j] < j + 1 || piv[j] > n) { printf("pivot out of range\n"); bad++; }
        orc_laswp(A, n, n, 0, 16, piv);
        orc_dgetf2_npv(n, 16, A, n, 0);
        orc_dtrsm_llnu(16, n - 16, A, n, A + (size_t)16 * n, n);
        free(back); free(P); free(A);
    }
    { /* element-wise helpers */
        uint16_t a[256], b[256], q[256];
        for (int i = 0; i < 256; ++i) { a[i] = (uint16_t)(i * 257u); b[i] = (uint16_t)(0x3c00u + i); }
        orc_hdiv_block(a, b, q, 256);
        (void)orc_f32_to_f16_hw(1.5f);
        int st[64];
        orc_rand_stream(1, st, 64);
        if (st[0] != 1804289383) { printf("rand stream KAT failed\n"); bad++; }
    }
    printf(bad ? "SAN DRIVER FAILED\n" : "san driver ok\n");
    return bad ? 1 : 0;
}

// mpf_bench -- the reference's benchmark workflow (benchmark.cpp:146-270) on top of libmpf_amd.so.
//
//   mpf_bench filename [-v] [--no-check] [-r PANEL]
//
// Reads a file written by the reference generator (or harness/mpf_matgen), and for every matrix in it:
//   * times MPF(data, n, 32, ipiv) exactly as benchmark.cpp:212-222 does (host buffers, identity IPIV, wall clock
//     around the whole call, transfers included);
//   * checks A == P * (L * U) to 1e-10 absolute per element (get_LU / multiply / reverse row_permute / compare,
//     benchmark.cpp:59-144);
//   * times LAPACK dgetrf on a copy (benchmark.cpp:239-242) -- LAPACKE_dgetrf is looked up at run time in
//     libmkl_rt / liblapacke / libopenblas; when none is installed an OpenMP partial-pivoting LU of this file is used
//     and the output says so;
//   * appends "n,mpf_time,lapack_time" (10 decimals) to benchmark_times.csv (benchmark.cpp:168-169,265).
// Nothing here is copied from the reference; it is a restatement of its observable behaviour.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>
#include "MPF.h"
#include "mpf_c.h"

typedef int (*lapacke_dgetrf_t)(int, int, int, double *, int, int *);
static const int LAPACK_COL_MAJOR_ = 102;

static lapacke_dgetrf_t find_lapacke(std::string &where) {
    // this program uses GNU OpenMP: tell MKL (if that is what we find) to use the same runtime, never libiomp5
    setenv("MKL_THREADING_LAYER", "GNU", 0);
    setenv("MKL_INTERFACE_LAYER", "LP64", 0);
    const char *libs[] = {"libmkl_rt.so", "libmkl_rt.so.1", "libmkl_rt.so.2", "/opt/conda/lib/libmkl_rt.so", "liblapacke.so",
                          "liblapacke.so.3", "libopenblas.so", "libopenblas.so.0"};
    for (const char *l : libs) {
        void *h = dlopen(l, RTLD_NOW | RTLD_LOCAL);
        if (!h) continue;
        if (void *f = dlsym(h, "LAPACKE_dgetrf")) { where = l; return (lapacke_dgetrf_t)f; }
    }
    return nullptr;
}

// fallback CPU baseline: right-looking partial-pivoting LU, column-major, OpenMP over columns
static int own_dgetrf(int n, double *a, int *ipiv) {
    for (int j = 0; j < n; ++j) {
        int p = j;
        double best = std::fabs(a[(size_t)j * n + j]);
        for (int i = j + 1; i < n; ++i)
            if (std::fabs(a[(size_t)j * n + i]) > best) { best = std::fabs(a[(size_t)j * n + i]); p = i; }
        ipiv[j] = p + 1;
        if (best == 0.0) return j + 1;
        if (p != j)
            for (int c = 0; c < n; ++c) std::swap(a[(size_t)c * n + j], a[(size_t)c * n + p]);
        const double piv = a[(size_t)j * n + j];
        for (int i = j + 1; i < n; ++i) a[(size_t)j * n + i] /= piv;
#pragma omp parallel for schedule(static)
        for (int c = j + 1; c < n; ++c) {
            const double u = a[(size_t)c * n + j];
            double *col = a + (size_t)c * n;
            const double *l = a + (size_t)j * n;
            for (int i = j + 1; i < n; ++i) col[i] -= l[i] * u;
        }
    }
    return 0;
}

// max |A - P (L U)| with the reference's conventions: unit-diagonal L below, U on and above the diagonal of `lu`,
// ipiv = 1-based sequential swaps undone from the last to the first (benchmark.cpp:84-95)
static void show(const char *title, const double *m, int n);

// -v dumps of the factors as the reference prints them for n < 10 (benchmark.cpp:27-57): L with its unit diagonal, then U
static void show_LU(const double *lu, int n) {
    if (n >= 10) return;
    std::cout << "L matrix:" << std::endl;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            if (i > j) std::cout << lu[(size_t)j * n + i] << " ";
            else std::cout << (i == j ? "1 " : "0 ");
        }
        std::cout << std::endl;
    }
    std::cout << std::endl << "U matrix:" << std::endl;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            if (i <= j) std::cout << lu[(size_t)j * n + i] << " ";
            else std::cout << "0 ";
        }
        std::cout << std::endl;
    }
    std::cout << std::endl;
}

// from this size on L * U is formed on the GPU (mpf_check_plu_host): the host triple loop is O(n^3) on a few cores
static const int DEVICE_CHECK_FROM = 2048;

static bool plu_matches(const double *A, const double *lu, const int *ipiv, int n, double tol, double *maxerr, bool verbose = false) {
    if (verbose) show_LU(lu, n);                                  // benchmark.cpp:114
    if (n >= DEVICE_CHECK_FROM) {
        double mx = 0, fro = 0;
        if (mpf_check_plu_host(A, lu, ipiv, n, &mx, &fro) == 0) {
            if (maxerr) *maxerr = mx;
            if (verbose) std::cout << "(L * U on the device) ||A - P L U||_F / ||A||_F = " << fro << "\nCorrectitude: " << (mx <= tol ? "True" : "False") << std::endl;
            return mx <= tol;
        }
        std::cout << "device-side check unavailable, checking on the host" << std::endl;
    }
    std::vector<double> P((size_t)n * n, 0.0);
#pragma omp parallel for schedule(dynamic, 8)
    for (int j = 0; j < n; ++j) {
        double *p = P.data() + (size_t)j * n;
        for (int k = 0; k <= j; ++k) {
            const double u = lu[(size_t)j * n + k];
            const double *l = lu + (size_t)k * n;
            p[k] += u;
            for (int i = k + 1; i < n; ++i) p[i] += l[i] * u;
        }
    }
    if (verbose) show("LU matrix:", P.data(), n);                 // benchmark.cpp:127
    for (int i = n - 1; i >= 0; --i) {
        const int pv = ipiv[i] - 1;
        if (pv != i)
            for (int j = 0; j < n; ++j) std::swap(P[(size_t)j * n + i], P[(size_t)j * n + pv]);
    }
    if (verbose) show("PLU matrix:", P.data(), n);                // benchmark.cpp:133
    double mx = 0.0;
    bool ok = true;
    for (size_t i = 0; i < (size_t)n * n; ++i) {
        const double d = std::fabs(A[i] - P[i]);
        if (!(d <= tol)) ok = false;
        if (d > mx || d != d) mx = d;
    }
    if (maxerr) *maxerr = mx;
    if (verbose) std::cout << "Correctitude: " << (ok ? "True" : "False") << std::endl;   // benchmark.cpp:137-139
    return ok;
}

static void show(const char *title, const double *m, int n) {
    if (n >= 10) return; // the reference prints matrices only below 10 x 10 (benchmark.cpp:15,28)
    std::cout << title << "\n";
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) std::cout << m[(size_t)j * n + i] << " ";
        std::cout << "\n";
    }
    std::cout << std::endl;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        std::cout << "Usage: " << argv[0] << " filename [-v] [--no-check] [-r panel_width]" << std::endl;
        return -1;
    }
    bool verbose = false, check = true;
    int r = 32; // the reference's caller hard-codes 32 (benchmark.cpp:220)
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-v") verbose = true;
        else if (a == "--no-check") check = false;
        else if (a == "-r" && i + 1 < argc) r = std::atoi(argv[++i]);
    }
    std::ifstream in(argv[1]);
    if (!in.is_open()) { std::cout << "Failed to open " << argv[1] << std::endl; return -1; }
    std::ofstream csv("benchmark_times.csv");
    csv << "matrix_size,mpf_time,lapack_time\n" << std::fixed << std::setprecision(10);
    int count = 0;
    in >> count;
    if (in.fail() || count <= 0) { std::cout << "Invalid number of matrices in " << argv[1] << std::endl; return -1; }
    if (verbose) std::cout << "Number of matrices: " << count << std::endl;
    std::string lapack_from;
    lapacke_dgetrf_t lapacke = find_lapacke(lapack_from);
    std::cout << "CPU baseline: " << (lapacke ? ("LAPACKE_dgetrf from " + lapack_from) : std::string("built-in OpenMP dgetrf (no LAPACKE found)")) << std::endl;

    int failures = 0;
    for (int mi = 0; mi < count; ++mi) {
        int n = 0;
        in >> n;
        if (in.fail() || n <= 0) { std::cout << "Invalid matrix size in " << argv[1] << " n: " << n << std::endl; return -1; }
        std::vector<double> orig((size_t)n * n);
        for (double &v : orig) in >> v; // tokens land linearly and are read as column-major (benchmark.cpp:192-194,19)
        if (in.fail()) { std::cout << "Error while reading matrix data in " << argv[1] << std::endl; return -1; }
        std::vector<double> a_mpf(orig), a_lap(orig);
        std::vector<int> ipiv(n), ipiv_lap(n);
        for (int i = 0; i < n; ++i) ipiv[i] = i + 1;
        if (verbose) show("Original matrix:", orig.data(), n);

        auto t0 = std::chrono::high_resolution_clock::now();
        MPF(a_mpf.data(), n, r, ipiv.data());
        auto t1 = std::chrono::high_resolution_clock::now();
        const double t_mpf = std::chrono::duration<double>(t1 - t0).count();
        if (verbose) std::cout << "MPF() time: " << t_mpf << " seconds\n" << std::endl;
        if (check) {
            double err = 0;
            std::cout << "Checking correctness of MPF results..." << std::endl;
            if (!plu_matches(orig.data(), a_mpf.data(), ipiv.data(), n, 1e-10, &err, verbose)) {
                std::cout << "MPF produced incorrect results." << std::endl;
                ++failures;
            }
            if (verbose) std::cout << "max |A - P L U| = " << err << std::endl;
        }
        std::cout << "Matrix size: " << n << std::endl;

        t0 = std::chrono::high_resolution_clock::now();
        const int info = lapacke ? lapacke(LAPACK_COL_MAJOR_, n, n, a_lap.data(), n, ipiv_lap.data()) : own_dgetrf(n, a_lap.data(), ipiv_lap.data());
        t1 = std::chrono::high_resolution_clock::now();
        const double t_lap = std::chrono::duration<double>(t1 - t0).count();
        if (info != 0) std::cout << "dgetrf failed with error code " << info << std::endl;
        if (verbose) std::cout << "dgetrf time: " << t_lap << " seconds\n" << std::endl;
        if (check && !plu_matches(orig.data(), a_lap.data(), ipiv_lap.data(), n, 1e-10, nullptr, verbose))
            std::cout << "dgetrf produced incorrect results." << std::endl;
        csv << n << "," << t_mpf << "," << t_lap << std::endl;
    }
    csv.close();
    return failures ? 1 : 0;
}

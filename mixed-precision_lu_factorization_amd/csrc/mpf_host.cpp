// C++ host side of libmpf_amd.so: context, the MPF panel loop (reference MPF.cu:66-256 re-architected
// as an asynchronous HIP stream of kernels with no per-panel host synchronisation), the refinement
// solve, the C ABI (include/mpf_c.h) and the drop-in C++ symbol MPF() (include/MPF.h).
#include "mpf_internal.h"
#include "../../include/MPF.h"
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <iostream>
#include <vector>
#include <mutex>
#include <functional>
#include <chrono>

static thread_local std::string g_noctx_err;

// ---- per-context options ---------------------------------------------------------------------------------------------
// One table: option name (mpf_set_option / mpf_get_option), the environment variable that gives its DEFAULT when a context
// is created, and where it lives in the context.  Nothing else in the library reads the environment.
namespace {
struct OptDesc { const char *name, *env; size_t off; bool wide; long long lo, hi; };
#define OPT_I(field, env, lo, hi) {#field, env, offsetof(MpfTuning, field), false, lo, hi}
#define OPT_L(field, env, lo, hi) {#field, env, offsetof(MpfTuning, field), true, lo, hi}
const OptDesc kOpts[] = {
    OPT_I(safe_pivots, "MPF_SAFE_PIVOTS", 0, 1),
    OPT_I(chain_pipeline, "MPF_CHAIN_PIPELINE", 0, 1),
    OPT_L(chain_pipeline_below, "MPF_CHAIN_PIPELINE_BELOW", 0, 1ll << 40),
    OPT_I(fp16_work32, "MPF_FP16_WORK32", 0, 1),
    OPT_I(superpanel_fp16, "MPF_SUPERPANEL", 0, 8),
    OPT_I(superpanel_fp64, "MPF_SUPERPANEL_FP64", 1, 8),
    OPT_I(no_lookahead, "MPF_NO_LOOKAHEAD", 0, 1),
    OPT_I(verbose, "MPF_VERBOSE", 0, 1),
    OPT_I(timeline, "MPF_TIMELINE", 0, 1),
    OPT_L(hp_spin_limit, "MPF_HP_SPIN_LIMIT", 1, 1ll << 31),
    OPT_L(hp_gate_ticks, "MPF_HP_GATE_TICKS", 0, 1ll << 40),
    OPT_I(hp_acq_fence, "MPF_HP_ACQ_FENCE", 0, 1),
    OPT_I(hp_window, "MPF_HP_WINDOW", -1, 1 << 30),
    OPT_I(hgemm_pad, "MPF_HGEMM_PAD", 0, 65536),
    OPT_I(hgemm_split_pad, "MPF_HGEMM_SPLIT_PAD", 0, 65536),
    OPT_I(hgemm_big, "MPF_HGEMM_BIG", 0, 1),
    OPT_I(hgemm_big_tile, "MPF_HGEMM_BIG_TILE", 0, 5),
    OPT_I(hgemm_mfma16, "MPF_HGEMM_MFMA16", 0, 1),
    OPT_I(dgemm_dma, "MPF_DGEMM_DMA", 0, 1),
    OPT_I(lazy_gather, "MPF_LAZY_GATHER", 0, 1),
    OPT_I(dpanel_fused_form, "MPF_DPANEL_FUSED", 0, 1),
    OPT_I(dist_instalments, "MPF_DIST_INSTALMENTS", 0, 1),
    OPT_L(dist_instalment_min_bytes, "MPF_DIST_INSTALMENT_MIN_BYTES", 0, 1ll << 40),
    OPT_I(generic_fused, "MPF_GENERIC_FUSED", 0, 1),
    OPT_I(fp64_rowmajor, "MPF_FP64_ROWMAJOR", 0, 1),
    OPT_L(fp64_rowmajor_min_n, "MPF_FP64_ROWMAJOR_MIN_N", 0, 1ll << 40),
    OPT_I(trsm_laswp_fused, "MPF_TRSM_LASWP_FUSED", 0, 1),
    OPT_I(fp64_two_lanes, "MPF_FP64_TWO_LANES", 0, 1 << 30),
    OPT_I(fp64_lane_a_pct, "MPF_FP64_LANE_A_PCT", 20, 90),
    OPT_I(event_timers, "MPF_EVENT_TIMERS", 0, 2),
    OPT_I(dist_world1_loop, "MPF_DIST_WORLD1_LOOP", 0, 1),
    OPT_I(gesv_fp64_tflops, "MPF_GESV_FP64_TFLOPS", 0, 1000),
    OPT_I(gate_wait_value, "MPF_GATE_WAIT_VALUE", 0, 1),
    OPT_I(host_sink, "MPF_HOST_SINK", 0, 1),
    OPT_I(sink_trace, "MPF_SINK_TRACE", 0, 1),
    OPT_I(hp_half_slabs, "MPF_HP_HALF_SLABS", 0, 1),
    OPT_I(hp_half_slabs_rows, "MPF_HP_HALF_SLABS_ROWS", 256, 32768),
    OPT_I(hp_local_xcd, "MPF_HP_LOCAL_XCD", 0, 2),
    OPT_I(host_late_parts, "MPF_HOST_LATE_PARTS", 0, 4),
    OPT_L(host_late_min_n, "MPF_HOST_LATE_MIN_N", 0, 1ll << 40),
    OPT_I(host_first_pct, "MPF_HOST_FIRST_PCT", 5, 100),
    OPT_I(host_late_q_pct, "MPF_HOST_LATE_Q_PCT", 1, 1000),
    OPT_L(host_sink_min_n, "MPF_HOST_SINK_MIN_N", 0, 1ll << 40),
    OPT_I(dist_solve_p2p, "MPF_DIST_SOLVE_P2P", 0, 1),
#ifdef MPF_PROBE
    OPT_I(hp_stamp, "MPF_HP_STAMP", 0, 1),
    OPT_I(hp_r256_upto, "MPF_HP_R256_UPTO", 0, 1 << 30),
    OPT_I(dgemm_w8, "MPF_DGEMM_W8", 0, 1),
    OPT_I(gemm_lds_pad, "MPF_GEMM_LDS_PAD", 0, 65536),
    OPT_I(hgemm_big_reg, "MPF_HGEMM_BIG_REG", 0, 1),
    OPT_I(hgemm_dbg, "MPF_HGEMM_DBG", 0, 8191),
#endif
};
#undef OPT_I
#undef OPT_L
void opt_store(MpfTuning &t, const OptDesc &d, long long v) {
    if (v < d.lo) v = d.lo;
    if (v > d.hi) v = d.hi;
    char *p = (char *)&t + d.off;
    if (d.wide) *(long long *)p = v; else *(int *)p = (int)v;
}
long long opt_load(const MpfTuning &t, const OptDesc &d) {
    const char *p = (const char *)&t + d.off;
    return d.wide ? *(const long long *)p : (long long)*(const int *)p;
}
void tuning_from_env(MpfTuning &t) {   // called by mpf_create only
    for (const OptDesc &d : kOpts)
        if (const char *e = getenv(d.env)) if (*e) opt_store(t, d, atoll(e));
}
} // namespace

static int fail(mpf_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg; else g_noctx_err = msg;
    return code;
}

extern "C" {

int mpf_create(mpf_ctx **out, int device) {
    if (!out) return -1;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(nullptr, -3, "No HIP devices available.");
    if (device < 0 || device >= ndev) return fail(nullptr, -1, "mpf_create: bad device index");
    mpf_ctx *c = new mpf_ctx();
    tuning_from_env(c->tune);
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return fail(nullptr, -2, "hipSetDevice failed"); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(nullptr, -2, "hipStreamCreate failed"); }
    c->own_stream = true;
    {
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi); // hi = numerically lowest = highest priority
        if (hipStreamCreateWithPriority(&c->pstream, hipStreamNonBlocking, hi) != hipSuccess) c->pstream = nullptr;
        if (hipStreamCreateWithPriority(&c->tstream, hipStreamNonBlocking, hi) != hipSuccess) c->tstream = nullptr;
    }
    if (hipMalloc((void **)&c->ws, sizeof(MpfWorkspace)) != hipSuccess) { hipStreamDestroy(c->stream); delete c; return fail(nullptr, -2, "hipMalloc(workspace) failed"); }
    hipMemset(c->ws, 0, sizeof(MpfWorkspace));
    // 8 bytes of signal memory for option gate_wait_value (a stream waits on the pivot kernel's progress word); optional
    if (hipExtMallocWithFlags((void **)&c->hp_signal, 8, hipMallocSignalMemory) != hipSuccess) { c->hp_signal = nullptr; (void)hipGetLastError(); }
    else hipMemset(c->hp_signal, 0, 8);
    hipEventCreate(&c->ev0);
    hipEventCreate(&c->ev1);
    *out = c;
    return 0;
}

int mpf_destroy(mpf_ctx *c) {
    if (!c) return 0;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->ws) hipFree(c->ws);
    if (c->hp_signal) hipFree(c->hp_signal);
    if (c->solve_buf) hipFree(c->solve_buf);
    if (c->perm_buf) hipFree(c->perm_buf);
    if (c->trsv_inv) hipFree(c->trsv_inv);
    if (c->trsv_inv256) hipFree(c->trsv_inv256);
    if (c->trsv_cnt) hipFree(c->trsv_cnt);
    if (c->res_part) hipFree(c->res_part);
    if (c->krylov) hipFree(c->krylov);
    mpf_rccl_destroy(c);
    for (auto *b : c->dist_buf) if (b) hipFree(b);
    if (c->dist_w32) hipFree(c->dist_w32);
    if (c->dist_spl) hipFree(c->dist_spl);
    if (c->dtiles) hipFree(c->dtiles);
    if (c->w32) hipFree(c->w32);
    if (c->r64) hipFree(c->r64);
    if (c->host_A) hipFree(c->host_A);
    if (c->host_P) hipFree(c->host_P);
    if (c->host_A0) hipFree(c->host_A0);
    sink_destroy(c);
    feed_destroy(c);
    if (c->late_flags) hipHostFree(c->late_flags);
    if (c->rm_tmp) hipFree(c->rm_tmp);
    if (c->rm_lt) hipFree(c->rm_lt);
    if (c->g16) hipFree(c->g16);
    if (c->gcand) hipFree(c->gcand);
    if (c->lists) hipFree(c->lists);
    if (c->Fmap) hipFree(c->Fmap);
    if (c->perm_tmp) hipFree(c->perm_tmp);
    if (c->h_L) hipFree(c->h_L);
    if (c->h_U) hipFree(c->h_U);
    for (auto *b : c->h_Lb) if (b) hipFree(b);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->pstream) { hipStreamSynchronize(c->pstream); hipStreamDestroy(c->pstream); }
    if (c->tstream) { hipStreamSynchronize(c->tstream); hipStreamDestroy(c->tstream); }
    if (c->xstream) { hipStreamSynchronize(c->xstream); hipStreamDestroy(c->xstream); }
    for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int mpf_set_stream(mpf_ctx *c, void *hip_stream) {
    if (!c) return -1;
    if (c->own_stream && c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return 0;
}

int mpf_synchronize(mpf_ctx *c) {
    if (!c) return -1;
    MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

const char *mpf_last_error(mpf_ctx *c) { return c ? c->err.c_str() : g_noctx_err.c_str(); }

int mpf_get_stats(mpf_ctx *c, mpf_stats *out) {
    if (!c || !out) return -1;
    *out = c->stats;
    return 0;
}

int mpf_set_option(mpf_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return -1;
    for (const OptDesc &d : kOpts)
        if (!strcmp(d.name, name)) {
            opt_store(c->tune, d, (long long)value);
            // kernel attributes (dynamic-LDS sizes, occupancy answers) depend on options: every family sets its own again
            c->attr_done = 0; c->attr_big = 0;
            return 0;
        }
    return fail(c, -1, std::string("mpf_set_option: unknown option '") + name + "'");
}
int mpf_get_option(mpf_ctx *c, const char *name, int64_t *value) {
    if (!c || !name || !value) return -1;
    for (const OptDesc &d : kOpts)
        if (!strcmp(d.name, name)) { *value = (int64_t)opt_load(c->tune, d); return 0; }
    return fail(c, -1, std::string("mpf_get_option: unknown option '") + name + "'");
}
int mpf_option_name(int32_t index, char *buf, int64_t buflen) { // enumerate: returns the number of options
    const int n = (int)(sizeof kOpts / sizeof kOpts[0]);
    if (index >= 0 && index < n && buf && buflen > 0) { strncpy(buf, kOpts[index].name, (size_t)buflen - 1); buf[buflen - 1] = 0; }
    return n;
}

int mpf_device_report(char *buf, int64_t buflen) {
    // HIP analogue of reference check_cooperative_groups.cu:4-48 (device properties + cooperative-launch support), plus what
    // this library's design depends on: LDS per CU (the pivot kernel's slab), CU count (its residency bound), L2 / Infinity
    // Cache, the RCCL version and the xGMI link matrix of the node
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    std::string s = "HIP devices: " + std::to_string(ndev) + "\n";
    int rt = 0, drv = 0;
    if (ndev > 0 && hipRuntimeGetVersion(&rt) == hipSuccess && hipDriverGetVersion(&drv) == hipSuccess)
        s += "HIP runtime " + std::to_string(rt) + " driver " + std::to_string(drv) + "\n";
    const int rv = ndev > 0 ? mpf_rccl_version() : -1;
    s += rv > 0 ? "RCCL version " + std::to_string(rv / 10000) + "." + std::to_string(rv / 100 % 100) + "." + std::to_string(rv % 100) + " (NCCL API, dlopen)\n"
                : std::string("RCCL: not loaded\n");
    for (int d = 0; d < ndev; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) != hipSuccess) continue;
        int coop = 0, lds_cu = 0, l2 = 0, clk = 0, memclk = 0, buswidth = 0;
        hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, d);
        hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, d);
        hipDeviceGetAttribute(&l2, hipDeviceAttributeL2CacheSize, d);
        hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, d);
        hipDeviceGetAttribute(&memclk, hipDeviceAttributeMemoryClockRate, d);
        hipDeviceGetAttribute(&buswidth, hipDeviceAttributeMemoryBusWidth, d);
        char line[768];
        snprintf(line, sizeof line,
                 "device %d: %s arch %s CUs %d maxThreadsPerBlock %d LDS/block %zu B LDS/CU %d B warpSize %d regs/block %d "
                 "L2 %d KiB clock %d MHz memclock %d MHz bus %d bit HBM %.1f GiB cooperativeLaunch %d pivot-kernel rows/launch %d\n",
                 d, p.name, p.gcnArchName, p.multiProcessorCount, p.maxThreadsPerBlock, p.sharedMemPerBlock, lds_cu, p.warpSize,
                 p.regsPerBlock, l2 / 1024, clk / 1000, memclk / 1000, buswidth, (double)p.totalGlobalMem / (1024.0 * 1024.0 * 1024.0), coop,
                 HP_R * (p.multiProcessorCount < HP_MAXG ? p.multiProcessorCount : HP_MAXG));
        s += line;
    }
    if (ndev > 1) { // xGMI topology: link type and hop count of every device pair, peer access
        s += "link matrix (type:hops, type 4 = xGMI; P = peer access):\n";
        for (int a = 0; a < ndev; ++a) {
            std::string row = "  dev " + std::to_string(a) + ":";
            for (int b = 0; b < ndev; ++b) {
                if (a == b) { row += "   -  "; continue; }
                uint32_t lt = 0, hops = 0;
                int peer = 0;
                hipDeviceCanAccessPeer(&peer, a, b);
                if (hipExtGetLinkTypeAndHopCount(a, b, &lt, &hops) != hipSuccess) { lt = 0; hops = 0; }
                char cell[32];
                snprintf(cell, sizeof cell, " %u:%u%s", lt, hops, peer ? "P" : " ");
                row += cell;
            }
            s += row + "\n";
        }
    }
    if (buf && buflen > 0) { strncpy(buf, s.c_str(), (size_t)buflen - 1); buf[buflen - 1] = 0; }
    return ndev;
}

// ---- step operators ----------------------------------------------------------------------------
int mpf_double_to_fp16(mpf_ctx *c, const double *d_in, uint16_t *d_out, int64_t n) {
    if (!c) return -1;
    return launch_double_to_fp16(c, d_in, d_out, n);
}
int mpf_hdiv(mpf_ctx *c, const uint16_t *a, const uint16_t *b, uint16_t *q, int64_t n) {
    if (!c) return -1;
    return launch_hdiv(c, a, b, q, n);
}
int mpf_hgetf2_pivots(mpf_ctx *c, const double *d_A, int64_t lda, int32_t rows, int32_t cols, int32_t ipiv_offset,
                      int32_t *d_ipiv, uint16_t *d_panel16_out) {
    if (!c || !d_A || !d_ipiv) return -1;
    if (lda < rows) return fail(c, -1, "hgetf2_pivots: lda < rows");
    if (cols > 65535) return fail(c, -1, "hgetf2_pivots: panel width > 65535");
    if (!safe_pivots(c) && hgetf2_lds_eligible(c, rows, cols))
        return launch_hgetf2(c, d_A, lda, nullptr, 0, rows, cols, ipiv_offset, d_ipiv, d_panel16_out, rows, nullptr);
    return launch_hgetf2_generic(c, d_A, lda, nullptr, 0, rows, cols, ipiv_offset, d_ipiv, d_panel16_out, rows);
}
int mpf_hgetf2(mpf_ctx *c, uint16_t *d_panel16, int64_t ld, int32_t rows, int32_t cols, int32_t *d_ipiv_panel) {
    if (!c || !d_panel16 || !d_ipiv_panel) return -1;
    if (ld < rows) return fail(c, -1, "hgetf2: ld < rows");
    if (!safe_pivots(c) && hgetf2_lds_eligible(c, rows, cols))
        return launch_hgetf2(c, nullptr, 0, d_panel16, ld, rows, cols, 0, d_ipiv_panel, nullptr, 0, nullptr);
    return launch_hgetf2_generic(c, nullptr, 0, d_panel16, ld, rows, cols, 0, d_ipiv_panel, nullptr, 0);
}
int mpf_laswp(mpf_ctx *c, double *d_A, int64_t lda, int64_t ncols, int32_t k, int32_t cols, const int32_t *d_ipiv) {
    if (!c || !d_A || !d_ipiv) return -1;
    if (cols > HP_MAXCOLS) return launch_laswp_seq(c, d_A, lda, ncols, k, cols, d_ipiv, lda); // any number of swaps, MPF.cu:42-59 as it stands
    return launch_laswp(c, d_A, lda, ncols, k, cols, d_ipiv);
}
int mpf_dgetf2_npv(mpf_ctx *c, double *d_P, int64_t ld, int32_t rows, int32_t cols, int32_t fused) {
    if (!c || !d_P) return -1;
    if (ld < rows) return fail(c, -1, "dgetf2_npv: ld < rows");
    return launch_dgetf2_npv(c, d_P, ld, rows, cols, fused, 0);
}
int mpf_dtrsm_llnu(mpf_ctx *c, int32_t m, int64_t n, const double *d_L, int64_t ldl, double *d_B, int64_t ldb) {
    if (!c) return -1;
    if (m > 0 && n > 0 && (ldl < m || ldb < m)) return fail(c, -1, "dtrsm: leading dimension < m");
    return launch_dtrsm_llnu(c, m, n, d_L, ldl, d_B, ldb);
}
int mpf_dgemm_minus(mpf_ctx *c, int64_t m, int64_t n, int32_t k, const double *d_A, int64_t lda, const double *d_B,
                    int64_t ldb, double *d_C, int64_t ldc) {
    if (!c) return -1;
    if (m > 0 && n > 0 && k > 0 && (lda < m || ldb < k || ldc < m)) return fail(c, -1, "dgemm: bad leading dimension");
    return launch_dgemm_minus(c, m, n, k, d_A, lda, d_B, ldb, d_C, ldc);
}

int mpf_ensure_h_images(mpf_ctx *c, int64_t rows, int kmax, bool big) {
    kmax = (kmax + 63) & ~63;
    if (c->h_L && c->h_rows >= rows && c->h_kmax >= kmax && (!big || c->h_Lb[0])) return 0;
    if (rows < c->h_rows) rows = c->h_rows;
    if (kmax < c->h_kmax) kmax = c->h_kmax;
    big = big || c->h_Lb[0];
    if (c->h_L) hipFree(c->h_L);
    if (c->h_U) hipFree(c->h_U);
    for (auto *&b : c->h_Lb) { if (b) hipFree(b); b = nullptr; }
    c->h_L = c->h_U = nullptr; c->h_rows = 0; c->h_kmax = 0;
    const size_t bytes = 2 * (size_t)rows * kmax * sizeof(unsigned short); // hi image, then lo image (split mode)
    MPF_HIP_TRY(c, hipMalloc((void **)&c->h_L, bytes));
    MPF_HIP_TRY(c, hipMalloc((void **)&c->h_U, bytes));
    if (big) for (auto *&b : c->h_Lb) MPF_HIP_TRY(c, hipMalloc((void **)&b, bytes));
    c->h_rows = rows; c->h_kmax = kmax;
    return 0;
}

int mpf_hgemm_minus(mpf_ctx *c, int64_t m, int64_t n, int32_t k, const double *d_A, int64_t lda, const double *d_B,
                    int64_t ldb, double *d_C, int64_t ldc, int32_t split) {
    if (!c) return -1;
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    if (k > 8 * HP_MAXCOLS) return fail(c, -1, "hgemm: k > 2048");
    if (lda < m || ldb < k || ldc < m) return fail(c, -1, "hgemm: bad leading dimension");
    int rc = mpf_ensure_h_images(c, m > n ? m : n, k, false);
    if (!rc) rc = launch_cvt_l21(c, d_A, lda, m, k, split);
    if (!rc) rc = launch_hgemm_minus(c, m, n, k, d_B, ldb, d_C, ldc, split);
    return rc;
}

int mpf_hgemm_minus_f32(mpf_ctx *c, int64_t m, int64_t n, int32_t k, const double *d_A, int64_t lda, const double *d_B,
                        int64_t ldb, float *d_C, int64_t ldc, int32_t split) {
    if (!c || !d_A || !d_B || !d_C) return -1;
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    if (k > 8 * HP_MAXCOLS) return fail(c, -1, "hgemm: k > 2048");
    if (lda < m || ldb < k || ldc < m) return fail(c, -1, "hgemm: bad leading dimension");
    int rc = mpf_ensure_h_images(c, m > n ? m : n, k, false);
    if (!rc) rc = launch_cvt_l21(c, d_A, lda, m, k, split);
    c->hgemm_standalone = true;   // (a call of its own: nothing of a panel chain waits for CUs beside it -- see launch_hgemm_ptrs)
    if (!rc) rc = launch_hgemm_minus_w32(c, m, n, k, d_B, ldb, d_C, ldc, split);
    c->hgemm_standalone = false;
    return rc;
}

int mpf_w32_from_f64(mpf_ctx *c, const double *d_A, int64_t lda, float *d_W, int64_t ldw, int64_t rows, int64_t cols) {
    if (!c || !d_A || !d_W) return -1;
    if (lda < rows || ldw < cols) return fail(c, -1, "w32_from_f64: bad leading dimension");
    return launch_cvt_f64_f32(c, d_A, lda, d_W, ldw, rows, cols);
}
int mpf_w32_to_f64(mpf_ctx *c, const float *d_W, int64_t ldw, double *d_A, int64_t lda, int64_t rows, int64_t cols) {
    if (!c || !d_A || !d_W) return -1;
    if (lda < rows || ldw < cols) return fail(c, -1, "w32_to_f64: bad leading dimension");
    return launch_cvt_f32_f64(c, d_W, ldw, d_A, lda, rows, cols);
}
int mpf_w32_laswp(mpf_ctx *c, float *d_W, int64_t ldw, int64_t ncols, int32_t k, int32_t cols, const int32_t *d_ipiv) {
    if (!c || !d_W || !d_ipiv) return -1;
    if (cols < 1 || cols > HP_MAXCOLS) return fail(c, -1, "w32_laswp: 1 <= cols <= 256");
    if (ldw < ncols) return fail(c, -1, "w32_laswp: ldw < ncols");
    const int64_t need = (int64_t)LASWP_MAXMOVED * ncols / 2 + 1;    // doubles: 2 * HP_MAXCOLS moved rows x ncols floats
    if (need > c->perm_cap) {
        MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->perm_tmp) hipFree(c->perm_tmp);
        c->perm_tmp = nullptr; c->perm_cap = 0;
        MPF_HIP_TRY(c, hipMalloc((void **)&c->perm_tmp, (size_t)need * sizeof(double)));
        c->perm_cap = need;
    }
    int rc = launch_laswp_plan(c, d_ipiv, k, cols, &c->ws->list0);
    if (!rc) rc = launch_laswp_from_list_f32(c, d_W, ldw, ncols, &c->ws->list0);
    return rc;
}

// ---- the panel loop (MPF.cu:100-242) -------------------------------------------------------------

// trailing GEMM of one panel in the selected mode (fp16 mode: the L21 image must already be in c->h_L)
static int trail_gemm(mpf_ctx *c, const mpf_opts &o, int64_t m, int64_t n, int pc, const double *L21, const double *U12,
                      double *C, int64_t lda) {
    if (o.trailing != MPF_TRAIL_FP64) return launch_hgemm_minus(c, m, n, pc, U12, lda, C, lda, o.trailing == MPF_TRAIL_FP16X3);
    return launch_dgemm_minus(c, m, n, pc, L21, lda, U12, lda, C, lda);
}

// Single-stream schedule with a host synchronisation after every phase (per-phase timers).
static int factor_sync_timed(mpf_ctx *c, double *d_A, int64_t lda, int64_t N, int32_t nb, int32_t *d_ipiv,
                             const mpf_opts &o, mpf_stats &st) {
    hipEvent_t pe0, pe1;
    hipEventCreate(&pe0); hipEventCreate(&pe1);
    auto phase = [&](double &acc, auto &&fn) -> int {
        hipEventRecord(pe0, c->stream);
        int rc = fn();
        hipEventRecord(pe1, c->stream);
        hipEventSynchronize(pe1);
        float ms = 0; hipEventElapsedTime(&ms, pe0, pe1);
        acc += ms;
        return rc;
    };
    int rc = 0;
    for (int64_t k = 0; k < N && rc == 0; k += nb) {
        const int pc = (int)((N - k) < nb ? (N - k) : nb);   // MPF.cu:101
        const int pr = (int)(N - k);                         // MPF.cu:102
        if (pr <= 1) break;                                  // MPF.cu:104 (1x1 tail: nothing to do)
        double *Ap = d_A + k * lda + k;
        MovedList *ml = c->lists + (k / nb);
        rc = phase(st.ms_hpanel, [&] { return launch_hgetf2(c, Ap, lda, nullptr, 0, pr, pc, (int)k, d_ipiv + k, nullptr, 0, ml, 0, o.trailing == MPF_TRAIL_FP64 ? HP_FP64_WINDOW_ROWS : 0); });
        if (rc) break;
        // MPF.cu:162 on the panel and everything right of it; the columns left of it are deferred (laswp.hip)
        rc = phase(st.ms_laswp, [&] { return launch_laswp_from_list(c, d_A + k * lda, lda, N - k, ml); });
        if (rc) break;
        rc = phase(st.ms_dpanel, [&] { return launch_dgetf2_npv(c, Ap, lda, pr, pc, o.fused_panel, (int)k); });
        if (rc) break;
        if (k + pc < N) {                                    // MPF.cu:203
            const int64_t n = N - k - pc;
            double *A12 = d_A + (k + pc) * lda + k;
            rc = phase(st.ms_trsm, [&] { return launch_dtrsm_llnu(c, pc, n, Ap, lda, A12, lda); });           // :215
            if (rc) break;
            rc = phase(st.ms_gemm, [&] {
                int e = o.trailing != MPF_TRAIL_FP64 ? launch_cvt_l21(c, Ap + pc, lda, n, pc, o.trailing == MPF_TRAIL_FP16X3) : 0;
                if (!e) e = trail_gemm(c, o, n, n, pc, Ap + pc, A12, A12 + pc, lda);
                return e; }); // :230
            if (rc) break;
            count_gemm(st, o, n, n, pc);
        }
        st.panels++;
        if (o.verbose) printf("panel k=%lld rows=%d cols=%d\n", (long long)k, pr, pc);
    }
    if (!rc) rc = phase(st.ms_laswp, [&] { return launch_lazy_left_swaps(c, d_A, lda, N, nb, (int)((N + nb - 1) / nb), c->lists); });
    hipEventDestroy(pe0); hipEventDestroy(pe1);
    return rc;
}

// Generic schedule: the reference's own loop (MPF.cu:100-242) step by step on one stream, for everything the tuned
// schedules do not cover -- panels wider than 256 columns, matrices taller than the LDS pivot kernel's 256 rows x #CUs,
// devices / callers that must not run a kernel whose workgroups wait for each other (o.pivot_path = 1).  Interchanges
// are the reference's sequential per-column swaps over ALL N columns (no deferred left-hand side), the TRSM is blocked
// by 256 rows, K is cut to what the update kernels address.  Same per-element operations in the same order as the
// tuned schedules: results are bit-identical (tests/test_gpu_generic.py).
static int factor_generic(mpf_ctx *c, double *d_A, int64_t lda, int64_t N, int32_t nb, int32_t *d_ipiv, const mpf_opts &o,
                          mpf_stats &st, bool force_generic_pivots) {
    EvPool ev(c);
    ev.keep = &st.ms_gemm;
    hipStream_t S = c->stream;
    int rc = 0;
    const bool split = o.trailing == MPF_TRAIL_FP16X3;
    for (int64_t k = 0; k < N && rc == 0; k += nb) {
        const int pc = (int)((N - k) < nb ? (N - k) : nb);   // MPF.cu:101
        const int pr = (int)(N - k);                         // MPF.cu:102
        if (pr <= 1) break;                                  // MPF.cu:104
        double *Ap = d_A + k * lda + k;
        rc = ev.timed(st.ms_hpanel, S, [&] {
            if (!force_generic_pivots && hgetf2_lds_eligible(c, pr, pc))
                return launch_hgetf2(c, Ap, lda, nullptr, 0, pr, pc, (int)k, d_ipiv + k, nullptr, 0, nullptr, 0, o.trailing == MPF_TRAIL_FP64 ? HP_FP64_WINDOW_ROWS : 0);
            st.pivot_path = 1;
            return launch_hgetf2_generic(c, Ap, lda, nullptr, 0, pr, pc, (int)k, d_ipiv + k, nullptr, 0); });
        if (rc) break;
        rc = ev.timed(st.ms_laswp, S, [&] { return launch_laswp_seq(c, d_A, lda, N, (int)k, pc, d_ipiv + k, N); });   // :162
        if (rc) break;
        rc = ev.timed(st.ms_dpanel, S, [&] { return launch_dgetf2_npv(c, Ap, lda, pr, pc, o.fused_panel, (int)k); });  // :183
        if (rc) break;
        if (k + pc < N) {                                    // MPF.cu:203
            const int64_t n = N - k - pc;
            double *A12 = d_A + (k + pc) * lda + k;
            rc = ev.timed(st.ms_trsm, S, [&] { return launch_dtrsm_llnu(c, pc, n, Ap, lda, A12, lda); });             // :215
            if (rc) break;
            rc = ev.timed(st.ms_gemm, S, [&] {                                                                         // :230
                if (o.trailing == MPF_TRAIL_FP64) return launch_dgemm_minus(c, n, n, pc, Ap + pc, lda, A12, lda, A12 + pc, lda);
                int e = 0;
                const int kcmax = c->h_kmax;                 // K capacity of the fp16 operand images
                for (int k0 = 0; k0 < pc && !e; k0 += kcmax) {
                    const int kc = (pc - k0) < kcmax ? (pc - k0) : kcmax;
                    e = launch_cvt_l21(c, Ap + pc + (int64_t)k0 * lda, lda, n, kc, split);
                    if (!e) e = launch_hgemm_minus(c, n, n, kc, A12 + k0, lda, A12 + pc, lda, split);
                }
                return e; });
            if (rc) break;
            count_gemm(st, o, n, n, pc);
        }
        st.panels++;
        if (o.verbose) printf("panel k=%lld rows=%d cols=%d (generic schedule)\n", (long long)k, pr, pc);
    }
    hipError_t se = hipStreamSynchronize(S);
    if (!rc && se != hipSuccess) return fail(c, -2, std::string("factorization failed: ") + hipGetErrorString(se));
    ev.collect();
    return rc;
}

// The chain of one panel (pivots, interchange of the panel's own columns, fp64 panel) with the fp64 panel FOLLOWING the pivot
// kernel instead of waiting for it.  The pivot kernel (stream P) publishes its progress every 32 columns; stream T runs, per
// 32-column sub-panel s: a gate (waits for the pivots of columns < 32 (s + 1)), the reference's sequential interchange of
// exactly those 32 pivots on the panel's columns (LASWP applied in instalments is LASWP: MPF.cu:47-57), and piece s of the fp64
// panel -- whose rows below the sub-panel are row-independent, so the swaps still to come only move finished rows around
// (what LAPACK's blocked dgetf2 does).  Same operations per element as the unpipelined chain: bit-identical.
// e1: the panel's columns are up to date (recorded on the main stream).  On return *e2p follows the pivot kernel (its moved-row
// list is complete) and *e2t the last fp64-panel piece.  Falls back to the plain chain (returns 1, nothing launched) when
// the shape has no pieces or there is no third stream.
static int chain_pipelined(mpf_ctx *c, EvPool &ev, mpf_stats &st, const mpf_opts &o, double *d_A, int64_t lda, int64_t N, int64_t nx,
                           int pc2, int32_t *d_ipiv, MovedList *ml, hipEvent_t e1, hipEvent_t *e2p, hipEvent_t *e2t, int *rc_out) {
    const int np = dgetf2_npv_pieces(c, pc2);
    if (!c->tune.chain_pipeline || np == 0 || !c->tstream || !c->pstream) return 1;
    // fp64 mode: while the update is the longer side (large trailing matrix) the chain hides under it anyway, and the extra
    // launches beside it only cost the update time (measured + 4 ms per factorization): pipeline the chain-bound panels only
    if (o.trailing == MPF_TRAIL_FP64 && (N - nx) > c->tune.chain_pipeline_below) return 1;
    // the gated interchange kernel's workgroups wait for the pivot kernel while sitting on CUs: the panel must fit beside them
    // (else the chain runs unpipelined: nobody waits for a kernel whose workgroups cannot all become resident)
    const int waiters = (c->tune.gate_wait_value && c->hp_signal) ? 0 : laswp_gated_grid(pc2);   // (a stream that waits holds no CU)
    if (!hgetf2_fits_beside(c, (int)(N - nx), pc2, waiters)) return 1;
    const int pref_win = o.trailing == MPF_TRAIL_FP64 ? HP_FP64_WINDOW_ROWS : 0;
    hipStream_t P = c->pstream, T = c->tstream;
    double *Anx = d_A + nx * lda + nx;
    int rc = 0;
    hipStreamWaitEvent(P, e1, 0);
    hipStreamWaitEvent(T, e1, 0);
    {
        StreamSwap sw(c, P);
        rc = ev.timed(st.ms_hpanel, P, [&] {
            return launch_hgetf2(c, Anx, lda, nullptr, 0, (int)(N - nx), pc2, (int)nx, d_ipiv + nx, nullptr, 0, ml, waiters, pref_win); });
        *e2p = ev.get();
        hipEventRecord(*e2p, P);
    }
    if (!rc) {
        StreamSwap sw(c, T);
        rc = ev.timed(st.ms_dpanel, T, [&] {
            int e = 0;
            for (int s = 0; s < np && !e; ++s) {
                e = launch_laswp_block_gated(c, d_A + nx * lda, lda, pc2, (int)nx + 32 * s, 32, d_ipiv + nx + 32 * s, N, 32 * (s + 1));
                if (!e) e = launch_dgetf2_npv_piece(c, Anx, lda, (int)(N - nx), pc2, o.fused_panel, (int)nx, s);
            }
            return e; });
        *e2t = ev.get();
        hipEventRecord(*e2t, T);
    }
    *rc_out = rc;
    return 0;
}


// Look-ahead schedule.  Main stream S: trailing updates and the row interchanges of everything outside the
// next panel.  Side stream P (high priority): the latency-bound chain of the NEXT panel -- fp16 pivots, the
// interchange of the panel's own columns, the fp64 panel -- which starts as soon as the update of panel k has
// reached the next panel's columns (the "strip") and runs under the rest of that update.
//   S: [swap_k others] [trsm_k strip][gemm_k strip] E1 [trsm_k rest][gemm_k rest] wait(E2) [swap_k+1 others] ...
//   P:                                   wait(E1) [pivots_k+1][swap_k+1 strip][dpanel_k+1] E2
// Per element the operations and their order are those of the single-stream schedule: results are identical.
// (Tried and dropped: confining S to 192 CUs and giving P the other 64 through hipExtStreamCreateWithCUMask for the
//  chain-bound panels.  In isolation the pivot kernel under a running hgemm went from 1766 us to 802 us, but in the
//  schedule the fp64 panel and the strip updates lost more on the smaller partitions than the pivot kernel gained:
//  fp64 537 vs 517 ms, fp16 322 vs 301 ms at N = 32768.)
static int factor_lookahead(mpf_ctx *c, double *d_A, int64_t lda, int64_t N, int32_t nb, int32_t *d_ipiv,
                            const mpf_opts &o, mpf_stats &st) {
    hipStream_t S = c->stream, P = c->pstream;
    EvPool ev(c);
    ev.keep = &st.ms_gemm;
    int rc = 0;
    const bool sink = sink_take(c, d_A, lda, N, nb);   // (mpf_factor_host: block rows leave as they become final, rowsink.hip)
    { // P must see everything already queued on S (the input matrix may still be in flight there)
        hipEvent_t e = ev.get();
        hipEventRecord(e, S);
        hipStreamWaitEvent(P, e, 0);
    }
    // panel 0 has nothing to hide under
    {
        const int pc = (int)(N < nb ? N : nb), pr = (int)N;
        if (pr > 1) {
            rc = ev.timed(st.ms_hpanel, S, [&] {
                int e = launch_hgetf2(c, d_A, lda, nullptr, 0, pr, pc, 0, d_ipiv, nullptr, 0, c->lists, 0, o.trailing == MPF_TRAIL_FP64 ? HP_FP64_WINDOW_ROWS : 0);
                const int64_t first = (int64_t)pc + nb < N ? (int64_t)pc + nb : N; // panel 0 and the strip right of it
                if (!e) e = launch_laswp_from_list(c, d_A, lda, first, c->lists);
                if (!e) e = launch_dgetf2_npv(c, d_A, lda, pr, pc, o.fused_panel, 0);
                return e;
            });
            st.panels++;
        }
    }
    for (int64_t k = 0; k < N && rc == 0; k += nb) {
        const int pc = (int)((N - k) < nb ? (N - k) : nb);
        if (N - k <= 1 || k + pc >= N) break;
        const int64_t n = N - k - pc;          // trailing size
        const int64_t nx = k + pc;             // first row/column of the next panel
        const int pc2 = (int)((N - nx) < nb ? (N - nx) : nb);
        const bool has_next = (N - nx) > 1;
        double *Ap = d_A + k * lda + k;
        double *A12 = d_A + nx * lda + k;      // U12 block row, starts at the strip
        const int64_t ns = has_next ? pc2 : n; // columns updated before the side stream may start
        // ---- strip (or everything, when no panel follows) ---------------------------------------------
        rc = ev.timed(st.ms_trsm, S, [&] { return launch_dtrsm_llnu(c, pc, ns, Ap, lda, A12, lda); });
        if (rc) break;
        rc = ev.timed(st.ms_gemm, S, [&] {
            int e = o.trailing != MPF_TRAIL_FP64 ? launch_cvt_l21(c, Ap + pc, lda, n, pc, o.trailing == MPF_TRAIL_FP16X3) : 0; // once per panel
            if (!e) e = trail_gemm(c, o, n, ns, pc, Ap + pc, A12, A12 + pc, lda);
            return e; });
        if (rc) break;
        count_gemm(st, o, n, ns, pc);
        if (!has_next) break;
        hipEvent_t e1 = ev.get(), e2 = ev.get();
        hipEventRecord(e1, S);
        // ---- side streams: the whole chain of panel k+1 ---------------------------------------------------
        hipEvent_t e2p = nullptr, e2t = nullptr;
        int rcp = 0;
        const bool piped = chain_pipelined(c, ev, st, o, d_A, lda, N, nx, pc2, d_ipiv, c->lists + (nx / nb), e1, &e2p, &e2t, &rcp) == 0;
        if (piped) rc = rcp;
        else {
            hipStreamWaitEvent(P, e1, 0);
            StreamSwap sw(c, P);
            double *Anx = d_A + nx * lda + nx;
            MovedList *ml = c->lists + (nx / nb);
            rc = ev.timed(st.ms_hpanel, P, [&] {
                return launch_hgetf2(c, Anx, lda, nullptr, 0, (int)(N - nx), pc2, (int)nx, d_ipiv + nx, nullptr, 0, ml, 0, o.trailing == MPF_TRAIL_FP64 ? HP_FP64_WINDOW_ROWS : 0); });
            if (!rc) rc = ev.timed(st.ms_dpanel, P, [&] {
                int e = launch_laswp_from_list(c, d_A + nx * lda, lda, pc2, ml);      // the panel's own columns
                if (!e) e = launch_dgetf2_npv(c, Anx, lda, (int)(N - nx), pc2, o.fused_panel, (int)nx);
                return e;
            });
        }
        if (rc) break;
        if (!piped) hipEventRecord(e2, P);
        st.panels++;
        // ---- main stream: interchanges of panel k on the columns right of the strip (the strip and the panel had
        //      theirs before the strip update), the rest of update k, then panel k+1's interchanges on the next strip
        if (n > pc2) {
            rc = ev.timed(st.ms_laswp, S, [&] {
                return launch_laswp_from_list(c, d_A + (nx + pc2) * lda, lda, N - nx - pc2, c->lists + (k / nb)); });
            if (rc) break;
            double *A12r = A12 + (int64_t)pc2 * lda;
            rc = ev.timed(st.ms_trsm, S, [&] { return launch_dtrsm_llnu(c, pc, n - pc2, Ap, lda, A12r, lda); });
            if (rc) break;
            if (sink) sink_notify(c, (int)(k / nb) + 1, S);   // block row k: pivoted, U solved
            rc = ev.timed(st.ms_gemm, S, [&] { return trail_gemm(c, o, n, n - pc2, pc, Ap + pc, A12r, A12r + pc, lda); });
            if (rc) break;
            count_gemm(st, o, n, n - pc2, pc);
        }
        if (piped) { hipStreamWaitEvent(S, e2p, 0); hipStreamWaitEvent(S, e2t, 0); }
        else hipStreamWaitEvent(S, e2, 0);
        rc = ev.timed(st.ms_laswp, S, [&] { // only the next strip now: it is all the next strip update needs
            const int64_t s0 = nx + pc2;
            const int64_t sw = (N - s0) < nb ? (N - s0) : nb;
            return sw > 0 ? launch_laswp_from_list(c, d_A + s0 * lda, lda, sw, c->lists + (nx / nb)) : 0;
        });
        if (o.verbose) printf("panel k=%lld rows=%lld cols=%d (look-ahead)\n", (long long)nx, (long long)(N - nx), pc2);
    }
    if (sink) {   // the last block rows; the sink has applied the left-hand interchanges on the way out, the device matrix keeps them owed
        hipEvent_t ep = ev.get(); hipEventRecord(ep, P); hipStreamWaitEvent(S, ep, 0);
        sink_notify(c, (int)((N + nb - 1) / nb), S);
    } else if (!rc) rc = ev.timed(st.ms_laswp, S, [&] { return launch_lazy_left_swaps(c, d_A, lda, N, nb, (int)((N + nb - 1) / nb), c->lists); });
    hipError_t se = sink_stream_wait(c, S);
    hipError_t sp = sink_stream_wait(c, P);
    if (c->tstream) { const hipError_t stt = sink_stream_wait(c, c->tstream); if (sp == hipSuccess) sp = stt; }
    if (!rc && (se != hipSuccess || sp != hipSuccess))
        return fail(c, -2, std::string("factorization failed: ") + hipGetErrorString(se != hipSuccess ? se : sp));
    ev.collect();
    st.lookahead = 1;
    return rc;
}

// Look-ahead schedule of the fp64 mode on a ROW-MAJOR working copy of the trailing matrix (round 3; default for N >= 8192).
// In the column-major matrix an interchange costs a 64-byte HBM sector each way for every moved row of every column right of
// the panel (~36 KB per column per panel, 39 ms of a factorization at N = 32768); the element (i, j) of the copy R lives at
// R[i * N + j], so an interchange moves contiguous row segments (two passes through a scratch image, laswp.hip).  Everything
// right of the current panel lives in R; the panels themselves (pivot kernel, fp64 panel) stay in the caller's column-major
// matrix A, which is also where the results end up:
//   per panel k:  L21 -> row-major scratch LT;  interchange on R (all columns right of the panel);
//                 U12 = L11^-1 R[k.., :]  (same TRSM kernel, addresses through strides);  U rows back to A;
//                 R[nx.., :] -= U12^T-form update: the SAME MFMA GEMM kernel run on the transposed problem
//                     C'(j, i) = R[(nx + i) N + nx + j],  A'(j, kk) = U(kk, j) = R[(k + kk) N + nx + j],  B'(kk, i) = L(i, kk) = LT[i * nb + kk]
//                 (per element the chain c = fma(-u, l, c), kk ascending: the product of the same two numbers as fma(-l, u, c));
//                 the next panel's columns go back to A (transpose) before its chain starts.
// Per element the operations and their order are those of factor_lookahead: bit-identical results (tests).
static int factor_lookahead_rm(mpf_ctx *c, double *d_A, int64_t lda, int64_t N, int32_t nb, int32_t *d_ipiv,
                               const mpf_opts &o, mpf_stats &st) {
    hipStream_t S = c->stream, P = c->pstream, T = c->tstream;
    EvPool ev(c);
    ev.keep = &st.ms_gemm;
    int rc = 0;
    const bool sink = sink_take(c, d_A, lda, N, nb);   // (mpf_factor_host: block rows leave as they become final, rowsink.hip)
    double *R = c->r64;
    const int64_t ldr = N;
    // mpf_factor_host with a LatePlan: columns from lp->c0[0] on are still on their way up.  `ce` = first column that is not here yet.
    LatePlan *lp = (c->late && !c->late->taken && c->late->nseg > 0 && lda == N) ? c->late : nullptr;
    if (lp) lp->taken = true;
    int lseg = 0;
    int64_t ce = lp ? lp->c0[0] : N;
    { hipEvent_t e = ev.get(); hipEventRecord(e, S); hipStreamWaitEvent(P, e, 0); if (T) hipStreamWaitEvent(T, e, 0); }
    const int pc0 = (int)(N < nb ? N : nb);
    // events after which the chain of the CURRENT panel is complete (what the main stream waited for at the end of the last turn)
    hipEvent_t chain_a = nullptr, chain_b = nullptr;
    if (N > 1) {
        // panel 0 (on the side stream) beside the transposition of everything right of it
        hipEvent_t e0 = ev.get();
        {
            StreamSwap sw(c, P);
            rc = ev.timed(st.ms_hpanel, P, [&] {
                int e = launch_hgetf2(c, d_A, lda, nullptr, 0, (int)N, pc0, 0, d_ipiv, nullptr, 0, c->lists, 0, o.trailing == MPF_TRAIL_FP64 ? HP_FP64_WINDOW_ROWS : 0);
                if (!e) e = launch_laswp_from_list(c, d_A, lda, pc0, c->lists);
                if (!e) e = launch_dgetf2_npv(c, d_A, lda, (int)N, pc0, o.fused_panel, 0);
                return e; });
            hipEventRecord(e0, P);
        }
        st.panels++;
        if (!rc && pc0 < ce) rc = ev.timed(st.ms_cvt, S, [&] { return launch_transpose64(c, d_A + (int64_t)pc0 * lda, lda, R + pc0, ldr, N, ce - pc0, true); });
        hipStreamWaitEvent(S, e0, 0);
        chain_a = e0;
    }
    // Two column lanes while the update is the longer side (trailing size > chain_pipeline_below: the chain is not pipelined there and
    // the helper stream T is free).  The columns right of the strip are split at `cm`: lane A = [nx, cm), lane B = [cm, N).  The main
    // stream then carries NOTHING BUT the two big updates, L(k), R(k), L(k + 1), ... back to back; every small launch -- a lane's
    // interchange, its TRSM, its U write-back, lane A's strip step that releases the next chain -- runs on the high-priority helper
    // stream T UNDER the other lane's update instead of between two updates:
    //     T:  a(k) | b(k) | a(k + 1) | ...     a(k) = lane A's small launches of panel k (needs L(k - 1) and chain k), b(k) = lane B's (needs R(k - 1))
    //     S:  ... R(k - 1) | wait a(k): L(k) | wait b(k): R(k) | ...
    // so a(k) runs under R(k - 1) and b(k) under L(k).  Per element nothing changes (each column still gets interchange, TRSM,
    // update, in this order): identical bits.  The split point only moves at a re-split (lane A's small launches then wait for
    // R(k - 1) too), because a column that changes lanes must have finished its update on the old one.
    int64_t cm = -1;                 // first column of lane B; -1: one lane
    hipEvent_t doneL = nullptr, doneR = nullptr, prevR = nullptr;   // L(k - 1), R(k - 1), R(k - 2) on the main stream
    hipEvent_t init_done = ev.get();
    hipEventRecord(init_done, S);    // the copy R is complete
    bool was_two = false;
    for (int64_t k = 0; k < N && rc == 0; k += nb) {
        const int pc = (int)((N - k) < nb ? (N - k) : nb);
        if (N - k <= 1 || k + pc >= N) break;
        const int64_t n = N - k - pc;          // trailing size
        const int64_t nx = k + pc;             // first row/column of the next panel
        const int pc2 = (int)((N - nx) < nb ? (N - nx) : nb);
        const bool has_next = (N - nx) > 1;
        double *Ap = d_A + k * lda + k;
        double *LT = c->rm_lt + ((k / nb) & 1) * N * (int64_t)nb;
        const int64_t ns = has_next ? pc2 : n; // columns updated before the side stream may start
        MovedList *lk = c->lists + (k / nb);
        // (not while the next pivot kernel needs (nearly) every CU -- one workgroup, one CU's LDS, per 256 rows: with a small launch or
        //  an update workgroup on a few of them it would hold all the others spinning until the update has drained: N = 65536 measured
        //  3583 ms with the lanes from the first panel on against 3336 without)
        // ---- a late column segment is due: it receives the panels [0, k / nb) one after the other on the main stream (behind the updates
        //      queued there), then the loop goes on with the wider matrix ------------------------------------------------------------------
        while (lp && lseg < lp->nseg && (k / nb >= lp->q[lseg] || nx + pc2 > ce || !has_next)) {
            const int64_t c_lo = lp->c0[lseg], c_hi = lseg + 1 < lp->nseg ? lp->c0[lseg + 1] : N, w = c_hi - c_lo;
            double *LT3 = c->rm_lt + 2 * N * (int64_t)nb;
            rc = launch_late_wait(c, lp->flags + lseg, lp->seq);
            if (!rc && lp->snapshot) {
                MPF_HIP_TRY(c, hipMemcpyAsync(lp->snapshot + c_lo * N, d_A + c_lo * lda, (size_t)w * N * sizeof(double), hipMemcpyDeviceToDevice, S));
                lp->snapped[lseg] = true;
            }
            if (!rc) rc = ev.timed(st.ms_cvt, S, [&] { return launch_transpose64(c, d_A + c_lo * lda, lda, R + c_lo, ldr, N, w, true); });
            // The replay, panel by panel, on the main stream.  (With the segment in two halves -- the main stream carrying only the updates, each
            // half's small launches on the helper stream under the other half's update, as in the two-lane loop below -- the call took the
            // same time within the boxes' spread: what the replay phases cost is not their small launches but the chain-bound panels in
            // front of them)
            for (int64_t kj = 0; kj < k && rc == 0; kj += nb) {
                const int64_t nxj = kj + nb, nj = N - nxj;
                double *Apj = d_A + kj * lda + kj;
                const MovedList *lj = c->lists + (kj / nb);
                rc = ev.timed(st.ms_cvt, S, [&] { return launch_transpose64(c, Apj + nb, lda, LT3, nb, nj, nb, true); });
                if (!rc) rc = ev.timed(st.ms_laswp, S, [&] { return launch_laswp_from_list_rm64(c, R + c_lo, ldr, w, lj, (int64_t)LASWP_MAXMOVED * c_lo); });
                if (!rc) rc = ev.timed(st.ms_trsm, S, [&] { return launch_dtrsm_llnu_strided(c, nb, w, Apj, lda, R + kj * ldr + c_lo, ldr, 1); });
                if (!rc) rc = ev.timed(st.ms_cvt, S, [&] { return launch_transpose64(c, d_A + c_lo * lda + kj, lda, R + kj * ldr + c_lo, ldr, nb, w, false); });
                if (!rc) rc = ev.timed(st.ms_gemm, S, [&] { return launch_dgemm_minus(c, w, nj, nb, R + kj * ldr + c_lo, ldr, LT3, nb, R + nxj * ldr + c_lo, ldr); });
                count_gemm(st, o, nj, w, nb);
                if (sink && !rc && c_hi == N) sink_notify(c, (int)(kj / nb) + 1, S);   // block row kj is complete now
            }
            ++lseg;
            ce = lseg < lp->nseg ? lp->c0[lseg] : N;
            cm = -1;                           // the lanes are cut anew, and lane A's small launches wait for everything queued here
            doneR = ev.get();
            hipEventRecord(doneR, S);
            doneL = nullptr;
        }
        if (rc) break;
        const int64_t cw = ce - nx - pc2;      // columns right of the strip (that are here)
        const bool pivots_fit = c->num_cus <= 0 || (N - nx + HP_R - 1) / HP_R <= (int64_t)c->num_cus * 4 / 5;
        const bool want_two = T && c->tune.fp64_two_lanes > 0 && has_next && pivots_fit && (N - nx) > c->tune.chain_pipeline_below &&
                              cw >= c->tune.fp64_two_lanes;
        bool resplit = false;
        if (!want_two) cm = -1;
        else if (cm < 0 || (cm - (nx + pc2)) * 100 < cw * (c->tune.fp64_lane_a_pct - 10)) {   // first split, or lane A's share has dropped
            cm = nx + pc2 + ((cw * c->tune.fp64_lane_a_pct / 100 + 127) / 128) * 128;
            if (cm > ce - 128) cm = ce - 128;
            resplit = true;
        }
        if (cm >= 0) {
            // ================================ two lanes ===========================================================================
            const int64_t aw = cm - (nx + pc2);
            if (!was_two && k > 0) {   // coming from one lane (the first pivot kernels needed every CU): everything the main stream holds
                doneL = ev.get();      // -- the whole update of the panel before -- is what lane A's small launches wait for
                hipEventRecord(doneL, S);
            }
            hipEvent_t lt_ready = ev.get();
            {   // L21 row-major, on the pivot stream right behind chain k (P ran it); this image was last read by L / R(k - 2)
                if (prevR) hipStreamWaitEvent(P, prevR, 0);
                StreamSwap sw(c, P);
                rc = ev.timed(st.ms_cvt, P, [&] { return launch_transpose64(c, Ap + pc, lda, LT, pc, n, pc, true); });
                hipEventRecord(lt_ready, P);
            }
            if (rc) break;
            hipEvent_t e1 = ev.get(), e2 = ev.get(), evA = ev.get(), evB = ev.get();
            {   // ---- a(k): lane A's interchange, the strip step, E1, lane A's TRSM and U write-back ---------------------------------
                if (chain_a) hipStreamWaitEvent(T, chain_a, 0);
                if (chain_b) hipStreamWaitEvent(T, chain_b, 0);
                if (doneL) hipStreamWaitEvent(T, doneL, 0); else hipStreamWaitEvent(T, init_done, 0);
                if (resplit && doneR) hipStreamWaitEvent(T, doneR, 0);
                StreamSwap sw(c, T);
                rc = ev.timed(st.ms_laswp, T, [&] { return launch_laswp_from_list_rm64(c, R + nx, ldr, cm - nx, lk); });
                if (!rc) rc = ev.timed(st.ms_trsm, T, [&] { return launch_dtrsm_llnu_strided(c, pc, ns, Ap, lda, R + k * ldr + nx, ldr, 1); });
                hipStreamWaitEvent(T, lt_ready, 0);
                if (!rc) rc = ev.timed(st.ms_gemm, T, [&] { return launch_dgemm_minus(c, ns, n, pc, R + k * ldr + nx, ldr, LT, pc, R + nx * ldr + nx, ldr); });
                count_gemm(st, o, n, ns, pc);
                // the next panel's columns return to the column-major matrix: rows nx.. (its U rows k..nx follow with lane A's below)
                if (!rc) rc = ev.timed(st.ms_cvt, T, [&] { return launch_transpose64(c, d_A + nx * lda + nx, lda, R + nx * ldr + nx, ldr, N - nx, pc2, false); });
                hipEventRecord(e1, T);
            }
            if (rc) break;
            {   // ---- chain of panel k + 1 on P --------------------------------------------------------------------------------------
                // E1 comes in the middle of R(k - 1).  A pivot kernel launched THERE would trickle onto the chip -- a workgroup needs a
                // whole CU's LDS, a CU only empties when the update's queue does -- and the workgroups that got in early would hold
                // their CUs spinning for milliseconds (measured: 3.5 instead of 1.1 ms per kernel, the update 17 % slower).  It starts
                // at the seam between R(k - 1) and L(k) instead: the chip is empty, the high-priority kernel takes its CUs first.
                hipStreamWaitEvent(P, e1, 0);
                if (doneR) hipStreamWaitEvent(P, doneR, 0);
                StreamSwap sw(c, P);
                double *Anx = d_A + nx * lda + nx;
                MovedList *ml = c->lists + (nx / nb);
                rc = ev.timed(st.ms_hpanel, P, [&] {
                    return launch_hgetf2(c, Anx, lda, nullptr, 0, (int)(N - nx), pc2, (int)nx, d_ipiv + nx, nullptr, 0, ml, 0, o.trailing == MPF_TRAIL_FP64 ? HP_FP64_WINDOW_ROWS : 0); });
                if (!rc) rc = ev.timed(st.ms_dpanel, P, [&] {
                    int e = launch_laswp_from_list(c, d_A + nx * lda, lda, pc2, ml);      // the panel's own columns
                    if (!e) e = launch_dgetf2_npv(c, Anx, lda, (int)(N - nx), pc2, o.fused_panel, (int)nx);
                    return e;
                });
                hipEventRecord(e2, P);
                st.panels++;
            }
            if (rc) break;
            {
                StreamSwap sw(c, T);
                rc = ev.timed(st.ms_trsm, T, [&] { return launch_dtrsm_llnu_strided(c, pc, aw, Ap, lda, R + k * ldr + nx + pc2, ldr, 1); });
                if (!rc) rc = ev.timed(st.ms_cvt, T, [&] { return launch_transpose64(c, d_A + nx * lda + k, lda, R + k * ldr + nx, ldr, pc, cm - nx, false); });
                hipEventRecord(evA, T);
                // ---- b(k): lane B's interchange, TRSM, U write-back (behind R(k - 1)) ----------------------------------------------
                if (doneR) hipStreamWaitEvent(T, doneR, 0);
                if (!rc) rc = ev.timed(st.ms_laswp, T, [&] { return launch_laswp_from_list_rm64(c, R + cm, ldr, ce - cm, lk, (int64_t)LASWP_MAXMOVED * (cm - nx)); });
                if (!rc) rc = ev.timed(st.ms_trsm, T, [&] { return launch_dtrsm_llnu_strided(c, pc, ce - cm, Ap, lda, R + k * ldr + cm, ldr, 1); });
                if (!rc) rc = ev.timed(st.ms_cvt, T, [&] { return launch_transpose64(c, d_A + cm * lda + k, lda, R + k * ldr + cm, ldr, pc, ce - cm, false); });
                hipEventRecord(evB, T);
                if (sink && !rc && ce == N) sink_notify(c, (int)(k / nb) + 1, T);   // block row k is complete in the column-major matrix
            }
            if (rc) break;
            // ---- main stream: the two updates ----------------------------------------------------------------------------------------
            hipStreamWaitEvent(S, evA, 0);
            rc = ev.timed(st.ms_gemm, S, [&] { return launch_dgemm_minus(c, aw, n, pc, R + k * ldr + nx + pc2, ldr, LT, pc, R + nx * ldr + nx + pc2, ldr); });
            if (rc) break;
            count_gemm(st, o, n, aw, pc);
            doneL = ev.get();
            hipEventRecord(doneL, S);
            hipStreamWaitEvent(S, evB, 0);
            rc = ev.timed(st.ms_gemm, S, [&] { return launch_dgemm_minus(c, ce - cm, n, pc, R + k * ldr + cm, ldr, LT, pc, R + nx * ldr + cm, ldr); });
            if (rc) break;
            count_gemm(st, o, n, ce - cm, pc);
            prevR = doneR;
            doneR = ev.get();
            hipEventRecord(doneR, S);
            chain_a = e2; chain_b = nullptr;
            was_two = true;
            if (o.verbose) printf("panel k=%lld rows=%lld cols=%d (look-ahead, row-major copy, two lanes split at %lld)\n", (long long)nx, (long long)(N - nx), pc2, (long long)cm);
            continue;
        }
        // ==================================== one lane =============================================================================
        if (was_two) {   // the lanes' last small launches and the chain of this panel (the main stream did not wait for either)
            hipEvent_t et = ev.get(); hipEventRecord(et, T); hipStreamWaitEvent(S, et, 0);
            if (chain_a) hipStreamWaitEvent(S, chain_a, 0);
            if (chain_b) hipStreamWaitEvent(S, chain_b, 0);
            was_two = false;
        }
        // ---- L21 row-major; interchanges of panel k on everything right of it (contiguous rows) ---------------------------------
        // (the transposition runs on the pivot stream, idle between two pivot kernels, beside the interchange and the strip's TRSM)
        hipEvent_t lt_ready = ev.get();
        {
            hipEvent_t eb = ev.get();
            hipEventRecord(eb, S);                 // chain k is complete (S has waited for it); the update before the last has read this LT image
            hipStreamWaitEvent(P, eb, 0);
            StreamSwap sw(c, P);
            rc = ev.timed(st.ms_cvt, P, [&] { return launch_transpose64(c, Ap + pc, lda, LT, pc, n, pc, true); });
            hipEventRecord(lt_ready, P);
        }
        if (!rc) rc = ev.timed(st.ms_laswp, S, [&] { return launch_laswp_from_list_rm64(c, R + nx, ldr, ce - nx, lk); });
        // ---- strip (or everything, when no panel follows) --------------------------------------------------------------------
        if (!rc) rc = ev.timed(st.ms_trsm, S, [&] { return launch_dtrsm_llnu_strided(c, pc, ns, Ap, lda, R + k * ldr + nx, ldr, 1); });
        hipStreamWaitEvent(S, lt_ready, 0);
        if (!rc) rc = ev.timed(st.ms_gemm, S, [&] { return launch_dgemm_minus(c, ns, n, pc, R + k * ldr + nx, ldr, LT, pc, R + nx * ldr + nx, ldr); });
        if (rc) break;
        count_gemm(st, o, n, ns, pc);
        if (!has_next) {   // the last rows of U and the last block go home
            rc = ev.timed(st.ms_cvt, S, [&] {
                int e = launch_transpose64(c, d_A + nx * lda + k, lda, R + k * ldr + nx, ldr, pc, N - nx, false);
                if (!e) e = launch_transpose64(c, d_A + nx * lda + nx, lda, R + nx * ldr + nx, ldr, N - nx, N - nx, false);
                return e; });
            break;
        }
        // the next panel's columns return to the column-major matrix: rows nx.. (its U rows k..nx follow with the others below)
        rc = ev.timed(st.ms_cvt, S, [&] { return launch_transpose64(c, d_A + nx * lda + nx, lda, R + nx * ldr + nx, ldr, N - nx, pc2, false); });
        if (rc) break;
        hipEvent_t e1 = ev.get(), e2 = ev.get();
        hipEventRecord(e1, S);
        // ---- side streams: the whole chain of panel k+1 ---------------------------------------------------------------------
        hipEvent_t e2p = nullptr, e2t = nullptr;
        int rcp = 0;
        const bool piped = chain_pipelined(c, ev, st, o, d_A, lda, N, nx, pc2, d_ipiv, c->lists + (nx / nb), e1, &e2p, &e2t, &rcp) == 0;
        if (piped) rc = rcp;
        else {
            hipStreamWaitEvent(P, e1, 0);
            StreamSwap sw(c, P);
            double *Anx = d_A + nx * lda + nx;
            MovedList *ml = c->lists + (nx / nb);
            rc = ev.timed(st.ms_hpanel, P, [&] {
                return launch_hgetf2(c, Anx, lda, nullptr, 0, (int)(N - nx), pc2, (int)nx, d_ipiv + nx, nullptr, 0, ml, 0, o.trailing == MPF_TRAIL_FP64 ? HP_FP64_WINDOW_ROWS : 0); });
            if (!rc) rc = ev.timed(st.ms_dpanel, P, [&] {
                int e = launch_laswp_from_list(c, d_A + nx * lda, lda, pc2, ml);      // the panel's own columns
                if (!e) e = launch_dgetf2_npv(c, Anx, lda, (int)(N - nx), pc2, o.fused_panel, (int)nx);
                return e;
            });
        }
        if (rc) break;
        if (!piped) hipEventRecord(e2, P);
        st.panels++;
        // ---- main stream: the rest of update k; the finished U rows of panel k go back to A -------------------------------------
        if (cw > 0) {
            rc = ev.timed(st.ms_trsm, S, [&] { return launch_dtrsm_llnu_strided(c, pc, cw, Ap, lda, R + k * ldr + nx + pc2, ldr, 1); });
            if (rc) break;
        }
        // (the write-back of the finished U rows runs on the chain's second stream, behind the chain's own launches: nothing
        //  waits for it before the end of the factorization, and the main stream goes straight on to the update)
        if (T) {
            hipEvent_t eu = ev.get();
            hipEventRecord(eu, S);
            hipStreamWaitEvent(T, eu, 0);
            StreamSwap sw(c, T);
            rc = ev.timed(st.ms_cvt, T, [&] { return launch_transpose64(c, d_A + nx * lda + k, lda, R + k * ldr + nx, ldr, pc, ce - nx, false); });
            if (sink && !rc && ce == N) sink_notify(c, (int)(k / nb) + 1, T);
        } else {
            rc = ev.timed(st.ms_cvt, S, [&] { return launch_transpose64(c, d_A + nx * lda + k, lda, R + k * ldr + nx, ldr, pc, ce - nx, false); });
            if (sink && !rc && ce == N) sink_notify(c, (int)(k / nb) + 1, S);
        }
        if (rc) break;
        if (cw > 0) {
            rc = ev.timed(st.ms_gemm, S, [&] { return launch_dgemm_minus(c, cw, n, pc, R + k * ldr + nx + pc2, ldr, LT, pc, R + nx * ldr + nx + pc2, ldr); });
            if (rc) break;
            count_gemm(st, o, n, cw, pc);
        }
        if (piped) { hipStreamWaitEvent(S, e2p, 0); hipStreamWaitEvent(S, e2t, 0); chain_a = e2p; chain_b = e2t; }
        else { hipStreamWaitEvent(S, e2, 0); chain_a = e2; chain_b = nullptr; }
        if (o.verbose) printf("panel k=%lld rows=%lld cols=%d (look-ahead, row-major copy)\n", (long long)nx, (long long)(N - nx), pc2);
    }
    if (!rc && lp && lseg < lp->nseg) rc = fail(c, -1, "factor_lookahead_rm: a late column segment was never brought up to date (plan beyond the matrix)");
    if (T) { hipEvent_t et = ev.get(); hipEventRecord(et, T); hipStreamWaitEvent(S, et, 0); }   // the last U write-backs / lane B
    { hipEvent_t ep = ev.get(); hipEventRecord(ep, P); hipStreamWaitEvent(S, ep, 0); }
    if (sink) sink_notify(c, (int)((N + nb - 1) / nb), S);   // the last block rows; the left-hand interchanges stay owed on the device (the sink applies them on the way out)
    else if (!rc) rc = ev.timed(st.ms_laswp, S, [&] { return launch_lazy_left_swaps(c, d_A, lda, N, nb, (int)((N + nb - 1) / nb), c->lists); });
    hipError_t se = sink_stream_wait(c, S);
    hipError_t sp = sink_stream_wait(c, P);
    if (T) { const hipError_t stt = sink_stream_wait(c, T); if (sp == hipSuccess) sp = stt; }
    if (!rc && (se != hipSuccess || sp != hipSuccess))
        return fail(c, -2, std::string("factorization failed: ") + hipGetErrorString(se != hipSuccess ? se : sp));
    ev.collect();
    st.lookahead = 1;
    return rc;
}

// Two-level schedule (default of the fp16 trailing modes).  The fp16 update streams the trailing matrix through the chip
// once per launch: with K = nb = 256 it is HBM-bound at ~5 % of the fp16 MFMA peak.  Here `sb` panels form a super-panel
// [c0, c1): inside it a panel only updates an INNER REGION of at most (sb + 1) * nb columns (right-looking, K = nb, on the
// fp64 matrix), and everything right of the inner region gets ONE update with K = sb * nb per super-panel -- after the
// interchanges of the sb panels and the U block-row
//   U[c0:c1, cols] = L_SS^-1 A[c0:c1, cols]      (one blocked fp64 TRSM over the super-panel's unit-lower block)
// so the trailing matrix is read and written N / (sb * nb) times instead of N / nb times.
//
// Inner region = the super-panel's own columns PLUS the next super-panel's first panel [c1, c1 + nb) (round 3).  That
// look-ahead panel is brought up to date panel by panel like the super-panel's own columns, so when the last panel of the
// super-panel is done the next pivot kernel starts at once: the heavy end-of-super-panel work (operand images, block-row,
// K = sb * nb update of the next inner region's columns, conversions) runs UNDER that chain instead of in front of it, and
// every panel of the factorization has the same short critical path (chain + one strip step).
//
// What is right of the inner region is owed the super-panel's update: the columns that join the next inner region
// [c1 + nb, c2 + nb) get it at the end of the super-panel, the rest as a pending job that the main stream works off in
// pieces, left to right, one under each panel chain of the next super-panel.  Two L images stay alive for that.
// fp16 modes: the matrix right of the inner region lives in an fp32 WORKING COPY W (same coordinates as A, but ROW-major:
// element (i, j) at W[i * N + j]): the K-updates then move 8 instead of 16 bytes of HBM per element and an interchange moves
// contiguous row segments instead of one element per 64-byte sector (laswp.hip).  A column range returns to fp64
// when it joins the inner region (panels, TRSMs and the finished factors are fp64); the block-row of a super-panel is
// converted just before its TRSM.  The products are fp16 x fp16 anyway (contract C6): an fp32 accumulator of the trailing
// matrix adds 2^-24 per update to an error of 2^-11 per product.  Option fp16_work32 = 0 updates the fp64 matrix in place.
// fp64 mode (option superpanel_fp64 > 1): same loop, updates straight from the matrix; per element the fma chain is the
// one-level schedule's (k ascending across the panels): identical bits.  overlap = false: same operations on one stream.
static int factor_superpanel(mpf_ctx *c, double *d_A, int64_t lda, int64_t N, int32_t nb, int32_t *d_ipiv, const mpf_opts &o,
                             mpf_stats &st, int sb, bool overlap) {
    // Three lanes (overlap = true; otherwise everything runs on the one stream, same operations):
    //   chain    P (+ T): pivots and fp64 panel of the next panel;
    //   inner    Ci (high priority): the short critical work between two chains -- interchanges of the super-panel's earlier
    //            columns, the strip step that releases the next chain (E1), the rest of the inner region;
    //   far      S: everything right of the inner region (block-row tasks, operand images, K = sb * nb updates, conversions).
    // The lanes meet only at events: a far launch never sits in front of a strip step in a queue, and far work that takes
    // longer than one chain spills under the next one instead of delaying it.
    hipStream_t S = c->stream, P = overlap ? c->pstream : c->stream;
    // (the inner lane shares the chain's second stream T: its work sits between two chains, when T is idle -- a fourth stream
    //  of its own would exceed the four hardware queues a process gets by default, and streams that share a queue serialise)
    hipStream_t Ci = (overlap && c->tstream) ? c->tstream : c->stream;
    const bool lanes = Ci != S;
    EvPool ev(c);
    ev.keep = &st.ms_gemm;
    int rc = 0;
    const bool split = o.trailing == MPF_TRAIL_FP16X3;
    const bool f64 = o.trailing == MPF_TRAIL_FP64;
    const bool use32 = !f64 && c->tune.fp16_work32 != 0 && N > (int64_t)(sb + 1) * nb;
    float *W = nullptr;
    const int64_t ldw = N;
    if (use32) {
        if (c->w32_n < N) {
            if (c->w32) (void)hipFree(c->w32);
            c->w32 = nullptr; c->w32_n = 0;
            MPF_HIP_TRY(c, hipMalloc((void **)&c->w32, (size_t)N * (size_t)N * sizeof(float)));
            c->w32_n = N;
        }
        W = c->w32;
    }
    if (overlap) {
        hipEvent_t e = ev.get();
        hipEventRecord(e, S);
        hipStreamWaitEvent(P, e, 0);
    }
    const int64_t sbw = (int64_t)sb * nb;
    const int kst = (int)((sbw + 63) & ~(int64_t)63);                       // row stride of the far U image (elements)
    const int64_t far_cap = N - (int64_t)(sb + 1) * nb;                     // most far columns any super-panel has
    const int64_t inner_u_off = (!f64 && far_cap > 0) ? far_cap * kst : 0;  // the inner steps' U image lives behind the far image
    const int64_t brow_l_off = (int64_t)N * c->h_kmax;                      // block-row L images: in the rows the image buffers hold beyond N
    auto width = [&](int64_t k) { return (int)((N - k) < nb ? (N - k) : nb); };
    auto chain = [&](int64_t kx, hipStream_t s) -> int { // pivots, own-column interchanges, fp64 panel of the panel at kx
        const int pcx = width(kx);
        const int prx = (int)(N - kx);
        if (prx <= 1) return 0;
        StreamSwap sw(c, s);
        double *Ax = d_A + kx * lda + kx;
        MovedList *ml = c->lists + (kx / nb);
        int e = ev.timed(st.ms_hpanel, s, [&] { return launch_hgetf2(c, Ax, lda, nullptr, 0, prx, pcx, (int)kx, d_ipiv + kx, nullptr, 0, ml, 0, o.trailing == MPF_TRAIL_FP64 ? HP_FP64_WINDOW_ROWS : 0); });
        if (!e) e = ev.timed(st.ms_dpanel, s, [&] {
            int e2 = launch_laswp_from_list(c, d_A + kx * lda, lda, pcx, ml);
            if (!e2) e2 = launch_dgetf2_npv(c, Ax, lda, prx, pcx, o.fused_panel, (int)kx);
            return e2; });
        st.panels++;
        return e;
    };
    auto side_chain = [&](int64_t kx, hipEvent_t e1, hipEvent_t &e2) -> int { // chain on P (and T) behind E1, E2 behind all of it
        if (!overlap) return chain(kx, S);
        e2 = ev.get();
        if (N - kx > 1) {
            hipEvent_t e2p = nullptr, e2t = nullptr;
            int rcp = 0;
            if (chain_pipelined(c, ev, st, o, d_A, lda, N, kx, width(kx), d_ipiv, c->lists + (kx / nb), e1, &e2p, &e2t, &rcp) == 0) {
                st.panels++;
                if (!rcp) { hipStreamWaitEvent(c->tstream, e2p, 0); hipEventRecord(e2, c->tstream); }
                return rcp;
            }
        }
        hipStreamWaitEvent(P, e1, 0);
        int e = chain(kx, P);
        hipEventRecord(e2, P);
        return e;
    };
    // one right-looking step of panel k (width pc) on the inner-region columns [col0, col0 + ncols), all in the fp64 matrix, on
    // the inner lane: interchange, TRSM with the panel's L11, K = pc update of the rows below.  need_img: convert the panel's L21
    // image first.
    // the panel's L21 image (fp16 modes): on the pivot stream, which is idle between two pivot kernels, beside the strip's TRSM
    hipEvent_t img_ready = nullptr;
    auto l21_image = [&](int64_t k, int pc) -> int {
        if (f64) return 0;
        const int64_t mrows = N - k - pc;
        if (mrows <= 0) return 0;
        hipStream_t Is = (lanes && P != S) ? P : Ci;
        if (Is != Ci) { hipEvent_t e = ev.get(); hipEventRecord(e, Ci); hipStreamWaitEvent(Is, e, 0); }   // the chain of panel k is done (Ci waited for it)
        StreamSwap sw(c, Is);
        int e = ev.timed(st.ms_cvt, Is, [&] { return launch_cvt_l21(c, d_A + k * lda + k + pc, lda, mrows, pc, split); });
        if (Is != Ci) { img_ready = ev.get(); hipEventRecord(img_ready, Is); } else img_ready = nullptr;
        return e;
    };
    auto inner_step = [&](int64_t k, int pc, int64_t col0, int64_t ncols, bool first) -> int {
        if (ncols <= 0) return 0;
        StreamSwap sw(c, Ci);
        double *Ap = d_A + k * lda + k, *A12 = d_A + col0 * lda + k;
        const int64_t mrows = N - k - pc;
        int e = ev.timed(st.ms_trsm, Ci, [&] { return launch_dtrsm_llnu(c, pc, ncols, Ap, lda, A12, lda); });
        if (e || mrows <= 0) return e;
        if (f64) e = ev.timed(st.ms_gemm, Ci, [&] { return launch_dgemm_minus(c, mrows, ncols, pc, Ap + pc, lda, A12, lda, A12 + pc, lda); });
        else {
            e = ev.timed(st.ms_cvt, Ci, [&] { return launch_cvt_u12(c, A12, lda, pc, ncols, split, inner_u_off); });
            if (first && img_ready) hipStreamWaitEvent(Ci, img_ready, 0);
            if (!e) e = ev.timed(st.ms_gemm, Ci, [&] { return launch_hgemm_images(c, mrows, ncols, pc, A12 + pc, lda, false, split, 0, 0, inner_u_off); });
        }
        count_gemm(st, o, mrows, ncols, pc);
        return e;
    };
    // ---- far lane: everything right of the inner region ("far" columns [f0, N) of the super-panel [s0, s1)) -----------------------
    // U block-row, one 256-row block per finished panel p of the super-panel (task T_p, full width: one launch of each kind per
    // block instead of one per column piece -- a 256-row TRSM costs the same ~90 us for 7000 columns as for 28000):
    //   interchange of panel p on the far columns; rows of block p lose L[p, < p] U[< p] (the update the one-level schedule
    //   would have given them panel by panel: fp16 modes through the fp16 MFMA kernel with the mode's operands, fp64 mode
    //   through the fp64 MFMA GEMM -- same fma chains as the blocked TRSM, identical bits); the block returns to fp64;
    //   TRSM with the panel's L11; the finished U rows are appended to the U image of the K = s1 - s0 update.
    // Then the update itself: operand image of L[s1.., s0..s1) once, the kernel on the columns that join the next inner region
    // first (they return to fp64: event NI releases the inner lane), then on all the rest in one launch.  U image:
    // Uh[n_far][kst] at the start of the context's U buffer; the inner region's steps keep their small image behind it.
    struct Far { bool on = false; int64_t s0 = 0, s1 = 0, f0 = 0; int done = 0, img = 1; } far;
    std::vector<hipEvent_t> panel_done;   // per panel of the current super-panel: recorded on the inner lane after its eager interchanges
    auto far_task = [&](int p) -> int {   // T_p of the current far job
        const int64_t kq = far.s0 + (int64_t)p * nb, f0 = far.f0, ncols = N - f0;
        if (ncols <= 0) return 0;
        if (lanes) hipStreamWaitEvent(S, panel_done[(size_t)p], 0);
        int e = ev.timed(st.ms_laswp, S, [&] {
            return use32 ? launch_laswp_from_list_f32(c, W + f0, ldw, ncols, c->lists + (kq / nb))      // row-major copy: contiguous rows
                         : launch_laswp_from_list(c, d_A + f0 * lda, lda, ncols, c->lists + (kq / nb)); });
        if (!e && p > 0) {
            const int K = p * nb;
            if (f64) e = ev.timed(st.ms_trsm, S, [&] {
                return launch_dgemm_minus(c, nb, ncols, K, d_A + far.s0 * lda + kq, lda, d_A + f0 * lda + far.s0, lda, d_A + f0 * lda + kq, lda); }, &st.ms_blockrow);
            else {
                e = ev.timed(st.ms_cvt, S, [&] { return launch_cvt_l21(c, d_A + far.s0 * lda + kq, lda, nb, K, split, 0, brow_l_off); });
                if (!e) e = ev.timed(st.ms_trsm, S, [&] {
                    return use32 ? launch_hgemm_images_rowmajor(c, nb, ncols, K, W + kq * ldw + f0, ldw, split, 0, brow_l_off, 0, 0, kst)
                                 : launch_hgemm_images(c, nb, ncols, K, d_A + f0 * lda + kq, lda, false, split, 0, brow_l_off, 0, 0, kst); }, &st.ms_blockrow);
            }
        }
        if (!e && use32) e = ev.timed(st.ms_cvt, S, [&] { return launch_cvt_f32_f64(c, W + kq * ldw + f0, ldw, d_A + f0 * lda + kq, lda, nb, ncols); });
        if (!e) e = ev.timed(st.ms_trsm, S, [&] { return launch_dtrsm_llnu(c, nb, ncols, d_A + kq * lda + kq, lda, d_A + f0 * lda + kq, lda); }, &st.ms_blockrow);
        if (!e && !f64) e = ev.timed(st.ms_cvt, S, [&] { return launch_cvt_u12(c, d_A + f0 * lda + kq, lda, nb, ncols, split, (int64_t)p * nb, kst); });
        return e;
    };
    auto far_tasks_upto = [&](int ready) -> int { // run the block-row tasks of the panels 0 .. ready - 1 that have not run yet
        int e = 0;
        while (far.on && !e && far.done < ready) { e = far_task(far.done); far.done++; }
        return e;
    };
    // the K = s1 - s0 update of the far job on the columns [col0, col0 + ncols) (block-row and U image complete)
    auto big_update = [&](const Far &fj, int64_t col0, int64_t ncols) -> int {
        const int64_t mrows = N - fj.s1;
        const int K = (int)(fj.s1 - fj.s0);
        if (ncols <= 0 || mrows <= 0) return 0;
        int e;
        if (f64) e = ev.timed(st.ms_gemm, S, [&] { return launch_dgemm_minus(c, mrows, ncols, K, d_A + fj.s0 * lda + fj.s1, lda, d_A + col0 * lda + fj.s0, lda, d_A + col0 * lda + fj.s1, lda); }, &st.ms_gemm_big);
        else e = ev.timed(st.ms_gemm, S, [&] {
                const int64_t uo = (col0 - fj.f0) * kst;
                return use32 ? launch_hgemm_images_rowmajor(c, mrows, ncols, K, W + fj.s1 * ldw + col0, ldw, split, fj.img, 0, uo, 0, kst)
                             : launch_hgemm_images(c, mrows, ncols, K, d_A + col0 * lda + fj.s1, lda, false, split, fj.img, 0, uo, 0, kst); }, &st.ms_gemm_big);
        const double cb = f64 ? 16.0 : (use32 ? 8.0 : 16.0), opb = f64 ? 8.0 : (split ? 4.0 : 2.0);
        count_gemm(st, o, mrows, ncols, K, cb);
        st.gemm_big_flops += 2.0 * (double)mrows * (double)ncols * K;
        st.gemm_big_bytes += cb * (double)mrows * (double)ncols + opb * K * (double)(mrows + ncols);
        st.gemm_big_launches++;
        return e;
    };
    // a column range that joins the next inner region returns to fp64: rows >= r0 (above them: finished U rows in A)
    auto back_to_f64 = [&](int64_t r0, int64_t col0, int64_t ncols) -> int {
        if (!use32 || ncols <= 0) return 0;
        return ev.timed(st.ms_cvt, S, [&] { return launch_cvt_f32_f64(c, W + r0 * ldw + col0, ldw, d_A + col0 * lda + r0, lda, N - r0, ncols); });
    };
    // Without lanes (single stream) the far work is issued in the same program order, so a pending job is just a deferred
    // launch: the rest of a super-panel's update is issued right after the part the next inner region needs.
    auto inner_end = [&](int64_t c1) { return (c1 + nb) < N ? (c1 + nb) : N; };   // inner region of the super-panel ending at c1
    {
        const int64_t e1 = inner_end(sbw < N ? sbw : N);
        if (use32 && e1 < N) rc = ev.timed(st.ms_cvt, S, [&] { return launch_cvt_f64_f32(c, d_A + e1 * lda, lda, W + e1, ldw, N, N - e1); });
    }
    if (!rc) rc = chain(0, S);
    hipEvent_t chain_done = nullptr;       // the chain of the panel the next iteration starts with
    if (lanes) { chain_done = ev.get(); hipEventRecord(chain_done, S); }
    hipEvent_t ni_ready = nullptr;         // the columns that joined the current inner region are back in fp64 and up to date (far lane)
    int next_img = 1;
    for (int64_t c0 = 0; c0 < N && rc == 0; c0 += sbw) {
        const int64_t c1 = (c0 + sbw) < N ? (c0 + sbw) : N;
        const int64_t e1c = inner_end(c1);
        const int npan = (int)((c1 - c0 + nb - 1) / nb);
        far = Far();
        if (e1c < N) {
            far.on = true; far.s0 = c0; far.s1 = c1; far.f0 = e1c; far.img = next_img; next_img = 3 - next_img;
            if (!f64 && nb % 64 != 0) {   // image blocks of odd width: the padding the kernels read beyond a block must be zero
                const size_t bytes = (size_t)(N - e1c) * kst * sizeof(unsigned short);
                MPF_HIP_TRY(c, hipMemsetAsync(c->h_U, 0, bytes, S));
                if (split) MPF_HIP_TRY(c, hipMemsetAsync(c->h_U + c->h_rows * c->h_kmax, 0, bytes, S));
            }
        }
        panel_done.assign((size_t)npan, nullptr);
        bool stop = false;
        int it = 0;
        for (int64_t k = c0; k < c1 && rc == 0; k += nb, ++it) {
            const int pc = width(k);
            const int64_t nx = k + pc;
            if (lanes) {
                hipStreamWaitEvent(Ci, chain_done, 0);
                if (it == 0 && ni_ready) hipStreamWaitEvent(Ci, ni_ready, 0);   // this super-panel's new inner columns
            }
            // ONE interchange launch for the whole inner region: the super-panel's earlier columns [c0, k) (now, not with the deferred
            // ones at the end: the K = c1 - c0 update reads them in final row order) and the columns [nx, e1) right of the panel
            {
                StreamSwap sw(c, Ci);
                const int64_t right = (N - k <= 1 || nx >= N) ? 0 : e1c - nx;
                if ((k - c0) + right > 0)
                    rc = ev.timed(st.ms_laswp, Ci, [&] { return launch_laswp_from_list_hole(c, d_A + c0 * lda, lda, (k - c0) + right, c->lists + (k / nb), k - c0, pc); });
            }
            if (rc) break;
            if (lanes) { panel_done[(size_t)it] = ev.get(); hipEventRecord(panel_done[(size_t)it], Ci); }
            if (N - k <= 1 || nx >= N) { stop = true; break; }
            hipEvent_t e2 = nullptr;
            const int64_t sw_ = (e1c - nx) < nb ? (e1c - nx) : nb;     // the strip = the next panel's columns: always inside the inner region
            rc = l21_image(k, pc);
            if (!rc) rc = inner_step(k, pc, nx, sw_, true);
            hipEvent_t e1 = nullptr;
            if (!rc && overlap) { e1 = ev.get(); hipEventRecord(e1, Ci); }   // E1: the next panel's columns are up to date
            // the rest of the inner region is issued BEFORE the chain's launches: on the shared stream it runs beside the first
            // 32 columns of the pivot kernel (stream P), and the fp64 panel's pieces (this stream) follow 32 columns behind anyway
            if (!rc) rc = inner_step(k, pc, nx + sw_, e1c - nx - sw_, false);
            if (!rc && N - nx > 1) rc = side_chain(nx, e1, e2);
            // ---- far lane: panels c0 .. k of this super-panel are complete ----------------------------------------------------
            if (!rc) rc = far_tasks_upto(it + 1);
            if (rc) break;
            if (nx >= c1 && far.on) {
                // the super-panel is complete: its update on the columns that join the next inner region first, then the rest
                const int64_t c2 = (c1 + sbw) < N ? (c1 + sbw) : N;
                const int64_t e2c = inner_end(c2);
                if (!rc && !f64) rc = ev.timed(st.ms_cvt, S, [&] { return launch_cvt_l21(c, d_A + c0 * lda + c1, lda, N - c1, (int)(c1 - c0), split, far.img); });
                if (!rc) rc = big_update(far, e1c, e2c - e1c);
                if (!rc) rc = back_to_f64(c1, e1c, e2c - e1c);
                if (lanes) { ni_ready = ev.get(); hipEventRecord(ni_ready, S); }
                if (!rc) rc = big_update(far, e2c, N - e2c);
                if (rc) break;
            }
            if (overlap) { if (e2) { if (lanes) chain_done = e2; else hipStreamWaitEvent(S, e2, 0); } }
            if (o.verbose) printf("panel k=%lld rows=%lld (super-panel [%lld, %lld))\n", (long long)nx, (long long)(N - nx), (long long)c0, (long long)c1);
        }
        if (stop) break;
    }
    if (lanes) { hipEvent_t e = ev.get(); hipEventRecord(e, Ci); hipStreamWaitEvent(S, e, 0); if (chain_done) hipStreamWaitEvent(S, chain_done, 0); }
    if (!rc) rc = ev.timed(st.ms_laswp, S, [&] { return launch_lazy_left_swaps(c, d_A, lda, N, nb, (int)((N + nb - 1) / nb), c->lists, sb); });
    hipError_t se = hipStreamSynchronize(S);
    hipError_t sp = overlap ? hipStreamSynchronize(P) : hipSuccess;
    if (overlap && c->tstream) { const hipError_t stt = hipStreamSynchronize(c->tstream); if (sp == hipSuccess) sp = stt; }
    if (lanes) { const hipError_t sc = hipStreamSynchronize(Ci); if (sp == hipSuccess) sp = sc; }
    if (!rc && (se != hipSuccess || sp != hipSuccess))
        return fail(c, -2, std::string("factorization failed: ") + hipGetErrorString(se != hipSuccess ? se : sp));
    ev.collect();
    st.lookahead = overlap ? 1 : 0;
    return rc;
}

// buffers of factor_lookahead_rm: the N x N fp64 row-major copy, the interchange scratch (2 * HP_MAXCOLS rows x N) and the
// panel's row-major L21 (N x nb).  Returns non-zero (and leaves the context usable) when the device has no room: the caller
// then runs the in-place schedule.
int mpf_ensure_rowmajor_copy(mpf_ctx *c, int64_t N, int64_t cols, int32_t nb) {   // N rows x cols columns (mpf_factor_dist: the local columns)
    auto grow = [&](double *&p, int64_t &cap, int64_t need) -> int {
        if (p && cap >= need) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc((void **)&p, (size_t)need * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return 1; }
        cap = need;
        return 0;
    };
    const int bad = grow(c->r64, c->r64_cap, N * cols) || grow(c->rm_tmp, c->rm_tmp_cap, (int64_t)LASWP_MAXMOVED * cols) ||
                    grow(c->rm_lt, c->rm_lt_cap, 3 * N * (int64_t)nb);   // two L21 images: panel k + 1's is written while update k still reads; the third: the replay on late column segments (LatePlan)
    if (bad) {   // no room: the in-place schedule runs, and nothing of this one stays resident
        if (c->r64) (void)hipFree(c->r64);
        if (c->rm_tmp) (void)hipFree(c->rm_tmp);
        if (c->rm_lt) (void)hipFree(c->rm_lt);
        c->r64 = c->rm_tmp = c->rm_lt = nullptr;
        c->r64_cap = c->rm_tmp_cap = c->rm_lt_cap = 0; c->r64_n = 0;
        return 1;
    }
    c->r64_n = N;
    return 0;
}

int mpf_factor_dev(mpf_ctx *c, double *d_A, int64_t lda, int64_t N, int32_t nb, int32_t *d_ipiv, const mpf_opts *opts) {
    if (!c || !d_A || !d_ipiv) return -1;
    if (N <= 0 || nb <= 0) return fail(c, -1, "mpf_factor: N and panel width must be positive");
    if (lda < N) return fail(c, -1, "mpf_factor: lda < N");
    if (N > INT_MAX / 2) return fail(c, -1, "mpf_factor: N too large");
    if (nb > 65535) return fail(c, -1, "mpf_factor: panel width > 65535");
    mpf_opts o{};
    if (opts) o = *opts;
    if (o.trailing < MPF_TRAIL_FP64 || o.trailing > MPF_TRAIL_FP16X3) return fail(c, -1, "mpf_factor: unknown trailing mode");
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    // tuned schedules need every panel to fit the LDS pivot kernel (<= 256 columns, all its workgroups resident at once);
    // anything else, and callers that ask for it, get the generic schedule
    const bool force_generic = o.pivot_path == 1 || safe_pivots(c);
    const bool generic = force_generic || !hgetf2_lds_eligible(c, (int)N, (int)(nb < N ? nb : N));
    // super-panels need equal-width column blocks left of every panel (deferred interchanges), i.e. N > sb * nb columns of nb
    int want_sb = o.trailing != MPF_TRAIL_FP64 ? c->tune.superpanel_fp16 : c->tune.superpanel_fp64;
    if (o.trailing != MPF_TRAIL_FP64 && want_sb == 0) want_sb = (o.trailing == MPF_TRAIL_FP16 && N >= 24576) ? 6 : 4;   // (MpfTuning::superpanel_fp16)
    if (o.superpanel > 0) want_sb = o.superpanel > 8 ? 8 : o.superpanel;
    const int sb = (!generic && !o.sync_timing && (int64_t)want_sb * nb < N) ? want_sb : 1;
    if (o.trailing != MPF_TRAIL_FP64) {
        const int64_t kimg = (int64_t)sb * nb < 8 * HP_MAXCOLS ? (int64_t)sb * nb : 8 * HP_MAXCOLS; // the generic schedule cuts K to the images
        int e = mpf_ensure_h_images(c, N + HP_MAXCOLS + 64, (int)kimg, sb > 1); if (e) return e;   // + room for the block-row L images
    }
    if (!generic) {   // per-panel moved-row lists + scratch of the deferred left-hand interchanges
        const int npanels = (int)((N + nb - 1) / nb);
        if (npanels > c->lists_cap) {
            if (c->lists) hipFree(c->lists);
            c->lists = nullptr; c->lists_cap = 0;
            MPF_HIP_TRY(c, hipMalloc((void **)&c->lists, (size_t)npanels * sizeof(MovedList)));
            c->lists_cap = npanels;
        }
        // N x nb doubles for the deferred left-hand interchanges; at least N x 256 so that the scratch also holds the
        // 2 * HP_MAXCOLS moved rows x N columns (fp32) of an interchange on the row-major working copy
        const int64_t pneed = N * (int64_t)(nb > HP_MAXCOLS ? nb : HP_MAXCOLS);
        if (pneed > c->perm_cap) {
            if (c->perm_tmp) hipFree(c->perm_tmp);
            c->perm_tmp = nullptr; c->perm_cap = 0;
            MPF_HIP_TRY(c, hipMalloc((void **)&c->perm_tmp, (size_t)pneed * sizeof(double)));
            c->perm_cap = pneed;
        }
        if (N > c->fmap_cap) {
            if (c->Fmap) hipFree(c->Fmap);
            c->Fmap = nullptr; c->fmap_cap = 0;
            MPF_HIP_TRY(c, hipMalloc((void **)&c->Fmap, (size_t)2 * N * sizeof(int)));   // the map and its inverse
            c->fmap_cap = N;
        }
        MPF_HIP_TRY(c, hipMemsetAsync(c->lists, 0, (size_t)npanels * sizeof(MovedList), c->stream));
    }
    const int imax = INT_MAX;
    MPF_HIP_TRY(c, hipMemcpyAsync(&c->ws->info, &imax, sizeof(int), hipMemcpyHostToDevice, c->stream));
    MPF_HIP_TRY(c, hipMemsetAsync(&c->ws->hp_timeouts, 0, sizeof(int), c->stream));
    mpf_stats st{};
    st.n = N; st.nb = nb; st.superpanel = sb;
    const bool lookahead = !o.sync_timing && !o.no_lookahead && !c->tune.no_lookahead && c->pstream != nullptr;
    // the row-major working copy of the fp64 mode is allocated BEFORE the clock starts (a context's first call pays hipMalloc
    // of N x N doubles once; ms_total is the factorization)
    const bool use_rm = !generic && sb <= 1 && lookahead && o.trailing == MPF_TRAIL_FP64 && c->tune.fp64_rowmajor &&
                        N >= c->tune.fp64_rowmajor_min_n && N > nb && mpf_ensure_rowmajor_copy(c, N, N, nb) == 0;
    if (c->late && !use_rm) {   // (mpf_factor_host planned on the row-major schedule and it is not the one that runs: everything has to be here first)
        const int e = feed_finish(c);
        c->late = nullptr;
        if (e) return e;
    }
    MPF_HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    int rc;
    if (generic) rc = factor_generic(c, d_A, lda, N, nb, d_ipiv, o, st, force_generic);
    else if (sb > 1) rc = factor_superpanel(c, d_A, lda, N, nb, d_ipiv, o, st, sb, lookahead);
    else if (use_rm) rc = factor_lookahead_rm(c, d_A, lda, N, nb, d_ipiv, o, st);
    else if (lookahead) rc = factor_lookahead(c, d_A, lda, N, nb, d_ipiv, o, st);
    else {
        mpf_opts o2 = o;
        rc = factor_sync_timed(c, d_A, lda, N, nb, d_ipiv, o2, st);
    }
    hipEventRecord(c->ev1, c->stream);
    hipError_t se = sink_stream_wait(c, c->stream);
    if (rc) return rc;
    if (se != hipSuccess) return fail(c, -2, std::string("factorization failed: ") + hipGetErrorString(se));
    float ms = 0;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    st.ms_total = ms;
    if (o.trailing == MPF_TRAIL_FP64 && N >= 8192 && ms > 0)   // what mpf_gesv prices an fp64 refactorization with
        c->fp64_rate_tflops = 2.0 / 3.0 * (double)N * (double)N * (double)N / (ms * 1e-3) / 1e12;
    int info = 0, flags0 = 0;
    MPF_HIP_TRY(c, hipMemcpy(&info, &c->ws->info, sizeof(int), hipMemcpyDeviceToHost));
    MPF_HIP_TRY(c, hipMemcpy(&flags0, &c->ws->hp_timeouts, sizeof(int), hipMemcpyDeviceToHost));
    st.info = info == INT_MAX ? 0 : info;
    st.hpanel_timeouts = flags0;
    const double h2d = c->stats.ms_h2d, d2h = c->stats.ms_d2h;
    c->stats = st;
    c->stats.ms_h2d = h2d; c->stats.ms_d2h = d2h;
    if (flags0) return fail(c, -4, "fp16 pivot kernel: inter-workgroup hand-off timed out (its workgroups were not all resident: GPU "
                                   "shared with another process?).  d_A and ipiv are invalid.  Use mpf_opts.pivot_path = 1 or MPF_SAFE_PIVOTS=1");
    return st.info;
}

// Device copies of the host entry point's buffers live in the context and only ever grow (round 4): the reference allocates and
// frees them inside every MPF() call (MPF.cu:80-94,250-255), which at N = 32768 cost 275 ms of a 1037-ms call here;
// benchmark.cpp:181-266 calls MPF() once per matrix of a file.  mpf_trim() gives the memory back.
static int ensure_host_copies(mpf_ctx *c, int64_t N) {
    const int64_t abytes = N * N * (int64_t)sizeof(double), pbytes = N * (int64_t)sizeof(int32_t); // 64-bit: the reference overflows int here (MPF.cu:81)
    if (c->host_A_cap < abytes) {
        if (c->host_A) (void)hipFree(c->host_A);
        c->host_A = nullptr; c->host_A_cap = 0;
        MPF_HIP_TRY(c, hipMalloc((void **)&c->host_A, (size_t)abytes));
        c->host_A_cap = abytes;
    }
    if (c->host_P_cap < pbytes) {
        if (c->host_P) (void)hipFree(c->host_P);
        c->host_P = nullptr; c->host_P_cap = 0;
        MPF_HIP_TRY(c, hipMalloc((void **)&c->host_P, (size_t)pbytes));
        c->host_P_cap = pbytes;
    }
    return 0;
}

int mpf_trim(mpf_ctx *c) {
    if (!c) return -1;
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
    auto drop = [](auto *&p) { if (p) (void)hipFree(p); p = nullptr; };
    drop(c->host_A); drop(c->host_P); drop(c->host_A0); c->host_A_cap = c->host_P_cap = c->host_A0_cap = 0;
    sink_trim(c);
    feed_trim(c);
    drop(c->r64); drop(c->rm_tmp); drop(c->rm_lt); c->r64_cap = c->rm_tmp_cap = c->rm_lt_cap = 0; c->r64_n = 0;
    drop(c->w32); c->w32_n = 0;
    return 0;
}

int mpf_factor_host(mpf_ctx *c, double *A_host, int64_t N, int32_t nb, int32_t *ipiv_host, const mpf_opts *opts) {
    if (!c || !A_host || !ipiv_host) return -1;
    if (N <= 0 || nb <= 0) return fail(c, -1, "mpf_factor: N and panel width must be positive");
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    { const int e = ensure_host_copies(c, N); if (e) return e; }
    double *dA = c->host_A;
    int32_t *dP = c->host_P;
    const size_t bytes = (size_t)N * (size_t)N * sizeof(double), pbytes = (size_t)N * sizeof(int32_t);
    // Block rows leave while the factorization runs (rowsink.hip) when the schedule is one that reports them -- the look-ahead
    // schedules of the fp64 mode, which is what MPF() runs.  The caller's buffer is then partly results before the call knows that it
    // succeeded, so the matrix as uploaded is kept on the device (a device-to-device copy, ~4 ms at N = 32768) for the one failure
    // the call recovers from by itself: a pivot kernel whose workgroups were not all resident (-4) -> once more on the generic path.
    const bool want_sink = c->tune.host_sink && N >= c->tune.host_sink_min_n && c->pstream != nullptr && !(opts && (opts->sync_timing || opts->no_lookahead)) &&
                           !(opts && opts->trailing != MPF_TRAIL_FP64) && !(opts && opts->pivot_path == 1);
    bool armed = false;
    if (want_sink) {
        if (c->host_A0_cap < (int64_t)(bytes + pbytes)) {
            if (c->host_A0) (void)hipFree(c->host_A0);
            c->host_A0 = nullptr; c->host_A0_cap = 0;
            if (hipMalloc((void **)&c->host_A0, bytes + pbytes) == hipSuccess) c->host_A0_cap = (int64_t)(bytes + pbytes);
            else (void)hipGetLastError();   // (no room for the snapshot: the plain way)
        }
        if (c->host_A0) {
            const int e = sink_attach(c, A_host, N, nb);
            if (e < 0) return e;
            armed = e == 0;
        }
    }
    // The way up (round 5): only the first part of the matrix goes up before the factorization starts; the rest follows in column
    // segments while the first panels are factored on what is there (LatePlan, factor_lookahead_rm; rowsink.hip for the transport).
    // Planned only where that schedule is the one that will run (fp64 mode, N >= fp64_rowmajor_min_n, panels the LDS pivot kernel takes).
    LatePlan plan;
    const int64_t npan = (N + nb - 1) / nb;
    int parts = c->tune.host_late_parts;
    if (!want_sink || !c->tune.fp64_rowmajor || N < c->tune.fp64_rowmajor_min_n || N < c->tune.host_late_min_n || npan < 32 || c->tune.no_lookahead || safe_pivots(c) ||
        !hgetf2_lds_eligible(c, (int)N, (int)nb) || c->tune.superpanel_fp64 > 1 || (opts && opts->superpanel > 1)) parts = 0;
    if (parts > 0 && !c->late_flags && hipHostMalloc((void **)&c->late_flags, 64) != hipSuccess) { (void)hipGetLastError(); parts = 0; }
    if (parts > 0 && mpf_ensure_rowmajor_copy(c, N, N, nb) != 0) parts = 0;
    if (parts > 0) {
        // first part: host_first_pct of the columns; the late segments share the rest equally; when a segment is due is decided below,
        // once the first part's upload has been timed
        const int64_t first = ((N * c->tune.host_first_pct / 100 + nb - 1) / nb) * nb;
        plan.nseg = parts;
        for (int i = 0; i < parts; ++i) {
            int64_t c0 = first + ((N - first) * i / parts / nb) * nb;
            if (c0 < 4 * (int64_t)nb) c0 = 4 * (int64_t)nb;
            plan.c0[i] = c0;
            plan.q[i] = 1;
            if (i && plan.c0[i] <= plan.c0[i - 1]) { plan.nseg = 0; break; }
        }
        if (plan.nseg && plan.c0[0] >= N) plan.nseg = 0;
        plan.flags = c->late_flags;
        plan.seq = ++c->late_seq;
        plan.snapshot = armed ? c->host_A0 : nullptr;
    }
    const int64_t up_cols = plan.nseg ? plan.c0[0] : N;
    const size_t up_bytes = (size_t)up_cols * (size_t)N * sizeof(double);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemsetAsync(&c->ws->flags[1], 0, sizeof(int), c->stream);
    hipEventRecord(e0, c->stream);
    const auto t_up0 = std::chrono::steady_clock::now();
    hipMemcpyAsync(dA, A_host, up_bytes, hipMemcpyHostToDevice, c->stream);   // (from pageable memory: returns when the copy is done)
    const double up_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up0).count();
    hipMemcpyAsync(dP, ipiv_host, pbytes, hipMemcpyHostToDevice, c->stream);
    hipEventRecord(e1, c->stream);
    if (plan.nseg) {
        // When is a segment due?  As soon as it is there -- earlier, the main stream waits for the link; later, more panels are replayed on
        // it one by one with their small launches in the open.  Arrival: the link's rate as just measured on the first part (the late
        // segments go through host threads and pinned buffers: not faster than that, 45 GB/s assumed at most).  The device's progress: the
        // updates' flops on the columns that are there at 58 TFLOP/s, a panel never faster than its pivot chain (2.4 us per column);
        // host_late_q_pct scales the estimate (100 = as computed; the default leans late: waiting costs more than replaying).
        double gbps = up_ms > 0 ? (double)up_bytes / up_ms / 1e6 : 45.0;
        if (gbps > 45.0) gbps = 45.0;
        if (gbps < 5.0) gbps = 5.0;
        const double fl_per_ms = 58e9, chain_ms = nb * 2.4e-3;
        double t = 0, arrive = 0;
        int64_t k = 0, ce = plan.c0[0];
        for (int i = 0; i < plan.nseg; ++i) {
            const int64_t hi = i + 1 < plan.nseg ? plan.c0[i + 1] : N, w = hi - plan.c0[i];
            arrive += (double)w * N * 8 / gbps / 1e6 * (c->tune.host_late_q_pct / 100.0);
            while (t < arrive && (k / nb + 3) * (int64_t)nb <= ce) {   // panel k on the columns [.., ce)
                const double rows = (double)(N - k - nb), cols = (double)(ce - k - nb);
                const double upd = 2.0 * rows * cols * nb / fl_per_ms;
                t += upd > chain_ms ? upd : chain_ms;
                k += nb;
            }
            plan.q[i] = (int)(k / nb) > 0 ? (int)(k / nb) : 1;
            for (int64_t kj = 0; kj < k; kj += nb) t += 2.0 * (double)(N - kj - nb) * (double)w * nb / fl_per_ms;   // the replay
            if (t < arrive) t = arrive;
            ce = hi;
        }
        if (c->tune.sink_trace) {
            fprintf(stderr, "late plan: first part %lld columns up in %.1f ms (%.1f GB/s)", (long long)plan.c0[0], up_ms, up_ms > 0 ? (double)up_bytes / up_ms / 1e6 : 0.0);
            for (int i = 0; i < plan.nseg; ++i) fprintf(stderr, "; segment from column %lld due at panel %d", (long long)plan.c0[i], plan.q[i]);
            fprintf(stderr, "\n");
        }
    }
    if (armed) {
        hipMemcpyAsync(c->host_A0, dA, up_bytes, hipMemcpyDeviceToDevice, c->stream);
        hipMemcpyAsync((char *)c->host_A0 + bytes, dP, pbytes, hipMemcpyDeviceToDevice, c->stream);
    }
    if (plan.nseg) {
        const int e = feed_start(c, A_host, dA, N, &plan);
        if (e) {   // (no upload threads to be had: the rest of the matrix goes up here and now, as without a plan)
            hipMemcpyAsync(dA + up_cols * N, A_host + up_cols * N, bytes - up_bytes, hipMemcpyHostToDevice, c->stream);
            if (armed) hipMemcpyAsync(c->host_A0 + up_cols * N, dA + up_cols * N, bytes - up_bytes, hipMemcpyDeviceToDevice, c->stream);
            plan.nseg = 0;
        } else c->late = &plan;
    }
    int rc = mpf_factor_dev(c, dA, N, N, nb, dP, opts);
    c->late = nullptr;
    {   // the upload thread has long finished when the factorization has; a plan the schedule did not take was waited for inside mpf_factor_dev
        const int e = feed_finish(c);
        if (e && rc >= 0) rc = e;
        int gave = 0;
        if (plan.nseg && hipMemcpy(&gave, &c->ws->flags[1], sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && gave && rc >= 0)
            rc = fail(c, -2, "mpf_factor_host: the factorization gave up waiting for a late column segment");
        if (armed && plan.nseg)      // segments the schedule did not reach (it failed before): their snapshot is taken now, from the untouched columns
            for (int i = 0; i < plan.nseg; ++i)
                if (!plan.snapped[i]) {
                    const int64_t lo = plan.c0[i], hi = i + 1 < plan.nseg ? plan.c0[i + 1] : N;
                    hipMemcpyAsync(c->host_A0 + lo * N, dA + lo * N, (size_t)(hi - lo) * N * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
                }
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ms_h2d = ms;
    const auto t_done = std::chrono::steady_clock::now();
    int sent = 0;
    const int npanels = (int)((N + nb - 1) / nb);
    int sk = armed ? sink_finish(c, &sent) : 1;   // 0: a schedule took the sink (sent block rows are in A_host), 1: nobody did, < 0: HIP error
    if (sk < 0) rc = sk;
    if (rc == -4 && sk == 0 && sent > 0) {
        // (the reference has no such failure mode: MPF.cu:126-140 is a cooperative launch)
        const std::string why = c->err;
        std::cerr << "mpf_factor_host: " << why << " -- " << sent << " block rows had left already; repeating on the generic pivot path from the uploaded matrix" << std::endl;
        hipMemcpyAsync(dA, c->host_A0, bytes, hipMemcpyDeviceToDevice, c->stream);
        hipMemcpyAsync(dP, (char *)c->host_A0 + bytes, pbytes, hipMemcpyDeviceToDevice, c->stream);
        mpf_opts o2{};
        if (opts) o2 = *opts;
        o2.pivot_path = 1;
        rc = mpf_factor_dev(c, dA, N, N, nb, dP, &o2);
        if (rc == -4) c->err = why;
        sent = 0;
        sk = 1;        // (this factorization did not go through the sink: its matrix goes home in one piece)
    }
    c->stats.ms_h2d = ms_h2d;
    c->stats.host_late_segments = (rc >= 0 && plan.taken) ? plan.nseg : 0;
    if (rc >= 0) {
        hipError_t se = hipSuccess;
        if (sk == 0 && sent == npanels) {   // every block row is home
            hipMemcpyAsync(ipiv_host, dP, pbytes, hipMemcpyDeviceToHost, c->stream);
            se = hipStreamSynchronize(c->stream);
            c->stats.host_rows_streamed = npanels;
        } else {
            // (a schedule that gave its rows to the sink has skipped its deferred left-hand interchanges: if the sink stopped early without
            //  an error of the factorization -- it cannot, short of a HIP error, but the matrix must never go home half-permuted -- they are due now)
            if (sk == 0 && sent < npanels) (void)launch_lazy_left_swaps(c, dA, N, N, nb, npanels, c->lists);
            hipMemcpyAsync(A_host, dA, bytes, hipMemcpyDeviceToHost, c->stream);
            hipMemcpyAsync(ipiv_host, dP, pbytes, hipMemcpyDeviceToHost, c->stream);
            se = hipStreamSynchronize(c->stream);
        }
        // what the call still spent on the way home after the factorization's last kernel (wall clock)
        c->stats.ms_d2h = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_done).count();
        if (se != hipSuccess) rc = fail(c, -2, std::string("D2H failed: ") + hipGetErrorString(se));
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    // rc < 0: with the sink (default for N >= host_sink_min_n in the fp64 mode) the caller's matrix may hold finished block rows
    // beside untouched ones; without it nothing has been copied back
    return rc;
}

// ---- refinement solve ----------------------------------------------------------------------------
// rows of the tallest panel the LDS pivot kernel takes (in either form) beside `waiters` workgroups that wait for it (0: alone);
// waiters < 0: beside the pipelined chain's gated interchange kernel on a panel of -waiters columns
int64_t mpf_hgetf2_capacity_rows(mpf_ctx *c, int32_t waiters, int32_t form) {
    if (!c || form < 0 || form > 2) return -1;
    if (hipSetDevice(c->device) != hipSuccess) return -2;
    return (int64_t)hgetf2_capacity_rows(c, waiters < 0 ? laswp_gated_grid(-waiters) : waiters, form);
}

int mpf_ensure_solve_buf(mpf_ctx *c, int64_t n) {
    if (c->solve_n >= n && c->solve_buf) return 0;
    if (c->solve_buf) hipFree(c->solve_buf);
    if (c->perm_buf) hipFree(c->perm_buf);
    if (c->trsv_inv) hipFree(c->trsv_inv);
    if (c->trsv_inv256) hipFree(c->trsv_inv256);
    if (c->trsv_cnt) hipFree(c->trsv_cnt);
    c->solve_buf = nullptr; c->perm_buf = nullptr; c->trsv_inv = nullptr; c->trsv_inv256 = nullptr; c->trsv_cnt = nullptr; c->solve_n = 0;
    MPF_HIP_TRY(c, hipMalloc((void **)&c->solve_buf, (size_t)(4 * n + 8) * sizeof(double)));
    MPF_HIP_TRY(c, hipMalloc((void **)&c->perm_buf, (size_t)n * sizeof(int32_t)));
    MPF_HIP_TRY(c, hipMalloc((void **)&c->trsv_inv, (size_t)(2 * ((n + 63) / 64)) * 64 * 64 * sizeof(double)));
    MPF_HIP_TRY(c, hipMalloc((void **)&c->trsv_inv256, (size_t)(2 * ((n + 255) / 256)) * 256 * 256 * sizeof(double)));
    MPF_HIP_TRY(c, hipMalloc((void **)&c->trsv_cnt, (size_t)(2 * ((n + 255) / 256) + 2) * sizeof(int)));
    c->solve_n = n;
    return 0;
}

int mpf_gesv(mpf_ctx *c, const double *d_A, int64_t lda, int64_t N, int32_t nb, double *d_work, int32_t *d_ipiv,
             const double *d_b, double *d_x, int32_t max_iter, double tol, int32_t try_fp16, mpf_gesv_stats *stats) {
    if (!c || !d_A || !d_work || !d_ipiv || !d_b || !d_x) return -1;
    if (N <= 0 || lda < N) return fail(c, -1, "gesv: bad N / lda");
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    mpf_gesv_stats gs{};
    std::vector<int32_t> ident((size_t)N);
    for (int64_t i = 0; i < N; ++i) ident[(size_t)i] = (int32_t)(i + 1); // benchmark.cpp:215-217
    auto t0 = std::chrono::steady_clock::now();
    auto attempt = [&](int mode, double &ms_fact, double &ms_ir, mpf_ir_stats &ir) -> int {
        MPF_HIP_TRY(c, hipMemcpy2DAsync(d_work, (size_t)N * 8, d_A, (size_t)lda * 8, (size_t)N * 8, (size_t)N,
                                        hipMemcpyDeviceToDevice, c->stream));
        MPF_HIP_TRY(c, hipMemcpyAsync(d_ipiv, ident.data(), (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        mpf_opts o{};
        o.trailing = mode;
        int rc = mpf_factor_dev(c, d_work, N, N, nb, d_ipiv, &o);
        if (rc < 0) return rc;
        gs.info = rc;
        ms_fact = c->stats.ms_total;
        rc = mpf_solve_ir(c, d_A, lda, d_work, N, d_ipiv, N, d_b, d_x, max_iter, tol, &ir);
        ms_ir = ir.ms_total;
        return rc;
    };
    int rc = 0;
    bool done = false;
    if (try_fp16) {
        rc = attempt(try_fp16 == 2 ? MPF_TRAIL_FP16X3 : MPF_TRAIL_FP16, gs.ms_factor_fp16, gs.ms_ir_fp16, gs.ir_fp16);
        if (rc < 0) return rc;
        if (gs.ir_fp16.converged) { gs.path = 1; gs.ir_final = gs.ir_fp16; done = true; }
        else if (try_fp16 == 3) {
            // plain refinement does not contract with these factors: keep them as the preconditioner of GMRES (GMRES-IR)
            // before paying for a second, fp64 factorization
            // ... but only for as long as that second factorization would take (round 4): 2/3 N^3 flops at ~50 TFLOP/s, and never less
            // than the fp16-mode factorization just timed (small matrices are bound by the pivot chain in every mode).  On the
            // reference generator's matrix at N = 32768 (195 inner iterations, 1.9 s, against 0.45 s) GMRES-IR then stops after
            // ~0.5 s and the fp64 path takes over; where the fp16 factors precondition well it converges inside the budget.
            // The rate: option gesv_fp64_tflops, else what this context's last fp64-mode factorization (N >= 8192) measured, else 50
            // (one MI355X at N = 32768) -- ADVICE r4: a constant misprices a slower or shared device.
            const double rate = c->tune.gesv_fp64_tflops > 0 ? (double)c->tune.gesv_fp64_tflops : (c->fp64_rate_tflops > 0 ? c->fp64_rate_tflops : 50.0);
            const double est_fp64_ms = 2.0 / 3.0 * (double)N * (double)N * (double)N / (rate * 1e12) * 1e3;
            c->gmres_budget_ms = est_fp64_ms > gs.ms_factor_fp16 ? est_fp64_ms : gs.ms_factor_fp16;
            gs.gmres_budget_ms = c->gmres_budget_ms;
            mpf_gmres_stats gm{};
            rc = mpf_solve_gmres_ir(c, d_A, lda, d_work, N, d_ipiv, N, d_b, d_x, max_iter, 30, tol, &gm);
            c->gmres_budget_ms = 0;
            if (rc < 0) return rc;
            gs.gmres_budget_expired = gm.budget_expired;
            gs.ms_ir_fp16 += gm.ms_total;
            if (gm.converged) {
                gs.path = 3;
                gs.ir_final = mpf_ir_stats{};
                gs.ir_final.converged = 1; gs.ir_final.iterations = gm.inner_iterations; gs.ir_final.rel_residual = gm.rel_residual;
                for (int i = 0; i < 32; ++i) gs.ir_final.history[i] = gm.history[i];
                gs.ir_final.ms_total = gm.ms_total;
                done = true;
            }
        }
    }
    if (!done) {
        rc = attempt(MPF_TRAIL_FP64, gs.ms_factor_fp64, gs.ms_ir_fp64, gs.ir_final);
        if (rc < 0) return rc;
        gs.path = 2;
    }
    gs.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = gs;
    return gs.ir_final.converged ? 0 : 1;
}

} // extern "C"

// ---- the reference's own symbol (MPF.h:3, C++ linkage) -------------------------------------------
// The reference builds and tears down everything inside every call (MPF.cu:69-97,250-255).  Here the context -- streams,
// workspace, the device copy of the matrix, the row-major working copy -- is created by the first call and lives until the
// process exits: benchmark.cpp:181-266 calls MPF() once per matrix in a loop, and the second call pays H2D + factor + D2H only.
// Calls are serialised (the reference is not re-entrant either: file-scope scratch, hgetf2_kernel.cu:6-7).
namespace {
std::mutex g_mpf_mu;
mpf_ctx *g_mpf_ctx = nullptr;
struct MpfAtExit { ~MpfAtExit() { if (g_mpf_ctx) { mpf_destroy(g_mpf_ctx); g_mpf_ctx = nullptr; } } } g_mpf_at_exit;
}  // namespace

void MPF(double *A, int N, int r, int *IPIV) {
    std::lock_guard<std::mutex> lk(g_mpf_mu);
    if (!g_mpf_ctx && mpf_create(&g_mpf_ctx, 0) != 0) { // reference MPF.cu:69-75: message on stderr, buffers untouched
        g_mpf_ctx = nullptr;
        std::cerr << (g_noctx_err.empty() ? "No HIP devices available." : g_noctx_err) << std::endl;
        return;
    }
    mpf_ctx *c = g_mpf_ctx;
    mpf_opts o{};
    o.verbose = c->tune.verbose; // the reference prints one line per panel (MPF.cu:137); off by default here (MPF_VERBOSE=1)
    int rc = mpf_factor_host(c, A, N, r, IPIV, &o);
    if (rc == -4) {
        // The LDS pivot kernel's workgroups were not all resident (something else holds CUs): the reference has no such failure
        // mode (MPF.cu:126-140 is a cooperative launch).  Nothing has been copied back, so the caller's buffers are intact: run
        // the call again on the generic pivot path, which never waits for another workgroup.
        std::cerr << "MPF: " << mpf_last_error(c) << " -- retrying on the generic pivot path" << std::endl;
        o.pivot_path = 1;
        rc = mpf_factor_host(c, A, N, r, IPIV, &o);
    }
    if (rc < 0) std::cout << "MPF error: " << mpf_last_error(c) << std::endl; // reference prints and continues (:134-138)
}

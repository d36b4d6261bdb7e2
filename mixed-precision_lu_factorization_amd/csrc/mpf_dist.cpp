// Multi-GPU MPF behind the C ABI: 1-D block-cyclic column layout, one process per GPU, ONE broadcast per panel.
//
// The reference is single-device (MPF.cu:77); this partition is the build's extension (BASELINE north_star
// "1-D block-column layout and RCCL broadcast of the factored panel over xGMI").  A column block of nb columns lives
// entirely on one rank (global block b on rank b % world, local index b / world), so the fp16 pivot panel, its pivot
// search and the fp64 no-pivot panel are local to the owner -- there is no cross-GPU argmax.  Per panel the only exchange
// is one broadcast owner -> all of {factored panel (rows k..N x nb, fp64), nb pivots (int32), the panel's moved-row list};
// every rank then interchanges / TRSMs / GEMMs the columns it owns.  Per element the arithmetic is that of the 1-GPU path
// (the partition only changes WHO computes a column block): IPIV and LU are bit-identical to mpf_factor_dev.
//
// Schedule = the look-ahead schedule of mpf_host.cpp with column indices translated: the owner of panel b+1 updates that
// block first ("strip"), runs its chain and posts the broadcast on the side stream P while every rank finishes update b on
// the main stream S.  Columns LEFT of a panel are interchanged once at the end (composite maps, laswp.hip).
//
// The broadcast is a callback (mpf_dist.bcast) so that any transport can carry it; the built-in one is RCCL
// (mpf_rccl_init: ncclBroadcast / ncclAllReduce on the context's communicator, resolved with dlopen so that single-GPU
// users never load RCCL).  Tests drive the same loop through a gloo-backed callback with several ranks on one GPU.
#include "mpf_internal.h"
#include <dlfcn.h>
#include <climits>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <algorithm>

// ---------------------------------------------------------------------------------------------------------------------
// RCCL through dlopen
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct NcclId { char internal[128]; };
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(void **, int, NcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;   // optional: the chain of the distributed triangular solves
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;       // optional (the bench line's record that RCCL saw N ranks)
    int (*CommUserRank)(void *, int *) = nullptr;
};
Rccl *rccl_load(std::string &err) {
    static Rccl r;
    static bool tried = false;
    if (r.h) return &r;
    if (tried) { err = "librccl not loadable"; return nullptr; }
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.h) break; } // the copy torch already loaded, if any
    if (!r.h) for (const char *n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
    if (!r.h) { err = std::string("dlopen(librccl) failed: ") + dlerror(); return nullptr; }
#define RSYM(field, name) r.field = (decltype(r.field))dlsym(r.h, name); if (!r.field) { err = std::string("librccl lacks ") + name; r.h = nullptr; return nullptr; }
    RSYM(GetUniqueId, "ncclGetUniqueId") RSYM(CommInitRank, "ncclCommInitRank") RSYM(CommDestroy, "ncclCommDestroy")
    RSYM(Broadcast, "ncclBroadcast") RSYM(AllReduce, "ncclAllReduce") RSYM(GetErrorString, "ncclGetErrorString")
    RSYM(GetVersion, "ncclGetVersion")
#undef RSYM
    r.Send = (decltype(r.Send))dlsym(r.h, "ncclSend");
    r.Recv = (decltype(r.Recv))dlsym(r.h, "ncclRecv");
    r.CommCount = (decltype(r.CommCount))dlsym(r.h, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.h, "ncclCommUserRank");
    return &r;
}
constexpr int NCCL_CHAR = 0, NCCL_DOUBLE = 8, NCCL_SUM = 0;

int rccl_bcast(void *user, void *d_buf, int64_t bytes, int32_t root, void *stream) {
    mpf_ctx *c = (mpf_ctx *)user;
    std::string err;
    Rccl *r = rccl_load(err);
    if (!r || !c->rccl_comm) { c->err = "RCCL communicator not initialised (mpf_rccl_init)"; return -5; }
    const int rc = r->Broadcast(d_buf, d_buf, (size_t)bytes, NCCL_CHAR, root, c->rccl_comm, (hipStream_t)stream);
    if (rc != 0) { c->err = std::string("ncclBroadcast: ") + r->GetErrorString(rc); return -5; }
    c->rccl_bcast_calls++; c->rccl_bcast_bytes += bytes;
    return 0;
}
int rccl_allreduce(void *user, double *d_buf, int64_t count, void *stream) {
    mpf_ctx *c = (mpf_ctx *)user;
    std::string err;
    Rccl *r = rccl_load(err);
    if (!r || !c->rccl_comm) { c->err = "RCCL communicator not initialised (mpf_rccl_init)"; return -5; }
    const int rc = r->AllReduce(d_buf, d_buf, (size_t)count, NCCL_DOUBLE, NCCL_SUM, c->rccl_comm, (hipStream_t)stream);
    if (rc != 0) { c->err = std::string("ncclAllReduce: ") + r->GetErrorString(rc); return -5; }
    c->rccl_allreduce_calls++;
    return 0;
}

int rccl_p2p(void *user, void *d_buf, int64_t bytes, int32_t peer, int32_t send, void *stream) {
    mpf_ctx *c = (mpf_ctx *)user;
    std::string err;
    Rccl *r = rccl_load(err);
    if (!r || !c->rccl_comm || !r->Send || !r->Recv) { c->err = "RCCL point-to-point not available"; return -5; }
    const int rc = send ? r->Send(d_buf, (size_t)bytes, NCCL_CHAR, peer, c->rccl_comm, (hipStream_t)stream)
                        : r->Recv(d_buf, (size_t)bytes, NCCL_CHAR, peer, c->rccl_comm, (hipStream_t)stream);
    if (rc != 0) { c->err = std::string(send ? "ncclSend: " : "ncclRecv: ") + r->GetErrorString(rc); return -5; }
    c->rccl_p2p_calls++; c->rccl_p2p_bytes += bytes;
    return 0;
}
bool rccl_has_p2p() { std::string err; Rccl *r = rccl_load(err); return r && r->Send && r->Recv; }

// ---- layout ----------------------------------------------------------------------------------------------------------
struct Layout {
    int64_t N; int nb, rank, world, nblocks;
    Layout(int64_t N_, int nb_, int rank_, int world_) : N(N_), nb(nb_), rank(rank_), world(world_), nblocks((int)((N_ + nb_ - 1) / nb_)) {}
    int owner(int b) const { return b % world; }
    bool mine(int b) const { return b % world == rank; }
    int width(int b) const { const int64_t w = N - (int64_t)b * nb; return (int)(w < nb ? w : nb); }
    int64_t lcol(int b) const { return (int64_t)(b / world) * nb; }           // first local column of (owned) block b
    int64_t local_cols() const { int64_t s = 0; for (int b = rank; b < nblocks; b += world) s += width(b); return s; }
    int64_t first_local_col_after(int b) const {                                // local columns whose global block is > b
        const int64_t cnt = b < rank ? 0 : (int64_t)(b - rank) / world + 1;
        const int64_t c0 = cnt * nb, lc = local_cols();
        return c0 < lc ? c0 : lc;
    }
    bool live(int b) const { return b < nblocks && N - (int64_t)b * nb > 1; }
};
// Panel message: the factored panel (rows k..N) with leading dimension ldp = rows + PANEL_PAD -- the padding at the bottom of
// column 32 q holds the 32 pivots (int32, global, 1-based) of sub-panel q, so a 32-column instalment of the message is ONE
// contiguous range that carries its own pivots -- then the panel's moved-row list.
constexpr int PANEL_PAD = 16;   // doubles = 128 bytes = 32 int32 pivots
size_t panel_buf_bytes(int64_t N, int nb) { return (size_t)(N + PANEL_PAD) * nb * 8 + sizeof(MovedList); }

int ensure_dist_bufs(mpf_ctx *c, int64_t N, int nb) {
    const size_t need = panel_buf_bytes(N, nb);
    if (c->dist_buf[0] && c->dist_buf_cap >= need) return 0;
    for (auto *&b : c->dist_buf) { if (b) hipFree(b); b = nullptr; }
    c->dist_buf_cap = 0;
    for (auto *&b : c->dist_buf) MPF_HIP_TRY(c, hipMalloc((void **)&b, need));
    c->dist_buf_cap = need;
    return 0;
}

// Panels wider than 256 columns (the LDS pivot kernel and the moved-row lists stop there): the reference's own order, panel by
// panel on one stream, with the generic (global-memory) pivot kernels and the sequential interchange of ALL local columns
// (MPF.cu:145-162) -- no look-ahead, no deferred left-hand side.  Message = the factored panel (leading dimension = its rows)
// followed by its pivots.  Per element the operations are those of factor_generic (mpf_host.cpp): identical bits.
int factor_dist_wide(mpf_ctx *c, double *d_Aloc, int64_t ldloc, int64_t N, int32_t nb, int32_t *d_ipiv, const mpf_dist *dist,
                     const mpf_opts &o, const Layout &L, mpf_bcast_fn bcast_fn, void *user, mpf_stats &st) {
    EvPool ev(c);
    ev.keep = &st.ms_gemm;
    hipStream_t S = c->stream;
    const int64_t lcols = L.local_cols();
    const bool split = o.trailing == MPF_TRAIL_FP16X3, f64 = o.trailing == MPF_TRAIL_FP64;
    int rc = 0;
    if (!f64) { rc = mpf_ensure_h_images(c, N, nb < 8 * HP_MAXCOLS ? nb : 8 * HP_MAXCOLS, false); if (rc) return rc; }
    for (int b = 0; L.live(b) && rc == 0; ++b) {
        const int64_t k = (int64_t)b * nb;
        const int pc = L.width(b), pr = (int)(N - k);
        double *buf = c->dist_buf[0];
        int *piv = (int *)(buf + (size_t)pr * pc);
        if (L.mine(b)) {
            double *Ap = d_Aloc + L.lcol(b) * ldloc + k;
            st.pivot_path = 1;
            rc = ev.timed(st.ms_hpanel, S, [&] { return launch_hgetf2_generic(c, Ap, ldloc, nullptr, 0, pr, pc, (int)k, d_ipiv + k, nullptr, 0); });
            if (!rc) rc = ev.timed(st.ms_laswp, S, [&] { return launch_laswp_seq(c, d_Aloc, ldloc, lcols, (int)k, pc, d_ipiv + k, N); });
            if (!rc) rc = ev.timed(st.ms_dpanel, S, [&] { return launch_dgetf2_npv(c, Ap, ldloc, pr, pc, o.fused_panel, (int)k); });
            if (rc) break;
            MPF_HIP_TRY(c, hipMemcpy2DAsync(buf, (size_t)pr * 8, Ap, (size_t)ldloc * 8, (size_t)pr * 8, (size_t)pc, hipMemcpyDeviceToDevice, S));
            MPF_HIP_TRY(c, hipMemcpyAsync(piv, d_ipiv + k, (size_t)pc * 4, hipMemcpyDeviceToDevice, S));
            st.panels++;
        }
        if (L.world > 1) {
            const int e = bcast_fn(user, buf, (int64_t)((size_t)pr * pc * 8 + (size_t)pc * 4), L.owner(b), (void *)S);
            if (e) return e < 0 ? e : -5;
        }
        if (!L.mine(b)) {
            MPF_HIP_TRY(c, hipMemcpyAsync(d_ipiv + k, piv, (size_t)pc * 4, hipMemcpyDeviceToDevice, S));
            if (lcols > 0) rc = ev.timed(st.ms_laswp, S, [&] { return launch_laswp_seq(c, d_Aloc, ldloc, lcols, (int)k, pc, d_ipiv + k, N); });
            if (rc) break;
        }
        const int64_t c0 = L.first_local_col_after(b), nc = lcols - c0, m = (int64_t)pr - pc;
        if (k + pc < N && nc > 0) {
            double *U12 = d_Aloc + c0 * ldloc + k;
            rc = ev.timed(st.ms_trsm, S, [&] { return launch_dtrsm_llnu(c, pc, nc, buf, pr, U12, ldloc); });
            if (!rc && m > 0) {
                rc = ev.timed(st.ms_gemm, S, [&] {
                    if (f64) return launch_dgemm_minus(c, m, nc, pc, buf + pc, pr, U12, ldloc, U12 + pc, ldloc);
                    int e = 0;
                    const int kcmax = c->h_kmax;                 // K capacity of the fp16 operand images
                    for (int k0 = 0; k0 < pc && !e; k0 += kcmax) {
                        const int kc = (pc - k0) < kcmax ? (pc - k0) : kcmax;
                        e = launch_cvt_l21(c, buf + pc + (int64_t)k0 * pr, pr, m, kc, split);
                        if (!e) e = launch_hgemm_minus(c, m, nc, kc, U12 + k0, ldloc, U12 + pc, ldloc, split);
                    }
                    return e; });
                count_gemm(st, o, m, nc, pc);
            }
        }
        if (o.verbose) printf("[rank %d] panel %d (k=%lld) owner %d (wide panels: generic schedule)\n", L.rank, b, (long long)k, L.owner(b));
    }
    const hipError_t se = hipStreamSynchronize(S);
    if (!rc && se != hipSuccess) { c->err = std::string("distributed factorization failed: ") + hipGetErrorString(se); return -2; }
    ev.collect();
    (void)dist;
    return rc;
}
} // namespace

extern "C" {

int mpf_rccl_unique_id(void *out128) {
    if (!out128) return -1;
    std::string err;
    Rccl *r = rccl_load(err);
    if (!r) return -5;
    NcclId id;
    if (r->GetUniqueId(&id) != 0) return -5;
    memcpy(out128, id.internal, 128);
    return 0;
}

int mpf_rccl_init(mpf_ctx *c, const void *id128, int32_t rank, int32_t world) {
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return -1;
    std::string err;
    Rccl *r = rccl_load(err);
    if (!r) { c->err = err; return -5; }
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    if (c->rccl_comm) { r->CommDestroy(c->rccl_comm); c->rccl_comm = nullptr; }
    NcclId id;
    memcpy(id.internal, id128, 128);
    const int rc = r->CommInitRank(&c->rccl_comm, world, id, rank);
    if (rc != 0) { c->rccl_comm = nullptr; c->err = std::string("ncclCommInitRank: ") + r->GetErrorString(rc); return -5; }
    c->rccl_rank = rank; c->rccl_world = world;
    c->rccl_bcast_calls = c->rccl_bcast_bytes = c->rccl_allreduce_calls = c->rccl_p2p_calls = c->rccl_p2p_bytes = 0;
    return 0;
}

int mpf_rccl_destroy(mpf_ctx *c) {
    if (!c) return -1;
    if (c->rccl_comm) {
        std::string err;
        Rccl *r = rccl_load(err);
        if (r) r->CommDestroy(c->rccl_comm);
        c->rccl_comm = nullptr;
    }
    return 0;
}

// one broadcast and one all-reduce of a small device buffer on the context's communicator: checks the whole RCCL call path
// (symbols, communicator, stream ordering) independently of the factorization; every rank of the communicator must call it
int mpf_rccl_selftest(mpf_ctx *c) {
    if (!c) return -1;
    if (!c->rccl_comm) { c->err = "RCCL communicator not initialised (mpf_rccl_init)"; return -5; }
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    double *d = nullptr, h[4] = {1.0 + c->rccl_rank, 2.0, 3.0, 4.0}, back[4] = {0, 0, 0, 0};
    MPF_HIP_TRY(c, hipMalloc((void **)&d, sizeof h));
    int rc = 0;
    if (hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = -2;
    if (!rc) rc = rccl_bcast(c, d, sizeof h, 0, c->stream);
    if (!rc) rc = rccl_allreduce(c, d, 4, c->stream);
    if (!rc && hipMemcpyAsync(back, d, sizeof h, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = -2;
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = -2;
    hipFree(d);
    if (rc) return rc;
    const double w = (double)c->rccl_world;
    if (back[0] != 1.0 * w || back[1] != 2.0 * w || back[3] != 4.0 * w) { c->err = "RCCL self-test: wrong values"; return -5; }
    return 0;
}

// What the context's communicator is and what has gone over it since mpf_rccl_init (the multi-GPU bench line's `rccl` object: the
// record that RCCL itself saw N ranks, not only the launcher).  Counters are host-side counts of the calls this library issued.
int mpf_rccl_info(mpf_ctx *c, mpf_rccl_info_t *out) {
    if (!c || !out) return -1;
    memset(out, 0, sizeof *out);
    std::string err;
    Rccl *r = rccl_load(err);
    out->version = -1; out->comm_count = -1; out->comm_rank = -1;
    if (r) { int v = 0; if (r->GetVersion(&v) == 0) out->version = v; }
    if (r && c->rccl_comm) {
        int v = 0;
        if (r->CommCount && r->CommCount(c->rccl_comm, &v) == 0) out->comm_count = v;
        if (r->CommUserRank && r->CommUserRank(c->rccl_comm, &v) == 0) out->comm_rank = v;
    }
    out->has_comm = c->rccl_comm ? 1 : 0;
    out->has_p2p = rccl_has_p2p() ? 1 : 0;
    out->device = c->device;
    out->bcast_calls = c->rccl_bcast_calls; out->bcast_bytes = c->rccl_bcast_bytes;
    out->allreduce_calls = c->rccl_allreduce_calls;
    out->p2p_calls = c->rccl_p2p_calls; out->p2p_bytes = c->rccl_p2p_bytes;
    // link to every other visible device: 0 = none / unknown, else hipExtGetLinkTypeAndHopCount's link type (xGMI = 4 on ROCm) and hops
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess) {
        out->visible_devices = ndev;
        for (int d = 0; d < ndev && d < 16; ++d) {
            if (d == c->device) continue;
            uint32_t lt = 0, hops = 0;
            int can = 0;
            if (hipExtGetLinkTypeAndHopCount(c->device, d, &lt, &hops) == hipSuccess) { out->link_type[d] = (int32_t)lt; out->link_hops[d] = (int32_t)hops; }
            if (hipDeviceCanAccessPeer(&can, c->device, d) == hipSuccess) out->peer_access[d] = can;
        }
    }
    return 0;
}
// `reps` broadcasts of `bytes` from `root` on the context's communicator, timed with HIP events on the context's stream (every rank
// calls it): milliseconds per broadcast -- the cost of one panel message of that size, apart from the factorization
int mpf_rccl_bcast_probe(mpf_ctx *c, int64_t bytes, int32_t root, int32_t reps, double *ms_per_bcast) {
    if (!c || !ms_per_bcast || bytes <= 0 || reps <= 0) return -1;
    if (!c->rccl_comm) { c->err = "RCCL communicator not initialised (mpf_rccl_init)"; return -5; }
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    void *d = nullptr;
    MPF_HIP_TRY(c, hipMalloc(&d, (size_t)bytes));
    hipMemsetAsync(d, 0, (size_t)bytes, c->stream);
    const long long calls0 = c->rccl_bcast_calls, bytes0 = c->rccl_bcast_bytes;
    int rc = rccl_bcast(c, d, bytes, root, c->stream);   // warm-up (connection set-up)
    hipEventRecord(c->ev0, c->stream);
    for (int i = 0; i < reps && !rc; ++i) rc = rccl_bcast(c, d, bytes, root, c->stream);
    hipEventRecord(c->ev1, c->stream);
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = -2;
    float ms = 0;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    *ms_per_bcast = ms / reps;
    c->rccl_bcast_calls = calls0; c->rccl_bcast_bytes = bytes0;   // (the probe is not part of a factorization's traffic)
    hipFree(d);
    return rc;
}

int mpf_rccl_version(void) {
    std::string err;
    Rccl *r = rccl_load(err);
    int v = 0;
    if (!r || r->GetVersion(&v) != 0) return -5;
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// factorization
// ---------------------------------------------------------------------------------------------------------------------
int mpf_factor_dist(mpf_ctx *c, double *d_Aloc, int64_t ldloc, int64_t N, int32_t nb, int32_t *d_ipiv, const mpf_dist *dist,
                    const mpf_opts *opts) {
    if (!c || !d_ipiv || !dist) return -1;
    if (N <= 0 || nb <= 0) { c->err = "mpf_factor_dist: N and panel width must be positive"; return -1; }
    if (dist->world < 1 || dist->rank < 0 || dist->rank >= dist->world) { c->err = "mpf_factor_dist: bad rank / world"; return -1; }
    if (ldloc < N) { c->err = "mpf_factor_dist: ldloc < N"; return -1; }
    if (nb > 65535) { c->err = "mpf_factor_dist: panel width > 65535"; return -1; }
    if (N > INT_MAX / 2) { c->err = "mpf_factor_dist: N too large"; return -1; }
    mpf_opts o{};
    if (opts) o = *opts;
    if (o.trailing < MPF_TRAIL_FP64 || o.trailing > MPF_TRAIL_FP16X3) { c->err = "mpf_factor_dist: unknown trailing mode"; return -1; }
    // fp16 modes: the full-slab pivot kernel (a CU per workgroup) except where a panel leaves fewer than 72 CUs free -- the gated
    // interchange kernel of the pipelined chain (64 workgroups) waits for the pivot kernel's progress, each of its workgroups keeps
    // a pivot workgroup off its CU (measured: 192 pivot workgroups + 64 run, 208 + 64 never become resident), and a pivot workgroup
    // that finds no CU never starts: N = 53 248 and up gave up after the bounded spin (-4).  The column-window form (two per CU)
    // has the room.
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    // One rank owns every column: the local matrix IS the matrix, and the single-GPU driver's schedules (row-major working copy in
    // two column lanes for fp64, three lanes for the fp16 modes) are the ones to run -- the N = 1 point of a scaling curve is the
    // single-GPU number.  (Option dist_world1_loop = 1 keeps one rank in the distributed loop: tests of that loop.)
    if (dist->world == 1 && !c->tune.dist_world1_loop) {
        if (!d_Aloc) return -1;
        return mpf_factor_dev(c, d_Aloc, ldloc, N, nb, d_ipiv, &o);
    }
    mpf_bcast_fn bcast_fn = dist->bcast ? dist->bcast : rccl_bcast;
    void *user = dist->bcast ? dist->user : (void *)c;
    if (!dist->bcast && dist->world > 1 && (!c->rccl_comm || c->rccl_world != dist->world || c->rccl_rank != dist->rank)) {
        c->err = "mpf_factor_dist: no broadcast callback and no matching RCCL communicator (mpf_rccl_init)"; return -5;
    }
    const Layout L(N, nb, dist->rank, dist->world);
    const int64_t lcols = L.local_cols();
    if (lcols > 0 && !d_Aloc) return -1;
    const bool split = o.trailing == MPF_TRAIL_FP16X3, f64 = o.trailing == MPF_TRAIL_FP64;
    int rc = ensure_dist_bufs(c, N, nb);
    if (rc) return rc;
    if (nb > HP_MAXCOLS) {   // wide panels: the generic schedule (reference order, one stream)
        const int imax0 = INT_MAX;
        MPF_HIP_TRY(c, hipMemcpyAsync(&c->ws->info, &imax0, sizeof(int), hipMemcpyHostToDevice, c->stream));
        mpf_stats stw{};
        stw.n = N; stw.nb = nb; stw.superpanel = 1;
        MPF_HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
        rc = factor_dist_wide(c, d_Aloc, ldloc, N, nb, d_ipiv, dist, o, L, bcast_fn, user, stw);
        hipEventRecord(c->ev1, c->stream);
        const hipError_t sew = hipStreamSynchronize(c->stream);
        if (rc) return rc;
        if (sew != hipSuccess) { c->err = std::string("distributed factorization failed: ") + hipGetErrorString(sew); return -2; }
        float msw = 0;
        hipEventElapsedTime(&msw, c->ev0, c->ev1);
        stw.ms_total = msw;
        int infow = 0;
        MPF_HIP_TRY(c, hipMemcpy(&infow, &c->ws->info, sizeof(int), hipMemcpyDeviceToHost));
        stw.info = infow == INT_MAX ? 0 : infow;
        c->stats = stw;
        return stw.info;
    }
    // Two-level schedule of the fp16 modes (factor_superpanel in mpf_host.cpp, carried over to the block-cyclic layout): sb panels
    // form a super-panel; a panel updates only the INNER blocks (the super-panel's own and the next super-panel's first one) of
    // this rank, panel by panel, in fp64; this rank's FAR columns live in an fp32 row-major working copy and get one update with
    // K = sb * nb per super-panel, after the U block-row tasks.  What the single-GPU schedule reads from the matrix left of the
    // panel -- the super-panel's earlier panels -- every rank keeps in a store of its own, assembled from the panel messages.
    const bool force_generic_ = o.pivot_path == 1 || safe_pivots(c);
    // (0 = automatic: the single-GPU rule, a function of N and the mode only -- the same on every rank, and the same summation grouping as
    //  mpf_factor_dev: the multi-rank result stays bit-identical to the single-GPU one)
    int want_sb = f64 ? 1 : (c->tune.superpanel_fp16 > 0 ? c->tune.superpanel_fp16 : ((o.trailing == MPF_TRAIL_FP16 && N >= 24576) ? 6 : 4));
    if (!f64 && o.superpanel > 0) want_sb = o.superpanel > 8 ? 8 : o.superpanel;
    const int sb = (!f64 && !force_generic_ && c->tune.fp16_work32 != 0 && hgetf2_lds_eligible(c, (int)N, (int)(nb < N ? nb : N)) &&
                    (int64_t)(want_sb + 1) * nb < N) ? want_sb : 1;
    const int64_t sbw = (int64_t)sb * nb;
    if (!f64) {
        const int64_t kimg = sbw < 8 * HP_MAXCOLS ? sbw : 8 * HP_MAXCOLS;
        rc = sb > 1 ? mpf_ensure_h_images(c, N + HP_MAXCOLS + 64, (int)kimg, true) : mpf_ensure_h_images(c, N, nb, false);
        if (rc) return rc;
    }
    if (sb > 1) {
        const int64_t wneed = N * (lcols > 0 ? lcols : 1), sneed = N * sbw;
        if (c->dist_w32_cap < wneed) {
            if (c->dist_w32) hipFree(c->dist_w32);
            c->dist_w32 = nullptr; c->dist_w32_cap = 0;
            MPF_HIP_TRY(c, hipMalloc((void **)&c->dist_w32, (size_t)wneed * sizeof(float)));
            c->dist_w32_cap = wneed;
        }
        if (c->dist_spl_cap < sneed) {
            if (c->dist_spl) hipFree(c->dist_spl);
            c->dist_spl = nullptr; c->dist_spl_cap = 0;
            MPF_HIP_TRY(c, hipMalloc((void **)&c->dist_spl, (size_t)sneed * sizeof(double)));
            c->dist_spl_cap = sneed;
        }
    }
    if (!c->xstream && dist->world > 1 && c->tune.dist_instalments) {   // exchange stream of the panel's instalments
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&c->xstream, hipStreamNonBlocking, hi) != hipSuccess) c->xstream = nullptr;
    }
    // ---- what every rank must decide alike ------------------------------------------------------------------------------
    // How many broadcasts a panel message travels in, and the one- / two-level choice, follow from rank-local state: which streams
    // could be created, what this device's pivot kernel can hold, per-context options.  Ranks that disagreed would issue different
    // collectives (a hang) or factor with different roundings.  Every rank's values go round once (one 64-byte broadcast per rank)
    // and everyone takes the minimum; `sb` differing is an error on every rank (the buffers above were sized for the local value).
    long long agreed[8];
    {
        hp_query_residency(c);
        // rows this device's pivot kernel can take beside the workgroups of the pipelined chain's gated interchange kernel (which wait for it)
        const long long lds_rows = hgetf2_capacity_rows(c, laswp_gated_grid(nb));
        const bool two_ = !(o.no_lookahead || !c->pstream);
        long long mine[8] = {two_ && c->tune.chain_pipeline != 0 && c->tstream != nullptr && !force_generic_ && c->tune.dist_instalments && c->xstream ? 1 : 0,
                             sb, -(long long)c->tune.dist_instalment_min_bytes, c->tune.dpanel_fused_form ? 1 : 0, lds_rows < HP_R * (long long)HP_MAXG ? lds_rows : HP_R * (long long)HP_MAXG,
                             two_ ? 1 : 0, 0, 0};
        for (int i = 0; i < 8; ++i) agreed[i] = mine[i];
        if (dist->world > 1) {
            long long *dv = (long long *)c->dist_buf[0];   // (free until the first panel message)
            for (int root = 0; root < dist->world; ++root) {
                if (root == dist->rank) MPF_HIP_TRY(c, hipMemcpyAsync(dv, mine, sizeof mine, hipMemcpyHostToDevice, c->stream));
                const int e = bcast_fn(user, dv, (int64_t)sizeof mine, root, (void *)c->stream);
                if (e) return e < 0 ? e : -5;
                long long got[8];
                MPF_HIP_TRY(c, hipMemcpyAsync(got, dv, sizeof got, hipMemcpyDeviceToHost, c->stream));
                MPF_HIP_TRY(c, hipStreamSynchronize(c->stream));
                for (int i = 0; i < 8; ++i) if (got[i] < agreed[i]) agreed[i] = got[i];
            }
        }
        if (agreed[1] != sb) { c->err = "mpf_factor_dist: the ranks disagree on the super-panel width (options superpanel_fp16 / fp16_work32 / safe_pivots must be the same on every rank)"; return -1; }
    }
    const bool inst_ok = agreed[0] != 0;
    const long long inst_min_bytes = -agreed[2], lds_rows_max = agreed[4];
    const bool fused_form_all = agreed[3] != 0;
    // fp64 mode: this rank's columns right of the current panel live in a ROW-major working copy (as factor_lookahead_rm's): an
    // interchange moves contiguous row segments instead of one 64-byte sector per moved row and column; the update is the same MFMA
    // kernel on the transposed problem (c = fma(-u, l, c), kk ascending: the same bits).  Rank-local choice: results do not depend on it.
    const bool rm = f64 && lcols > 0 && c->tune.fp64_rowmajor && N >= c->tune.fp64_rowmajor_min_n && N > nb &&
                    mpf_ensure_rowmajor_copy(c, N, lcols, nb) == 0;
    double *Rl = rm ? c->r64 : nullptr;
    const int64_t ldr = lcols > 0 ? lcols : 1;
    {   // per-panel moved-row lists + scratch of the deferred left-hand interchanges (as mpf_factor_dev)
        const int npanels = L.nblocks;
        if (npanels > c->lists_cap) {
            if (c->lists) hipFree(c->lists);
            c->lists = nullptr; c->lists_cap = 0;
            MPF_HIP_TRY(c, hipMalloc((void **)&c->lists, (size_t)npanels * sizeof(MovedList)));
            c->lists_cap = npanels;
        }
        // N x nb doubles for the deferred left-hand interchanges; at least N x 256 so that the scratch also holds the
        // 2 * HP_MAXCOLS moved rows x local columns (fp32) of an interchange on the row-major working copy
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
    hipStream_t S = c->stream, P = (o.no_lookahead || !c->pstream) ? c->stream : c->pstream;
    const bool two = P != S;
    EvPool ev(c);
    ev.keep = &st.ms_gemm;
    MPF_HIP_TRY(c, hipEventRecord(c->ev0, S));
    if (two) { hipEvent_t e = ev.get(); hipEventRecord(e, S); hipStreamWaitEvent(P, e, 0); }

    auto buf_of = [&](int b) { return (char *)c->dist_buf[b & 1]; };
    auto rows_of = [&](int b) { return N - (int64_t)b * nb; };
    auto ldp_of = [&](int b) { return rows_of(b) + PANEL_PAD; };
    auto list_off = [&](int b) { return (size_t)ldp_of(b) * L.width(b) * 8; };
    auto piv_ptr = [&](int b, int q) { return (int *)((double *)buf_of(b) + (size_t)(32 * q) * ldp_of(b) + rows_of(b)); };  // pivots of sub-panel q
    const bool piped_ok = c->tune.chain_pipeline != 0 && c->tstream != nullptr;
    const bool force_generic = force_generic_;   // as mpf_factor_dev: GPUs shared with other processes
    // Number of 32-column instalments the message of panel b travels in (0: one broadcast after the whole chain).  A function
    // of the shape and of options that must be the same on every rank (chain_pipeline, pivot_path / safe_pivots, dist_instalments).
    // Small panels go in one piece: an instalment costs a collective's latency.
    auto pieces_of = [&](int b) -> int {
        const int pc = L.width(b), pr = (int)rows_of(b);
        if (!inst_ok) return 0;
        if (pc > HP_MAXCOLS || pr > lds_rows_max) return 0;                       // (every rank's pivot kernel can hold the panel)
        if ((long long)pr * pc * 8 < inst_min_bytes) return 0;
        return (fused_form_all && pc % 32 == 0 && pc >= 64) ? pc / 32 : 0;        // = dgetf2_npv_pieces with the agreed option
    };
    // owner only: pivots, interchange of the panel's own columns, fp64 panel, pack -- on stream s (and T).  With instalments
    // (np > 0) ev_piece[q] is recorded once instalment q (32 packed columns + their pivots; the last one: + the moved-row list)
    // is complete in the message buffer.
    std::vector<hipEvent_t> ev_piece;
    auto chain = [&](int b, hipStream_t s, int np) -> int {
        const int64_t k = (int64_t)b * nb;
        const int pc = L.width(b), pr = (int)(N - k);
        const int64_t ldp = ldp_of(b);
        StreamSwap sw(c, s);
        double *Ap = d_Aloc + L.lcol(b) * ldloc + k;
        MovedList *ml = c->lists + b;
        char *buf = buf_of(b);
        hipEvent_t before_pivots = ev.get();
        hipEventRecord(before_pivots, s);
        const bool lds = !force_generic && hgetf2_lds_eligible(c, pr, pc);
        // the fp64 panel follows the pivot kernel 32 columns behind on the helper stream (see chain_pipelined in mpf_host.cpp) -- where
        // the panel's workgroups fit beside the gated interchange kernel's, which wait for it while sitting on CUs
        const int waiters = (c->tune.gate_wait_value && c->hp_signal) ? 0 : laswp_gated_grid(pc);
        const bool will_pipe = piped_ok && lds && hgetf2_fits_beside(c, pr, pc, waiters);
        int e = ev.timed(st.ms_hpanel, s, [&] {
            if (lds) return launch_hgetf2(c, Ap, ldloc, nullptr, 0, pr, pc, (int)k, d_ipiv + k, nullptr, 0, ml, will_pipe ? waiters : 0, f64 ? HP_FP64_WINDOW_ROWS : 0);
            st.pivot_path = 1; // generic pivots, then the sequential swap list resolved into a moved-row list (laswp.hip)
            int e2 = launch_hgetf2_generic(c, Ap, ldloc, nullptr, 0, pr, pc, (int)k, d_ipiv + k, nullptr, 0);
            if (!e2) e2 = launch_laswp_plan(c, d_ipiv + k, (int)k, pc, ml);
            return e2; });
        if (e) return e;
        hipEvent_t pivots_done = ev.get();
        hipEventRecord(pivots_done, s);
        const int npp = will_pipe ? dgetf2_npv_pieces(c, pc) : 0;
        if (np > 0 && npp != np) { c->err = "mpf_factor_dist: instalment count differs from the agreed one"; return -1; }
        auto pack = [&](hipStream_t ps, int c0, int nc) -> int {   // columns [c0, c0 + nc) of the panel + their sub-panels' pivots
            MPF_HIP_TRY(c, hipMemcpy2DAsync(buf + (size_t)c0 * ldp * 8, (size_t)ldp * 8, Ap + (int64_t)c0 * ldloc, (size_t)ldloc * 8, (size_t)pr * 8, (size_t)nc,
                                            hipMemcpyDeviceToDevice, ps));
            for (int q = c0 / 32; q * 32 < c0 + nc; ++q) {
                const int w = (pc - 32 * q) < 32 ? (pc - 32 * q) : 32;
                MPF_HIP_TRY(c, hipMemcpyAsync(piv_ptr(b, q), d_ipiv + k + 32 * q, (size_t)w * 4, hipMemcpyDeviceToDevice, ps));
            }
            return 0;
        };
        if (npp > 0) {
            hipStream_t T = c->tstream;
            hipStreamWaitEvent(T, before_pivots, 0);
            {
                StreamSwap swt(c, T);
                hipEvent_t t0 = ev.get(), t1 = ev.get();
                hipEventRecord(t0, T);
                for (int q = 0; q < npp && !e; ++q) {
                    e = launch_laswp_block_gated(c, d_Aloc + L.lcol(b) * ldloc, ldloc, pc, (int)k + 32 * q, 32, d_ipiv + k + 32 * q, N, 32 * (q + 1));
                    if (!e) e = launch_dgetf2_npv_piece(c, Ap, ldloc, pr, pc, o.fused_panel, (int)k, q);
                    if (!e && np > 0) {   // instalment q leaves as it is: the interchanges still to come are applied to the message copy
                        e = pack(T, 32 * q, 32);
                        // (the matrix holds the UNfactored diagonal tile until the last piece has run: dpanel.hip)
                        if (!e && q < np - 1) e = launch_dpanel_tile_copy(c, (double *)buf + (size_t)(32 * q) * ldp + 32 * q, ldp, q);
                        if (!e && q == np - 1) {
                            hipStreamWaitEvent(T, pivots_done, 0);   // the moved-row list is complete when the pivot kernel has ended
                            MPF_HIP_TRY(c, hipMemcpyAsync(buf + list_off(b), ml, sizeof(MovedList), hipMemcpyDeviceToDevice, T));
                        }
                        if (!e) { ev_piece[(size_t)q] = ev.get(); hipEventRecord(ev_piece[(size_t)q], T); }
                    }
                }
                hipEventRecord(t1, T);
                if (c->tune.timeline || c->tune.event_timers >= 2) ev.pairs.push_back({t0, t1, &st.ms_dpanel, nullptr});
            }
            hipEvent_t eb = ev.get();
            hipEventRecord(eb, T);
            hipStreamWaitEvent(s, eb, 0);
        } else e = ev.timed(st.ms_dpanel, s, [&] {
            int e2 = launch_laswp_from_list(c, d_Aloc + L.lcol(b) * ldloc, ldloc, pc, ml);
            if (!e2) e2 = launch_dgetf2_npv(c, Ap, ldloc, pr, pc, o.fused_panel, (int)k);
            return e2; });
        if (e) return e;
        if (np == 0) {   // the whole message at once, from the finished panel
            e = pack(s, 0, pc);
            if (!e) MPF_HIP_TRY(c, hipMemcpyAsync(buf + list_off(b), ml, sizeof(MovedList), hipMemcpyDeviceToDevice, s));
        }
        st.panels++;
        return e;
    };
    // every rank: the exchange of panel b on stream s, then (non-owners) pivots and moved list out of the message.
    // np == 0: ONE broadcast.  np > 0: np instalments; after instalment q every rank applies sub-panel q's 32 interchanges to
    // the instalments already there (what the owner's matrix got from laswp_block while the later instalments were still being
    // factored): the assembled message is the finished panel.
    auto exchange = [&](int b, hipStream_t s, int np) -> int {
        const int pc = L.width(b);
        const int64_t k = (int64_t)b * nb, ldp = ldp_of(b);
        char *buf = buf_of(b);
        if (np == 0) {
            if (L.world > 1) {
                const int e = bcast_fn(user, buf, (int64_t)(list_off(b) + sizeof(MovedList)), L.owner(b), (void *)s);
                if (e) return e < 0 ? e : -5;
            }
        } else {
            StreamSwap sw(c, s);
            for (int q = 0; q < np; ++q) {
                if (L.mine(b)) hipStreamWaitEvent(s, ev_piece[(size_t)q], 0);
                if (L.world > 1) {
                    const size_t off = (size_t)(32 * q) * ldp * 8;
                    const size_t bytes = q == np - 1 ? list_off(b) + sizeof(MovedList) - off : (size_t)32 * ldp * 8;
                    const int e = bcast_fn(user, buf + off, (int64_t)bytes, L.owner(b), (void *)s);
                    if (e) return e < 0 ? e : -5;
                }
                if (q > 0) {   // rows of the message are rows k.. of the matrix: address it from the matrix' row 0
                    const int e = launch_laswp_block(c, (double *)buf - k, ldp, 32 * q, (int)k + 32 * q, 32, piv_ptr(b, q), N);
                    if (e) return e;
                }
            }
        }
        if (!L.mine(b)) {
            const int nq = (pc + 31) / 32;
            MPF_HIP_TRY(c, hipMemcpy2DAsync(d_ipiv + k, 128, piv_ptr(b, 0), (size_t)32 * ldp * 8, 128, (size_t)(pc / 32), hipMemcpyDeviceToDevice, s));
            if (pc % 32) MPF_HIP_TRY(c, hipMemcpyAsync(d_ipiv + k + 32 * (nq - 1), piv_ptr(b, nq - 1), (size_t)(pc % 32) * 4, hipMemcpyDeviceToDevice, s));
            MPF_HIP_TRY(c, hipMemcpyAsync(c->lists + b, buf + list_off(b), sizeof(MovedList), hipMemcpyDeviceToDevice, s));
        }
        return 0;
    };
    const int kst = (int)((sbw + 63) & ~(int64_t)63);                                      // row stride of the far U image (elements)
    const int64_t far_cap = sb > 1 ? lcols - L.first_local_col_after(sb) : 0;              // most far columns any super-panel has here
    const int64_t inner_u_off = far_cap > 0 ? far_cap * kst : 0;                           // the inner steps' U image lives behind the far image
    const int64_t brow_l_off = (int64_t)N * c->h_kmax;                                     // block-row L images: rows the image buffers hold beyond N
    // interchange + TRSM + trailing update of panel b on the local columns [c0, c0 + nc) (all right of block b)
    bool image_ready = false;
    auto update = [&](int b, int64_t c0, int64_t nc) -> int {
        if (nc <= 0) return 0;
        const int64_t k = (int64_t)b * nb;
        const int pc = L.width(b);
        const int64_t pr = N - k, m = pr - pc;
        const double *Pb = (const double *)buf_of(b);
        const int64_t ldp = ldp_of(b);
        if (rm) {
            // interchange of contiguous row segments, TRSM through strides, L21 row-major once per panel, the update on the transposed
            // problem, then the finished U rows of these columns go home to the column-major matrix
            double *LT = c->rm_lt;
            int e = ev.timed(st.ms_laswp, S, [&] { return launch_laswp_from_list_rm64(c, Rl + c0, ldr, nc, c->lists + b); });
            if (!e) e = ev.timed(st.ms_trsm, S, [&] { return launch_dtrsm_llnu_strided(c, pc, nc, Pb, ldp, Rl + k * ldr + c0, ldr, 1); });
            if (!e && m > 0) {
                if (!image_ready) { e = ev.timed(st.ms_cvt, S, [&] { return launch_transpose64(c, (double *)Pb + pc, ldp, LT, pc, m, pc, true); }); image_ready = true; }
                if (!e) e = ev.timed(st.ms_gemm, S, [&] { return launch_dgemm_minus(c, nc, m, pc, Rl + k * ldr + c0, ldr, LT, pc, Rl + (k + pc) * ldr + c0, ldr); });
                count_gemm(st, o, m, nc, pc);
            }
            if (!e) e = ev.timed(st.ms_cvt, S, [&] { return launch_transpose64(c, d_Aloc + c0 * ldloc + k, ldloc, Rl + k * ldr + c0, ldr, pc, nc, false); });
            return e;
        }
        int e = ev.timed(st.ms_laswp, S, [&] { return launch_laswp_from_list(c, d_Aloc + c0 * ldloc, ldloc, nc, c->lists + b); });
        double *U12 = d_Aloc + c0 * ldloc + k;
        if (!e) e = ev.timed(st.ms_trsm, S, [&] { return launch_dtrsm_llnu(c, pc, nc, Pb, ldp, U12, ldloc); });
        if (!e && m > 0) {
            e = ev.timed(st.ms_gemm, S, [&] {
                if (f64) return launch_dgemm_minus(c, m, nc, pc, Pb + pc, ldp, U12, ldloc, U12 + pc, ldloc);
                int e2 = 0;
                if (!image_ready) { e2 = launch_cvt_l21(c, Pb + pc, ldp, m, pc, split); image_ready = true; } // once per panel
                if (sb == 1) { if (!e2) e2 = launch_hgemm_minus(c, m, nc, pc, U12, ldloc, U12 + pc, ldloc, split); }
                else {   // the inner steps' U image lives behind the far columns' image
                    if (!e2) e2 = launch_cvt_u12(c, U12, ldloc, pc, nc, split, inner_u_off);
                    if (!e2) e2 = launch_hgemm_images(c, m, nc, pc, U12 + pc, ldloc, false, split, 0, 0, inner_u_off);
                }
                return e2; });
            count_gemm(st, o, m, nc, pc);
        }
        return e;
    };
    // ---- two-level schedule: this rank's far columns (sb > 1) -------------------------------------------------------------
    float *W = c->dist_w32;
    double *SPL = c->dist_spl;
    const int64_t ldw = lcols > 0 ? lcols : 1;
    auto sp_first = [&](int b) { return (b / sb) * sb; };                                        // first block of b's super-panel
    auto sp_end = [&](int b) { const int e = sp_first(b) + sb; return e < L.nblocks ? e : L.nblocks; };   // one past its last block
    auto inner_end_blk = [&](int b) { const int e = sp_end(b) + 1; return e < L.nblocks ? e : L.nblocks; };  // one past the look-ahead block
    auto far0 = [&](int b) { return sb > 1 ? L.first_local_col_after(inner_end_blk(b) - 1) : lcols; };     // first local far column
    int far_img = 1;
    // the super-panel's panels as received: eager interchanges of panel b on the earlier ones, then panel b itself
    auto spl_add = [&](int b) -> int {
        const int p = b - sp_first(b), pc = L.width(b);
        const int64_t k = (int64_t)b * nb;
        int e = 0;
        if (p > 0) e = ev.timed(st.ms_laswp, S, [&] { return launch_laswp_from_list(c, SPL, N, (int64_t)p * nb, c->lists + b); });
        if (!e) MPF_HIP_TRY(c, hipMemcpy2DAsync(SPL + (int64_t)p * nb * N + k, (size_t)N * 8, buf_of(b), (size_t)ldp_of(b) * 8, (size_t)(N - k) * 8, (size_t)pc,
                                                hipMemcpyDeviceToDevice, S));
        return e;
    };
    // block-row task of panel b on the far columns (T_p of factor_superpanel): interchange on the working copy; the rows of block
    // b lose L[b, earlier panels of the super-panel] U; they return to fp64; TRSM with the panel's L11; appended to the U image
    auto far_task = [&](int b) -> int {
        const int64_t f0 = far0(b), ncols = lcols - f0, kq = (int64_t)b * nb;
        if (ncols <= 0) return 0;
        const int p = b - sp_first(b), pc = L.width(b);
        int e = spl_add(b);
        if (!e) e = ev.timed(st.ms_laswp, S, [&] { return launch_laswp_from_list_f32(c, W + f0, ldw, ncols, c->lists + b); });
        if (!e && p > 0) {
            const int K = p * nb;
            e = ev.timed(st.ms_cvt, S, [&] { return launch_cvt_l21(c, SPL + kq, N, pc, K, split, 0, brow_l_off); });
            if (!e) e = ev.timed(st.ms_trsm, S, [&] {
                return launch_hgemm_images_rowmajor(c, pc, ncols, K, W + kq * ldw + f0, ldw, split, 0, brow_l_off, 0, 0, kst); }, &st.ms_blockrow);
        }
        if (!e) e = ev.timed(st.ms_cvt, S, [&] { return launch_cvt_f32_f64(c, W + kq * ldw + f0, ldw, d_Aloc + f0 * ldloc + kq, ldloc, pc, ncols); });
        if (!e) e = ev.timed(st.ms_trsm, S, [&] {
            return launch_dtrsm_llnu(c, pc, ncols, (const double *)buf_of(b), ldp_of(b), d_Aloc + f0 * ldloc + kq, ldloc); }, &st.ms_blockrow);
        if (!e) e = ev.timed(st.ms_cvt, S, [&] { return launch_cvt_u12(c, d_Aloc + f0 * ldloc + kq, ldloc, pc, ncols, split, (int64_t)p * nb, kst); });
        return e;
    };
    // the K = (super-panel width) update of the local far columns [col0, col0 + ncols); f0 = first far column of that super-panel
    auto big_update = [&](int64_t s0, int64_t s1, int64_t f0, int64_t col0, int64_t ncols) -> int {
        const int64_t mrows = N - s1;
        const int K = (int)(s1 - s0);
        if (ncols <= 0 || mrows <= 0) return 0;
        const int e = ev.timed(st.ms_gemm, S, [&] {
            return launch_hgemm_images_rowmajor(c, mrows, ncols, K, W + s1 * ldw + col0, ldw, split, far_img, 0, (col0 - f0) * kst, 0, kst); }, &st.ms_gemm_big);
        const double opb = split ? 4.0 : 2.0;
        count_gemm(st, o, mrows, ncols, K, 8.0);
        st.gemm_big_flops += 2.0 * (double)mrows * (double)ncols * K;
        st.gemm_big_bytes += 8.0 * (double)mrows * (double)ncols + opb * K * (double)(mrows + ncols);
        st.gemm_big_launches++;
        return e;
    };
    // end of b's super-panel: the operand image of L[s1.., s0..s1) once, the update on the columns that join the next inner region
    // first (they return to fp64), then on the rest
    auto superpanel_end = [&](int b) -> int {
        const int64_t f0 = far0(b);
        if (f0 >= lcols) return 0;
        const int64_t s0 = (int64_t)sp_first(b) * nb, s1 = (int64_t)sp_end(b) * nb;
        const int64_t f0n = far0(sp_end(b));                          // first far column of the next super-panel
        int e = ev.timed(st.ms_cvt, S, [&] { return launch_cvt_l21(c, SPL + s1, N, N - s1, (int)(s1 - s0), split, far_img); });
        if (!e) e = big_update(s0, s1, f0, f0, f0n - f0);
        if (!e && f0n > f0) e = ev.timed(st.ms_cvt, S, [&] { return launch_cvt_f32_f64(c, W + s1 * ldw + f0, ldw, d_Aloc + f0 * ldloc + s1, ldloc, N - s1, f0n - f0); });
        if (!e) e = big_update(s0, s1, f0, f0n, lcols - f0n);
        far_img = 3 - far_img;
        return e;
    };
    auto superpanel_begin = [&](int b) -> int {   // image blocks of odd width: the padding the kernels read beyond a block must be zero
        const int64_t f0 = far0(b);
        if (f0 >= lcols || nb % 64 == 0) return 0;
        const size_t bytes = (size_t)(lcols - f0) * kst * sizeof(unsigned short);
        MPF_HIP_TRY(c, hipMemsetAsync(c->h_U, 0, bytes, S));
        if (split) MPF_HIP_TRY(c, hipMemsetAsync(c->h_U + c->h_rows * c->h_kmax, 0, bytes, S));
        return 0;
    };

    rc = 0;
    if (sb > 1 && far0(0) < lcols)   // this rank's far columns of the first super-panel go into the working copy
        rc = ev.timed(st.ms_cvt, S, [&] { return launch_cvt_f64_f32(c, d_Aloc + far0(0) * ldloc, ldloc, W + far0(0), ldw, N, lcols - far0(0)); });
    if (!rc && L.live(0)) {
        if (L.mine(0)) rc = chain(0, S, 0);
        if (!rc) rc = exchange(0, S, 0);
    }
    if (!rc && rm)   // everything this rank owns goes into the row-major copy (block 0, if it is here, is finished: its copy is never read)
        rc = ev.timed(st.ms_cvt, S, [&] { return launch_transpose64(c, d_Aloc, ldloc, Rl, ldr, N, lcols, true); });
    hipStream_t X = c->xstream ? c->xstream : P;   // exchange stream of the instalments (the owner's P runs the pivot kernel meanwhile)
    for (int b = 0; L.live(b) && rc == 0; ++b) {
        const int64_t k = (int64_t)b * nb;
        const int pc = L.width(b);
        const int nxt = b + 1;
        const bool has_next = L.live(nxt), own_next = has_next && L.mine(nxt), trailing = k + pc < N;
        int64_t rest0 = L.first_local_col_after(b);
        image_ready = false;
        hipEvent_t e2 = nullptr;
        if (trailing && own_next) { // my block of panel b+1 first, then its chain on the side stream
            rc = update(b, L.lcol(nxt), L.width(nxt));
            if (!rc && rm)   // the next panel's columns return to the column-major matrix before its chain (rows k + pc ..; its U rows went with the update)
                rc = ev.timed(st.ms_cvt, S, [&] { return launch_transpose64(c, d_Aloc + L.lcol(nxt) * ldloc + k + pc, ldloc, Rl + (k + pc) * ldr + L.lcol(nxt), ldr,
                                                                            N - k - pc, L.width(nxt), false); });
            if (rc) break;
            rest0 = L.lcol(nxt) + L.width(nxt);
        }
        if (has_next) {
            const int np = pieces_of(nxt);
            hipStream_t xs = np > 0 ? X : P;
            if (two) { // strip done; receive buffer free (its last reader was update b - 1, in front of this event on S)
                hipEvent_t e1 = ev.get(); hipEventRecord(e1, S); hipStreamWaitEvent(P, e1, 0);
                if (xs != P) hipStreamWaitEvent(xs, e1, 0);
            }
            ev_piece.assign((size_t)(np > 0 ? np : 1), nullptr);
            if (own_next) rc = chain(nxt, P, np);
            if (!rc) rc = exchange(nxt, xs, np);
            if (rc) break;
            if (two) {
                e2 = ev.get(); hipEventRecord(e2, xs);
                if (xs != P && own_next) { hipEvent_t e3 = ev.get(); hipEventRecord(e3, P); hipStreamWaitEvent(S, e3, 0); }   // the owner's chain itself
            }
        }
        if (sb == 1) { if (trailing && rest0 < lcols) rc = update(b, rest0, lcols - rest0); }
        else if (trailing) {
            // inner blocks right of the strip, panel by panel in fp64; far columns: block-row task now, K = sb * nb update at the
            // end of the super-panel (it runs under the chain of the next super-panel's first panel, launched above)
            const int64_t f0 = far0(b);
            if (b == sp_first(b)) rc = superpanel_begin(b);
            if (!rc && rest0 < f0) rc = update(b, rest0, f0 - rest0);
            if (!rc) rc = far_task(b);
            if (!rc && b + 1 == sp_end(b)) rc = superpanel_end(b);
        }
        if (e2) hipStreamWaitEvent(S, e2, 0);
        if (o.verbose) printf("[rank %d] panel %d (k=%lld) owner %d\n", L.rank, b, (long long)k, L.owner(b));
    }
    if (!rc) rc = ev.timed(st.ms_laswp, S, [&] { return launch_lazy_left_swaps(c, d_Aloc, ldloc, N, nb, L.nblocks, c->lists, 1, L.world, L.rank); });
    hipEventRecord(c->ev1, S);
    hipError_t se = hipStreamSynchronize(S);
    hipError_t sp = two ? hipStreamSynchronize(P) : hipSuccess;
    if (two && c->xstream) { const hipError_t sx = hipStreamSynchronize(c->xstream); if (sp == hipSuccess) sp = sx; }
    if (two && c->tstream) { const hipError_t stt = hipStreamSynchronize(c->tstream); if (sp == hipSuccess) sp = stt; }
    if (rc) return rc;
    if (se != hipSuccess || sp != hipSuccess) { c->err = std::string("distributed factorization failed: ") + hipGetErrorString(se != hipSuccess ? se : sp); return -2; }
    ev.collect();
    float ms = 0;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    st.ms_total = ms;
    st.lookahead = two ? 1 : 0;
    int info = 0, flags0 = 0;
    MPF_HIP_TRY(c, hipMemcpy(&info, &c->ws->info, sizeof(int), hipMemcpyDeviceToHost));
    MPF_HIP_TRY(c, hipMemcpy(&flags0, &c->ws->hp_timeouts, sizeof(int), hipMemcpyDeviceToHost));
    st.info = info == INT_MAX ? 0 : info;
    st.hpanel_timeouts = flags0;
    c->stats = st;
    if (flags0) { c->err = "fp16 pivot kernel: inter-workgroup hand-off timed out (buffers invalid; use pivot_path = 1)"; return -4; }
    return st.info; // this rank's panels only: the caller combines (min over the positive values)
}

int mpf_dist_set_p2p(mpf_ctx *c, mpf_p2p_fn fn, void *user) {
    if (!c) return -1;
    c->p2p_fn = fn; c->p2p_user = user;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// distributed refinement solve: residual = local GEMV + all-reduce; triangular solves walk the column blocks, the owner
// of a block applies it to the (replicated) vector and broadcasts the vector on
// ---------------------------------------------------------------------------------------------------------------------
int mpf_solve_ir_dist(mpf_ctx *c, const double *d_Aloc, int64_t lda, const double *d_LUloc, int64_t ldlu, const int32_t *d_ipiv,
                      int64_t N, int32_t nb, const double *d_b, double *d_x, int32_t max_iter, double tol, const mpf_dist *dist,
                      mpf_ir_stats *stats) {
    if (!c || !d_ipiv || !d_b || !d_x || !dist) return -1;
    if (N <= 0 || nb <= 0) { c->err = "solve_dist: bad N / nb"; return -1; }
    if (max_iter > 31) max_iter = 31;
    MPF_HIP_TRY(c, hipSetDevice(c->device));
    mpf_bcast_fn bcast_fn = dist->bcast ? dist->bcast : rccl_bcast;
    mpf_allreduce_fn ar_fn = dist->allreduce ? dist->allreduce : rccl_allreduce;
    void *user = (dist->bcast || dist->allreduce) ? dist->user : (void *)c;
    const Layout L(N, nb, dist->rank, dist->world);
    const int64_t lcols = L.local_cols();
    int rc = mpf_ensure_solve_buf(c, N);
    if (rc) return rc;
    const int64_t SN = c->solve_n;
    double *r = c->solve_buf, *d = c->solve_buf + SN, *xloc = c->solve_buf + 2 * SN, *scal = c->solve_buf + 4 * SN;
    hipStream_t S = c->stream;
    std::vector<int32_t> ip((size_t)N), perm((size_t)N);
    MPF_HIP_TRY(c, hipMemcpyAsync(ip.data(), d_ipiv, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost, S));
    MPF_HIP_TRY(c, hipStreamSynchronize(S));
    for (int64_t i = 0; i < N; ++i) perm[(size_t)i] = (int32_t)i;
    for (int64_t i = 0; i < N; ++i) {
        const int64_t p = (int64_t)ip[(size_t)i] - 1;
        if (p < 0 || p >= N) { c->err = "solve_dist: ipiv entry out of range"; return -1; }
        if (p != i) std::swap(perm[(size_t)i], perm[(size_t)p]);
    }
    MPF_HIP_TRY(c, hipMemcpyAsync(c->perm_buf, perm.data(), (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice, S));
    mpf_ir_stats st{};
    hipEventRecord(c->ev0, S);
    // inverted diagonal blocks of the factors, for the blocks this rank owns (global block-of-64 index = position in trsv_inv)
    for (int b = L.rank; b < L.nblocks; b += L.world) {
        rc = launch_trsv_prepare_cols(c, d_LUloc + L.lcol(b) * ldlu, ldlu, N, (int64_t)b * nb, L.width(b));
        if (rc) return rc;
    }
    auto vec_bcast = [&](double *v, int64_t off, int64_t cnt, int root) -> int {
        if (L.world == 1 || cnt <= 0) return 0;
        const int e = bcast_fn(user, v + off, cnt * 8, root, (void *)S);
        return e ? (e < 0 ? e : -5) : 0;
    };
    // Point-to-point transport (round 4): the context's RCCL communicator (ncclSend / ncclRecv) with the built-in transport, or the
    // callback registered with mpf_dist_set_p2p next to the caller's own broadcast / all-reduce.
    mpf_p2p_fn p2p_fn = nullptr;
    void *p2p_user = nullptr;
    if (L.world > 1) {
        if (c->p2p_fn) { p2p_fn = c->p2p_fn; p2p_user = c->p2p_user; }
        else if (c->tune.dist_solve_p2p && !dist->bcast && !dist->allreduce && c->rccl_comm && rccl_has_p2p()) { p2p_fn = rccl_p2p; p2p_user = (void *)c; }
        // The choice follows from rank-local state (an option, a registered callback): ranks that chose differently would issue
        // different collectives and hang.  Every rank votes; the chain is taken only if ALL of them have it.
        double vote[2] = {p2p_fn ? 1.0 : 0.0, 1.0};
        MPF_HIP_TRY(c, hipMemcpyAsync(scal, vote, sizeof vote, hipMemcpyHostToDevice, S));
        { const int e = ar_fn(user, scal, 2, (void *)S); if (e) return e < 0 ? e : -5; }
        MPF_HIP_TRY(c, hipMemcpyAsync(vote, scal, sizeof vote, hipMemcpyDeviceToHost, S));
        MPF_HIP_TRY(c, hipStreamSynchronize(S));
        if (vote[0] != vote[1]) { p2p_fn = nullptr; p2p_user = nullptr; }
    }
    double *up = xloc, *rbuf = c->solve_buf + 3 * SN;   // (xloc is only used by the residual, between two solves)
    auto p2p = [&](double *v, int64_t off, int64_t cnt, int peer, int send) -> int {
        if (cnt <= 0) return 0;
        const int e = p2p_fn(p2p_user, v + off, cnt * 8, peer, send, (void *)S);
        return e ? (e < 0 ? e : -5) : 0;
    };
    // out = U^-1 L^-1 P rhs, replicated on every rank.
    // With a point-to-point transport the running vector travels from owner to owner (a block's update needs nothing but what the
    // blocks before it have done to the rows below): 2 (nblocks - 1) sends of the tail / head + ONE all-reduce per solve, each rank
    // touching the wire only for its own blocks -- round 3 broadcast the vector to every rank after every block (2 nblocks
    // collectives per solve).  Lower sweep: the tail out[k + w .. N) goes on.  Upper sweep: every rank starts from its OWN segments
    // of y (zero elsewhere), adds what arrives, applies its block, sends the head [0, k) on and zeroes it locally (what has been sent
    // is the next owner's to hold); at the end the ranks' vectors are disjoint pieces of x: one all-reduce replicates it.
    auto lu_solve_chain = [&](const double *rhs, double *out) -> int {
        int e = launch_gather_rows(c, rhs, c->perm_buf, out, N);
        for (int b = 0; b < L.nblocks && !e; ++b) {
            if (!L.mine(b)) continue;
            const int64_t k = (int64_t)b * nb;
            const int w = L.width(b);
            if (b > 0 && L.owner(b - 1) != L.rank) e = p2p(out, k, N - k, L.owner(b - 1), 0);
            if (!e) e = launch_trsv_lower_cols(c, d_LUloc + L.lcol(b) * ldlu, ldlu, out, N, k, w);
            if (!e && b + 1 < L.nblocks && L.owner(b + 1) != L.rank) e = p2p(out, k + w, N - k - w, L.owner(b + 1), 1);
        }
        if (e) return e;
        MPF_HIP_TRY(c, hipMemsetAsync(up, 0, (size_t)N * 8, S));
        for (int b = L.rank; b < L.nblocks; b += L.world)
            MPF_HIP_TRY(c, hipMemcpyAsync(up + (int64_t)b * nb, out + (int64_t)b * nb, (size_t)L.width(b) * 8, hipMemcpyDeviceToDevice, S));
        for (int b = L.nblocks - 1; b >= 0 && !e; --b) {
            if (!L.mine(b)) continue;
            const int64_t k = (int64_t)b * nb;
            const int w = L.width(b);
            if (b + 1 < L.nblocks && L.owner(b + 1) != L.rank) {
                e = p2p(rbuf, 0, k + w, L.owner(b + 1), 0);
                if (!e) e = launch_axpy(c, 1.0, rbuf, up, k + w);
            }
            if (!e) e = launch_trsv_upper_cols(c, d_LUloc + L.lcol(b) * ldlu, ldlu, up, N, k, w);
            if (!e && b > 0 && L.owner(b - 1) != L.rank) {
                e = p2p(up, 0, k, L.owner(b - 1), 1);
                if (!e) MPF_HIP_TRY(c, hipMemsetAsync(up, 0, (size_t)k * 8, S));
            }
        }
        if (!e) { e = ar_fn(user, up, N, (void *)S); if (e) e = e < 0 ? e : -5; }
        if (!e) MPF_HIP_TRY(c, hipMemcpyAsync(out, up, (size_t)N * 8, hipMemcpyDeviceToDevice, S));
        return e;
    };
    auto lu_solve = [&](const double *rhs, double *out) -> int {
        if (p2p_fn) return lu_solve_chain(rhs, out);
        // broadcast-only transport: the owner applies a block to the replicated vector and broadcasts it on
        int e = launch_gather_rows(c, rhs, c->perm_buf, out, N);
        for (int b = 0; b < L.nblocks && !e; ++b) {
            const int64_t k = (int64_t)b * nb;
            if (L.mine(b)) e = launch_trsv_lower_cols(c, d_LUloc + L.lcol(b) * ldlu, ldlu, out, N, k, L.width(b));
            if (!e) e = vec_bcast(out, k, N - k, L.owner(b));
        }
        for (int b = L.nblocks - 1; b >= 0 && !e; --b) {
            const int64_t k = (int64_t)b * nb;
            if (L.mine(b)) e = launch_trsv_upper_cols(c, d_LUloc + L.lcol(b) * ldlu, ldlu, out, N, k, L.width(b));
            if (!e) e = vec_bcast(out, 0, k + L.width(b), L.owner(b));
        }
        return e;
    };
    auto norm = [&](const double *v, double &out) -> int {
        int e = launch_norm2(c, v, N, scal);
        if (e) return e;
        double h = 0;
        MPF_HIP_TRY(c, hipMemcpyAsync(&h, scal, sizeof(double), hipMemcpyDeviceToHost, S));
        MPF_HIP_TRY(c, hipStreamSynchronize(S));
        out = std::sqrt(h);
        return 0;
    };
    auto residual = [&](const double *x, double *rr) -> int { // rr = b - A x: own columns, then the sum over ranks
        int e = 0;
        for (int b = L.rank; b < L.nblocks && !e; b += L.world) // x restricted to the columns this rank owns
            MPF_HIP_TRY(c, hipMemcpyAsync(xloc + L.lcol(b), x + (int64_t)b * nb, (size_t)L.width(b) * 8, hipMemcpyDeviceToDevice, S));
        e = launch_residual_rect(c, d_Aloc, lda, xloc, L.rank == 0 ? d_b : nullptr, rr, N, lcols);
        if (!e && L.world > 1) { e = ar_fn(user, rr, N, (void *)S); if (e) e = e < 0 ? e : -5; }
        return e;
    };
    double nb2 = 0;
    rc = norm(d_b, nb2);
    if (rc) return rc;
    if (nb2 == 0) nb2 = 1;
    rc = lu_solve(d_b, d_x);
    if (rc) return rc;
    for (int it = 0;; ++it) {
        rc = residual(d_x, r);
        if (rc) return rc;
        double nr = 0;
        rc = norm(r, nr);
        if (rc) return rc;
        st.rel_residual = nr / nb2;
        st.history[it] = st.rel_residual;
        st.iterations = it;
        if (st.rel_residual <= tol) { st.converged = 1; break; }
        if (it >= max_iter || !(st.rel_residual == st.rel_residual)) break;
        if (it >= 2 && st.history[it] > 0.7 * st.history[it - 1] && st.history[it - 1] > 0.7 * st.history[it - 2]) { st.stalled = 1; break; }
        rc = lu_solve(r, d);
        if (rc) return rc;
        rc = launch_axpy(c, 1.0, d, d_x, N);
        if (rc) return rc;
    }
    hipEventRecord(c->ev1, S);
    MPF_HIP_TRY(c, hipStreamSynchronize(S));
    float ms = 0;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    st.ms_total = ms;
    if (stats) *stats = st;
    return 0;
}

} // extern "C"

// mpf_factor_host's way home (round 5; the reference copies the whole matrix back after the last panel, MPF.cu:245-247, and
// benchmark.cpp:219-222 times that copy with the call): finished BLOCK ROWS of the factors go to the caller's host matrix while the
// factorization runs.
//
// What is final when.  After panel p (rows / columns [k, k + pc), k = p * nb) has been pivoted and its U block row solved and stored,
// nothing touches rows [k, k + pc) of the result again: later panels interchange rows >= their own first row only.  In the device
// matrix the part of those rows RIGHT of column k (diagonal block, U) is in place; the part LEFT of it (L) is not, because the
// schedules defer the left-hand interchanges to one pass at the end (laswp.hip, launch_lazy_left_swaps): column block j < p still
// has its rows in the order they had when panel j was done.  So the sink keeps the row bookkeeping forward:
//   Id[pos]  = which row of the matrix-as-of-panel-0 stands at position pos after panels 0..p   (updated from panel p's moved-row list)
//   Pos_j[id] = where that row stood after panel j                                              (a snapshot per panel, N ints each)
// and final row r of column block j is the device row Pos_j[Id[r]] -- the same rows the deferred pass would bring there, which a
// schedule that has given its rows to the sink therefore skips.
//
// The way of a block row (each step is there for something that was measured, profiles/r05_xfer_probe*.log, r05_sink_trace*.log):
//   1. assembled on the device into a contiguous staging block [N columns][pc] by one gather kernel -- queued by the SCHEDULE'S thread on
//      the stream on which the block row became final, in program order.  (A stream of the sink's own shares a hardware queue with
//      one of the schedule's streams once a process has more than four streams, and its kernels then wait behind everything the
//      schedule has queued ahead: every block row left after the last kernel.)
//   2. copied into a PINNED bounce buffer by the sink's thread (one plain copy on a stream that carries nothing else).  (A copy into
//      the caller's pageable memory goes through the runtime's pinning path, which waited for the launching thread: 370 ms for the
//      first block row while the schedule was still being queued, 301 ms beside a thread parked in hipStreamSynchronize.)
//   3. scattered into the caller's matrix (N runs of pc * 8 bytes) by four host threads (one thread: 4.1 ms per 67-MB block row,
//      the link delivers one in 1.2).
// The schedule only pays for step 1's launches; it records an event and pushes "block rows below p are final behind this event".
// A pivot kernel whose hand-off timed out (-4) leaves wrong rows behind: the gather kernel stores the give-up counter with the block
// row, the thread stops sending when it is set; rows sent before are right (they were final), and mpf_factor_host repeats the call
// from a device snapshot.
#include "mpf_internal.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

namespace {
constexpr int SINK_BOUNCE = 3;         // pinned bounce buffers (one being filled, one or two being scattered)
constexpr int SINK_WORKERS = 4;        // host threads that scatter a block row
constexpr int SINK_TAIL = 16;          // doubles behind a staging block: [0] = the pivot kernels' give-up counter as the block row was assembled
}

struct RowSink {
    mpf_ctx *c = nullptr;
    // per call
    double *host = nullptr;            // the caller's N x N column-major matrix
    const double *A = nullptr;         // the device matrix the schedule factors
    int64_t lda = 0, N = 0;
    int nb = 0, npanels = 0;
    int64_t slot_doubles = 0;          // N * nb + SINK_TAIL
    bool armed = false, taken = false;
    int next_asm = 0;                  // (schedule's thread) block rows whose assembly has been queued
    std::thread th;
    std::vector<std::thread> workers;
    std::mutex mu;                     // the notification queue
    std::condition_variable cv;
    std::deque<std::pair<int, hipEvent_t>> q;
    bool closing = false;
    struct Task { const double *src; double *dst; int64_t c0, c1; int pc; int slot; };
    std::mutex wmu;                    // the scatter tasks
    std::condition_variable wcv, wdone;
    std::deque<Task> tasks;
    int pending[SINK_BOUNCE] = {0, 0, 0};
    bool wclosing = false;
    size_t ev_used = 0;
    int sent = 0;                      // block rows that are home
    bool gave_up = false;              // stopped on the pivot kernels' give-up counter
    hipError_t err = hipSuccess;
    // for the life of the context
    hipStream_t cs = nullptr;          // carries the bounce copies and nothing else
    std::vector<hipEvent_t> events;
    int *maps = nullptr;               // Id (N) | Pos (N) | Pos_j, j = 0 .. npanels - 1 (N each)
    int64_t maps_cap = 0;              // ints
    double *stage = nullptr;           // one staging block per panel
    int64_t stage_cap = 0;             // doubles
    double *bounce = nullptr;          // SINK_BOUNCE pinned blocks
    int64_t bounce_cap = 0;            // doubles
};

namespace {
__global__ void sink_init_kernel(int *Id, int *Pos, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { Id[i] = (int)i; Pos[i] = (int)i; }
}
// one panel's interchanges: the row at src[t] moves to dst[t] (laswp.hip); all reads before all writes
__global__ __launch_bounds__(512) void sink_step_kernel(int *Id, int *Pos, const MovedList *ml) {
    int n = ml->n;
    if (n > LASWP_MAXMOVED) n = LASWP_MAXMOVED;
    const int t = threadIdx.x;
    int v = 0;
    if (t < n) v = Id[ml->src[t]];
    __syncthreads();
    if (t < n) { const int d = ml->dst[t]; Id[d] = v; Pos[v] = d; }
}
// block row [k, k + pc) of the finished factors -> stage[col * pc + r]; thread = row (coalesced writes, and coalesced reads right
// of column k; left of it every element is its own sector, as in the deferred pass this replaces)
__global__ __launch_bounds__(256) void sink_block_row_kernel(const double *__restrict__ A, long long lda, long long N, long long k, int pc, int nb,
                                                            const int *__restrict__ Id, const int *__restrict__ PosAll, double *__restrict__ stage,
                                                            const int *__restrict__ gave_up, long long tail_at) {
    const long long c0 = (long long)blockIdx.x * 16;
    if (blockIdx.x == 0 && threadIdx.x == 0) ((int *)(stage + tail_at))[0] = *gave_up;
    for (int r = threadIdx.x; r < pc; r += 256) {
        const int id = Id[k + r];
        long long jprev = -1, src = k + r;
#pragma unroll 4
        for (long long col = c0; col < c0 + 16 && col < N; ++col) {
            if (col < k) {
                const long long j = col / nb;
                if (j != jprev) { src = PosAll[j * N + id]; jprev = j; }
            } else src = k + r;
            stage[col * pc + r] = A[col * lda + src];
        }
    }
}

void sink_worker(RowSink *s) {
    for (;;) {
        RowSink::Task t;
        {
            std::unique_lock<std::mutex> lk(s->wmu);
            s->wcv.wait(lk, [&] { return !s->tasks.empty() || s->wclosing; });
            if (s->tasks.empty()) return;
            t = s->tasks.front();
            s->tasks.pop_front();
        }
        const size_t run = (size_t)t.pc * sizeof(double);
        for (int64_t col = t.c0; col < t.c1; ++col) memcpy(t.dst + col * s->N, t.src + col * t.pc, run);
        {
            std::lock_guard<std::mutex> lk(s->wmu);
            if (--s->pending[t.slot] == 0) s->wdone.notify_all();
        }
    }
}

void sink_thread(RowSink *s) {
    mpf_ctx *c = s->c;
    auto fail = [&](hipError_t e) { if (s->err == hipSuccess) s->err = e; };
    if (hipSetDevice(c->device) != hipSuccess) { fail(hipErrorInvalidDevice); return; }
    const int64_t N = s->N;
    const bool trace = c->tune.sink_trace != 0;   // (MPF_SINK_TRACE=1: one line per block row on stderr)
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_now = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    auto nap = [] { std::this_thread::sleep_for(std::chrono::microseconds(20)); };
    int next = 0, issued = 0;
    for (;;) {
        std::pair<int, hipEvent_t> job;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [&] { return !s->q.empty() || s->closing; });
            if (s->q.empty()) break;
            job = s->q.front();
            s->q.pop_front();
        }
        if (s->err != hipSuccess || s->gave_up) continue;   // (keep draining the queue)
        hipError_t e;
        while ((e = hipEventQuery(job.second)) == hipErrorNotReady) nap();   // (polling: see sink_stream_wait)
        if (e != hipSuccess) { fail(e); continue; }
        const double t_final = trace ? ms_now() : 0;
        for (int p = next; p < job.first && p < s->npanels; ++p) {
            const int64_t k = (int64_t)p * s->nb;
            const int pc = (int)((N - k) < s->nb ? (N - k) : s->nb);
            const int slot = p % SINK_BOUNCE;
            double *bn = s->bounce + (int64_t)slot * s->slot_doubles;
            { std::unique_lock<std::mutex> lk(s->wmu); s->wdone.wait(lk, [&] { return s->pending[slot] == 0; }); }
            e = hipMemcpyAsync(bn, s->stage + (int64_t)p * s->slot_doubles, (size_t)s->slot_doubles * sizeof(double), hipMemcpyDeviceToHost, s->cs);
            if (e != hipSuccess) { fail(e); break; }
            while ((e = hipStreamQuery(s->cs)) == hipErrorNotReady) nap();
            if (e != hipSuccess) { fail(e); break; }
            if (((const int *)(bn + (int64_t)N * s->nb))[0] != 0) { s->gave_up = true; break; }
            {
                std::lock_guard<std::mutex> lk(s->wmu);
                s->pending[slot] = SINK_WORKERS;
                for (int w = 0; w < SINK_WORKERS; ++w)
                    s->tasks.push_back({bn, s->host + k, N * w / SINK_WORKERS, N * (w + 1) / SINK_WORKERS, pc, slot});
            }
            s->wcv.notify_all();
            issued = p + 1;
            if (trace) fprintf(stderr, "sink: block row %d final by %.2f ms, in the bounce buffer at %.2f ms\n", p, t_final, ms_now());
        }
        if (job.first > next) next = job.first;
    }
    { std::unique_lock<std::mutex> lk(s->wmu); s->wdone.wait(lk, [&] { for (int v : s->pending) if (v) return false; return true; }); }
    s->sent = issued;
    if (trace) fprintf(stderr, "sink: %d block rows home at %.2f ms\n", issued, ms_now());
}
}  // namespace

// Before mpf_factor_dev: the next factorization of this context may give its block rows to A_host (N x N, column-major, leading
// dimension N).  Allocations happen here, outside the factorization's clock.  Returns 0 (armed), 1 (not armed: the plain copy at the
// end), or < 0 with c->err set.
int sink_attach(mpf_ctx *c, double *A_host, int64_t N, int nb) {
    if (!c->sink) { c->sink = new RowSink(); c->sink->c = c; }
    RowSink *s = c->sink;
    s->armed = false;
    const int npanels = (int)((N + nb - 1) / nb);
    if (!s->cs) MPF_HIP_TRY(c, hipStreamCreateWithFlags(&s->cs, hipStreamNonBlocking));
    const int64_t slot = N * (int64_t)nb + SINK_TAIL;
    const int64_t need_maps = (int64_t)(npanels + 2) * N, need_stage = (int64_t)npanels * slot, need_bounce = SINK_BOUNCE * slot;
    if (need_maps * 4 > (4ll << 30)) return 1;   // (very narrow panels on a very large matrix)
    auto grow = [&](auto *&ptr, int64_t &cap, int64_t need, size_t elem, bool pinned) -> int {
        if (cap >= need) return 0;
        if (ptr) (void)(pinned ? hipHostFree(ptr) : hipFree(ptr));
        ptr = nullptr; cap = 0;
        const hipError_t e = pinned ? hipHostMalloc((void **)&ptr, (size_t)need * elem) : hipMalloc((void **)&ptr, (size_t)need * elem);
        if (e != hipSuccess) { (void)hipGetLastError(); ptr = nullptr; return 1; }   // (no room: the plain copy at the end)
        cap = need;
        return 0;
    };
    if (grow(s->maps, s->maps_cap, need_maps, sizeof(int), false)) return 1;
    if (grow(s->stage, s->stage_cap, need_stage, sizeof(double), false)) return 1;
    if (grow(s->bounce, s->bounce_cap, need_bounce, sizeof(double), true)) return 1;
    while ((int)s->events.size() < npanels + 4) {
        hipEvent_t e;
        MPF_HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s->events.push_back(e);
    }
    s->host = A_host; s->N = N; s->nb = nb; s->npanels = npanels; s->slot_doubles = slot;
    s->A = nullptr; s->lda = 0;
    s->taken = false; s->closing = false; s->wclosing = false; s->ev_used = 0; s->sent = 0; s->gave_up = false; s->err = hipSuccess; s->next_asm = 0;
    s->q.clear(); s->tasks.clear();
    for (int &v : s->pending) v = 0;
    s->armed = true;
    return 0;
}

static void sink_join(RowSink *s);
// A schedule that defers its left-hand interchanges and reports its block rows (sink_notify) takes the sink here; it then skips the
// deferred pass.  false: no sink armed for this factorization.  `stream`: where the row bookkeeping starts (the schedule's main stream).
bool sink_take(mpf_ctx *c, const double *d_A, int64_t lda, int64_t N, int nb) {
    RowSink *s = c->sink;
    if (!s || !s->armed || s->taken || s->N != N || s->nb != nb) return false;
    s->A = d_A; s->lda = lda;
    sink_init_kernel<<<(int)((N + 255) / 256), 256, 0, c->stream>>>(s->maps, s->maps + N, N);   // (every schedule stream waits for this one's head)
    if (hipGetLastError() != hipSuccess) return false;
    s->taken = true;
    try {
        for (int w = 0; w < SINK_WORKERS; ++w) s->workers.emplace_back(sink_worker, s);
        s->th = std::thread(sink_thread, s);
    } catch (...) {   // (no threads to be had: the matrix goes home in one piece, and the schedule keeps its deferred interchanges)
        sink_join(s);
        s->armed = false;
        return false;
    }
    return true;
}

// What the launching thread waits for a stream with while the sink's thread is at work.  A thread parked in hipStreamSynchronize holds
// up another thread's transfers (tools/src/xfer_probe2.cpp, profiles/r05_xfer_probe2.log: a block row into pageable memory takes
// 301 ms instead of 1.3 beside a 300-ms kernel); a thread that polls does not.
hipError_t sink_stream_wait(mpf_ctx *c, hipStream_t stream) {
    RowSink *s = c->sink;
    if (!s || !s->taken) return hipStreamSynchronize(stream);
    hipError_t e;
    while ((e = hipStreamQuery(stream)) == hipErrorNotReady) std::this_thread::sleep_for(std::chrono::microseconds(50));
    return e;
}

// Block rows of the panels [0, upto_panel) are final once everything queued on `stream` so far has run: their assembly is queued
// on that stream here (the row bookkeeping of a panel, then its gather), and the sink's thread is told.
void sink_notify(mpf_ctx *c, int upto_panel, hipStream_t stream) {
    RowSink *s = c->sink;
    if (!s || !s->taken) return;
    if (upto_panel > s->npanels) upto_panel = s->npanels;
    if (upto_panel <= s->next_asm || s->ev_used >= s->events.size()) return;
    const int64_t N = s->N;
    int *Id = s->maps, *Pos = s->maps + N, *PosAll = s->maps + 2 * N;
    for (int p = s->next_asm; p < upto_panel; ++p) {
        const int64_t k = (int64_t)p * s->nb;
        const int pc = (int)((N - k) < s->nb ? (N - k) : s->nb);
        sink_step_kernel<<<1, 512, 0, stream>>>(Id, Pos, c->lists + p);
        (void)hipMemcpyAsync(PosAll + (int64_t)p * N, Pos, (size_t)N * sizeof(int), hipMemcpyDeviceToDevice, stream);
        sink_block_row_kernel<<<(int)((N + 15) / 16), 256, 0, stream>>>(s->A, s->lda, N, k, pc, s->nb, Id, PosAll, s->stage + (int64_t)p * s->slot_doubles,
                                                                          &c->ws->hp_timeouts, N * s->nb);
    }
    s->next_asm = upto_panel;
    hipEvent_t e = s->events[s->ev_used++];
    if (hipEventRecord(e, stream) != hipSuccess) return;
    { std::lock_guard<std::mutex> lk(s->mu); s->q.emplace_back(upto_panel, e); }
    s->cv.notify_one();
}

static void sink_join(RowSink *s) {
    { std::lock_guard<std::mutex> lk(s->mu); s->closing = true; }
    s->cv.notify_one();
    if (s->th.joinable()) s->th.join();
    { std::lock_guard<std::mutex> lk(s->wmu); s->wclosing = true; }
    s->wcv.notify_all();
    for (auto &w : s->workers) if (w.joinable()) w.join();
    s->workers.clear();
    s->taken = false;
}

// After mpf_factor_dev: waits for the sink's threads.  *panels_sent = block rows that reached the host (npanels: all of them).
// Returns 0; 1 when no schedule took the sink (nothing was sent: copy the matrix the plain way); -2 on a HIP error.
int sink_finish(mpf_ctx *c, int *panels_sent) {
    RowSink *s = c->sink;
    if (panels_sent) *panels_sent = 0;
    if (!s || !s->armed) return 1;
    s->armed = false;
    if (!s->taken) return 1;
    sink_join(s);
    if (panels_sent) *panels_sent = s->sent;
    if (s->err != hipSuccess) { c->err = std::string("block-row copy to the host: ") + hipGetErrorString(s->err); return -2; }
    return 0;
}

void sink_trim(mpf_ctx *c) {
    RowSink *s = c->sink;
    if (!s || s->taken) return;
    if (s->maps) hipFree(s->maps);
    if (s->stage) hipFree(s->stage);
    if (s->bounce) hipHostFree(s->bounce);
    s->maps = nullptr; s->stage = nullptr; s->bounce = nullptr; s->maps_cap = s->stage_cap = s->bounce_cap = 0;
}

void sink_destroy(mpf_ctx *c) {
    RowSink *s = c->sink;
    if (!s) return;
    if (s->taken) sink_join(s);
    sink_trim(c);
    if (s->cs) { hipStreamSynchronize(s->cs); hipStreamDestroy(s->cs); }
    for (hipEvent_t e : s->events) hipEventDestroy(e);
    delete s;
    c->sink = nullptr;
}

// ---- the way up: late column segments (mpf_factor_host with a LatePlan) ------------------------------------------------------------------
// The first part of the matrix goes up in one blocking copy from the caller's pageable memory (the runtime's own path: 56 GB/s);
// the rest follows while the factorization has started on that part: host threads gather 64-MB slabs of whole columns into two
// pinned bounce buffers, the feed thread sends one while the next is gathered (one copy in flight, completion polled with
// hipStreamQuery: an event would be a packet in a hardware queue that may be shared with the schedule's main stream), and writes the
// segment's flag -- a pinned host word the schedule's wait kernel polls -- when its last slab is in place.
namespace {
constexpr int FEED_WORKERS = 6;
}
struct ColFeed {
    mpf_ctx *c = nullptr;
    const double *host = nullptr;
    double *dA = nullptr;
    int64_t N = 0;
    LatePlan *lp = nullptr;
    std::thread th;
    std::vector<std::thread> workers;
    struct Task { double *dst; const double *src; size_t bytes; int slot; };
    std::mutex wmu;
    std::condition_variable wcv, wdone;
    std::deque<Task> tasks;
    int pending[2] = {0, 0};
    bool wclosing = false, running = false;
    hipError_t err = hipSuccess;
    hipStream_t fs = nullptr;
    double *bounce = nullptr;          // two pinned slabs
    int64_t slab_doubles = 0, bounce_cap = 0;
};

namespace {
__global__ void late_wait_kernel(const unsigned *flag, unsigned seq, int *gave_up, unsigned long long limit_ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > limit_ticks) { atomicAdd(gave_up, 1); return; }
        __builtin_amdgcn_s_sleep(64);
    }
}

void feed_worker(ColFeed *f) {
    for (;;) {
        ColFeed::Task t;
        {
            std::unique_lock<std::mutex> lk(f->wmu);
            f->wcv.wait(lk, [&] { return !f->tasks.empty() || f->wclosing; });
            if (f->tasks.empty()) return;
            t = f->tasks.front();
            f->tasks.pop_front();
        }
        memcpy(t.dst, t.src, t.bytes);
        {
            std::lock_guard<std::mutex> lk(f->wmu);
            if (--f->pending[t.slot] == 0) f->wdone.notify_all();
        }
    }
}

void feed_thread(ColFeed *f) {
    mpf_ctx *c = f->c;
    if (hipSetDevice(c->device) != hipSuccess) { f->err = hipErrorInvalidDevice; return; }
    const int64_t N = f->N;
    LatePlan *lp = f->lp;
    const int64_t slab_cols = f->slab_doubles / N;
    const auto t_feed0 = std::chrono::steady_clock::now();
    auto nap = [] { std::this_thread::sleep_for(std::chrono::microseconds(20)); };
    auto gather = [&](int64_t col0, int64_t ncols, int slot) {
        const size_t bytes = (size_t)ncols * N * sizeof(double);
        const char *src = (const char *)(f->host + col0 * N);
        char *dst = (char *)(f->bounce + (int64_t)slot * f->slab_doubles);
        std::lock_guard<std::mutex> lk(f->wmu);
        f->pending[slot] = FEED_WORKERS;
        for (int w = 0; w < FEED_WORKERS; ++w) {
            const size_t a = bytes * w / FEED_WORKERS / 4096 * 4096, b = w + 1 == FEED_WORKERS ? bytes : bytes * (w + 1) / FEED_WORKERS / 4096 * 4096;
            f->tasks.push_back({(double *)(dst + a), (const double *)(src + a), b - a, slot});
        }
        f->wcv.notify_all();
    };
    for (int sg = 0; sg < lp->nseg && f->err == hipSuccess; ++sg) {
        const int64_t c_lo = lp->c0[sg], c_hi = sg + 1 < lp->nseg ? lp->c0[sg + 1] : N;
        const int64_t nslab = (c_hi - c_lo + slab_cols - 1) / slab_cols;
        auto cols_of = [&](int64_t i) { const int64_t a = c_lo + i * slab_cols; return std::pair<int64_t, int64_t>(a, (c_hi - a) < slab_cols ? (c_hi - a) : slab_cols); };
        if (nslab > 0) { const auto s0 = cols_of(0); gather(s0.first, s0.second, 0); }
        for (int64_t i = 0; i < nslab; ++i) {
            const int slot = (int)(i & 1);
            { std::unique_lock<std::mutex> lk(f->wmu); f->wdone.wait(lk, [&] { return f->pending[slot] == 0; }); }
            if (i + 1 < nslab) { const auto s1 = cols_of(i + 1); gather(s1.first, s1.second, slot ^ 1); }   // (its last copy was waited for below)
            const auto si = cols_of(i);
            hipError_t e = hipMemcpyAsync(f->dA + si.first * N, f->bounce + (int64_t)slot * f->slab_doubles, (size_t)si.second * N * sizeof(double), hipMemcpyHostToDevice, f->fs);
            if (e == hipSuccess) while ((e = hipStreamQuery(f->fs)) == hipErrorNotReady) nap();
            if (e != hipSuccess) { f->err = e; break; }
        }
        if (f->err == hipSuccess) __atomic_store_n(&lp->flags[sg], lp->seq, __ATOMIC_RELEASE);
        if (c->tune.sink_trace) fprintf(stderr, "feed: segment %d (columns %lld..%lld, %.2f GB) in place %.2f ms after the feed's start\n", sg, (long long)c_lo, (long long)c_hi,
                                        (double)(c_hi - c_lo) * N * 8 / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_feed0).count());
    }
    { std::unique_lock<std::mutex> lk(f->wmu); f->wdone.wait(lk, [&] { return f->pending[0] == 0 && f->pending[1] == 0; }); }
}
}  // namespace

int launch_late_wait(mpf_ctx *c, const unsigned *flag, unsigned seq) {
    unsigned *dflag = nullptr;
    MPF_HIP_TRY(c, hipHostGetDevicePointer((void **)&dflag, (void *)flag, 0));
    late_wait_kernel<<<1, 1, 0, c->stream>>>(dflag, seq, &c->ws->flags[1], 20ull * 100000000ull);   // 20 s of the 100-MHz clock
    MPF_HIP_TRY(c, hipGetLastError());
    return 0;
}

int feed_start(mpf_ctx *c, const double *A_host, double *d_A, int64_t N, LatePlan *lp) {
    if (!c->feed) { c->feed = new ColFeed(); c->feed->c = c; }
    ColFeed *f = c->feed;
    if (f->running) { c->err = "feed_start: already running"; return -1; }
    if (!f->fs) MPF_HIP_TRY(c, hipStreamCreateWithFlags(&f->fs, hipStreamNonBlocking));
    int64_t slab_cols = (64ll << 20) / (N * 8);
    if (slab_cols < 1) slab_cols = 1;
    const int64_t need = 2 * slab_cols * N;
    if (f->bounce_cap < need) {
        if (f->bounce) (void)hipHostFree(f->bounce);
        f->bounce = nullptr; f->bounce_cap = 0;
        MPF_HIP_TRY(c, hipHostMalloc((void **)&f->bounce, (size_t)need * sizeof(double)));
        f->bounce_cap = need;
    }
    f->slab_doubles = slab_cols * N;
    f->host = A_host; f->dA = d_A; f->N = N; f->lp = lp; f->err = hipSuccess; f->wclosing = false;
    f->tasks.clear(); f->pending[0] = f->pending[1] = 0;
    f->running = true;
    try {
        for (int w = 0; w < FEED_WORKERS; ++w) f->workers.emplace_back(feed_worker, f);
        f->th = std::thread(feed_thread, f);
    } catch (...) {
        { std::lock_guard<std::mutex> lk(f->wmu); f->wclosing = true; }
        f->wcv.notify_all();
        for (auto &w : f->workers) if (w.joinable()) w.join();
        f->workers.clear();
        f->running = false;
        c->err = "feed_start: could not start the upload threads";
        return -2;
    }
    return 0;
}

int feed_finish(mpf_ctx *c) {
    ColFeed *f = c->feed;
    if (!f || !f->running) return 0;
    if (f->th.joinable()) f->th.join();
    { std::lock_guard<std::mutex> lk(f->wmu); f->wclosing = true; }
    f->wcv.notify_all();
    for (auto &w : f->workers) if (w.joinable()) w.join();
    f->workers.clear();
    f->running = false;
    if (f->err != hipSuccess) { c->err = std::string("upload of the late column segments: ") + hipGetErrorString(f->err); return -2; }
    return 0;
}

void feed_trim(mpf_ctx *c) {
    ColFeed *f = c->feed;
    if (!f || f->running) return;
    if (f->bounce) hipHostFree(f->bounce);
    f->bounce = nullptr; f->bounce_cap = 0;
}

void feed_destroy(mpf_ctx *c) {
    ColFeed *f = c->feed;
    if (!f) return;
    (void)feed_finish(c);
    feed_trim(c);
    if (f->fs) { hipStreamSynchronize(f->fs); hipStreamDestroy(f->fs); }
    delete f;
    c->feed = nullptr;
}

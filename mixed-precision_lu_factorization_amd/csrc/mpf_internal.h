// Internal declarations shared by the HIP kernels and the C++ host driver of libmpf_amd.so.
// gfx950 (MI355X) only: 64-lane wavefronts, 160 KiB LDS per CU, 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <functional>
#include "../../include/mpf_c.h"

// ---- fp16 pivot panel geometry (fp16_panel.hip) ---------------------------------------------
constexpr int HP_R = 256;              // panel rows owned by one workgroup (its LDS slab)
constexpr int HP_T = 512;              // threads per workgroup
constexpr int HP_MAXG = 256;           // max workgroups = CUs: all must be co-resident
constexpr int HP_MAXCOLS = 256;        // max panel width
constexpr int LASWP_MAXMOVED = 2 * HP_MAXCOLS;

// Rows that really move under one panel's interchanges: content of row src[i] goes to row dst[i] (global rows).
struct MovedList {
    int n;
    int pad[3];
    int src[LASWP_MAXMOVED];
    int dst[LASWP_MAXMOVED];
};

// Device workspace owned by a context, zeroed once at mpf_create.  Hand-off granules carry the launch sequence number of the
// pivot kernel in their tags, so nothing here is cleared between launches.
struct MpfWorkspace {
    unsigned long long cand[2][HP_MAXG];   // {tag:31 | abs:15 | ~tiekey:17} per workgroup, tag = launch sequence << 9 | column + 1
    int flags[16];                         // [0]: give-ups of the triangular solves' bounded waits (ir.hip; must stay 0); rest spare
    unsigned long long rowbuf[2][HP_MAXG][HP_MAXCOLS / 2]; // candidate pivot rows: {tag32 | 2 x fp16} granules
    int hp_timeouts;                       // spin give-ups inside the pivot kernel (must stay 0)
    int pad0[3];
    MovedList list0;                       // moved-row list of the stand-alone mpf_laswp (built by laswp_plan)
    int info;                              // first zero pivot in the fp64 panel (1-based) or INT_MAX
    int pad[3];
    unsigned long long hp_stamps[8];       // diagnostic build of the pivot kernel (MPF_HP_STAMP=1): cycles per segment
    unsigned long long hp_progress;        // {launch sequence:32 | columns whose pivots are final:32}, published by workgroup 0
    unsigned long long hp_xcd_target;      // single-XCD form of the pivot kernel: {launch sequence:32 | XCC id + 1} of the XCD whose workgroups take the panel
    unsigned long long hp_xcd_roles;       // ... {launch sequence:32 | workgroups of that XCD that have taken a slab}
};

// Behaviour switches of a context.  Defaults come from the environment ONCE, at mpf_create (the MPF_* variable named with
// each field); after that they are plain per-context state, changed through mpf_set_option (include/mpf_c.h) -- two contexts
// on two host threads never share or race on them.
struct MpfTuning {
    int safe_pivots = 0;                 // MPF_SAFE_PIVOTS=1: generic (never-waiting) pivot path and schedule always
    int chain_pipeline = 1;              // MPF_CHAIN_PIPELINE=0: the fp64 panel waits for the whole pivot kernel
    long long chain_pipeline_below = 10240; // MPF_CHAIN_PIPELINE_BELOW: fp64 mode pipelines the chain only below this trailing size (re-tuned in round 4
                                            // after the pivot kernel got faster: 8192 .. 14336 within 1 ms of each other, 18432 + 3 ms, 0 + 9 ms)
    int fp16_work32 = 1;                 // MPF_FP16_WORK32=0: fp16 modes update the fp64 matrix in place (no fp32 working copy)
    int superpanel_fp16 = 0;             // MPF_SUPERPANEL: panels per super-panel of the fp16 modes (mpf_opts.superpanel = 0): 1 .. 8, or 0 = automatic --
                                         // plain fp16 operands from N = 24576 on: 6, else 4 (fp16x3: 4; mpf_factor_dist: the same rule).  A wider super-panel feeds
                                         // the big-K update a longer K (fewer passes over the fp32 copy: 520 -> 628 TFLOP/s per launch in the schedule
                                         // at N = 32768) and costs fp64 work in the inner region (125.4 -> 127.9 ms there, 467.9 -> 456.9 at N = 65536;
                                         // fp16x3 loses at every width above 4): profiles/r05_superpanel_width.log
    int superpanel_fp64 = 1;             // MPF_SUPERPANEL_FP64: the same for the fp64 mode
    int no_lookahead = 0;                // MPF_NO_LOOKAHEAD=1: single-stream schedule
    int verbose = 0;                     // MPF_VERBOSE=1: per-panel line (MPF.cu:137) from the drop-in MPF()
    int timeline = 0;                    // MPF_TIMELINE=1: every timed region as (start, end) on stderr
    long long hp_spin_limit = 1ll << 21; // MPF_HP_SPIN_LIMIT: bound of every cross-workgroup spin of the LDS pivot kernel (polls)
    long long hp_gate_ticks = 200000000ll; // MPF_HP_GATE_TICKS: bound of a gate's wait for the pivot kernel (100 MHz ticks: 2 s)
    int hp_window = -1;                  // MPF_HP_WINDOW: pivot panels on the column-window kernel (76 KB of LDS, shares its CU) instead of the
                                         // full-slab one (a CU per workgroup): 0 never, n >= 1 for panels of at least n rows, -1 automatic --
                                         // 20 000 rows under an fp64 trailing update (whose 68-KB workgroups fit beside it: -1 % factor time), under the fp16
                                         // ones only for panels that leave fewer than 72 CUs free (above (CUs - 72) x 256 = 47 104 rows), where the
                                         // full-slab form deadlocks against the chain's gated kernels (measured)
    int hp_acq_fence = 0;                // MPF_HP_ACQ_FENCE=1: agent-scope acquire after the hand-off poll (debug aid)
    int hgemm_pad = 0;                   // MPF_HGEMM_PAD: unused dynamic LDS (bytes) of the plain fp16 update kernel (occupancy cap)
    int hgemm_split_pad = 32768;         // MPF_HGEMM_SPLIT_PAD: the same for the split-operand kernel (two workgroups per CU)
    int hgemm_big = 1;                   // MPF_HGEMM_BIG=0: the 128 x 128-tile fp16 update kernel for every shape (A/B switch)
    int hgemm_mfma16 = 1;                // MPF_HGEMM_MFMA16=0: plain-operand fp16 updates on round 4's v_mfma_f32_32x32x16_f16 kernels (trailing_f16.hip,
                                         // hgemm_pp.hip) instead of the 16x16x32 ones (hgemm16.hip); A/B switch -- the two families differ in bits
                                         // (different summation order inside an MFMA), each is self-consistent across its kernels
    int hgemm_big_tile = 0;              // MPF_HGEMM_BIG_TILE: form of the big-K fp16 update (launch_hgemm_ptrs): 0 = automatic, 1 / 2 = 128 x 256 tiles (one / two
                                         // workgroups per CU), 3 = round 3's kernel, 4 = hgemm_big_kernel everywhere, 5 = hgemm_pp_kernel everywhere
    int dgemm_dma = 1;                   // MPF_DGEMM_DMA=0: register-staged eight-wave fp64 update kernel (same bits)
    int lazy_gather = 1;                 // MPF_LAZY_GATHER=0: deferred left-hand interchanges as scattered writes
    int dpanel_fused_form = 1;           // MPF_DPANEL_FUSED=0: fp64 panel without the fused update + sub-panel launches
    int dist_instalments = 1;            // MPF_DIST_INSTALMENTS=0: the panel message of mpf_factor_dist always travels in one broadcast
    long long dist_instalment_min_bytes = 8ll << 20; // MPF_DIST_INSTALMENT_MIN_BYTES: panels below this go in one broadcast
    int generic_fused = 1;               // MPF_GENERIC_FUSED=0: generic pivot path with four launches per column instead of two
    int fp64_rowmajor = 1;               // MPF_FP64_ROWMAJOR=0: fp64 mode updates the column-major matrix in place (no row-major working copy)
    long long fp64_rowmajor_min_n = 8192;// MPF_FP64_ROWMAJOR_MIN_N: smaller matrices stay in place (the copy's extra launches cost more than they save)
    int trsm_laswp_fused = 1;            // MPF_TRSM_LASWP_FUSED=0: interchange and TRSM right of the strip as two launches
    int fp64_two_lanes = 8192;           // MPF_FP64_TWO_LANES: fp64 row-major schedule splits the update over two lanes while at least this many
                                         // columns lie right of the strip and the chain is not pipelined (0: always one lane)
    int dist_solve_p2p = 0;              // MPF_DIST_SOLVE_P2P=1: the distributed triangular solves pass the vector from owner to owner over ncclSend / ncclRecv
                                         // instead of broadcasting after every block.  Off by default until a run on two or more GPUs has covered it
                                         // (ADVICE r4); the ranks vote, and the chain is taken only when every rank has it
    int host_sink = 1;                   // MPF_HOST_SINK=0: mpf_factor_host / MPF() copy the factors back in one piece after the factorization (as the
                                         // reference does) instead of block row by block row while it runs (rowsink.hip)
    int hp_local_xcd = 1;                // MPF_HP_LOCAL_XCD: panels of at most (CUs / 8) slabs run their pivot kernel on the workgroups of ONE XCD (hand-offs through that
                                         // XCD's L2: plain stores; fp16_panel.hip): 0 never, 1 in the fp16 modes' schedules and the step operator, 2 always
    int hp_half_slabs = 1;               // MPF_HP_HALF_SLABS=0: the pivot kernel keeps 256-row slabs everywhere (1: 128-row slabs for 256-column panels of <= 16384 rows in the
                                         // fp16 modes' schedules and the step operator; fp16_panel.hip)
    int hp_half_slabs_rows = 16384;      // MPF_HP_HALF_SLABS_ROWS: ... up to this many rows
    int sink_trace = 0;                  // MPF_SINK_TRACE=1: the block-row sink prints one line per block row on stderr (when final, when home)
    int host_late_parts = 3;             // MPF_HOST_LATE_PARTS: column segments of the matrix that go up WHILE mpf_factor_host's factorization has started on the
                                         // first part (0: the whole matrix first, as MPF.cu:82); fp64 row-major schedule only
    long long host_late_min_n = 16384;   // MPF_HOST_LATE_MIN_N: ... and from this size on
    int host_first_pct = 25;             // MPF_HOST_FIRST_PCT: share of the columns in that first part
    int host_late_q_pct = 110;           // MPF_HOST_LATE_Q_PCT: percent of the estimated arrival time of a late segment at which it is planned in (mpf_factor_host)
    long long host_sink_min_n = 4096;    // MPF_HOST_SINK_MIN_N: smaller matrices always go back in one piece
    int gate_wait_value = 0;             // MPF_GATE_WAIT_VALUE=1: the pipelined chain's gated interchange as hipStreamWaitValue64 on the pivot kernel's progress
                                         // word (signal memory) + an ungated kernel, instead of a kernel that spins on CUs (measured round 5: DESIGN 4.1)
    int gesv_fp64_tflops = 0;            // MPF_GESV_FP64_TFLOPS: fp64 factorization rate mpf_gesv(try_fp16 = 3) prices GMRES-IR's time limit with; 0 = this
                                         // context's last measured fp64-mode factorization (N >= 8192), 50 before there is one
    int dist_world1_loop = 0;            // MPF_DIST_WORLD1_LOOP=1: mpf_factor_dist with ONE rank runs the distributed loop (tests) instead of handing over to mpf_factor_dev
    int event_timers = 1;                // MPF_EVENT_TIMERS: HIP-event pairs around 2 = every timed region (mpf_stats.ms_hpanel ... ms_cvt), 1 = the
                                         // trailing updates only (ms_gemm, ms_gemm_big; default), 0 = none; option timeline implies 2
    int fp64_lane_a_pct = 50;            // MPF_FP64_LANE_A_PCT: share of those columns in lane A at a (re-)split; re-split 10 points below.  Lane B's
                                         // update has to cover lane A's small launches of the next panel (measured: 60 % and more lose)
#ifdef MPF_PROBE                         // libmpf_probe.so only (tools/): measured-slower variants and diagnostics
    int hp_stamp = 0;                    // MPF_HP_STAMP=1: cycle-stamped build of the pivot kernel
    int hp_r256_upto = 1 << 30;          // MPF_HP_R256_UPTO: panels above that many rows use 128-row workgroups
    int dgemm_w8 = 1;                    // MPF_DGEMM_W8=0: four-wave fp64 update kernel everywhere
    int gemm_lds_pad = 0;                // MPF_GEMM_LDS_PAD: extra dynamic LDS of the fp64 update kernels
    int hgemm_dbg = 0;                   // MPF_HGEMM_DBG: big fp16 update 1 = K loop only, 2 = C stream only (timing probes; results wrong)
    int hgemm_big_reg = 0;               // MPF_HGEMM_BIG_REG=1: the big fp16 update stages its operands through registers instead of by LDS-DMA
#endif
};

struct mpf_ctx {
    MpfTuning tune;
    unsigned attr_done = 0;            // hipFuncSetAttribute done for this context's device, one bit per kernel family
    unsigned attr_big = 0;             // the same for the instantiations of hgemm_big_kernel (bit = launch slot, + 8 for the fp32 copy)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t pstream = nullptr;     // high-priority stream of the look-ahead panel chain
    hipStream_t tstream = nullptr;     // second chain stream: the fp64 panel follows the pivot kernel 32 columns behind
    hipStream_t xstream = nullptr;     // mpf_factor_dist: exchange stream of the panel message's instalments (created on demand)
    std::vector<hipEvent_t> ev_pool;   // reusable events (dependencies + per-launch timing)
    MpfWorkspace *ws = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    mpf_stats stats{};
    int num_cus = 0;
    // scratch for the solve path (grown on demand)
    double *solve_buf = nullptr;
    int64_t solve_n = 0;
    unsigned short *h_L = nullptr, *h_U = nullptr; // fp16 operand images of the fp16 trailing mode
    int64_t h_rows = 0;
    unsigned short *h_Lb[2] = {nullptr, nullptr}; // L images of the deferred K = sb * nb updates (two super-panels in flight)
    int h_kmax = 0;                     // K capacity (columns) of the fp16 operand images
    unsigned hp_seq = 0;               // launch sequence number of the pivot kernel (row-granule tags)
    unsigned long long *hp_signal = nullptr;   // 8 bytes of signal memory (hipMallocSignalMemory): the progress word for hipStreamWaitValue64
    int32_t *perm_buf = nullptr;
    MovedList *lists = nullptr;        // one moved-row list per panel of the running factorization
    int lists_cap = 0;
    int *Fmap = nullptr;               // composite row map of the deferred left-hand interchanges
    double *perm_tmp = nullptr;        // N x nb scratch of the same
    int64_t perm_cap = 0, fmap_cap = 0;
    double *trsv_inv = nullptr;        // inverted 64x64 diagonal blocks of L then of U (solve path)
    double *trsv_inv256 = nullptr;     // full inverses of the 256 x 256 diagonal blocks of L then of U (single-GPU solve)
    int *trsv_cnt = nullptr;           // per-step counters of the solve's launches (near workgroups done), L steps then U steps
    double *krylov = nullptr;          // GMRES-IR: (restart + 1) basis vectors
    size_t krylov_cap = 0;             // doubles
    double *res_part = nullptr;        // per-column-chunk partial sums of the residual (deterministic reduction)
    size_t res_part_cap = 0;           // doubles
    // factored 32x32 diagonal tiles of the fp64 panel, parked here until every workgroup of the sub-panel launches has
    // read the UNfactored tile from the matrix (dpanel.hip); one tile per 32 panel columns, grown on demand
    double *dtiles = nullptr;
    int dtiles_cap = 0;                // tiles
    float *w32 = nullptr;              // fp32 working copy of the trailing matrix (fp16 trailing modes, two-level schedule)
    int64_t w32_n = 0;
    double *r64 = nullptr;             // fp64 ROW-major working copy of the trailing matrix (fp64 mode, factor_lookahead_rm)
    int64_t r64_n = 0;                 // the size it was last used for
    int64_t r64_cap = 0;               // its capacity (doubles)
    bool hgemm_standalone = false;     // set around the fp16 update of a step operator (no panel chain beside it: the persistent kernel may take every CU)
    mpf_p2p_fn p2p_fn = nullptr;       // point-to-point transport of the distributed solves (mpf_dist_set_p2p); null: RCCL's, or none
    void *p2p_user = nullptr;
    double gmres_budget_ms = 0;        // wall-clock limit of mpf_solve_gmres_ir while mpf_gesv runs it (0: none)
    double fp64_rate_tflops = 0;       // last measured fp64-mode factorization rate of this context (N >= 8192; 0: none yet)
    double *host_A = nullptr;          // mpf_factor_host's device copy of the caller's matrix, kept between calls (grow-only)
    int64_t host_A_cap = 0;            // bytes
    int32_t *host_P = nullptr;         // ... and of the pivot vector
    int64_t host_P_cap = 0;
    double *host_A0 = nullptr;         // ... and the matrix as uploaded, while block rows leave during the factorization (rowsink.hip): what a
    int64_t host_A0_cap = 0;           // repeated call on the generic pivot path starts from when the caller's buffer is already partly results
    unsigned *late_flags = nullptr;    // 16 pinned host words: LatePlan::flags
    unsigned late_seq = 0;
    struct ColFeed *feed = nullptr;    // mpf_factor_host's upload of late column segments (rowsink.hip)
    struct LatePlan *late = nullptr;   // set by mpf_factor_host around mpf_factor_dev: column segments still on their way up (factor_lookahead_rm takes it)
    struct RowSink *sink = nullptr;    // mpf_factor_host's block-row copies (rowsink.hip); null until the first call that uses it
    double *rm_tmp = nullptr;          // its scratch: moved rows of an interchange (2 * HP_MAXCOLS x N) / the panel's L21 row-major
    int64_t rm_tmp_cap = 0;            // doubles
    double *rm_lt = nullptr;           // L21 of the current panel, row-major [rows][nb]
    int64_t rm_lt_cap = 0;
    // generic (global-memory) fp16 pivot path, fp16_panel_generic.hip: packed fp16 panel + per-block candidates
    unsigned short *g16 = nullptr;
    size_t g16_cap = 0;                // elements
    unsigned long long *gcand = nullptr;
    int gcand_cap = 0;
    // distributed path (mpf_dist.cpp): RCCL communicator (dlopen'ed), two panel message buffers
    void *rccl_comm = nullptr;
    int rccl_rank = 0, rccl_world = 0;
    long long rccl_bcast_calls = 0, rccl_bcast_bytes = 0, rccl_allreduce_calls = 0, rccl_p2p_calls = 0, rccl_p2p_bytes = 0;   // since mpf_rccl_init
    double *dist_buf[2] = {nullptr, nullptr};
    size_t dist_buf_cap = 0;           // bytes
    // mpf_factor_dist, two-level schedule of the fp16 modes: this rank's fp32 working copy (N rows x local columns, row-major) and
    // the current super-panel's panels as every rank received them (N x sb * nb doubles, column-major, leading dimension N)
    float *dist_w32 = nullptr;
    int64_t dist_w32_cap = 0;          // floats
    double *dist_spl = nullptr;
    int64_t dist_spl_cap = 0;          // doubles
    int hp_resident_per_cu = -1;       // occupancy query of the LDS pivot kernel (cached)
    int hp_win_per_cu = 0;             // ... of its column-window form
    int hp_full_beside_waiter = 0;     // pivot workgroups (full-slab / column-window form) that still fit on a CU that holds one workgroup of
    int hp_win_beside_waiter = 0;      // the gated interchange kernel (from the kernels' own LDS / register footprints)
};

#define MPF_HIP_TRY(ctx, expr)                                                        \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);           \
            return -2;                                                                \
        }                                                                             \
    } while (0)

enum { ATTR_HP = 1, ATTR_DGEMM = 2, ATTR_HGEMM = 4, ATTR_HGEMM256 = 8, ATTR_HGEMM_PP = 16, ATTR_HGEMM16_BIG32 = 32, ATTR_HGEMM16_BIG64 = 64 };
inline bool safe_pivots(const mpf_ctx *c) { return c->tune.safe_pivots != 0; }

// ---- launchers implemented in the .hip files (all asynchronous on `s`) -----------------------
int launch_double_to_fp16(mpf_ctx *c, const double *in, uint16_t *out, int64_t n);
int launch_hdiv(mpf_ctx *c, const uint16_t *a, const uint16_t *b, uint16_t *q, int64_t n);
// LDS-resident pivot kernel (cols <= 256, rows <= 256 x resident workgroups); hgetf2_lds_eligible says whether it may run
// waiters: workgroups of kernels that will WAIT for this launch's progress while it runs (the pipelined chain's gated interchange:
// laswp_gated_grid); they hold CUs the pivot kernel's workgroups -- which must all be resident at once -- cannot use.
// prefer_window_rows > 0: panels of at least that many rows take the column-window form where it fits (fp64 schedules: it leaves
// half of each CU to the update).  The form is chosen per call from these and the occupancy figures: no state is kept between calls.
int launch_hgetf2(mpf_ctx *c, const double *A64, int64_t lda, uint16_t *P16, int64_t ld16, int rows,
                  int cols, int ipiv_offset, int *d_ipiv, uint16_t *out16, int64_t ldo, MovedList *moved, int waiters = 0, int prefer_window_rows = 0);
bool hgetf2_lds_eligible(mpf_ctx *c, int rows, int cols);
// can a panel of that shape run (in either form) beside `waiters` waiting workgroups?  Rows either form can take beside them.
bool hgetf2_fits_beside(mpf_ctx *c, int rows, int cols, int waiters);
long long hgetf2_capacity_rows(mpf_ctx *c, int waiters, int form = 0);
int laswp_gated_footprint(int *lds_bytes, int *vgprs, int *threads);
int laswp_gated_grid(int64_t ncols);
constexpr int HP_FP64_WINDOW_ROWS = 20000;   // fp64 schedules: the column-window form from that many panel rows on (-1 % factor time at N = 32768)
int hp_query_residency(mpf_ctx *c);
// generic pivot path (any shape, no cross-workgroup spinning) and the reference-style sequential interchange that goes with it
// wait (on c->stream, bounded) until the most recent launch_hgetf2 of this context has fixed the pivots of `target` columns
int launch_hgetf2_gate(mpf_ctx *c, int target);
// cols (<= 256) sequential swaps (row k + j <-> d_ipiv[j] - 1) on ncols columns, resolved and applied in one launch
int launch_laswp_block(mpf_ctx *c, double *A, int64_t lda, int64_t ncols, int k, int cols, const int *d_ipiv, int64_t nrows);
int launch_laswp_block_gated(mpf_ctx *c, double *A, int64_t lda, int64_t ncols, int k, int cols, const int *d_ipiv, int64_t nrows, int target);
// one piece of launch_dgetf2_npv: piece 0 = the first 32-column sub-panel, piece s = fused update + sub-panel s; the last piece
// also puts the parked diagonal tiles back.  Only for cols % 32 == 0, cols >= 64 (dgetf2_npv_pieces(cols) > 0).
int dgetf2_npv_pieces(mpf_ctx *c, int cols);
int launch_dgetf2_npv_piece(mpf_ctx *c, double *P, int64_t ld, int rows, int cols, int fused, int info_base, int piece);
int launch_dpanel_tile_copy(mpf_ctx *c, double *dst, int64_t ld, int piece);   // parked factored diagonal tile of a finished piece
int launch_hgetf2_generic(mpf_ctx *c, const double *A64, int64_t lda, uint16_t *P16, int64_t ld16, int rows, int cols,
                          int ipiv_offset, int *d_ipiv, uint16_t *out16, int64_t ldo);
int launch_laswp_seq(mpf_ctx *c, double *A, int64_t lda, int64_t ncols, int k, int cols, const int *d_ipiv, int64_t nrows);
// row interchange of `ncols` columns from a moved-row list left by the pivot kernel (fused LASWP plan)
int launch_laswp_from_list(mpf_ctx *c, double *A, int64_t lda, int64_t ncols, const MovedList *ml);
int launch_laswp_from_list_hole(mpf_ctx *c, double *A, int64_t lda, int64_t ncols, const MovedList *ml, int64_t hole_from, int64_t hole_len);
// deferred interchanges of everything LEFT of each panel: one composite permutation per column block, applied at
// the end of the factorization (lists[p] = moved rows of panel p, p = 0..npanels-1)
int launch_lazy_left_swaps(mpf_ctx *c, double *A, int64_t lda, int64_t N, int nb, int npanels, const MovedList *lists, int sb = 1,
                           int world = 1, int rank = 0);
// mpf_factor_host -> factor_lookahead_rm: the matrix's columns from c0[0] on are still being uploaded when the factorization starts.
// Segment i = columns [c0[i], c0[i + 1]) (the last one ends at N); the panels [0, q[i]) are factored on the columns left of it, then
// the segment -- in d_A once flags[i] == seq -- receives those panels' interchange / TRSM / update one after the other (per element
// the same operations in the same order: same bits), and the loop goes on with the wider matrix.
struct LatePlan {
    int nseg = 0;
    int64_t c0[4] = {0, 0, 0, 0};
    int q[4] = {0, 0, 0, 0};
    unsigned *flags = nullptr;         // pinned host words the device can read
    unsigned seq = 0;
    double *snapshot = nullptr;        // where the uploaded matrix is kept (mpf_ctx::host_A0), or null
    bool taken = false;
    bool snapped[4] = {false, false, false, false};
};
int launch_late_wait(mpf_ctx *c, const unsigned *flag, unsigned seq);   // the stream waits (bounded) until *flag == seq; a give-up lands in ws->flags[1]
int feed_start(mpf_ctx *c, const double *A_host, double *d_A, int64_t N, LatePlan *lp);   // uploads the plan's segments from a thread of its own
int feed_finish(mpf_ctx *c);           // joins it; 0 or -2
void feed_destroy(mpf_ctx *c);
void feed_trim(mpf_ctx *c);
// rowsink.hip: finished block rows of the factors to the caller's host matrix while the factorization runs (mpf_factor_host)
int sink_attach(mpf_ctx *c, double *A_host, int64_t N, int nb);
bool sink_take(mpf_ctx *c, const double *d_A, int64_t lda, int64_t N, int nb);
void sink_notify(mpf_ctx *c, int upto_panel, hipStream_t stream);
hipError_t sink_stream_wait(mpf_ctx *c, hipStream_t stream);   // hipStreamSynchronize, or polling while the sink's thread copies
int sink_finish(mpf_ctx *c, int *panels_sent);
void sink_destroy(mpf_ctx *c);
void sink_trim(mpf_ctx *c);
int launch_laswp(mpf_ctx *c, double *A, int64_t lda, int64_t ncols, int k, int cols, const int *d_ipiv);
// resolve a panel's sequential swap list (<= 256 swaps, global 1-based pivots) into a moved-row list
int launch_laswp_plan(mpf_ctx *c, const int *d_ipiv, int k, int cols, MovedList *out);
int launch_dgetf2_npv(mpf_ctx *c, double *P, int64_t ld, int rows, int cols, int fused, int info_base);
int launch_dtrsm_llnu(mpf_ctx *c, int m, int64_t n, const double *L, int64_t ldl, double *B, int64_t ldb);
int launch_dgemm_minus(mpf_ctx *c, int64_t m, int64_t n, int k, const double *A, int64_t lda,
                       const double *B, int64_t ldb, double *C, int64_t ldc);
// fp16-in / fp32-accumulate trailing update (trailing_f16.hip); operand images live in c->h_L / c->h_U
int launch_cvt_l21(mpf_ctx *c, const double *A, int64_t lda, int64_t m, int K, int split, int img = 0, int64_t elem_off = 0);
int launch_hgemm_minus(mpf_ctx *c, int64_t m, int64_t n, int K, const double *B, int64_t ldb, double *C, int64_t ldc,
                       int split, int img = 0, int64_t elem_off = 0);
int launch_hgemm_minus_w32(mpf_ctx *c, int64_t m, int64_t n, int K, const double *B, int64_t ldb, float *C, int64_t ldc,
                           int split, int img = 0, int64_t elem_off = 0);
// the two halves of launch_hgemm_minus[_w32]: U12 -> fp16 image in c->h_U, then the MFMA kernel on ready images
// (images may be blocks of wider ones: element offsets into the buffers and row strides; 0 = the padded K itself)
int launch_cvt_u12(mpf_ctx *c, const double *B, int64_t ldb, int K, int64_t n, int split, int64_t elem_off = 0, int kstride = 0);
int launch_hgemm_images(mpf_ctx *c, int64_t m, int64_t n, int K, void *C, int64_t ldc, bool c32, int split, int img = 0, int64_t elem_off = 0,
                        int64_t u_off = 0, int ksL = 0, int ksU = 0);
struct HgemmImages { const unsigned short *Lh = nullptr, *Ll = nullptr, *Uh = nullptr, *Ul = nullptr; int ksL = 0, ksU = 0; };
int launch_hgemm_ptrs(mpf_ctx *c, int64_t m, int64_t n, int K, const HgemmImages &im, void *C, int64_t ldc, bool c32, int split);
// hgemm16.hip: the plain-operand updates on v_mfma_f32_16x16x32_f16 (big-K tile kernel; fp32 copy or fp64 matrix)
int launch_hgemm16_big(mpf_ctx *c, int64_t m, int64_t n, int Kp, const HgemmImages &im, void *C, int64_t ldc, bool c32);
int launch_hgemm16_ring(mpf_ctx *c, int64_t m, int64_t n, int Kp, const HgemmImages &im, void *C, int64_t ldc, bool c32);
// hgemm_pp.hip: the big-K update on the fp32 copy (persistent workgroups, ping-pong wave groups); Kp = padded K
int launch_hgemm_pp(mpf_ctx *c, int64_t m, int64_t n, int Kp, const HgemmImages &im, float *C, int64_t ldc);
// C is ROW-major fp32 (element (i, j) at Crm[i * ldrow + j]): the fp32 working copy of the two-level schedule
int launch_hgemm_images_rowmajor(mpf_ctx *c, int64_t m, int64_t n, int K, float *Crm, int64_t ldrow, int split, int img = 0, int64_t elem_off = 0,
                                 int64_t u_off = 0, int ksL = 0, int ksU = 0);
int launch_laswp_from_list_f32(mpf_ctx *c, float *A, int64_t lda, int64_t ncols, const MovedList *ml);
// fp64 row-major working copy of the fp64 mode (factor_lookahead_rm): interchange of contiguous rows, window transposes
int launch_laswp_from_list_rm64(mpf_ctx *c, double *R, int64_t ldr, int64_t ncols, const MovedList *ml, int64_t scratch_off = 0);
int launch_transpose64(mpf_ctx *c, double *A, int64_t lda, double *R, int64_t ldr, int64_t rows, int64_t cols, bool to_rowmajor);
// dtrsm_llnu on a right-hand side with arbitrary strides: element (row, col) at B[row * rs + col * cs]
int launch_dtrsm_llnu_strided(mpf_ctx *c, int m, int64_t n, const double *L, int64_t ldl, double *B, int64_t rs, int64_t cs);
int launch_cvt_f64_f32(mpf_ctx *c, const double *A, int64_t lda, float *W, int64_t ldw, int64_t rows, int64_t cols);
int launch_cvt_f32_f64(mpf_ctx *c, const float *W, int64_t ldw, double *A, int64_t lda, int64_t rows, int64_t cols);
// solve helpers (ir.hip)
int launch_gather_rows(mpf_ctx *c, const double *in, const int *perm, double *out, int64_t n);
int launch_residual(mpf_ctx *c, const double *A, int64_t lda, const double *x, const double *b, double *r,
                    int64_t n);
int launch_residual_rect(mpf_ctx *c, const double *A, int64_t lda, const double *x, const double *b, double *r, int64_t n, int64_t ncols);
int launch_trsv_prepare_cols(mpf_ctx *c, const double *LUb, int64_t ld, int64_t n, int64_t k0, int w);
int launch_trsv_lower_cols(mpf_ctx *c, const double *LUb, int64_t ld, double *x, int64_t n, int64_t k0, int w);
int launch_trsv_upper_cols(mpf_ctx *c, const double *LUb, int64_t ld, double *x, int64_t n, int64_t k0, int w);
extern "C" {
int mpf_ensure_h_images(mpf_ctx *c, int64_t rows, int kmax, bool big); // internal (not in mpf_c.h): fp16 operand images
int mpf_ensure_solve_buf(mpf_ctx *c, int64_t n);                        // internal: solve scratch
int mpf_ensure_rowmajor_copy(mpf_ctx *c, int64_t rows, int64_t cols, int32_t nb);   // internal: fp64 row-major working copy + its scratch (non-zero: no room)
}
int launch_trsv_prepare(mpf_ctx *c, const double *LU, int64_t ld, int64_t n);
int launch_trsv_lower_unit(mpf_ctx *c, const double *LU, int64_t ld, double *x, int64_t n);
int launch_trsv_upper(mpf_ctx *c, const double *LU, int64_t ld, double *x, int64_t n);
int launch_axpy(mpf_ctx *c, double alpha, const double *x, double *y, int64_t n);
int launch_norm2(mpf_ctx *c, const double *x, int64_t n, double *d_out);
int launch_dot(mpf_ctx *c, const double *x, const double *y, int64_t n, double *d_out);
int launch_scal(mpf_ctx *c, double alpha, double *x, int64_t n);

// ---- host-side helpers shared by the schedules (mpf_host.cpp, mpf_dist.cpp) --------------------------------------------
struct StreamSwap { // launch_* helpers use c->stream: point it at another stream for a scope
    mpf_ctx *c; hipStream_t saved;
    StreamSwap(mpf_ctx *c_, hipStream_t s) : c(c_), saved(c_->stream) { c->stream = s; }
    ~StreamSwap() { c->stream = saved; }
};
struct EvPool { // events are recycled across calls; timing pairs are read after the final synchronise
    mpf_ctx *c; size_t next = 0;
    struct Pair { hipEvent_t a, b; double *acc, *acc2; };
    std::vector<Pair> pairs;
    const double *keep = nullptr;   // the timer that stays on at event_timers = 1 (the schedule's ms_gemm)
    explicit EvPool(mpf_ctx *c_) : c(c_) {}
    hipEvent_t get() {
        if (next == c->ev_pool.size()) { hipEvent_t e; hipEventCreate(&e); c->ev_pool.push_back(e); }
        return c->ev_pool[next++];
    }
    int timed(double &acc, hipStream_t s, const std::function<int()> &fn, double *also = nullptr) {
        // option event_timers: 2 = every region, 1 = only the trailing updates (default: what the roofline needs), 0 = none.  A pair of
        // events around every small launch of the chain costs ~8 ms per factorization at N = 32768 (fp16 mode: 152 -> 145 ms)
        const int lvl = c->tune.timeline ? 2 : c->tune.event_timers;
        if (lvl == 0 || (lvl == 1 && &acc != keep)) return fn();
        hipEvent_t a = get(), b = get();
        hipEventRecord(a, s);
        int rc = fn();
        hipEventRecord(b, s);
        pairs.push_back({a, b, &acc, also});   // `also`: a second timer the same region is booked under (a sub-total)
        return rc;
    }
    void collect() {
        for (auto &p : pairs) { float ms = 0; if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { *p.acc += ms; if (p.acc2) *p.acc2 += ms; } }
        if (!pairs.empty() && c->tune.timeline) { // diagnostic: every timed region as (start, end) in ms since the first one
            const double *base = pairs[0].acc;
            for (auto &p : pairs) if (p.acc < base) base = p.acc;
            for (auto &p : pairs) {
                float t0 = 0, t1 = 0;
                hipEventElapsedTime(&t0, pairs[0].a, p.a);
                hipEventElapsedTime(&t1, pairs[0].a, p.b);
                fprintf(stderr, "TL %d %.4f %.4f\n", (int)(p.acc - base), t0, t1);
            }
        }
    }
};

// bookkeeping of one trailing-update launch timed under ms_gemm: flops and algorithmic HBM bytes (every fp64 element of the
// block read and written once + the operands in the form the kernel reads them: fp64, fp16 images, or hi + lo images)
inline void count_gemm(mpf_stats &st, const mpf_opts &o, int64_t m, int64_t n, int64_t k, double c_bytes = 16.0) {
    if (m <= 0 || n <= 0 || k <= 0) return;
    st.gemm_flops += 2.0 * (double)m * (double)n * (double)k;
    const double opb = o.trailing == MPF_TRAIL_FP64 ? 8.0 : (o.trailing == MPF_TRAIL_FP16X3 ? 4.0 : 2.0);
    st.gemm_bytes += c_bytes * (double)m * (double)n + opb * (double)k * (double)(m + n);   // c_bytes = 8: fp32 working copy
    st.gemm_launches++;
}


// gretel_hip.hip -- host side of libgretel_hip.so: the C ABI of include/gretel_hip.h over the gfx950 kernels
// of kernels.hpp (which documents every kernel and table).
//
// Data in HBM per window:
//   band    T[(N+2)][7][W][7]      element (a, b) of cell (i, j=i+d), d in 1..W, at bidx(W, i, d, a, b) = ((i*7+a)*W+(d-1))*7+b ; T = float | double
//   cnt, marg  f64[(N+2)][8]       c_s(p) = sum_t H[s,t,p,p+1] (+ total), c_s/total          (lookup API)
//   nvalid, cmask                  V(p), candidate bitmask
//   minfo   f64[(N+2)][16]         log10 marginal x5, marginal x5, candidate bits, log10 ORIGINAL marginal x5
//   G (lt)  f64[(N+16)][6][L][5]   source-major log10 conditionals, over symbols or over candidate ranks (k_lt)
//
// Launch sequence of one path of a spin (gretel/cmd.py:148-179):  k_lt (full build for the first path, then only a
// check of the candidate-mask flags) -> k_walk_spec -> k_marg<T,true> (reweight + marginals + the table rows the path
// changed); gh_spin reduces the removed mass of all paths with one k_reweight_finish_all at the end, gh_batch_spin
// launches each kernel over all windows (k_reweight_finish per path).
//
// Arithmetic contract (identical to oracle/hansel_ref.py): row/column sums accumulate
// sequentially in the storage dtype, everything else is IEEE binary64 with NO fma
// contraction (build with -ffp-contract=off) and gh_log10 from gh_detlog.h.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "gretel_hip.h"
#include "gh_detlog.h"


// ---------------------------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(GH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                        __FILE__, __LINE__);                                                  \
    } while (0)

extern "C" const char *gh_last_error(void) { return g_err; }

#include "kernels.hpp"
#include "segwalk.hpp"
#include "cwalk.hpp"
#include "wpipe.hpp"

// ---------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------
struct prof_slot {
    std::vector<hipEvent_t> ev;   // start/stop pairs
    size_t used = 0;
    double ms = 0.0;
    int64_t launches = 0;
    double bytes = 0.0;
    int64_t seq = 0;              // launches seen since profiling was switched on
    bool open = false;            // prof_begin recorded a start event for the launch in flight
};

struct gh_handle {
    gh_config cfg;
    int N, W, L;
    int dev;
    hipStream_t stream;
    size_t n_cells;
    void *band;
    double *cnt, *marg, *minfo;
    int32_t *nvalid;
    uint32_t *cmask;
    double *rinfo;                // [(N+2)][8] log10 marginal / marginal by candidate rank (k_marg, k_rw)
    unsigned long long *pipe_pk;  // [N+2] the window pipeline's packed candidate words (wpipe.hpp), allocated by the first batch that takes it
    double *pipe_gp;              // ... and its compact table, (N+LT_PAD) sources of L x 128 bytes (wpipe.hpp: pipe_gp_piece)
    size_t pipe_gp_bytes;
    double *pipe_lm;              // ... and, with the marginal term, the candidates' log-marginals by rank [N+2][4]
    double *pipe_gw;              // ... and, for a window with a few five-candidate positions, their side table + the chunk directory (wpipe.hpp)
    size_t pipe_gw_bytes;
    bool need_rinfo;              // ... kept only where somebody reads it: with the marginal term (k_seg, k_cwalk add it in front of x1) and
                                  // for the three-launch spins (GH_FUSE at creation); nullptr goes to the kernels otherwise (C5: k_rw is bound by its stores)
    symmap sm;                    // compact index <-> symbol (gh_config.cand_order)
    double *lt;
    bool ht_stale;                // an incremental refresh left the depth-2 walker's tables (ht, yt) out: a full rebuild before they are read
    bool lt_baked;                // lt holds the marginal term in its lag-1 entries (built for a serial walker; only with cfg.marginal_term)
    double *ht, *yt;              // depth-2 walker tables derived from lt (k_lt), when walk_depth2_ok(L)
    unsigned long long fill_seen[6];   // host mirror of dstate->fill as last read (the counters only move under this handle's calls)
    uint8_t *spin_paths;          // gh_spin's device results, kept between calls: [spin_cap][N+1]
    gh_path_rec *spin_recs;       // [spin_cap]
    int spin_cap;
    int lt_L;
    bool dirty_marg, dirty_lt, have_orig;
    const uint8_t *lt_inc_path;   // non-null: the ONLY mutation since G was last built is a path reweight whose fused
                                  // kernel has rewritten the rows it changed (k_lt then only checks the mask flags)
    uint8_t *d_rw_path;           // [N+1] device copy of the path of gh_reweight_path
    dev_state *dstate;
    double *partial;       // reweight block partial sums
    int partial_cap;
    uint8_t *d_path;       // [N+1] scratch path
    gh_path_rec *d_rec;    // 1 scratch record
    // segment-parallel walk (segwalk.hpp), sized for seg_L by alloc_seg
    uint32_t *seg_hist;    // picks of every (segment, entry state)
    uint16_t *seg_maps, *seg_pmaps, *seg_gmaps;
    double *seg_min;       // [CW_MAX_SEG]
    double *seg_smin, *seg_gmin;   // minimum marginal per (segment, entry state) / per (group, entry state)
    size_t seg_smin_bytes;
    uint8_t *cm5snap;      // [N+2] candidate bits as the last k_seg saw them
    size_t fuse_lds;       // LDS of k_rw's fused prologue for this spin's state space
    bool band_zero;        // the tensor holds nothing but zeros (gh_create, gh_clear; until something is added): k_fill_own may store instead of add
    void *tband;           // column conditionals, lane groups of 16 / 32: the band once more, TO-major (tband[bidx(W, p, d, b, a)] = band[bidx(W, p, d, a, b)])
    bool lt_inc_seg;       // lt_inc_path was left by a reweight behind a segment-parallel walk (k_rw / k_rwseg keep the table under every conditional)
    uint64_t band_epoch, tband_epoch;      // tband mirrors the band iff equal: everything that writes the band counts, k_rw<.., COL> keeps both
    void *seg_halo;        // k_rwseg: per segment, the band blocks of the L positions in front of it (k_emit's copy)
    size_t seg_halo_bytes;
    bool rws;              // inside a gh_spin whose paths run as k_rwseg + k_scan + k_emit (segwalk.hpp)
    bool fuse;             // inside a gh_spin over the enumerated states: no k_emit, k_rw chains the maps itself (segwalk.hpp)
    double *lmsel1;        // [N+1] selected log-marginals of a lone gh_generate_path
    double *spin_lmsel;    // [spin_cap][N+1] the same for every path of a spin
    int seg_L;
    // candidate-pool segment walk (cwalk.hpp)
    cw_key *cw_keys, *cw_exits, *cw_pend, *cw_pend_exit;
    int32_t *cw_pend_ready;
    uint32_t *cw_phist;
    uint32_t *cw_hist;
    int32_t *cw_npend;
    int32_t *cw_last_hit, *cw_npool, *cw_true;
    uint8_t *cw_walked;
    int8_t *cw_nxt;
    bool cw_ready;         // the pools hold states of this tensor (seeded by a serial walk since the last fill / L change)
    bool cw_off;           // a position with five candidates and more than CW_MAX_L5 lags: serial walkers only
    bool cw_wide;          // the conditional table is over the symbols, not over candidate ranks: k_cwalk<L, 5>
    bool cw_pool_wide;     // ... and what the pools' states are made of
    bool cw_no_rw;         // inside gh_generate_path: no reweight follows the path (k_cemit leaves the window's flags standing)
    uint8_t *cw_keys_d, *cw_exits_d, *cw_pend_d;      // k_cwalkg: the states as bytes, [S][CW_K][cw_LD] (cw_pend_d: two sets, as the request lists)
    uint8_t *cw_pend_exit_d;                          // ... and the exit states run-on requests arrive with, two sets
    int cw_LD;
    int cw_rounds;         // walk/scan rounds queued per path (adapts to how often chains stay open)
    int cw_stamp;
    int cw_pp;             // k_cwalk launches so far: which of the two request-list sets this launch appends to (cwalk.hpp)
    size_t cw_S;           // segments the pool buffers are sized for (the second set of request lists lies cw_S entries behind the first)
    int64_t cw_stat[4];    // paths through the pools, paths handed to the serial walker, rounds queued, re-queues
    int force_stale_at;    // GH_SEG_FORCE_STALE=k at creation (tests): path k of every gh_spin finds the table stale once
    uint8_t *stage;        // pinned host staging for the results of a spin
    size_t stage_cap;
    double *ew_buf;        // gh_edge_weights_at: seven weights and the candidate mask
    bool seg6;             // inside a gh_spin at L = 6 whose table is ranked: every state of every segment (4^6), not pools
    int cw_round_cap;      // GH_CW_ROUND_CAP=k at creation (tests): never more than k rounds per launch, so that chains stay open and the serial fallback runs
    int spin_partial_stride;   // doubles between two paths' partial sums of the removed mass in a spin (0 outside spins)
    int spin_requeues;     // how often the last gh_spin rebuilt the table and queued the remaining paths again
    gh_fill_stats stats;
    int wmode;                    // WM_*: which path extension (GH_WALK at creation)
    bool lt_full;                 // GH_LT_FULL=1 at creation: rebuild the conditional table in full before every path (A/B)
    int prof;                     // 0 = off, k = bracket every k-th launch of each kernel
    prof_slot ps[GH_K_COUNT];
};

// GH_WALK (read when a handle is created): see the walker selection further down
enum { WM_SEG = 0, WM_SPEC = 1, WM_SPEC1 = 2, WM_SRC = 3 };

static int walk_mode_from_env()
{
    const char *m = getenv("GH_WALK");
    if (!m || !*m || !strcmp(m, "seg")) return WM_SEG;
    if (!strcmp(m, "spec1")) return WM_SPEC1;
    if (!strcmp(m, "src")) return WM_SRC;
    return WM_SPEC;
}

struct gh_reads {
    int dev;
    int64_t n_reads, n_bases;
    int32_t *rank;
    int64_t *off;
    uint8_t *bases;
    int max_k;
    bool sorted;      // ranks ascend: k_fill_sorted applies
    int span_pos;     // sorted tables: widest run of positions one workgroup of k_fill_sorted (FILL_RPB reads) covers
    int64_t dens128;  // sorted tables: the most reads whose ranks fall into any 128 consecutive positions (k_fill_own's counter width)
    int64_t *first_at; // sorted tables: [n_first] first_at[p] = the first read whose rank is >= p (first_at[n_first - 1] = n_reads)
    int n_first;
};
#define FILL_RPB 2048     /* reads per workgroup of k_fill_sorted */
#define FILL_PAIRS_MIN_K 10   /* k_fill_pairs (32 lanes per read) from this many SNPs in the longest read */

static inline size_t esize(const gh_handle *h) { return h->cfg.storage == GH_STORAGE_F64 ? 8 : 4; }

static int set_dev(const gh_handle *h)
{
    HIPCHK(hipSetDevice(h->dev));
    return GH_OK;
}

// launch check: GH_DEBUG_SYNC=1 synchronises after every launch so that a fault names its kernel
static int post_launch(gh_handle *h, const char *what)
{
    static const bool dbg = getenv("GH_DEBUG_SYNC") && atoi(getenv("GH_DEBUG_SYNC"));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && dbg) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(GH_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return GH_OK;
}

// profiling brackets --------------------------------------------------------------------------
static void prof_begin(gh_handle *h, int k)
{
    if (!h->prof) return;
    prof_slot &s = h->ps[k];
    s.open = false;
    // (k_seg samples half a stride away from the bracket around the whole extension it is nested in, so that neither
    // times the other's event bubbles)
    if ((s.seq++ % h->prof) != (k == GH_K_SEG ? h->prof / 2 : 0)) return;
    if (s.used + 2 > s.ev.size()) {
        for (int q = 0; q < 2; q++) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            s.ev.push_back(e);
        }
    }
    hipEventRecord(s.ev[s.used], h->stream);
    s.open = true;
}

static void prof_end(gh_handle *h, int k, double bytes)
{
    if (!h->prof) return;
    prof_slot &s = h->ps[k];
    s.bytes = bytes;
    if (!s.open) return;
    s.open = false;
    hipEventRecord(s.ev[s.used + 1], h->stream);
    s.used += 2;
}

static void prof_collect(gh_handle *h)
{
    for (int k = 0; k < GH_K_COUNT; k++) {
        prof_slot &s = h->ps[k];
        for (size_t q = 0; q + 1 < s.used; q += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s.ev[q], s.ev[q + 1]) == hipSuccess) {
                s.ms += ms;
                s.launches++;
            }
        }
        s.used = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int init_constants()
{
    int8_t lut[256];
    memset(lut, -1, sizeof lut);
    lut['A'] = 0; lut['C'] = 1; lut['G'] = 2; lut['T'] = 3; lut['N'] = 4; lut['-'] = 5; lut['_'] = 6;
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_sym_of_char), lut, sizeof lut));
    return GH_OK;
}

extern "C" int gh_device_count(int *n)
{
    if (!n) return fail(GH_ERR_ARG, "null argument");
    HIPCHK(hipGetDeviceCount(n));
    return GH_OK;
}

extern "C" int gh_device_clock_khz(int device, int *khz)
{
    if (!khz) return fail(GH_ERR_ARG, "null argument");
    if (device < 0) HIPCHK(hipGetDevice(&device));
    HIPCHK(hipDeviceGetAttribute(khz, hipDeviceAttributeClockRate, device));
    return GH_OK;
}

// log10 as the kernels evaluate it (include/gh_detlog.h), over an array: what tests/test_gpu_detlog.py holds to the host's libm
__global__ void k_log10_many(const double *__restrict__ x, double *__restrict__ y, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = gh_log10(x[i]);
}

extern "C" int gh_log10_device(int device, const double *x, double *y, int64_t n)
{
    if (!x || !y || n < 0) return fail(GH_ERR_ARG, "null argument");
    if (n == 0) return GH_OK;
    if (device >= 0) HIPCHK(hipSetDevice(device));
    double *dx = nullptr, *dy = nullptr;
    if (hipMalloc(&dx, n * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); return fail(GH_ERR_NOMEM, "hipMalloc of %lld doubles failed", (long long)n); }
    if (hipMalloc(&dy, n * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); hipFree(dx); return fail(GH_ERR_NOMEM, "hipMalloc of %lld doubles failed", (long long)n); }
    int rc = GH_OK;
    if (hipMemcpy(dx, x, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = fail(GH_ERR_HIP, "hipMemcpy");
    if (rc == GH_OK) {
        k_log10_many<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(dx, dy, n);
        if (hipGetLastError() != hipSuccess || hipMemcpy(y, dy, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(GH_ERR_HIP, "k_log10_many");
    }
    hipFree(dx); hipFree(dy);
    return rc;
}

// the same function compiled for the host (no device touched): the restatement itself against the running libm
extern "C" int gh_log10_host(const double *x, double *y, int64_t n)
{
    if (!x || !y || n < 0) return fail(GH_ERR_ARG, "null argument");
    for (int64_t i = 0; i < n; i++) y[i] = gh_log10(x[i]);
    return GH_OK;
}

static void free_handle(gh_handle *h)
{
    if (!h) return;
    hipSetDevice(h->dev);
    if (h->stream) hipStreamSynchronize(h->stream);
    hipFree(h->band); hipFree(h->tband); hipFree(h->cnt); hipFree(h->marg); hipFree(h->minfo);
    hipFree(h->pipe_pk); hipFree(h->pipe_gp); hipFree(h->pipe_lm); hipFree(h->pipe_gw);
    hipFree(h->nvalid); hipFree(h->cmask); hipFree(h->rinfo); hipFree(h->lt); hipFree(h->ht); hipFree(h->yt); hipFree(h->dstate); hipFree(h->partial);
    hipFree(h->spin_paths); hipFree(h->spin_recs);
    hipFree(h->d_path); hipFree(h->d_rw_path); hipFree(h->d_rec);
    hipFree(h->seg_hist); hipFree(h->seg_maps); hipFree(h->seg_pmaps); hipFree(h->seg_gmaps); hipFree(h->seg_min); hipFree(h->lmsel1); hipFree(h->spin_lmsel);
    hipFree(h->seg_smin); hipFree(h->seg_gmin); hipFree(h->cm5snap); hipFree(h->seg_halo);
    if (h->stage) hipHostFree(h->stage);
    hipFree(h->ew_buf);
    hipFree(h->cw_keys_d); hipFree(h->cw_exits_d); hipFree(h->cw_pend_d); hipFree(h->cw_pend_exit_d);
    hipFree(h->cw_pend_exit); hipFree(h->cw_pend_ready); hipFree(h->cw_phist);
    hipFree(h->cw_keys); hipFree(h->cw_exits); hipFree(h->cw_hist); hipFree(h->cw_last_hit); hipFree(h->cw_npool); hipFree(h->cw_walked); hipFree(h->cw_nxt); hipFree(h->cw_true); hipFree(h->cw_pend); hipFree(h->cw_npend);
    for (int k = 0; k < GH_K_COUNT; k++)
        for (hipEvent_t e : h->ps[k].ev) hipEventDestroy(e);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int gh_create(const gh_config *cfg, gh_t **out)
{
    if (!cfg || !out) return fail(GH_ERR_ARG, "null argument");
    if (cfg->n_snps < 1) return fail(GH_ERR_ARG, "n_snps must be >= 1 (got %d)", cfg->n_snps);
    if (cfg->band < 1) return fail(GH_ERR_ARG, "band must be >= 1 (got %d)", cfg->band);
    if (cfg->storage != GH_STORAGE_F32 && cfg->storage != GH_STORAGE_F64)
        return fail(GH_ERR_ARG, "bad storage %d", cfg->storage);
    if (cfg->cond_mode < 0 || cfg->cond_mode > GH_COND_E) return fail(GH_ERR_ARG, "bad cond_mode %d", cfg->cond_mode);
    uint8_t order[5] = {0, 1, 2, 3, 5};
    {
        bool all_zero = true;
        for (int q = 0; q < 8; q++) all_zero = all_zero && cfg->cand_order[q] == 0;
        if (!all_zero) {
            unsigned seen = 0;
            for (int q = 0; q < 5; q++) {
                if (cfg->cand_order[q] > 5 || cfg->cand_order[q] == 4) return fail(GH_ERR_ARG, "cand_order[%d] = %d is not a valid symbol index", q, cfg->cand_order[q]);
                seen |= 1u << cfg->cand_order[q];
                order[q] = cfg->cand_order[q];
            }
            if (seen != 0x2Fu) return fail(GH_ERR_ARG, "cand_order is not a permutation of the valid symbols {0,1,2,3,5}");
        }
    }
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(GH_ERR_HIP, "no HIP device visible");
    int dev = cfg->device;
    if (dev < 0) HIPCHK(hipGetDevice(&dev));
    if (dev >= ndev) return fail(GH_ERR_ARG, "device %d out of range (%d devices)", dev, ndev);
    HIPCHK(hipSetDevice(dev));
    int rc = init_constants();
    if (rc) return rc;

    gh_handle *h = new (std::nothrow) gh_handle();
    if (!h) return fail(GH_ERR_NOMEM, "host allocation failed");
    h->cfg = *cfg;
    h->cfg.device = dev;
    h->cfg.offer_zero = cfg->offer_zero ? 1 : 0;
    memset(h->cfg.cand_order, 0, sizeof h->cfg.cand_order);
    memcpy(h->cfg.cand_order, order, 5);
    h->sm = make_symmap(order);
    h->rinfo = nullptr; h->pipe_pk = nullptr; h->pipe_gp = nullptr; h->pipe_gp_bytes = 0; h->pipe_lm = nullptr; h->pipe_gw = nullptr; h->pipe_gw_bytes = 0; h->lt_baked = false; h->ht_stale = false;
    h->need_rinfo = cfg->marginal_term != 0 || (getenv("GH_FUSE") && atoi(getenv("GH_FUSE")) >= 1);
    h->dev = dev;
    h->N = cfg->n_snps;
    h->W = cfg->band;
    h->L = 1;
    h->n_cells = (size_t)(h->N + 2) * h->W;
    h->lt = nullptr; h->ht = nullptr; h->yt = nullptr; h->lt_L = 0;
    h->spin_paths = nullptr; h->spin_recs = nullptr; h->spin_cap = 0;
    h->seg_hist = nullptr; h->seg_maps = nullptr; h->seg_pmaps = nullptr; h->seg_gmaps = nullptr; h->seg_min = nullptr; h->lmsel1 = nullptr;
    h->seg_smin = nullptr; h->seg_gmin = nullptr; h->cm5snap = nullptr; h->fuse = false; h->seg_smin_bytes = 0;
    h->seg_halo = nullptr; h->seg_halo_bytes = 0; h->rws = false;
    h->spin_lmsel = nullptr; h->seg_L = 0; h->spin_requeues = 0; h->spin_partial_stride = 0;
    h->cw_keys = nullptr; h->cw_exits = nullptr; h->cw_hist = nullptr; h->cw_last_hit = nullptr; h->cw_npool = nullptr;
    h->cw_pend_exit = nullptr; h->cw_pend_ready = nullptr; h->cw_phist = nullptr;
    h->cw_walked = nullptr; h->cw_nxt = nullptr; h->cw_true = nullptr; h->cw_pend = nullptr; h->cw_npend = nullptr; h->cw_ready = false; h->cw_off = false; h->cw_wide = false; h->cw_pool_wide = false; h->cw_rounds = 2; h->cw_stamp = 0; h->cw_pp = 0; h->cw_S = 0;
    h->cw_keys_d = nullptr; h->cw_exits_d = nullptr; h->cw_pend_d = nullptr; h->cw_pend_exit_d = nullptr; h->cw_LD = 0; h->cw_no_rw = false;
    memset(h->cw_stat, 0, sizeof h->cw_stat);
    h->force_stale_at = getenv("GH_SEG_FORCE_STALE") ? atoi(getenv("GH_SEG_FORCE_STALE")) : -1;
    h->cw_round_cap = getenv("GH_CW_ROUND_CAP") ? atoi(getenv("GH_CW_ROUND_CAP")) : 0;
    h->stage = nullptr; h->stage_cap = 0;
    h->seg6 = false;
    h->ew_buf = nullptr;
    memset(h->fill_seen, 0, sizeof h->fill_seen);
    h->dirty_marg = h->dirty_lt = true; h->lt_inc_path = nullptr; h->band_zero = true;      // (the allocation is zeroed below / above: gh_create)
    h->tband = nullptr; h->band_epoch = 1; h->tband_epoch = 0; h->lt_inc_seg = false;
    h->have_orig = false;
    h->lt_inc_path = nullptr; h->d_rw_path = nullptr;
    h->prof = 0;
    h->wmode = walk_mode_from_env();
    h->lt_full = getenv("GH_LT_FULL") && atoi(getenv("GH_LT_FULL"));
    h->partial = nullptr; h->partial_cap = 0;
    memset(&h->stats, 0, sizeof h->stats);
    h->stats.L = 1;
    const size_t np = (size_t)h->N + 2;
#define ALLOC(ptr, bytes)                                                                     \
    do {                                                                                      \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes));                                  \
        if (e_ != hipSuccess) {                                                               \
            free_handle(h);                                                                   \
            return fail(GH_ERR_NOMEM, "hipMalloc(%zu) failed: %s", (size_t)(bytes),           \
                        hipGetErrorString(e_));                                               \
        }                                                                                     \
    } while (0)
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        free_handle(h);
        return fail(GH_ERR_HIP, "hipStreamCreate failed");
    }
    ALLOC(h->band, h->n_cells * CELL * esize(h));
    ALLOC(h->cnt, np * 8 * sizeof(double));
    ALLOC(h->marg, np * 8 * sizeof(double));
    ALLOC(h->minfo, np * MINFO * sizeof(double));
    ALLOC(h->nvalid, np * sizeof(int32_t));
    ALLOC(h->cmask, np * sizeof(uint32_t));
    ALLOC(h->rinfo, np * RINFO * sizeof(double));
    ALLOC(h->dstate, sizeof(dev_state));
    ALLOC(h->d_path, np);
    ALLOC(h->d_rw_path, np);
    ALLOC(h->d_rec, sizeof(gh_path_rec));
#undef ALLOC
    hipMemsetAsync(h->band, 0, h->n_cells * CELL * esize(h), h->stream);
    hipMemsetAsync(h->dstate, 0, sizeof(dev_state), h->stream);
    hipMemsetAsync(h->minfo, 0, np * MINFO * sizeof(double), h->stream);
    hipMemsetAsync(h->rinfo, 0, np * RINFO * sizeof(double), h->stream);
    hipMemsetAsync(h->cmask, 0, np * sizeof(uint32_t), h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    *out = h;
    return GH_OK;
}

extern "C" int gh_destroy(gh_t *h)
{
    free_handle(h);
    return GH_OK;
}

extern "C" int gh_sync(gh_t *h)
{
    if (!h) return fail(GH_ERR_ARG, "null handle");
    if (set_dev(h)) return GH_ERR_HIP;
    HIPCHK(hipStreamSynchronize(h->stream));
    return GH_OK;
}

extern "C" int gh_clear(gh_t *h)
{
    if (!h) return fail(GH_ERR_ARG, "null handle");
    if (set_dev(h)) return GH_ERR_HIP;
    HIPCHK(hipMemsetAsync(h->band, 0, h->n_cells * CELL * esize(h), h->stream));
    HIPCHK(hipMemsetAsync(h->dstate, 0, sizeof(dev_state), h->stream));
    memset(h->fill_seen, 0, sizeof h->fill_seen);
    memset(&h->stats, 0, sizeof h->stats);
    h->stats.L = 1;
    h->L = 1;
    h->dirty_marg = h->dirty_lt = true; h->lt_inc_path = nullptr; h->band_zero = true; h->band_epoch++;
    h->have_orig = false;
    h->cw_ready = false; h->cw_off = false;
    return GH_OK;
}

extern "C" int gh_copy(const gh_t *src, gh_t **out)
{
    if (!src || !out) return fail(GH_ERR_ARG, "null argument");
    gh_t *h = nullptr;
    int rc = gh_create(&src->cfg, &h);
    if (rc) return rc;
    hipStreamSynchronize(src->stream);
    // device-to-device copies are asynchronous to the host: order it on the new handle's stream
    hipError_t e = hipMemcpyAsync(h->band, src->band, src->n_cells * CELL * esize(src), hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { free_handle(h); return fail(GH_ERR_HIP, "copy failed: %s", hipGetErrorString(e)); }
    h->L = src->L;
    h->stats = src->stats;
    h->band_zero = src->band_zero;
    *out = h;
    return GH_OK;
}

extern "C" int gh_set_L(gh_t *h, int32_t L)
{
    if (!h) return fail(GH_ERR_ARG, "null handle");
    if (L < 1) return fail(GH_ERR_ARG, "L must be >= 1 (got %d)", L);
    if (L > 32767) return fail(GH_ERR_ARG, "L > 32767 unsupported (got %d)", L);
    if (L != h->L) { h->dirty_lt = true; h->cw_ready = false; }
    h->L = L;
    h->stats.L = L;
    return GH_OK;
}

extern "C" int gh_get_L(const gh_t *h, int32_t *L)
{
    if (!h || !L) return fail(GH_ERR_ARG, "null argument");
    *L = h->L;
    return GH_OK;
}

extern "C" int gh_get_fill_stats(const gh_t *h, gh_fill_stats *out)
{
    if (!h || !out) return fail(GH_ERR_ARG, "null argument");
    *out = h->stats;
    out->L = h->L;
    return GH_OK;
}

extern "C" int gh_set_fill_stats(gh_t *h, const gh_fill_stats *in)
{
    if (!h || !in) return fail(GH_ERR_ARG, "null argument");
    h->stats.n_slices = in->n_slices;
    h->stats.n_crumbs = in->n_crumbs;
    h->stats.covered_snps = in->covered_snps;
    return GH_OK;
}

// reads ---------------------------------------------------------------------------------------
extern "C" int gh_reads_upload(const gh_t *h, const int32_t *rank, const int64_t *off,
                               const uint8_t *bases, int64_t n_reads, gh_reads_t **out)
{
    if (!h || !out || n_reads < 0 || (n_reads > 0 && (!rank || !off))) return fail(GH_ERR_ARG, "bad argument");
    if (set_dev(h)) return GH_ERR_HIP;
    gh_reads *r = new (std::nothrow) gh_reads();
    if (!r) return fail(GH_ERR_NOMEM, "host allocation failed");
    r->dev = h->dev;
    r->n_reads = n_reads;
    r->n_bases = n_reads ? off[n_reads] : 0;
    r->rank = nullptr; r->off = nullptr; r->bases = nullptr;
    r->max_k = 0;
    r->sorted = true;
    r->span_pos = 0;
    r->dens128 = 0;
    r->first_at = nullptr; r->n_first = 0;
    // The table's properties (longest read, off[] ascending, ranks ascending and then span / density / first_at) are found on the
    // device behind the copies (k_reads_meta): the host looks at the two ends of rank[] only, to size first_at for a table that
    // turns out to be sorted.  Page-locked sources (gh_host_alloc: what gretel_amd.util.load_from_bam decodes into) are read by DMA.
    int64_t top = 0;
    if (n_reads > 0 && rank[0] >= 0 && rank[n_reads - 1] >= rank[0]) {
        top = (int64_t)rank[n_reads - 1] + 2;
        if (top > ((int64_t)1 << 28)) top = 0;
    }
    reads_meta *dm = nullptr;
    reads_meta hm;
    hm.max_k = 0; hm.unsorted = 0; hm.bad_off = 0x7fffffffffffffffll; hm.span = 0; hm.dens128 = 0;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc((void **)&r->rank, (size_t)(n_reads ? n_reads : 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&r->off, (size_t)(n_reads + 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&r->bases, (size_t)(r->n_bases ? r->n_bases : 1));
    if (e == hipSuccess && top > 0) e = hipMalloc((void **)&r->first_at, (size_t)top * 8);
    if (e == hipSuccess && n_reads) e = hipMalloc((void **)&dm, sizeof(reads_meta));
    if (e == hipSuccess && n_reads) e = hipMemcpyAsync(dm, &hm, sizeof hm, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && n_reads) e = hipMemcpyAsync(r->rank, rank, (size_t)n_reads * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && n_reads) e = hipMemcpyAsync(r->off, off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && n_reads) {
        const int grid = (int)std::min<int64_t>((n_reads + 256) / 256, 2048);
        k_reads_meta<<<grid, 256, 0, h->stream>>>(r->rank, r->off, n_reads, dm);
        if (top > 0) k_reads_first_at<<<grid, 256, 0, h->stream>>>(r->rank, n_reads, r->first_at, top, dm);
        k_reads_meta2<<<grid, 256, 0, h->stream>>>(r->rank, n_reads, r->first_at, FILL_RPB, dm);
        e = hipGetLastError();
    }
    if (e == hipSuccess && r->n_bases) e = hipMemcpyAsync(r->bases, bases, (size_t)r->n_bases, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && n_reads) e = hipMemcpyAsync(&hm, dm, sizeof hm, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(dm);
    if (e != hipSuccess) {
        hipFree(r->rank); hipFree(r->off); hipFree(r->bases); hipFree(r->first_at);
        delete r;
        return fail(GH_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
    }
    if (hm.bad_off != 0x7fffffffffffffffll) {
        hipFree(r->rank); hipFree(r->off); hipFree(r->bases); hipFree(r->first_at);
        delete r;
        return fail(GH_ERR_ARG, "off[] not monotone at read %lld", hm.bad_off);
    }
    r->max_k = hm.max_k;
    r->sorted = !hm.unsorted;
    if (r->sorted && n_reads > 0) {
        r->span_pos = hm.span + r->max_k + 1;             // the slice k_fill_sorted counts in LDS
        if (r->first_at) { r->n_first = (int)top; r->dens128 = hm.dens128; }
        else {
            // (ranks ascend from a negative one, or reach beyond 2^28: no first_at; the density the slow way)
            int64_t lo = 0;
            for (int64_t q = 0; q < n_reads; q++) {
                while (rank[q] - rank[lo] >= 128) lo++;
                if (q - lo + 1 > r->dens128) r->dens128 = q - lo + 1;
            }
        }
    } else if (r->first_at) {
        hipFree(r->first_at);
        r->first_at = nullptr;
    }
    *out = r;
    return GH_OK;
}

extern "C" int gh_reads_max_k(const gh_reads_t *r, int32_t *max_k)
{
    if (!r || !max_k) return fail(GH_ERR_ARG, "bad argument");
    *max_k = r->max_k;
    return GH_OK;
}

extern "C" int gh_reads_info(const gh_reads_t *r, int64_t info[5], int64_t *first_at)
{
    if (!r || !info) return fail(GH_ERR_ARG, "bad argument");
    info[0] = r->max_k; info[1] = r->sorted ? 1 : 0; info[2] = r->span_pos; info[3] = r->dens128; info[4] = r->first_at ? r->n_first : 0;
    if (first_at && r->first_at) {
        HIPCHK(hipSetDevice(r->dev));
        HIPCHK(hipMemcpy(first_at, r->first_at, (size_t)r->n_first * 8, hipMemcpyDeviceToHost));
    }
    return GH_OK;
}

extern "C" int gh_reads_free(gh_reads_t *r)
{
    if (!r) return GH_OK;
    hipSetDevice(r->dev);
    hipFree(r->rank); hipFree(r->off); hipFree(r->bases); hipFree(r->first_at);
    delete r;
    return GH_OK;
}

static int pull_fill_state(gh_handle *h, const char *what)
{
    dev_state hs;
    HIPCHK(hipMemcpyAsync(&hs, h->dstate, sizeof hs, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    unsigned long long before[6];
    memcpy(before, h->fill_seen, sizeof before);
    memcpy(h->fill_seen, hs.fill, sizeof before);
    h->stats.n_slices += (int64_t)(hs.fill[0] - before[0]);
    h->stats.n_crumbs += (int64_t)(hs.fill[1] - before[1]);
    h->stats.covered_snps += (int64_t)(hs.fill[2] - before[2]);
    if (hs.fill[3] != before[3])
        return fail(GH_ERR_SYMBOL, "%s: %llu item(s) carry a symbol outside \"ACGTN-_\"", what,
                    hs.fill[3] - before[3]);
    if (hs.fill[4] != before[4])
        return fail(GH_ERR_BAND, "%s: %llu observation(s) outside band %d / positions [0,%d]", what,
                    hs.fill[4] - before[4], h->W, h->N + 1);
    return GH_OK;
}

extern "C" int gh_fill(gh_t *h, const gh_reads_t *r, int use_end_sentinels, gh_fill_stats *out)
{
    if (!h || !r) return fail(GH_ERR_ARG, "null argument");
    if (r->dev != h->dev) return fail(GH_ERR_ARG, "reads live on device %d, handle on %d", r->dev, h->dev);
    if (set_dev(h)) return GH_ERR_HIP;
    if (r->n_reads > 0) {
        const int block = 256;
        int64_t nb = (r->n_reads + block - 1) / block;
        if (nb > 256 * 32) nb = 256 * 32;
        const double bytes = 8.0 * 0 + (double)r->n_reads * 12.0 + (double)r->n_bases;   // + 8*adds, added below
        prof_begin(h, GH_K_FILL);
        // sorted tables: LDS-privatised counting when a run of reads stays inside a slice that fits in LDS
        static const bool no_sorted = getenv("GH_FILL_SCATTER") && atoi(getenv("GH_FILL_SCATTER"));
        const int rpb = FILL_RPB;                                 // reads per workgroup
        int max_pos = 0;
        if (r->sorted && !no_sorted) {
            // LDS for the widest slice any workgroup needs (known from the upload), at most 96 KB: several
            // workgroups per CU when the slices are narrow; what falls outside a slice goes to global atomics
            const size_t per_pos = (size_t)h->W * CELL * sizeof(unsigned);
            max_pos = (int)((96 * 1024) / per_pos);
            if (max_pos > 1024) max_pos = 1024;
            if (r->span_pos > 0 && r->span_pos < max_pos) max_pos = r->span_pos > r->max_k + 8 ? r->span_pos : r->max_k + 8;
        }
        // rank-sorted tables: the tensor cut by owner (k_fill_own) -- no global atomics -- when a slice of from-positions wide enough
        // to keep the overlap between neighbours small fits the LDS: P positions x W x 49 counters of 2 bytes (4 where a workgroup
        // may see 65 536 reads).  GH_FILL_OWN=0 keeps the older fills (the tests run them against each other).
        int own_P = 0;
        bool own_half = true;
        if (r->sorted && r->first_at && !no_sorted && !(getenv("GH_FILL_OWN") && atoi(getenv("GH_FILL_OWN")) == 0)) {
            int P = (h->N + 2 + 1023) / 1024;                    // about a thousand workgroups ...
            if (P < 2 * r->max_k) P = 2 * r->max_k;              // ... that visit a read 1.5 times at most
            if (P < 8) P = 8;
            // the reads one workgroup can see: ranks within P + max_k positions
            const int64_t win = (int64_t)((P + r->max_k + 127) / 128 + 1) * r->dens128;
            own_half = win < 65536 && !(getenv("GH_FILL_OWN_WIDE") && atoi(getenv("GH_FILL_OWN_WIDE")));      // (the tests force the 4-byte counters)
            const size_t per_pos = (size_t)h->W * CELL * (own_half ? 2 : 4);
            const int P_lds = (int)((150 * 1024 - FILL_OWN_SYMS) / per_pos);
            if (P > P_lds) P = P_lds;
            if (P >= r->max_k && P >= 8 && r->max_k <= FILL_OWN_SYMS) own_P = P;               // (narrower: every read would be visited by many workgroups)
        }
        if (own_P > 0) {
            const unsigned gb = (unsigned)((h->N + 2 + own_P - 1) / own_P);
            const size_t sym_off = (((size_t)own_P * h->W * CELL * (own_half ? 2 : 4)) + 15) & ~(size_t)15;
            const size_t lds = sym_off + FILL_OWN_SYMS;
#define FILL_OWN(T_, CT_, Z_, G_) do {                                                                                                   \
                hipFuncSetAttribute((const void *)k_fill_own<T_, CT_, Z_, G_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
                hipLaunchKernelGGL((k_fill_own<T_, CT_, Z_, G_>), dim3(gb), dim3(1024), lds, h->stream, (T_ *)h->band, h->N, h->W, r->rank,  \
                                   r->off, r->bases, r->n_reads, own_P, r->max_k, use_end_sentinels, h->dstate, r->first_at, r->n_first, (int)sym_off); \
            } while (0)
            // lanes per read: one for short reads, four / eight for long ones (a lane takes every G-th from-index of its read)
#define FILL_OWN_G(T_, CT_, Z_) do { if (r->max_k <= 8) FILL_OWN(T_, CT_, Z_, 1); else if (r->max_k <= 32) FILL_OWN(T_, CT_, Z_, 4); else FILL_OWN(T_, CT_, Z_, 8); } while (0)
#define FILL_OWN_Z(T_, CT_) do { if (h->band_zero) FILL_OWN_G(T_, CT_, true); else FILL_OWN_G(T_, CT_, false); } while (0)
            if (h->cfg.storage == GH_STORAGE_F64) { if (own_half) FILL_OWN_Z(double, uint16_t); else FILL_OWN_Z(double, uint32_t); }
            else { if (own_half) FILL_OWN_Z(float, uint16_t); else FILL_OWN_Z(float, uint32_t); }
#undef FILL_OWN_G
#undef FILL_OWN_Z
#undef FILL_OWN
        } else if (max_pos >= r->max_k + 8) {
            const unsigned gb = (unsigned)((r->n_reads + rpb - 1) / rpb);
            const size_t lds = (size_t)max_pos * h->W * CELL * sizeof(unsigned);
            if (h->cfg.storage == GH_STORAGE_F64) {
                hipFuncSetAttribute((const void *)k_fill_sorted<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(k_fill_sorted<double>, dim3(gb), dim3(block), lds, h->stream, (double *)h->band, h->N, h->W,
                                   r->rank, r->off, r->bases, r->n_reads, rpb, max_pos, r->max_k, use_end_sentinels, h->dstate);
            } else {
                hipFuncSetAttribute((const void *)k_fill_sorted<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(k_fill_sorted<float>, dim3(gb), dim3(block), lds, h->stream, (float *)h->band, h->N, h->W,
                                   r->rank, r->off, r->bases, r->n_reads, rpb, max_pos, r->max_k, use_end_sentinels, h->dstate);
            }
        } else if (r->max_k >= FILL_PAIRS_MIN_K && r->max_k <= 32 && !(getenv("GH_FILL_PAIRS") && atoi(getenv("GH_FILL_PAIRS")) == 0)) {
            // long reads, sparse tensor: 32 lanes per read (k_fill_pairs)
            int64_t nbp = (r->n_reads * 32 + block - 1) / block;
            if (nbp > 256 * 64) nbp = 256 * 64;
            if (h->cfg.storage == GH_STORAGE_F64)
                hipLaunchKernelGGL(k_fill_pairs<double>, dim3((unsigned)nbp), dim3(block), 0, h->stream, (double *)h->band,
                                   h->N, h->W, r->rank, r->off, r->bases, r->n_reads, use_end_sentinels, h->dstate);
            else
                hipLaunchKernelGGL(k_fill_pairs<float>, dim3((unsigned)nbp), dim3(block), 0, h->stream, (float *)h->band,
                                   h->N, h->W, r->rank, r->off, r->bases, r->n_reads, use_end_sentinels, h->dstate);
        } else if (h->cfg.storage == GH_STORAGE_F64)
            hipLaunchKernelGGL(k_fill<double>, dim3((unsigned)nb), dim3(block), 0, h->stream, (double *)h->band,
                               h->N, h->W, r->rank, r->off, r->bases, r->n_reads, use_end_sentinels, h->dstate);
        else
            hipLaunchKernelGGL(k_fill<float>, dim3((unsigned)nb), dim3(block), 0, h->stream, (float *)h->band,
                               h->N, h->W, r->rank, r->off, r->bases, r->n_reads, use_end_sentinels, h->dstate);
        prof_end(h, GH_K_FILL, bytes);
        { int rc_ = post_launch(h, "k_fill"); if (rc_) return rc_; }
    }
    h->dirty_marg = h->dirty_lt = true; h->lt_inc_path = nullptr; if (r->n_reads > 0) h->band_zero = false; h->band_epoch++;
    h->cw_ready = false; h->cw_off = false;
    int rc = pull_fill_state(h, "gh_fill");
    if (h->stats.n_slices > 0) {                                   // util.py:333
        int L = (int)std::ceil((double)h->stats.covered_snps / (double)h->stats.n_slices);
        if (L < 1) L = 1;
        h->L = L;
        h->stats.L = L;
    }
    if (h->prof) {
        // algorithmic bytes: one 4/8-byte read-modify-write per add_observation + the table itself
        h->ps[GH_K_FILL].bytes = 2.0 * esize(h) * (double)h->stats.n_crumbs * 1.0 + (double)r->n_reads * 12.0 + (double)r->n_bases;
    }
    if (out) { *out = h->stats; out->L = h->L; }
    return rc;
}

// one-cell API --------------------------------------------------------------------------------
static int cell_index(const gh_handle *h, int a, int b, int i, int j, size_t *idx)
{
    if (a < 0 || a >= NSYM || b < 0 || b >= NSYM) return fail(GH_ERR_SYMBOL, "symbol index out of range (%d,%d)", a, b);
    int d = j - i;
    if (i < 0 || j > h->N + 1 || d < 1 || d > h->W) return 1;    // outside the band: a zero cell
    *idx = bidx(h->W, (size_t)i, d, a, b);
    return 0;
}

extern "C" int gh_get(gh_t *h, int a, int b, int i, int j, double *out)
{
    if (!h || !out) return fail(GH_ERR_ARG, "null argument");
    if (set_dev(h)) return GH_ERR_HIP;
    size_t idx;
    int rc = cell_index(h, a, b, i, j, &idx);
    if (rc < 0) return rc;
    if (rc == 1) { *out = 0.0; return GH_OK; }
    if (h->cfg.storage == GH_STORAGE_F64) {
        HIPCHK(hipMemcpyAsync(out, (double *)h->band + idx, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    } else {
        float f;
        HIPCHK(hipMemcpyAsync(&f, (float *)h->band + idx, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        *out = (double)f;
    }
    return GH_OK;
}

extern "C" int gh_add_batch(gh_t *h, const uint8_t *a, const uint8_t *b, const int32_t *i, const int32_t *j, int64_t n)
{
    if (!h || n < 0 || (n > 0 && (!a || !b || !i || !j))) return fail(GH_ERR_ARG, "bad argument");
    if (n == 0) return GH_OK;
    if (set_dev(h)) return GH_ERR_HIP;
    uint8_t *da = nullptr, *db = nullptr;
    int32_t *di = nullptr, *dj = nullptr;
    hipError_t e = hipMalloc((void **)&da, n);
    if (e == hipSuccess) e = hipMalloc((void **)&db, n);
    if (e == hipSuccess) e = hipMalloc((void **)&di, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&dj, n * 4);
    if (e == hipSuccess) e = hipMemcpy(da, a, n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db, b, n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(di, i, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dj, j, n * 4, hipMemcpyHostToDevice);
    int rc = GH_OK;
    if (e != hipSuccess) rc = fail(GH_ERR_HIP, "gh_add_batch staging failed: %s", hipGetErrorString(e));
    if (rc == GH_OK) {
        const int block = 256;
        const unsigned nb = (unsigned)((n + block - 1) / block);
        if (h->cfg.storage == GH_STORAGE_F64)
            hipLaunchKernelGGL(k_add_batch<double>, dim3(nb), dim3(block), 0, h->stream, (double *)h->band, h->N, h->W, da, db, di, dj, n, h->dstate);
        else
            hipLaunchKernelGGL(k_add_batch<float>, dim3(nb), dim3(block), 0, h->stream, (float *)h->band, h->N, h->W, da, db, di, dj, n, h->dstate);
        h->dirty_marg = h->dirty_lt = true; h->lt_inc_path = nullptr; h->band_zero = false; h->band_epoch++;
        int64_t s0 = h->stats.n_slices, c0 = h->stats.n_crumbs, v0 = h->stats.covered_snps;
        rc = pull_fill_state(h, "gh_add_batch");
        h->stats.n_slices = s0; h->stats.n_crumbs = c0; h->stats.covered_snps = v0;
    }
    hipFree(da); hipFree(db); hipFree(di); hipFree(dj);
    return rc;
}

extern "C" int gh_add(gh_t *h, int a, int b, int i, int j)
{
    if (a < 0 || a >= NSYM || b < 0 || b >= NSYM) return fail(GH_ERR_SYMBOL, "symbol index out of range (%d,%d)", a, b);
    uint8_t ua = (uint8_t)a, ub = (uint8_t)b;
    int32_t ii = i, jj = j;
    return gh_add_batch(h, &ua, &ub, &ii, &jj, 1);
}

extern "C" int gh_reweight_obs(gh_t *h, int a, int b, int i, int j, double ratio, double *removed)
{
    if (!h) return fail(GH_ERR_ARG, "null handle");
    if (set_dev(h)) return GH_ERR_HIP;
    size_t idx;
    int rc = cell_index(h, a, b, i, j, &idx);
    if (rc < 0) return rc;
    double rem = 0.0;
    if (rc == 0) {
        double *d_rem = &h->d_rec->magnitude;
        if (h->cfg.storage == GH_STORAGE_F64)
            hipLaunchKernelGGL(k_reweight_one<double>, dim3(1), dim3(1), 0, h->stream, (double *)h->band + idx, ratio, d_rem);
        else
            hipLaunchKernelGGL(k_reweight_one<float>, dim3(1), dim3(1), 0, h->stream, (float *)h->band + idx, ratio, d_rem);
        HIPCHK(hipMemcpyAsync(&rem, d_rem, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->dirty_marg = h->dirty_lt = true; h->lt_inc_path = nullptr; h->band_epoch++;
    }
    if (removed) *removed = rem;
    return GH_OK;
}

// tables --------------------------------------------------------------------------------------
static int ensure_marg(gh_handle *h)
{
    if (!h->dirty_marg) return GH_OK;
    const int threads = (h->N + 1) * 8;
    const int block = 256;
    // re-arm the "first SNP without a candidate" word that k_marg min-reduces into
    HIPCHK(hipMemsetAsync(&h->dstate->first_hole, 0x7f, 4 * sizeof(int), h->stream));   // first_hole, nodel, cm_same, narrow
    prof_begin(h, GH_K_MARG);
    if (h->cfg.storage == GH_STORAGE_F64)
        hipLaunchKernelGGL((k_marg<double, false>), dim3((threads + block - 1) / block), dim3(block), 0, h->stream,
                           (double *)h->band, h->N, h->W, h->cnt, h->marg, h->nvalid, h->cmask, h->minfo, h->dstate, (const win_desc *)nullptr,
                           (const uint8_t *)nullptr, 0.0, 0, (double *)nullptr, 0, (double *)nullptr, 0, 0, (const double *)nullptr, (gh_path_rec *)nullptr,
                           h->sm, h->cfg.offer_zero, h->need_rinfo ? h->rinfo : (double *)nullptr);
    else
        hipLaunchKernelGGL((k_marg<float, false>), dim3((threads + block - 1) / block), dim3(block), 0, h->stream,
                           (float *)h->band, h->N, h->W, h->cnt, h->marg, h->nvalid, h->cmask, h->minfo, h->dstate, (const win_desc *)nullptr,
                           (const uint8_t *)nullptr, 0.0, 0, (double *)nullptr, 0, (double *)nullptr, 0, 0, (const double *)nullptr, (gh_path_rec *)nullptr,
                           h->sm, h->cfg.offer_zero, h->need_rinfo ? h->rinfo : (double *)nullptr);
    // algorithmic bytes: read the (p,p+1) cell, write cnt/marg (2x64), minfo (88), nvalid+cmask (8)
    prof_end(h, GH_K_MARG, (double)(h->N + 1) * (CELL * esize(h) + 2 * 64 + 88 + 8));
    { int rc_ = post_launch(h, "k_marg"); if (rc_) return rc_; }
    h->dirty_marg = false;
    return GH_OK;
}

static int alloc_lt(gh_handle *h);
// may the fused reweight keep the conditional table current (k_marg<T,true> rewrites the rows a path changes)?
// k_marg<T,true> (behind the serial walkers, gh_reweight_path): rows only -- conditionals whose denominator is a row sum or a
// marginal count, and no marginal term baked into the table
static bool lt_incremental_ok(const gh_handle *h)
{
    return h->cfg.cond_mode != GH_COND_C && h->cfg.cond_mode != GH_COND_E && !h->cfg.marginal_term && !h->lt_full;
}
// k_rw (behind the segment-parallel walks, which add the marginal term themselves): every conditional -- rows, or columns for C and E
static bool rw_incremental_ok(const gh_handle *h) { return !h->lt_full && !h->lt_baked; }
static bool walk_depth2_ok(int wm, int L);
static bool walk_ranked_ok(int wm, int L);
static bool seg_ok(int wm, int L);

// baked: the table is for a serial walker, which reads whole rows: with the marginal term its lag-1 entries hold lm + x1
// (k_lt); the segment-parallel walks want the bare conditionals and add lm themselves
// derived = false (the candidate pools' looks): the depth-2 serial walker's tables Ht / Yt are left alone -- nobody reads them
// until a serial walk, in front of which the table is rebuilt in full anyway (45 us per look at C5 otherwise)
// the mixed-radix state space (segmix.hpp) for this handle?  L = 5 over the segment-parallel extension; not where every symbol is
// offered everywhere (five candidates at every position: nothing to gain) and not inside a spin of the opt-in GH_FUSE flow, whose
// k_seg / k_scan / k_rw carry the minima over the enumerated states
static bool mixed_allowed(const gh_handle *h)
{
    static const bool off = getenv("GH_MIXED") && atoi(getenv("GH_MIXED")) == 0;
    return !off && !h->fuse && h->L == SEGM_L && h->wmode == WM_SEG && !h->cfg.offer_zero;
}

static int ensure_lt(gh_handle *h, bool baked = false, bool derived = true)
{
    int rc = ensure_marg(h);
    if (rc) return rc;
    const bool want_baked = baked && h->cfg.marginal_term;
    const bool need_derived = derived && h->ht_stale;
    if (!h->dirty_lt && h->lt && h->lt_L == h->L && h->lt_baked == want_baked && !need_derived) return GH_OK;
    if ((rc = alloc_lt(h))) return rc;
    // what kept the table current since the last build: k_marg<T,true> behind a serial walker (rows, conditionals A / B without the
    // marginal term), or k_rw / k_rwseg behind a segment-parallel walk (every conditional: rows or columns; the derived tables of the
    // depth-2 serial walker are refreshed by ROWS only, so under a column conditional or the marginal term a caller that wants those
    // gets a full build).  Round 3 rebuilt the whole table at every look of a pool spin under C / E / the marginal term.
    const bool col_or_mt = h->cfg.cond_mode == GH_COND_C || h->cfg.cond_mode == GH_COND_E || h->cfg.marginal_term;
    const bool inc_rule = h->lt_inc_seg ? (rw_incremental_ok(h) && (!col_or_mt || !derived)) : lt_incremental_ok(h);
    const bool inc_ok = h->lt_inc_path && inc_rule && h->lt_baked == want_baked && !need_derived;
    const uint8_t *inc = inc_ok ? h->lt_inc_path : nullptr;
    const size_t total = inc ? (size_t)h->N * 4 : (size_t)(h->N + LT_PAD) * h->L * LT_BLK;
    const int block = 256;
    size_t nb = (total + block - 1) / block;
    if (nb > 256 * 16) nb = 256 * 16;
    // (column conditionals: the to-major copy, while it mirrors the band, makes the column sums contiguous reads)
    const void *tb_lt = (h->tband && h->tband_epoch == h->band_epoch) ? h->tband : nullptr;
    prof_begin(h, GH_K_LT);
    if (h->cfg.storage == GH_STORAGE_F64)
        hipLaunchKernelGGL(k_lt<double>, dim3((unsigned)nb), dim3(block), 0, h->stream, (const double *)h->band,
                           h->N, h->W, h->L, h->cfg.cond_mode, want_baked ? 1 : 0, h->cnt, h->nvalid, h->cmask,
                           h->minfo, h->lt, h->dstate, inc, (const win_desc *)nullptr, 0, walk_ranked_ok(h->wmode, h->L),
                           derived ? h->ht : (double *)nullptr, derived ? h->yt : (double *)nullptr, h->sm, (const double *)tb_lt);
    else
        hipLaunchKernelGGL(k_lt<float>, dim3((unsigned)nb), dim3(block), 0, h->stream, (const float *)h->band,
                           h->N, h->W, h->L, h->cfg.cond_mode, want_baked ? 1 : 0, h->cnt, h->nvalid, h->cmask,
                           h->minfo, h->lt, h->dstate, inc, (const win_desc *)nullptr, 0, walk_ranked_ok(h->wmode, h->L),
                           derived ? h->ht : (double *)nullptr, derived ? h->yt : (double *)nullptr, h->sm, (const float *)tb_lt);
    const int wl = h->W < h->L ? h->W : h->L;
    // algorithmic bytes: full = read the band cells within reach + write G; after a fused reweight = the two flags
    prof_end(h, GH_K_LT, inc ? 8.0
                             : (double)h->N * ((double)wl * CELL * esize(h) + (double)h->L * LT_BLK * 8.0));
    { int rc_ = post_launch(h, "k_lt"); if (rc_) return rc_; }
    // mixed radix for the segment-parallel extension (segmix.hpp): how many states enter a target at most, taken whenever the table
    // may have been rebuilt (k_lt zeroes it then; behind an incremental refresh the old bound stands: candidates only disappear)
    if (mixed_allowed(h)) {
        hipLaunchKernelGGL(k_classify, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, (const uint32_t *)h->cmask, h->N, h->L, h->dstate);
        int rc_ = post_launch(h, "k_classify"); if (rc_) return rc_;
    }
    h->dirty_lt = false;
    h->lt_inc_path = nullptr;
    h->lt_baked = want_baked;
    if (inc && !derived && h->ht) h->ht_stale = true;       // (a full build writes them from the band whatever `derived` says -- k_lt)
    else if (!inc) h->ht_stale = !derived && h->ht != nullptr;
    return GH_OK;
}

extern "C" int gh_counts_at(gh_t *h, int p, double out[8])
{
    if (!h || !out) return fail(GH_ERR_ARG, "null argument");
    if (p < 0 || p > h->N) return fail(GH_ERR_ARG, "position %d outside [0,%d]", p, h->N);
    if (set_dev(h)) return GH_ERR_HIP;
    int rc = ensure_marg(h);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, h->cnt + (size_t)p * 8, 64, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return GH_OK;
}

extern "C" int gh_marginal_of_at(gh_t *h, int s, int p, double *out)
{
    if (!h || !out) return fail(GH_ERR_ARG, "null argument");
    if (s < 0 || s >= NSYM) return fail(GH_ERR_SYMBOL, "symbol index %d out of range", s);
    if (p < 0 || p > h->N) return fail(GH_ERR_ARG, "position %d outside [0,%d]", p, h->N);
    if (set_dev(h)) return GH_ERR_HIP;
    int rc = ensure_marg(h);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, h->marg + (size_t)p * 8 + s, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return GH_OK;
}

extern "C" int gh_edge_weights_at(gh_t *h, int p, const uint8_t *path, double w[GH_NSYM], int *cand_mask)
{
    if (!h || !path || !w || !cand_mask) return fail(GH_ERR_ARG, "null argument");
    if (p < 1 || p > h->N) return fail(GH_ERR_ARG, "position %d outside [1,%d]", p, h->N);
    if (set_dev(h)) return GH_ERR_HIP;
    int rc = ensure_marg(h);
    if (rc) return rc;
    const int lmax = h->L < p ? h->L : p;
    std::vector<uint8_t> hist(lmax);
    for (int l = 1; l <= lmax; l++) {
        if (path[p - l] >= NSYM) return fail(GH_ERR_SYMBOL, "path[%d] = %d is not a symbol index", p - l, path[p - l]);
        hist[l - 1] = path[p - l];
    }
    // (the handle's own scratch: the lone-path buffer holds the history, eight doubles the answer -- no allocation per call)
    uint8_t *d_hist = h->d_path;
    if (!h->ew_buf && hipMalloc((void **)&h->ew_buf, 8 * sizeof(double)) != hipSuccess) return fail(GH_ERR_NOMEM, "hipMalloc failed");
    double *d_w = h->ew_buf;
    hipError_t e = hipMemcpyAsync(d_hist, hist.data(), lmax, hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return fail(GH_ERR_HIP, "gh_edge_weights_at failed: %s", hipGetErrorString(e));
    if (h->cfg.storage == GH_STORAGE_F64)
        hipLaunchKernelGGL(k_edge_weights<double>, dim3(1), dim3(64), 0, h->stream, (const double *)h->band, h->W,
                           h->cfg.cond_mode, p, h->L, h->cfg.marginal_term, h->cnt, h->marg, h->nvalid, h->cmask,
                           d_hist, d_w, (int *)(d_w + 7));
    else
        hipLaunchKernelGGL(k_edge_weights<float>, dim3(1), dim3(64), 0, h->stream, (const float *)h->band, h->W,
                           h->cfg.cond_mode, p, h->L, h->cfg.marginal_term, h->cnt, h->marg, h->nvalid, h->cmask,
                           d_hist, d_w, (int *)(d_w + 7));
    double hw[8];
    e = hipMemcpyAsync(hw, d_w, sizeof hw, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(GH_ERR_HIP, "gh_edge_weights_at failed: %s", hipGetErrorString(e));
    memcpy(w, hw, 7 * sizeof(double));
    memcpy(cand_mask, &hw[7], sizeof(int));
    return GH_OK;
}

extern "C" int gh_gap_check(gh_t *h, int *first_gap)
{
    if (!h || !first_gap) return fail(GH_ERR_ARG, "null argument");
    if (set_dev(h)) return GH_ERR_HIP;
    int rc = ensure_marg(h);
    if (rc) return rc;
    int *d_gap = reinterpret_cast<int *>(&h->dstate->fill[5]);      // (a spare word of the state)
    int big = 0x7fffffff;
    HIPCHK(hipMemcpyAsync(d_gap, &big, 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_gap, dim3((h->N + 1 + 255) / 256), dim3(256), 0, h->stream, h->cnt, h->N, d_gap);
    int g;
    HIPCHK(hipMemcpyAsync(&g, d_gap, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *first_gap = (g == big) ? -1 : g;
    return GH_OK;
}

extern "C" int gh_export_cmask(gh_t *h, uint32_t *out)
{
    if (!h || !out) return fail(GH_ERR_ARG, "null argument");
    if (set_dev(h)) return GH_ERR_HIP;
    int rc = ensure_marg(h);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, h->cmask, (size_t)(h->N + 1) * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int p = 0; p <= h->N; p++) out[p] = CM_CAND(out[p]);      // (the device word also carries the symbols seen: kernels.hpp)
    return GH_OK;
}

extern "C" int gh_snapshot_original(gh_t *h)
{
    if (!h) return fail(GH_ERR_ARG, "null handle");
    if (set_dev(h)) return GH_ERR_HIP;
    int rc = ensure_marg(h);
    if (rc) return rc;
    hipLaunchKernelGGL(k_snapshot, dim3(((h->N + 1) * 8 + 255) / 256), dim3(256), 0, h->stream, h->minfo, h->minfo, h->N, (const win_desc *)nullptr);
    { int rc_ = post_launch(h, "k_snapshot"); if (rc_) return rc_; }
    h->have_orig = true;
    return GH_OK;
}

static int alloc_lt(gh_handle *h)
{
    if (h->lt && h->lt_L == h->L) return GH_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->lt) hipFree(h->lt);
    if (h->ht) hipFree(h->ht);
    if (h->yt) hipFree(h->yt);
    h->lt = nullptr; h->ht = nullptr; h->yt = nullptr;
    size_t bytes = (size_t)(h->N + LT_PAD) * h->L * LT_BLK * sizeof(double);
    hipError_t e = hipMalloc((void **)&h->lt, bytes);
    if (e == hipSuccess && walk_depth2_ok(h->wmode, h->L) && !seg_ok(h->wmode, h->L)) {
        // the tables the depth-2 walker's loaders copy (k_lt keeps them in step with lt)
        e = hipMalloc((void **)&h->ht, (size_t)(h->N + WALK_TPAD) * 64 * sizeof(double));
        const size_t ypos = (size_t)16 * deep_nyp(h->L);
        if (e == hipSuccess && ypos) e = hipMalloc((void **)&h->yt, (size_t)(h->N + WALK_TPAD) * ypos * sizeof(double));
    }
    if (e != hipSuccess) return fail(GH_ERR_NOMEM, "hipMalloc(%zu) for the conditional table failed", bytes);
    h->lt_L = h->L;
    h->dirty_lt = true; h->lt_inc_path = nullptr;
    return GH_OK;
}

// path extension / reweight -------------------------------------------------------------------
static int walk_threads()
{
    static const int n = getenv("GH_WALK_THREADS") ? atoi(getenv("GH_WALK_THREADS")) : 512;
    return (n >= 192 && n <= 512 && n % 64 == 0) ? n : 512;
}
#define WALK_THREADS walk_threads()
#define WALK_MAX_LC 16

// Which path extension runs.  GH_WALK pins one for A/B measurements and tests; it is read when a handle is created:
//   unset / "seg"   segment-parallel (segwalk.hpp) for L <= SEG_MAX_L, else the serial walkers below
//   "spec"          k_walk_spec: one wavefront, speculation depth 2 where the window allows it, else depth 1
//   "spec1"         k_walk_spec at depth 1       "src"  k_walk_src (no speculation)
static bool seg_ok(int wm, int L) { return wm == WM_SEG && L >= 1 && L <= SEG_MAX_L; }

// Whether the depth-2 walker can run for this L: then k_lt may build G over candidate ranks (kernels.hpp).
static bool walk_depth2_ok(int wm, int L)
{
    const bool off = wm == WM_SRC || wm == WM_SPEC1;
    return !off && L >= 2 && L <= WALK_MAX_LC && walk_chunk(L, false) > 0 && walk_chunk(L, true) > 0 && WALK_THREADS == 512;
}

// candidate-pool segments (cwalk.hpp) for the lag counts above: spins only, narrow windows only (checked on the device)
static bool cw_ok(int wm, int L) { return wm == WM_SEG && L >= CW_MIN_L && L <= CW_MAX_LG; }
// beyond what a 64-bit state holds (2 bits per pick over ranks, 3 over symbols): states as bytes next to their hash, k_cwalkg
static bool cw_digit_mode(const gh_handle *h) { return h->cw_wide ? h->L > CW_MAX_L5 : h->L > CW_MAX_L; }
// ... of which 33..64 lags over ranks and 22..40 over the symbols are walked by k_cwalk2 (registers and an unrolled block, as
// k_cwalk) instead of k_cwalkg (GH_CWALK2=0: k_cwalkg, for the tests and A/B; read per launch)
static bool cw2_ok(const gh_handle *h)
{
    if (getenv("GH_CWALK2") && atoi(getenv("GH_CWALK2")) == 0) return false;
    return h->cw_wide ? (h->L > CW_MAX_L5 && h->L <= CW2_MAX_L5) : (h->L > CW_MAX_L && h->L <= CW2_MAX_L);
}

// single windows: may k_lt build the ranked layout?  (the segment-parallel walk reads either layout)
static bool walk_ranked_ok(int wm, int L) { return seg_ok(wm, L) || cw_ok(wm, L) || walk_depth2_ok(wm, L); }

template <int LC>
static void launch_walk_lc(bool spec, size_t lds, hipStream_t stream, const walk_params &P, int grid, const win_desc *wd, int spin)
{
    if (spec) {
        hipFuncSetAttribute((const void *)k_walk_spec<LC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_walk_spec<LC>), dim3(grid), dim3(WALK_THREADS), lds, stream, P, wd, spin);
    } else {
        hipFuncSetAttribute((const void *)k_walk_src<LC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_walk_src<LC>), dim3(grid), dim3(WALK_THREADS), lds, stream, P, wd, spin);
    }
}

static void launch_walk_src(int LC, bool spec, size_t lds, hipStream_t stream, const walk_params &P, int grid = 1,
                            const win_desc *wd = nullptr, int spin = 0)
{
#define GH_WALK_CASE(n) case n: launch_walk_lc<n>(spec, lds, stream, P, grid, wd, spin); break;
    switch (LC) {
        GH_WALK_CASE(1) GH_WALK_CASE(2) GH_WALK_CASE(3) GH_WALK_CASE(4)
        GH_WALK_CASE(5) GH_WALK_CASE(6) GH_WALK_CASE(7) GH_WALK_CASE(8)
        GH_WALK_CASE(9) GH_WALK_CASE(10) GH_WALK_CASE(11) GH_WALK_CASE(12)
        GH_WALK_CASE(13) GH_WALK_CASE(14) GH_WALK_CASE(15) GH_WALK_CASE(16)
    }
#undef GH_WALK_CASE
}

// launches the path-extension kernel for `grid` windows (grid == 1: the handle's own buffers in P)
static void launch_walk_any(int wm, int N, int L, walk_params P, hipStream_t stream, int grid, const win_desc *wd, int spin)
{
    const int chunk = L <= WALK_MAX_LC ? walk_chunk(L, false) : 0;
    if (chunk > 0) {
        P.chunk = chunk;                    // k_walk_src; k_walk_spec takes walk_chunk(L, variant) itself
        size_t lds = walk_lds_bytes(L, false);
        if (walk_depth2_ok(wm, L) && walk_lds_bytes(L, true) > lds) lds = walk_lds_bytes(L, true);
        const bool spec = wm != WM_SRC;
        P.depth2 = walk_depth2_ok(wm, L);
        launch_walk_src(L, spec, lds, stream, P, grid, wd, spin);
    } else {
        int hl = 16;
        while (hl <= L) hl <<= 1;
        P.chunk = 0; P.depth2 = 0;
        hipLaunchKernelGGL(k_walk_global, dim3(grid), dim3(64), (size_t)hl, stream, P, hl, wd, spin);
    }
}

// ---- segment-parallel walk ---------------------------------------------------------------------
static size_t max2(size_t a, size_t b) { return a > b ? a : b; }
// the largest value of f(class) over the state spaces a lag count can be walked in: 4 (ranked), 5 (symbols, L <= 5), 6 (mixed radix, L = 5);
// which one applies is decided on the device (seg_class), the host sizes for all
template <typename F> static size_t max_cls(int L, F f)
{
    size_t m = f(4);
    if (seg_radix_ok(5, L)) m = max2(m, f(5));
    if (L == SEGM_L) m = max2(m, f(SEG_CLS_MIXED));
    return m;
}

static int alloc_seg(gh_handle *h)
{
    if (h->seg_hist && h->seg_L == h->L) return GH_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    hipFree(h->seg_hist); hipFree(h->seg_maps); hipFree(h->seg_pmaps); hipFree(h->seg_gmaps); hipFree(h->seg_min); hipFree(h->lmsel1);
    hipFree(h->seg_smin); hipFree(h->seg_gmin); hipFree(h->cm5snap);
    h->seg_hist = nullptr; h->seg_maps = nullptr; h->seg_pmaps = nullptr; h->seg_gmaps = nullptr; h->seg_min = nullptr; h->lmsel1 = nullptr;
    h->seg_smin = nullptr; h->seg_gmin = nullptr; h->cm5snap = nullptr;
    // the layout is decided on the device (st->ranked): size for both (L = 6: ranked tables only, see gh_spin)
    const int N_ = h->N, L_ = h->L;
    const size_t hist_b = max_cls(L_, [&](int R) { const seg_geom g = seg_geometry(N_, L_, R); return (size_t)g.S * g.NW * g.NS; }) * 4;
    const size_t maps_b = max_cls(L_, [&](int R) { const seg_geom g = seg_geometry(N_, L_, R); return (size_t)g.S * g.NS; }) * 2;
    const size_t gmaps_b = max_cls(L_, [&](int R) { const seg_geom g = seg_geometry(N_, L_, R); return (size_t)g.G1 * g.NS; }) * 2;
    hipError_t e = hipMalloc((void **)&h->seg_hist, hist_b);
    if (e == hipSuccess) e = hipMalloc((void **)&h->seg_maps, maps_b);
    if (e == hipSuccess) e = hipMalloc((void **)&h->seg_pmaps, maps_b + 16);        // (+ slack: 16-byte copies in k_rw)
    if (e == hipSuccess) e = hipMalloc((void **)&h->seg_gmaps, gmaps_b + 16);      // (+ slack: k_rw copies the group maps four bytes at a time)
    if (e == hipSuccess) e = hipMalloc((void **)&h->seg_smin, maps_b * 4);          // doubles where the maps hold 2-byte states
    h->seg_smin_bytes = maps_b * 4;
    if (e == hipSuccess) e = hipMalloc((void **)&h->seg_gmin, gmaps_b * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&h->cm5snap, (size_t)h->N + 2);
    // (behind the segment minima: k_rwseg's flag words, one int2 per segment)
    if (e == hipSuccess) e = hipMalloc((void **)&h->seg_min, CW_MAX_SEG * (sizeof(double) + sizeof(int2)));
    if (e == hipSuccess) e = hipMalloc((void **)&h->lmsel1, ((size_t)h->N + 2) * sizeof(double));
    if (e != hipSuccess) return fail(GH_ERR_NOMEM, "hipMalloc for the segment-parallel walk failed: %s", hipGetErrorString(e));
    h->seg_L = h->L;
    return GH_OK;
}

template <int LC>
static void launch_seg_lc(gh_handle *h, const seg_params &P)
{
    hipStream_t stream = h->stream;
    const int N = h->N, dev = h->dev;
    const size_t lds_seg = max_cls(LC, [&](int R) { return seg_lds_total(R, LC); });
    const size_t lds_scan = max_cls(LC, [&](int R) { return scan_lds_bytes(N, LC, R); });
    const size_t lds_emit = max_cls(LC, [&](int R) { return emit_lds_bytes(N, LC, R); });
    // per instantiation and device: raise the dynamic-LDS limit once, not on every launch
    static std::atomic<size_t> set_seg[64], set_scan[64], set_emit[64], set_segt[64], set_scant[64];      // (gh_batch_spin runs gh_spin on several host threads)
    const int dv = dev & 63;
    const int S = (int)max_cls(LC, [&](int R) { return (size_t)seg_geometry(N, LC, R).S; }), G1 = (int)max_cls(LC, [&](int R) { return (size_t)seg_geometry(N, LC, R).G1; });
    if (h->fuse) {
        // spins without k_emit: k_seg / k_scan also carry the minimum marginals (TRACK), k_rw chains the group maps itself
        if (lds_seg > set_segt[dv]) { hipFuncSetAttribute((const void *)k_seg<LC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_seg); set_segt[dv] = lds_seg; }
        if (lds_scan > set_scant[dv]) { hipFuncSetAttribute((const void *)k_scan<LC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_scan); set_scant[dv] = lds_scan; }
        prof_begin(h, GH_K_SEG);
        hipLaunchKernelGGL((k_seg<LC, true>), dim3(S), dim3(SEG_THREADS), lds_seg, stream, P);
        prof_end(h, GH_K_SEG, (double)N * (double)LC * CELL * esize(h));
        hipLaunchKernelGGL((k_scan<LC, true>), dim3(G1), dim3(SEG_THREADS), lds_scan, stream, P);
        return;
    }
    if (lds_seg > set_seg[dv]) { hipFuncSetAttribute((const void *)k_seg<LC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_seg); set_seg[dv] = lds_seg; }
    if (lds_scan > set_scan[dv]) { hipFuncSetAttribute((const void *)k_scan<LC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_scan); set_scan[dv] = lds_scan; }
    if (lds_emit > set_emit[dv]) { hipFuncSetAttribute((const void *)k_emit<LC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_emit); set_emit[dv] = lds_emit; }
    prof_begin(h, GH_K_SEG);
    hipLaunchKernelGGL((k_seg<LC, false>), dim3(S), dim3(SEG_THREADS), lds_seg, stream, P);
    // algorithmic bytes of k_seg: the conditional lookups of the extension (L history cells per step)
    prof_end(h, GH_K_SEG, (double)N * (double)LC * CELL * esize(h));
    // short memories / small windows: every segment map fits the LDS of the emitting workgroup, which composes them itself
    const size_t lds_small = max_cls(LC, [&](int R) { return emit_small_lds_bytes(N, LC, R); });
    static const bool no_small = getenv("GH_EMIT_SMALL") && atoi(getenv("GH_EMIT_SMALL")) == 0;
    if (lds_small <= 64 * 1024 && !no_small) {
        static std::atomic<size_t> set_small[64];
        if (lds_small > set_small[dv]) { hipFuncSetAttribute((const void *)k_emit_small<LC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small); set_small[dv] = lds_small; }
        hipLaunchKernelGGL((k_emit_small<LC>), dim3(S), dim3(SEG_THREADS), lds_small, stream, P);
        return;
    }
    hipLaunchKernelGGL((k_scan<LC, false>), dim3(G1), dim3(SEG_THREADS), lds_scan, stream, P);
    hipLaunchKernelGGL((k_emit<LC>), dim3(S), dim3(SEG_THREADS), lds_emit, stream, P);
}

static int launch_seg_walk(gh_handle *h, uint8_t *d_path, double *d_lmsel, int rearm, int check_masks)
{
    int rc = alloc_seg(h);
    if (rc) return rc;
    seg_params P;
    P.N = h->N; P.L = h->L; P.rearm = rearm; P.check_masks = check_masks;
    P.G = h->lt; P.minfo = h->minfo; P.rinfo = h->rinfo; P.mt = h->cfg.marginal_term; P.nanp = h->cfg.marginal_term && h->cfg.offer_zero; P.sm = h->sm; P.st = h->dstate;
    P.hist = h->seg_hist; P.maps = h->seg_maps; P.pmaps = h->seg_pmaps; P.gmaps = h->seg_gmaps; P.segmin = h->seg_min;
    P.smin = h->seg_smin; P.gmin = h->seg_gmin; P.cm5snap = h->cm5snap;
    P.rws = 0; P.W = h->W; P.esz = (int)esize(h); P.band = h->band; P.halo = h->rws ? h->seg_halo : nullptr; P.patch_off = 0;
    P.rwflags = reinterpret_cast<int2 *>(h->seg_min + CW_MAX_SEG);
    P.path_out = d_path; P.lmsel = d_lmsel ? d_lmsel : h->lmsel1;      // (lmsel1 exists only behind alloc_seg)
    if (h->L < 1 || h->L > SEG_MAX_L_NARROW) return fail(GH_ERR_STATE, "segment-parallel walk needs L <= %d", SEG_MAX_L_NARROW);
    prof_begin(h, GH_K_WALK);
    switch (h->L) {
        case 1: launch_seg_lc<1>(h, P); break;
        case 2: launch_seg_lc<2>(h, P); break;
        case 3: launch_seg_lc<3>(h, P); break;
        case 4: launch_seg_lc<4>(h, P); break;
        case 5: launch_seg_lc<5>(h, P); break;
        case 6: launch_seg_lc<6>(h, P); break;      // (ranked tables only: gh_spin decides)
    }
    prof_end(h, GH_K_WALK, (double)h->N * ((1.0 + (double)h->L) * CELL * esize(h) + 28.0));
    return post_launch(h, "k_seg/k_scan/k_emit");
}

// ---- k_rwseg: reweight of the path before + this path's k_seg in one launch, then k_scan, k_emit (segwalk.hpp) ----------
static size_t rws_lds_bytes(int LC, bool five)
{
    (void)five;
    size_t b = max_cls(LC, [&](int R) { return seg_lds_total(R, LC); });
    if (b < (size_t)SEG_THREADS * 8) b = (size_t)SEG_THREADS * 8;      // the reweight phase's reduction scratch
    return (b + 15) & ~(size_t)15;
}

// where the patch lies in k_rwseg's dynamic LDS for this handle (behind k_seg's regions for either radix and, under the column
// conditionals, behind the staged band blocks of the workgroup's positions); the kernel needs this + sizeof(seg_patch)
static size_t rws_patch_off(const gh_handle *h, int LC)
{
    const bool five = seg_radix_ok(5, LC);
    const seg_geom g4 = seg_geometry(h->N, LC, 4), g5 = five ? seg_geometry(h->N, LC, 5) : g4;
    size_t off = rws_lds_bytes(LC, five);
    if (h->cfg.cond_mode == GH_COND_C || h->cfg.cond_mode == GH_COND_E) {
        const int longest = g4.seglen > g5.seglen ? g4.seglen : g5.seglen;
        const size_t need = (size_t)SEG_THREADS * 8 + (size_t)(longest + LC + 1) * NSYM * h->W * NSYM * esize(h);
        if (need > off) off = (need + 15) & ~(size_t)15;
    }
    return off;
}
#define RWS_LDS_MAX (160 * 1024 - 3 * 1024 - 512)      /* what a workgroup may have of the CU's 160 KB, less the kernel's static variables */

template <typename T, int LC, bool COL>
static void launch_rwseg_lc(gh_handle *h, seg_params P, const rws_params &Q)
{
    constexpr bool five = seg_radix_ok(5, LC);
    const int N = h->N;
    (void)five;
    const int S = (int)max_cls(LC, [&](int R) { return (size_t)seg_geometry(N, LC, R).S; }), G1 = (int)max_cls(LC, [&](int R) { return (size_t)seg_geometry(N, LC, R).G1; });
    const size_t off = rws_patch_off(h, LC);
    const size_t lds = off + sizeof(seg_patch);
    const size_t lds_scan = max_cls(LC, [&](int R) { return scan_lds_bytes(N, LC, R); });
    const size_t lds_emit = max_cls(LC, [&](int R) { return emit_lds_bytes(N, LC, R); });
    static std::atomic<size_t> set_rws[64], set_scan[64], set_emit[64];
    const int dv = h->dev & 63;
    if (lds > set_rws[dv]) { hipFuncSetAttribute((const void *)k_rwseg<T, LC, COL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); set_rws[dv] = lds; }
    if (lds_scan > set_scan[dv]) { hipFuncSetAttribute((const void *)k_scan<LC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_scan); set_scan[dv] = lds_scan; }
    if (lds_emit > set_emit[dv]) { hipFuncSetAttribute((const void *)k_emit<LC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_emit); set_emit[dv] = lds_emit; }
    P.patch_off = (int)off;
    P.rws = 1;
    prof_begin(h, GH_K_RWSEG);
    hipLaunchKernelGGL((k_rwseg<T, LC, COL>), dim3(S), dim3(SEG_THREADS), lds, h->stream, P, Q);
    // the extension's conditional lookups + what the reweight of a path reads and writes (the figure GH_K_REWEIGHT quotes)
    const int wl = h->W < LC ? h->W : LC;
    prof_end(h, GH_K_RWSEG, (double)N * (double)LC * CELL * esize(h) +
             (double)(N + 1) * ((double)h->W * 2.0 * esize(h) + 1.0 + CELL * esize(h) + 2 * 64 + 88 + 8) +
             (double)N * ((double)wl * 7 * esize(h) + (double)LC * LT_ROW * 8.0));
    // short memories / small windows: every segment map fits the LDS of the emitting workgroup, which composes them itself and
    // takes k_scan's look at what the reweight found as well: two launches per path
    const size_t lds_small = max_cls(LC, [&](int R) { return emit_small_lds_bytes(N, LC, R); });
    if (lds_small <= 64 * 1024 && !(getenv("GH_EMIT_SMALL") && atoi(getenv("GH_EMIT_SMALL")) == 0)) {
        static std::atomic<size_t> set_small[64];
        if (lds_small > set_small[dv]) { hipFuncSetAttribute((const void *)k_emit_small<LC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small); set_small[dv] = lds_small; }
        hipLaunchKernelGGL((k_emit_small<LC>), dim3(S), dim3(SEG_THREADS), lds_small, h->stream, P);
        return;
    }
    hipLaunchKernelGGL((k_scan<LC, false>), dim3(G1), dim3(SEG_THREADS), lds_scan, h->stream, P);
    hipLaunchKernelGGL((k_emit<LC>), dim3(S), dim3(SEG_THREADS), lds_emit, h->stream, P);
}

// reweights along d_prev (record d_prev_rec, removed mass into slot prev_slot) and walks the next path into d_path
static int launch_rwseg(gh_handle *h, const uint8_t *d_prev, gh_path_rec *d_prev_rec, int prev_slot, double min_remove,
                        uint8_t *d_path, double *d_lmsel, int check_masks)
{
    int rc = alloc_seg(h);
    if (rc) return rc;
    seg_params P;
    P.N = h->N; P.L = h->L; P.rearm = 1; P.check_masks = check_masks;
    P.G = h->lt; P.minfo = h->minfo; P.rinfo = h->rinfo; P.mt = h->cfg.marginal_term; P.nanp = h->cfg.marginal_term && h->cfg.offer_zero; P.sm = h->sm; P.st = h->dstate;
    P.hist = h->seg_hist; P.maps = h->seg_maps; P.pmaps = h->seg_pmaps; P.gmaps = h->seg_gmaps; P.segmin = h->seg_min;
    P.smin = h->seg_smin; P.gmin = h->seg_gmin; P.cm5snap = h->cm5snap;
    P.rws = 1; P.W = h->W; P.esz = (int)esize(h); P.band = h->band; P.halo = h->seg_halo; P.patch_off = 0;
    P.rwflags = reinterpret_cast<int2 *>(h->seg_min + CW_MAX_SEG);
    P.path_out = d_path; P.lmsel = d_lmsel;
    rws_params Q;
    Q.band = h->band; Q.cnt = h->cnt; Q.marg = h->marg; Q.minfo = h->minfo; Q.rinfo = h->need_rinfo ? h->rinfo : nullptr; Q.G = h->lt;
    Q.nvalid = h->nvalid; Q.cmask = h->cmask; Q.path = d_prev; Q.min_remove = min_remove;
    Q.partial = h->partial + (size_t)prev_slot * h->spin_partial_stride;
    Q.rec = d_prev_rec; Q.cond_mode = h->cfg.cond_mode; Q.offer_zero = h->cfg.offer_zero;
    if (h->L < 1 || h->L > SEG_MAX_L_NARROW) return fail(GH_ERR_STATE, "k_rwseg needs L <= %d", SEG_MAX_L_NARROW);   // before the bracket opens
    prof_begin(h, GH_K_WALK);
    const bool f64 = h->cfg.storage == GH_STORAGE_F64;
    const bool col = h->cfg.cond_mode == GH_COND_C || h->cfg.cond_mode == GH_COND_E;
    switch (h->L) {
#define RWS_CASE(n) case n: if (f64) { if (col) launch_rwseg_lc<double, n, true>(h, P, Q); else launch_rwseg_lc<double, n, false>(h, P, Q); } \
                            else { if (col) launch_rwseg_lc<float, n, true>(h, P, Q); else launch_rwseg_lc<float, n, false>(h, P, Q); } break;
        RWS_CASE(1) RWS_CASE(2) RWS_CASE(3) RWS_CASE(4) RWS_CASE(5) RWS_CASE(6)
#undef RWS_CASE
        default: return fail(GH_ERR_STATE, "k_rwseg needs L <= %d", SEG_MAX_L_NARROW);
    }
    prof_end(h, GH_K_WALK, (double)h->N * ((1.0 + (double)h->L) * CELL * esize(h) + 28.0));
    // (what launch_reweight_marg notes behind a fused reweight: the table is current up to the rows this path's reweight wrote)
    h->lt_inc_path = d_prev;
    h->lt_inc_seg = true;
    h->dirty_lt = true;
    h->dirty_marg = false;
    h->band_epoch++;
    return post_launch(h, "k_rwseg/k_scan/k_emit");
}

// serial walkers: the record is closed by the walker itself.  Segment-parallel: by the k_marg<T,true> that follows
// (spins) or by k_seg_fin (lone gh_generate_path); d_lmsel receives the selected log-marginals for k_hp.
static int launch_walk(gh_handle *h, uint8_t *d_path, gh_path_rec *d_rec, double min_remove, int rearm, double *d_lmsel = nullptr,
                       int check_masks = 0)
{
    if (seg_ok(h->wmode, h->L) || h->seg6) return launch_seg_walk(h, d_path, d_lmsel, rearm, check_masks);
    walk_params P;
    P.N = h->N; P.L = h->L; P.chunk = 0; P.rearm = rearm;
    P.G = h->lt; P.Ht = h->ht; P.Yt = h->yt; P.minfo = h->minfo;
    P.path_out = d_path; P.rec = d_rec; P.st = h->dstate; P.min_remove = min_remove; P.sm = h->sm;
    prof_begin(h, GH_K_WALK);
    launch_walk_any(h->wmode, h->N, h->L, P, h->stream, 1, nullptr, 0);
    // algorithmic bytes, SURVEY 8(d): per step the marginal cell + L history cells (49 elements each) + the original
    // marginals (7 x 4 B).  What this build's layout needs per step is one table row: N * (L * 40 + 25) bytes.
    prof_end(h, GH_K_WALK, (double)h->N * ((1.0 + (double)h->L) * CELL * esize(h) + 28.0));
    { int rc_ = post_launch(h, "k_walk"); if (rc_) return rc_; }
    return GH_OK;
}

// reweight along d_path AND refresh the marginal tables in one pass (k_marg<T, true>)
// room for `slots` sets of per-block partial sums of the removed mass
static int ensure_partial(gh_handle *h, int nb, int slots)
{
    const size_t need = (size_t)nb * (slots > 0 ? slots : 1);
    if (need > (size_t)h->partial_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->partial) hipFree(h->partial);
        h->partial = nullptr; h->partial_cap = 0;
        HIPCHK(hipMalloc((void **)&h->partial, need * sizeof(double)));
        h->partial_cap = (int)need;
    }
    return GH_OK;
}

// slot < 0: reduce the removed mass right behind the pass (k_reweight_finish).  slot >= 0 (gh_spin): keep this
// path's partial sums in their own slot, the caller reduces all paths with one k_reweight_finish_all at the end.
// k_rw's lane group: 32 lanes per position where the band or the lag count exceeds 8 (all distances and lags in one
// round), else 8; blocks per path accordingly (also the stride of the per-path partial sums of the removed mass)
// (16: row conditionals with bands up to 32 and at most 16 lags -- the distances beyond 16 take a second round, the table
// entries are dealt out over the group anyway, and the marginals, 7 lanes of every group, cost a wavefront half as much)
static bool rw_col(const gh_handle *h) { return h->cfg.cond_mode == GH_COND_C || h->cfg.cond_mode == GH_COND_E; }
// column conditionals beyond the 8-lane groups: k_rw reads its columns from a to-major copy of the band (GH_RW_TBAND=0: from the staged block)
static bool rw_tband(const gh_handle *h)
{
    static const bool off = getenv("GH_RW_TBAND") && atoi(getenv("GH_RW_TBAND")) == 0;
    return rw_col(h) && (h->W > 8 || h->L > 8) && !off;
}
static int rw_lanes(const gh_handle *h)
{
    if (!(h->W > 8 || h->L > 8)) return 8;
    static const bool no16 = getenv("GH_RW_LP16") && atoi(getenv("GH_RW_LP16")) == 0;
    return (h->W <= 32 && h->L <= 16 && !no16 && (!rw_col(h) || rw_tband(h))) ? 16 : 32;
}
static int rw_blocks(const gh_handle *h, bool seg) { return (int)(((size_t)(h->N + 1) * (seg ? rw_lanes(h) : 8) + 255) / 256); }

// seg: the walk just before was segment-parallel: the kernel reduces the minimum marginal itself, clamps it to `ratio`
// (= min_remove) and closes the record
static int launch_reweight_marg(gh_handle *h, const uint8_t *d_path, double ratio, int use_state, gh_path_rec *d_rec, int slot = -1,
                                bool seg = false, bool chained = false, int nseg_arg = 0, double *d_lmsel = nullptr)
{
    const int block = 256;
    const int nb = rw_blocks(h, seg);
    // (a spin strides its per-path partial sums by the widest kernel it may launch: see gh_spin)
    const int stride = (slot >= 0 && h->spin_partial_stride > nb) ? h->spin_partial_stride : nb;
    if (slot < 0) { int rc_ = ensure_partial(h, nb, 1); if (rc_) return rc_; }
    double *partial = h->partial + (slot > 0 ? (size_t)slot * stride : 0);
    if (slot >= 0 && stride > nb && !h->rws) HIPCHK(hipMemsetAsync(partial + nb, 0, sizeof(double) * (size_t)(stride - nb), h->stream));
    // in a spin the walker re-armed the flags when it finished; a lone reweight does it here
    if (!use_state) hipLaunchKernelGGL(k_rearm, dim3(1), dim3(64), 0, h->stream, h->dstate, (const win_desc *)nullptr, 0);
    // with a valid conditional table (conditional A or B, no marginal term) the kernel also rewrites the table rows
    // this path changes; k_lt then only has to confirm that no candidate mask moved
    // (chained: a spin without k_lt between its paths -- the table is current up to the rows the previous reweight rewrote, and
    // the k_seg that ran before this reweight has checked the candidate masks)
    const bool lt_ok = (!h->dirty_lt || chained) && h->lt && h->lt_L == h->L && (seg ? rw_incremental_ok(h) : lt_incremental_ok(h));
    double *lt_rows = lt_ok ? h->lt : nullptr;
    prof_begin(h, GH_K_REWEIGHT);
    if (seg) {
        // behind a segment-parallel walk (L <= SEG_MAX_L <= 8: one table row per lane): segwalk.hpp's k_rw
#define GH_RW_LAUNCH(T, LP, COL, FZ)                                                                                              \
    do {                                                                                                                          \
        /* COL: the band block of the workgroup's positions staged in LDS when it fits (k_rw) */                                   \
        const size_t blk_b = (size_t)(256 / LP) * NSYM * h->W * NSYM * sizeof(T);                                                 \
        /* | 2: `cnt` holds the row sums of the band as it stands (k_rw takes the rows a reweight does not touch from there) */     \
        const int stage = ((COL && !use_tb && blk_b <= 64 * 1024 && !(getenv("GH_RW_STAGE") && atoi(getenv("GH_RW_STAGE")) == 0)) ? 1 : 0) |  \
                          ((!h->dirty_marg && !(getenv("GH_RW_CNT") && atoi(getenv("GH_RW_CNT")) == 0)) ? 2 : 0) | ((COL && use_tb) ? 4 : 0); \
        const size_t lds_b = fuse_lds + ((stage & 1) ? blk_b : 0);                                                                \
        static std::atomic<size_t> set_lds[64];                                                                                   \
        if (lds_b > set_lds[h->dev & 63]) {                                                                                       \
            hipFuncSetAttribute((const void *)k_rw<T, LP, COL, FZ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);      \
            set_lds[h->dev & 63] = lds_b;                                                                                         \
        }                                                                                                                         \
        hipLaunchKernelGGL((k_rw<T, LP, COL, FZ>), dim3(nb), dim3(block), lds_b, h->stream, (T *)h->band, h->N, h->W, h->cnt, h->marg, h->nvalid, \
                           h->cmask, h->minfo, h->dstate, d_path, ratio, partial, lt_rows, h->L, h->cfg.cond_mode,                \
                           (const double *)h->seg_min, d_rec, nseg_arg, h->sm, h->cfg.offer_zero, h->need_rinfo ? h->rinfo : (double *)nullptr, stage, fz, (int)fuse_lds, \
                           (T *)(use_tb ? h->tband : nullptr));                                                                   \
    } while (0)
#define GH_RW_LAUNCH2(T, LP) do { if (col) GH_RW_LAUNCH(T, LP, true, false); else GH_RW_LAUNCH(T, LP, false, false); } while (0)
#define GH_RW_LAUNCHF(T) do { if (col) GH_RW_LAUNCH(T, 8, true, true); else GH_RW_LAUNCH(T, 8, false, true); } while (0)
        const bool wide = rw_lanes(h) == 32, mid = rw_lanes(h) == 16;
        const bool col = rw_col(h);                         // the table entries a reweighted cell feeds: a column
        // the to-major copy: made (or made again, when something else wrote the band since) right here, kept by the kernel
        const bool use_tb = rw_tband(h) && !(h->fuse && nseg_arg == 0);
        if (use_tb) {
            const size_t nel = h->n_cells * CELL;
            if (!h->tband && hipMalloc(&h->tband, nel * esize(h)) != hipSuccess) { h->tband = nullptr; return fail(GH_ERR_NOMEM, "hipMalloc for the to-major band failed"); }
            if (h->tband_epoch != h->band_epoch) {
                const unsigned nbt = (unsigned)((nel + 255) / 256);
                if (h->cfg.storage == GH_STORAGE_F64) hipLaunchKernelGGL(k_band_to_major<double>, dim3(nbt), dim3(256), 0, h->stream, (const double *)h->band, (double *)h->tband, nel, h->W);
                else hipLaunchKernelGGL(k_band_to_major<float>, dim3(nbt), dim3(256), 0, h->stream, (const float *)h->band, (float *)h->tband, nel, h->W);
            }
        }
        // behind k_seg + k_scan without a k_emit (h->fuse): the kernel finds its picks itself and writes the path to d_path / d_lmsel
        fuse_params fz;
        memset(&fz, 0, sizeof fz);
        size_t fuse_lds = 0;
        if (h->fuse && nseg_arg == 0) {
            fz.hist = h->seg_hist; fz.pmaps = h->seg_pmaps; fz.gmaps = h->seg_gmaps; fz.gmin = h->seg_gmin; fz.cm5snap = h->cm5snap;
            fz.path_out = const_cast<uint8_t *>(d_path); fz.lmsel = d_lmsel;
            fuse_lds = h->fuse_lds;
        }
        if (fz.hist) {                                      // (three-launch spins: lane groups of 8 only, gh_spin decides)
            if (h->cfg.storage == GH_STORAGE_F64) GH_RW_LAUNCHF(double); else GH_RW_LAUNCHF(float);
        } else if (mid) { if (h->cfg.storage == GH_STORAGE_F64) GH_RW_LAUNCH2(double, 16); else GH_RW_LAUNCH2(float, 16); }
        else if (h->cfg.storage == GH_STORAGE_F64) { if (wide) GH_RW_LAUNCH2(double, 32); else GH_RW_LAUNCH2(double, 8); }
        else { if (wide) GH_RW_LAUNCH2(float, 32); else GH_RW_LAUNCH2(float, 8); }
#undef GH_RW_LAUNCHF
#undef GH_RW_LAUNCH2
#undef GH_RW_LAUNCH
    } else if (h->cfg.storage == GH_STORAGE_F64)
        hipLaunchKernelGGL((k_marg<double, true>), dim3(nb), dim3(block), 0, h->stream, (double *)h->band, h->N, h->W,
                           h->cnt, h->marg, h->nvalid, h->cmask, h->minfo, h->dstate, (const win_desc *)nullptr,
                           d_path, ratio, use_state, partial, 0, lt_rows, h->L, h->cfg.cond_mode,
                           (const double *)nullptr, (gh_path_rec *)nullptr, h->sm, h->cfg.offer_zero, h->need_rinfo ? h->rinfo : (double *)nullptr);
    else
        hipLaunchKernelGGL((k_marg<float, true>), dim3(nb), dim3(block), 0, h->stream, (float *)h->band, h->N, h->W,
                           h->cnt, h->marg, h->nvalid, h->cmask, h->minfo, h->dstate, (const win_desc *)nullptr,
                           d_path, ratio, use_state, partial, 0, lt_rows, h->L, h->cfg.cond_mode,
                           (const double *)nullptr, (gh_path_rec *)nullptr, h->sm, h->cfg.offer_zero, h->need_rinfo ? h->rinfo : (double *)nullptr);
    if (slot < 0)
        hipLaunchKernelGGL(k_reweight_finish, dim3(1), dim3(256), 0, h->stream, partial, nb, h->dstate, use_state, d_rec,
                           (const win_desc *)nullptr, 0);
    // algorithmic bytes: the reweighted elements (read+write) + the marginal pass (read cell (p,p+1), write the tables)
    // (+ one band row read and L x 5 table entries written per (source, lag) when the table rows are rewritten too)
    const int wl = h->W < h->L ? h->W : h->L;
    prof_end(h, GH_K_REWEIGHT, (double)(h->N + 1) * ((double)h->W * 2.0 * esize(h) + 1.0 + CELL * esize(h) + 2 * 64 + 88 + 8) +
                               (lt_ok ? (double)h->N * ((double)wl * 7 * esize(h) + (double)h->L * LT_ROW * 8.0) : 0.0));
    { int rc_ = post_launch(h, "k_marg<reweight>"); if (rc_) return rc_; }
    h->lt_inc_path = lt_ok ? d_path : nullptr;
    h->lt_inc_seg = seg;
    h->dirty_lt = true;
    h->dirty_marg = false;
    h->band_epoch++;
    if (seg && rw_tband(h) && !(h->fuse && nseg_arg == 0)) h->tband_epoch = h->band_epoch;      // (the kernel wrote both)
    return GH_OK;
}

// the control words a spin starts from (stop, hole_at, n_done; lt_stale, cw_unres) and, for k_rwseg, the partial sums of the
// removed mass, in one launch instead of two small copies and a memset with a dispatch gap each
__global__ void k_spin_reset(dev_state *st, double *partial, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride) partial[q] = 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->stop = 0; st->hole_at = 0; st->n_done = 0; st->lt_stale = 0; st->cw_unres = 0; }
}

static int reset_spin_state(gh_handle *h, double *partial = nullptr, size_t n = 0)
{
    const int blocks = n ? (int)((n + 1023) / 1024 > 256 ? 256 : (n + 1023) / 1024) : 1;
    hipLaunchKernelGGL(k_spin_reset, dim3(blocks), dim3(256), 0, h->stream, h->dstate, partial, n);
    return post_launch(h, "k_spin_reset");
}

// ---- candidate-pool segment walk (cwalk.hpp): L = 6 .. 16, spins -------------------------------
static int alloc_cw(gh_handle *h)
{
    if (h->cw_keys) return GH_OK;
    // sized for whichever geometry a lag count gives this N (h->L can change between spins)
    const cw_geom ga = cw_geometry(h->N, CW_MIN_L), gb = cw_geometry(h->N, CW_MAX_L);
    const size_t S = (size_t)(ga.S > gb.S ? ga.S : gb.S);
    const size_t SNW = (size_t)ga.S * ga.NW5 > (size_t)gb.S * gb.NW5 ? (size_t)ga.S * ga.NW5 : (size_t)gb.S * gb.NW5;      // (8 picks per word is the wider one)
    hipError_t e = hipMalloc((void **)&h->cw_keys, sizeof(cw_key) * S * CW_K);
    if (e == hipSuccess) e = hipMalloc((void **)&h->cw_exits, sizeof(cw_key) * S * CW_K);
    if (e == hipSuccess) e = hipMalloc((void **)&h->cw_last_hit, sizeof(int32_t) * S * CW_K);
    if (e == hipSuccess) e = hipMalloc((void **)&h->cw_hist, sizeof(uint32_t) * SNW * CW_K);
    if (e == hipSuccess) e = hipMalloc((void **)&h->cw_npool, sizeof(int32_t) * S);
    // (the request lists twice: a launch appends to one set and consumes the other, cwalk.hpp)
    h->cw_S = S;
    if (e == hipSuccess) e = hipMalloc((void **)&h->cw_pend, sizeof(cw_key) * 2 * S * CW_K);
    if (e == hipSuccess) e = hipMalloc((void **)&h->cw_pend_exit, sizeof(cw_key) * 2 * S * CW_K);
    if (e == hipSuccess) e = hipMalloc((void **)&h->cw_pend_ready, sizeof(int32_t) * 2 * S * CW_K);
    if (e == hipSuccess) e = hipMemsetAsync(h->cw_pend_ready, 0, sizeof(int32_t) * 2 * S * CW_K, h->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&h->cw_phist, sizeof(uint32_t) * 2 * SNW * CW_K);
    if (e == hipSuccess) e = hipMalloc((void **)&h->cw_npend, sizeof(int32_t) * 2 * S);
    if (e == hipSuccess) e = hipMemsetAsync(h->cw_npend, 0, sizeof(int32_t) * 2 * S, h->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&h->cw_walked, S * CW_K);
    if (e == hipSuccess) e = hipMalloc((void **)&h->cw_nxt, S * CW_K);
    if (e == hipSuccess) e = hipMalloc((void **)&h->cw_true, sizeof(int32_t) * S);
    if (e == hipSuccess) e = hipMemsetAsync(h->cw_npool, 0, sizeof(int32_t) * S, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->cw_walked, 0, S * CW_K, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->cw_nxt, 0xff, S * CW_K, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->cw_last_hit, 0, sizeof(int32_t) * S * CW_K, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->cw_keys, 0, sizeof(cw_key) * S * CW_K, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->cw_exits, 0, sizeof(cw_key) * S * CW_K, h->stream);
    if (e == hipSuccess && !h->seg_min) e = hipMalloc((void **)&h->seg_min, CW_MAX_SEG * sizeof(double));      // (alloc_seg sizes for L <= 5 only)
    if (e != hipSuccess) {
        // all or nothing: a later call must not find cw_keys set next to pool buffers that never came to be
        hipStreamSynchronize(h->stream);
        hipFree(h->cw_pend_exit); hipFree(h->cw_pend_ready); hipFree(h->cw_phist);
        h->cw_pend_exit = nullptr; h->cw_pend_ready = nullptr; h->cw_phist = nullptr;
        hipFree(h->cw_keys); hipFree(h->cw_exits); hipFree(h->cw_last_hit); hipFree(h->cw_hist); hipFree(h->cw_npool); hipFree(h->cw_pend);
        hipFree(h->cw_npend); hipFree(h->cw_walked); hipFree(h->cw_nxt); hipFree(h->cw_true);
        h->cw_keys = nullptr; h->cw_exits = nullptr; h->cw_last_hit = nullptr; h->cw_hist = nullptr; h->cw_npool = nullptr; h->cw_pend = nullptr;
        h->cw_npend = nullptr; h->cw_walked = nullptr; h->cw_nxt = nullptr; h->cw_true = nullptr;
        return fail(GH_ERR_NOMEM, "hipMalloc for the candidate pools failed: %s", hipGetErrorString(e));
    }
    return GH_OK;
}

static cw_params cw_make_params(gh_handle *h, uint8_t *d_path, double *d_lmsel)
{
    cw_params P;
    memset(&P, 0, sizeof P);
    P.N = h->N; P.L = h->L; P.rearm = h->cw_no_rw ? 0 : 1; P.stamp = h->cw_stamp;
    P.G = h->lt; P.minfo = h->minfo; P.rinfo = h->rinfo; P.mt = h->cfg.marginal_term; P.sm = h->sm; P.st = h->dstate;
    P.keys = h->cw_keys; P.exits = h->cw_exits; P.last_hit = h->cw_last_hit; P.npool = h->cw_npool; P.walked = h->cw_walked; P.nxt = h->cw_nxt; P.pend = h->cw_pend; P.npend = h->cw_npend;
    P.hist = h->cw_hist; P.true_idx = h->cw_true; P.segmin = h->seg_min; P.path_out = d_path; P.lmsel = d_lmsel;
    // run-on (cwalk.hpp): how far a walker follows a new track beyond its own segment -- some two hundred positions: the launch
    // lasts as long as its longest walker (GH_CW_RUNON=k pins the number of segments, 0 = off; read per launch: the tests switch it)
    const cw_geom ggr = cw_geometry(h->N, h->L);
    int runon = ggr.seglen <= 64 ? 256 / ggr.seglen : 0;      // (C5, 98 positions per segment: 168 us per path without, 175 with two segments of run-on)
    if (runon > CW_RUNON) runon = CW_RUNON;
    // (states as bytes, beyond 32 lags: a step of k_cwalkg costs several times k_cwalk's, and a launch lasts as long as its longest
    // walker -- 736 us per path without run-on at L = 33, 855..897 with one to three segments of it)
    // (33..64 lags over ranks and 22..40 over the symbols go through k_cwalk2 since round 6: k_cwalk's step, and its run-on)
    if (cw_digit_mode(h) && !cw2_ok(h)) runon = 0;
    if (getenv("GH_CW_RUNON")) runon = atoi(getenv("GH_CW_RUNON"));
    if (runon > CW_RUNON) runon = CW_RUNON;
    const bool no_runon = runon <= 0;
    P.runon = runon;
    P.pend_c = P.pend; P.npend_c = P.npend;
    {
        // the set this launch appends to / the set it consumes (what the launch before appended to)
        const cw_geom gg = cw_geometry(h->N, h->L);
        const size_t nw = (size_t)(h->cw_wide ? gg.NW5 : gg.NW);
        const size_t a = (size_t)(h->cw_pp & 1), c = a ^ 1;
        P.pend = h->cw_pend + a * h->cw_S * CW_K;            P.pend_c = h->cw_pend + c * h->cw_S * CW_K;
        P.npend = h->cw_npend + a * h->cw_S;                 P.npend_c = h->cw_npend + c * h->cw_S;
        if (!no_runon) {
            P.pend_exit = h->cw_pend_exit + a * h->cw_S * CW_K;   P.pend_exit_c = h->cw_pend_exit + c * h->cw_S * CW_K;
            P.pend_ready = h->cw_pend_ready + a * h->cw_S * CW_K; P.pend_ready_c = h->cw_pend_ready + c * h->cw_S * CW_K;
            P.phist = h->cw_phist + a * gg.S * nw * CW_K;         P.phist_c = h->cw_phist + c * gg.S * nw * CW_K;
        }
        if (cw_digit_mode(h)) {
            const size_t bytes = (size_t)gg.S * CW_K * h->cw_LD;
            P.pend_d = h->cw_pend_d + a * bytes;             P.pend_d_c = h->cw_pend_d + c * bytes;
            P.pend_exit_d = h->cw_pend_exit_d + a * bytes;   P.pend_exit_d_c = h->cw_pend_exit_d + c * bytes;
        }
    }
    if (cw_digit_mode(h)) {
        P.keys_d = h->cw_keys_d; P.exits_d = h->cw_exits_d; P.LD = h->cw_LD;
        unsigned long long hh = 0xcbf29ce484222325ull;      // cw_hash_digits of L zero bytes: the start state
        for (int l = 0; l < h->L; l++) { hh ^= 0u; hh *= 0x100000001b3ull; }
        hh ^= hh >> 32; hh *= 0x9e3779b97f4a7c15ull; hh ^= hh >> 29;
        P.key0 = hh;
    }
    return P;
}

template <int LC, int R>
static void launch_cwalk_lc(const cw_params &P, hipStream_t stream, int S, int dev)
{
    static std::atomic<bool> set[64];
    if (!set[dev & 63]) {
        hipFuncSetAttribute((const void *)k_cwalk<LC, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cw_lds_bytes(LC, R));
        set[dev & 63] = true;
    }
    hipLaunchKernelGGL((k_cwalk<LC, R>), dim3(S), dim3(CW_K * cw_lanes(R)), cw_lds_bytes(LC, R), stream, P);
}

template <int LC, int R = 4>
static void launch_cwalk2_lc(const cw_params &P, hipStream_t stream, int S, int dev)
{
    static std::atomic<bool> set[64];
    if (!set[dev & 63]) {
        hipFuncSetAttribute((const void *)k_cwalk2<LC, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cw2_lds_bytes(LC, R));
        set[dev & 63] = true;
    }
    hipLaunchKernelGGL((k_cwalk2<LC, R>), dim3(S), dim3(CW_K * cw_lanes(R)), cw2_lds_bytes(LC, R), stream, P);
}

// the kernels of one path: `rounds` x (walk what is new, link + chain), emit
static int launch_cw_path(gh_handle *h, uint8_t *d_path, double *d_lmsel, int rounds, int check_masks, bool resume = false)
{
    const cw_geom g = cw_geometry(h->N, h->L);
    cw_params P = cw_make_params(h, d_path, d_lmsel);
    if (h->cw_round_cap > 0 && rounds > h->cw_round_cap) rounds = h->cw_round_cap;
    const bool skip0 = !resume && !(getenv("GH_CW_SKIP0") && atoi(getenv("GH_CW_SKIP0")) == 0);      // (GH_CW_SKIP0=0: A/B, tests)
    if (!cw_digit_mode(h) && (h->L < CW_MIN_L || h->L > CW_MAX_L)) return fail(GH_ERR_STATE, "candidate-pool walk needs %d <= L <= %d", CW_MIN_L, CW_MAX_LG);
    prof_begin(h, GH_K_WALK);
    for (int r = 0; r < rounds; r++) {
        P.round = resume ? r + 1 : r;                       // (a resumed path continues behind the rounds already run)
        P.check_masks = (r == 0 && !resume) ? check_masks : 0;
        P.last_round = r == rounds - 1;
        {
            // every k_cwalk launch appends to the request lists the launch before consumed, and the other way round
            h->cw_pp++;
            const cw_params Q = cw_make_params(h, d_path, d_lmsel);
            P.pend = Q.pend; P.npend = Q.npend; P.pend_exit = Q.pend_exit; P.pend_ready = Q.pend_ready; P.phist = Q.phist;
            P.pend_c = Q.pend_c; P.npend_c = Q.npend_c; P.pend_exit_c = Q.pend_exit_c; P.pend_ready_c = Q.pend_ready_c; P.phist_c = Q.phist_c;
            P.pend_d = Q.pend_d; P.pend_d_c = Q.pend_d_c; P.pend_exit_d = Q.pend_exit_d; P.pend_exit_d_c = Q.pend_exit_d_c;
        }
        if (r == 0) prof_begin(h, GH_K_SEG);                // (bench.py: the pool walker alone, first round of a path)
        if (cw_digit_mode(h) && cw2_ok(h) && h->cw_wide) {
            switch (cw2_lc(h->L, 5)) {
                case 24: launch_cwalk2_lc<24, 5>(P, h->stream, g.S, h->dev); break;
                case 28: launch_cwalk2_lc<28, 5>(P, h->stream, g.S, h->dev); break;
                case 32: launch_cwalk2_lc<32, 5>(P, h->stream, g.S, h->dev); break;
                case 36: launch_cwalk2_lc<36, 5>(P, h->stream, g.S, h->dev); break;
                default: launch_cwalk2_lc<40, 5>(P, h->stream, g.S, h->dev); break;
            }
        } else if (cw_digit_mode(h) && cw2_ok(h)) {
            switch (cw2_lc(h->L)) {
                case 36: launch_cwalk2_lc<36>(P, h->stream, g.S, h->dev); break;
                case 40: launch_cwalk2_lc<40>(P, h->stream, g.S, h->dev); break;
                case 44: launch_cwalk2_lc<44>(P, h->stream, g.S, h->dev); break;
                case 48: launch_cwalk2_lc<48>(P, h->stream, g.S, h->dev); break;
                case 52: launch_cwalk2_lc<52>(P, h->stream, g.S, h->dev); break;
                case 56: launch_cwalk2_lc<56>(P, h->stream, g.S, h->dev); break;
                case 60: launch_cwalk2_lc<60>(P, h->stream, g.S, h->dev); break;
                default: launch_cwalk2_lc<64>(P, h->stream, g.S, h->dev); break;
            }
        } else if (cw_digit_mode(h)) {
            static std::atomic<size_t> set_g[2][64];
            const int R = h->cw_wide ? 5 : 4;
            const size_t lds_g = cwg_lds_bytes(h->L, R);
            if (lds_g > set_g[R - 4][h->dev & 63]) {
                hipFuncSetAttribute(R == 4 ? (const void *)k_cwalkg<4> : (const void *)k_cwalkg<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g);
                set_g[R - 4][h->dev & 63] = lds_g;
            }
            if (!h->cw_wide) hipLaunchKernelGGL((k_cwalkg<4>), dim3(g.S), dim3(CW_K * cw_lanes(4)), lds_g, h->stream, P);
            else hipLaunchKernelGGL((k_cwalkg<5>), dim3(g.S), dim3(CW_K * cw_lanes(5)), lds_g, h->stream, P);
        } else
        switch (h->L) {
            // (the table over the symbols: 3 bits per pick, CW_MAX_L5 lags in a state)
#define CW_CASE(n) case n: if (!h->cw_wide) launch_cwalk_lc<n, 4>(P, h->stream, g.S, h->dev); else launch_cwalk_lc<(n <= CW_MAX_L5 ? n : CW_MAX_L5), 5>(P, h->stream, g.S, h->dev); break;
            CW_CASE(6) CW_CASE(7) CW_CASE(8) CW_CASE(9) CW_CASE(10) CW_CASE(11) CW_CASE(12) CW_CASE(13) CW_CASE(14) CW_CASE(15) CW_CASE(16)
            CW_CASE(17) CW_CASE(18) CW_CASE(19) CW_CASE(20) CW_CASE(21) CW_CASE(22) CW_CASE(23) CW_CASE(24)
            // (25..32 lags: over ranks only -- cw_digit_mode sends the symbol table beyond CW_MAX_L5 to k_cwalkg)
            CW_CASE(25) CW_CASE(26) CW_CASE(27) CW_CASE(28) CW_CASE(29) CW_CASE(30) CW_CASE(31) CW_CASE(32)
#undef CW_CASE
            default: return fail(GH_ERR_STATE, "candidate-pool walk needs %d <= L <= %d", CW_MIN_L, CW_MAX_LG);
        }
        if (r == 0) prof_end(h, GH_K_SEG, (double)h->N * (double)h->L * CELL * esize(h));
        // (skip0: hardly any chain closes in the round that walks everything -- the states a reweight moved are only found by
        // it -- so its link + scan launches, 15 us of a path, are left out and the chain is first followed behind round 1)
        if (r == 0 && skip0 && rounds >= 2) continue;
        hipLaunchKernelGGL(k_clink, dim3(g.S), dim3(CW_K), 0, h->stream, P);
        hipLaunchKernelGGL(k_cscan, dim3(1), dim3(1024), 0, h->stream, P);
    }
    hipLaunchKernelGGL(k_cemit, dim3(g.S), dim3(256), 0, h->stream, P);
    prof_end(h, GH_K_WALK, (double)h->N * ((1.0 + (double)h->L) * CELL * esize(h) + 28.0));
    return post_launch(h, "k_cwalk/k_cscan/k_cemit");
}

// one path through the serial walker, its boundary states into the pools (merge: keep what is there)
static int cw_serial_path(gh_handle *h, uint8_t *d_path, gh_path_rec *d_rec, double *d_lmsel, double min_remove, int slot, int merge, bool reweight = true)
{
    // the serial depth-2 walker reads tables that k_lt keeps; behind pool paths (no k_lt between them) they are stale
    h->lt_inc_path = nullptr;
    h->dirty_lt = true;
    int rc = ensure_lt(h, true);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(d_lmsel, 0, sizeof(double), h->stream));      // k_hp: the walker sums this path itself
    if ((rc = launch_walk(h, d_path, d_rec, min_remove, 1, nullptr, 0))) return rc;
    cw_params P = cw_make_params(h, d_path, d_lmsel);
    const cw_geom g = cw_geometry(h->N, h->L);
    hipLaunchKernelGGL(k_cseed, dim3((g.S + 255) / 256), dim3(256), 0, h->stream, P, (const uint8_t *)d_path, merge);
    if ((rc = post_launch(h, "k_cseed"))) return rc;
    if (reweight) rc = launch_reweight_marg(h, d_path, 0.0, 1, d_rec, slot, false, false, 0);
    h->cw_stat[1]++;
    return rc;
}

// device buffers for the paths, records and selected log-marginals of a spin of max_paths paths
static int ensure_spin_buffers(gh_handle *h, int max_paths)
{
    const size_t n1 = (size_t)h->N + 1;
    if (max_paths > h->spin_cap) {
        // (a handle is sized for the reference's default spin -- 100 paths, gretel/cmd.py:148 -- the first time it spins, however
        // few paths that first call asks for: growing later costs three frees, three allocations and the pinned staging buffer
        // over again, 0.7 ms on the caller's clock -- a fifth of a C3 spin -- unless the window is so long that the floor is real memory)
        const int floor_paths = 128;
        if (max_paths < floor_paths && n1 * 9 * (size_t)floor_paths <= ((size_t)256 << 20)) max_paths = floor_paths;
        HIPCHK(hipStreamSynchronize(h->stream));
        hipFree(h->spin_paths); hipFree(h->spin_recs);
        h->spin_paths = nullptr; h->spin_recs = nullptr; h->spin_cap = 0;
        hipFree(h->spin_lmsel); h->spin_lmsel = nullptr;
        hipError_t ea = hipMalloc((void **)&h->spin_paths, n1 * max_paths);
        if (ea == hipSuccess) ea = hipMalloc((void **)&h->spin_recs, sizeof(gh_path_rec) * max_paths);
        // the selected symbols' log-marginals of every path, for the likelihood sums behind the loop (k_hp)
        if (ea == hipSuccess) ea = hipMalloc((void **)&h->spin_lmsel, sizeof(double) * n1 * max_paths);
        if (ea != hipSuccess) {
            hipFree(h->spin_paths); h->spin_paths = nullptr;
            hipFree(h->spin_recs); h->spin_recs = nullptr;
            return fail(GH_ERR_NOMEM, "hipMalloc failed");
        }
        h->spin_cap = max_paths;
    }
    return GH_OK;
}

struct spin_io {
    int max_paths;
    double min_remove;
    size_t n1;              // bytes per path
    int nb;                 // stride of the per-path partial sums
    uint8_t *d_paths;
    gh_path_rec *d_recs;
    uint8_t *paths_out;
    gh_path_rec *recs;
    bool no_reweight;       // gh_generate_path: the path is only recovered (its record closed by k_seg_fin), the tensor stays
};
static int spin_candidate_pools(gh_handle *h, const spin_io &io, dev_state &hs, int *first_out, bool *gave_up);

extern "C" int gh_generate_path(gh_t *h, const gh_t *original, uint8_t *path_out, double *hp_current,
                                double *hp_original, double *min_marginal, int *hole_at)
{
    if (!h || !path_out || !hole_at) return fail(GH_ERR_ARG, "null argument");
    if (set_dev(h)) return GH_ERR_HIP;
    const bool pools = cw_ok(h->wmode, h->L) && !h->cw_off;
    int rc = ensure_lt(h, !seg_ok(h->wmode, h->L) && !pools);      // (a serial walker wants the marginal term in the table)
    if (rc) return rc;
    const int nb8 = ((h->N + 1) * 8 + 255) / 256;
    if (original && original != h) {
        gh_handle *o = const_cast<gh_handle *>(original);
        if (o->N != h->N) return fail(GH_ERR_ARG, "original has %d SNPs, hansel has %d", o->N, h->N);
        if (o->dev != h->dev) return fail(GH_ERR_ARG, "original lives on another device");
        if ((rc = ensure_marg(o))) return rc;
        HIPCHK(hipStreamSynchronize(o->stream));
        hipLaunchKernelGGL(k_snapshot, dim3(nb8), dim3(256), 0, h->stream, h->minfo, o->minfo, h->N, (const win_desc *)nullptr);
        h->have_orig = true;
    } else if (!h->have_orig) {
        hipLaunchKernelGGL(k_snapshot, dim3(nb8), dim3(256), 0, h->stream, h->minfo, h->minfo, h->N, (const win_desc *)nullptr);
    }
    if ((rc = reset_spin_state(h))) return rc;
    if (cw_ok(h->wmode, h->L) && !h->cw_off) {
        // lag counts of the candidate pools: the path comes out of them (one path, no reweight: the tensor stays as it is,
        // the pools keep what they learnt for the next call); only a window they cannot take goes on to the serial walker
        if ((rc = ensure_spin_buffers(h, 1))) return rc;
        gh_path_rec rec1;
        spin_io io;
        io.max_paths = 1; io.min_remove = 0.0; io.n1 = (size_t)h->N + 1; io.nb = 0;
        io.d_paths = h->spin_paths; io.d_recs = h->spin_recs; io.paths_out = path_out; io.recs = &rec1; io.no_reweight = true;
        dev_state hs1;
        memset(&hs1, 0, sizeof hs1);
        int first1 = 0;
        bool gave_up = false;
        h->cw_no_rw = true;
        rc = spin_candidate_pools(h, io, hs1, &first1, &gave_up);
        h->cw_no_rw = false;
        if (rc) return rc;
        if (!gave_up) {
            *hole_at = hs1.stop ? hs1.hole_at : 0;
            if (hs1.stop) {
                // (the prefix that was walked: the reference returns what it has)
                HIPCHK(hipMemcpy(path_out, h->spin_paths, (size_t)h->N + 1, hipMemcpyDeviceToHost));
            } else {
                if (hp_current) *hp_current = rec1.hp_current;
                if (hp_original) *hp_original = rec1.hp_original;
                if (min_marginal) *min_marginal = rec1.min_marginal;
            }
            return GH_OK;
        }
        if ((rc = reset_spin_state(h))) return rc;
        if ((rc = ensure_lt(h, true))) return rc;                   // the pools gave the window up: the serial walker's table
    }
    if ((rc = launch_walk(h, h->d_path, h->d_rec, 0.0, 0))) return rc;
    if (seg_ok(h->wmode, h->L)) {
        // close the record (hole / minimum marginal), then the two likelihood sums
        seg_params P;
        memset(&P, 0, sizeof P);
        P.N = h->N; P.L = h->L; P.st = h->dstate; P.segmin = h->seg_min;
        hipLaunchKernelGGL(k_seg_fin, dim3(1), dim3(256), 0, h->stream, P, h->d_rec, 0.0, 0);
        hipLaunchKernelGGL(k_hp, dim3(1, 2), dim3(64), 0, h->stream, (const double *)h->lmsel1, (size_t)0, (const uint8_t *)h->d_path,
                           (size_t)0, (const double *)h->minfo, h->N, (const dev_state *)h->dstate, h->d_rec, h->sm);
        if ((rc = post_launch(h, "k_seg_fin/k_hp"))) return rc;
    }
    dev_state hs;
    gh_path_rec rec;
    HIPCHK(hipMemcpyAsync(&hs, h->dstate, sizeof hs, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(&rec, h->d_rec, sizeof rec, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(path_out, h->d_path, (size_t)h->N + 1, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *hole_at = hs.stop ? hs.hole_at : 0;
    if (!hs.stop) {
        if (hp_current) *hp_current = rec.hp_current;
        if (hp_original) *hp_original = rec.hp_original;
        if (min_marginal) *min_marginal = rec.min_marginal;
    }
    return GH_OK;
}

extern "C" int gh_reweight_path(gh_t *h, const uint8_t *path, double ratio, double *removed)
{
    if (!h || !path) return fail(GH_ERR_ARG, "null argument");
    if (set_dev(h)) return GH_ERR_HIP;
    for (int q = 0; q <= h->N; q++)
        if (path[q] >= NSYM) return fail(GH_ERR_SYMBOL, "path[%d] = %d is not a symbol index", q, path[q]);
    HIPCHK(hipMemcpyAsync(h->d_rw_path, path, (size_t)h->N + 1, hipMemcpyHostToDevice, h->stream));
    int rc = launch_reweight_marg(h, h->d_rw_path, ratio, 0, h->d_rec);
    if (rc) return rc;
    gh_path_rec rec;
    HIPCHK(hipMemcpyAsync(&rec, h->d_rec, sizeof rec, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (removed) *removed = rec.magnitude;
    return GH_OK;
}

// results to the caller's (pageable) buffers through pinned staging: one asynchronous copy each at PCIe rate and a
// host memcpy, instead of the runtime's chunked pageable path
static int results_to_host(gh_handle *h, uint8_t *paths_out, const uint8_t *d_paths, size_t path_bytes, gh_path_rec *recs,
                           const gh_path_rec *d_recs, size_t rec_bytes)
{
    const size_t need = path_bytes + rec_bytes;
    if (need > h->stage_cap) {
        if (h->stage) hipHostFree(h->stage);
        h->stage = nullptr; h->stage_cap = 0;
        const size_t cap = need + need / 2 + 4096;
        if (hipHostMalloc((void **)&h->stage, cap, hipHostMallocDefault) != hipSuccess) {
            // no pinned memory to be had: the plain copies
            (void)hipGetLastError();
            h->stage = nullptr;
            HIPCHK(hipMemcpy(paths_out, d_paths, path_bytes, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(recs, d_recs, rec_bytes, hipMemcpyDeviceToHost));
            return GH_OK;
        }
        h->stage_cap = cap;
    }
    HIPCHK(hipMemcpyAsync(h->stage, d_paths, path_bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->stage + path_bytes, d_recs, rec_bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(paths_out, h->stage, path_bytes);
    memcpy(recs, h->stage + path_bytes, rec_bytes);
    return GH_OK;
}

// The end of a spin in ONE wait: the device state and the results of the `launched` paths go to pinned staging together
// (how many of them are complete is only known from the state); stage_deliver then hands n_done of them to the caller.
// Returns 1 when there is no pinned memory to be had (the caller takes the two-step way).
static int state_and_results_to_stage(gh_handle *h, dev_state *hs, const uint8_t *d_paths, size_t n1, const gh_path_rec *d_recs, int launched)
{
    const size_t pb = n1 * (size_t)launched, rb = sizeof(gh_path_rec) * (size_t)launched, need = pb + rb + sizeof(dev_state) + 64;
    if (need > h->stage_cap) {
        if (h->stage) hipHostFree(h->stage);
        h->stage = nullptr; h->stage_cap = 0;
        // (pinned memory is the dearest allocation of a spin: sized once for what the device buffers hold -- ensure_spin_buffers)
        const size_t lc = (size_t)(launched > h->spin_cap ? launched : h->spin_cap);
        const size_t cap = (n1 + sizeof(gh_path_rec)) * lc + sizeof(dev_state) + 64 + 4096;
        if (hipHostMalloc((void **)&h->stage, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h->stage = nullptr; return 1; }
        h->stage_cap = cap;
    }
    const size_t rec_off = (pb + 63) & ~(size_t)63;
    HIPCHK(hipMemcpyAsync(h->stage + rec_off + rb, h->dstate, sizeof(dev_state), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->stage + rec_off, d_recs, rb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->stage, d_paths, pb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(hs, h->stage + rec_off + rb, sizeof(dev_state));
    return GH_OK;
}

static void stage_deliver(gh_handle *h, uint8_t *paths_out, size_t n1, gh_path_rec *recs, int launched, int n_done)
{
    const size_t rec_off = (n1 * (size_t)launched + 63) & ~(size_t)63;
    memcpy(paths_out, h->stage, n1 * (size_t)n_done);
    memcpy(recs, h->stage + rec_off, sizeof(gh_path_rec) * (size_t)n_done);
}

// The spin of lag counts 6 .. 24: segments walked from candidate pools (cwalk.hpp).  The paths are queued a few at a
// time: a path whose chain stays open after the queued rounds idles the kernels behind it; the host then queues more
// rounds for that one path, and if its chain is still open hands it to the serial walker (whose states join the pools)
// before queueing on.  Returns with *gave_up set when the window turned out to have a position with five candidates:
// paths *first .. go the serial walkers' way (the caller's loop).

static int spin_candidate_pools(gh_handle *h, const spin_io &io, dev_state &hs, int *first_out, bool *gave_up)
{
    const int max_paths = io.max_paths;
    const double min_remove = io.min_remove;
    const size_t n1 = io.n1;
    const int nb = io.nb;
    uint8_t *d_paths = io.d_paths, *paths_out = io.paths_out;
    gh_path_rec *d_recs = io.d_recs, *recs = io.recs;
    hipError_t e = hipSuccess;
    int rc = GH_OK, first = 0;
    bool cw_gave_up = false;
    rc = alloc_cw(h);
    const cw_geom cg = cw_geometry(h->N, h->L);
    // what follows the kernels of a path: the fused reweight (which also closes the path's record), or only the closing
    auto finish_path = [&](uint8_t *pth, gh_path_rec *rec, int slot, bool chained) -> int {
        if (!io.no_reweight) return launch_reweight_marg(h, pth, min_remove, 1, rec, slot, true, chained, cg.S);
        seg_params SP;
        memset(&SP, 0, sizeof SP);
        SP.N = h->N; SP.L = h->L; SP.st = h->dstate; SP.segmin = h->seg_min;
        hipLaunchKernelGGL(k_seg_fin, dim3(1), dim3(256), 0, h->stream, SP, rec, min_remove, cg.S);
        return post_launch(h, "k_seg_fin");
    };
    const int zero2[2] = {0, 0};
    if (rc == GH_OK) {
        e = hipMemcpyAsync(&h->dstate->lt_stale, zero2, sizeof zero2, hipMemcpyHostToDevice, h->stream);      // lt_stale, cw_unres
        if (e != hipSuccess) rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e));
    }
    // which layout did k_lt choose for this tensor?  (ranks when every position has at most four candidates, else symbols:
    // the walker's instantiation, the bits per pick and what a pool entry means follow from it)
    auto look_at_layout = [&](int ranked) {
        h->cw_wide = !ranked;
        if (h->cw_pool_wide != h->cw_wide) { h->cw_ready = false; h->cw_pool_wide = h->cw_wide; }

        if (cw_digit_mode(h) && !h->cw_off) {
            const int LD = (h->L + 3) & ~3;
            if (h->cw_LD != LD) {
                hipStreamSynchronize(h->stream);
                hipFree(h->cw_keys_d); hipFree(h->cw_exits_d); hipFree(h->cw_pend_d); hipFree(h->cw_pend_exit_d);
                h->cw_keys_d = h->cw_exits_d = h->cw_pend_d = h->cw_pend_exit_d = nullptr;
                const cw_geom gg = cw_geometry(h->N, h->L);
                const size_t bytes = (size_t)gg.S * CW_K * LD;
                if (hipMalloc((void **)&h->cw_keys_d, bytes) != hipSuccess || hipMalloc((void **)&h->cw_exits_d, bytes) != hipSuccess ||
                    hipMalloc((void **)&h->cw_pend_d, 2 * bytes) != hipSuccess || hipMalloc((void **)&h->cw_pend_exit_d, 2 * bytes) != hipSuccess) {
                    rc = fail(GH_ERR_NOMEM, "hipMalloc for the candidate pools failed");
                    h->cw_LD = 0;
                    return;
                }
                hipMemsetAsync(h->cw_keys_d, 0, bytes, h->stream); hipMemsetAsync(h->cw_exits_d, 0, bytes, h->stream); hipMemsetAsync(h->cw_pend_d, 0, 2 * bytes, h->stream); hipMemsetAsync(h->cw_pend_exit_d, 0, 2 * bytes, h->stream);
                h->cw_LD = LD;
                h->cw_ready = false;
            }
        }
    };
    if (rc == GH_OK) rc = ensure_lt(h);
    if (rc == GH_OK) {
        dev_state look;
        e = hipMemcpyAsync(&look, h->dstate, sizeof look, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e));
        else look_at_layout(look.ranked);
    }
    int done = 0;
    int CHUNK = 8, clean = 0;        // paths queued between two looks at the device state: grows while every chain closes
    while (rc == GH_OK && done < max_paths) {
        if (h->cw_off) {
            // a position with five candidates: the rest of the spin goes the serial walkers' way (the loop below)
            cw_gave_up = true;
            first = done;
            if (done < max_paths) {
                e = hipMemset2DAsync(h->spin_lmsel + n1 * done, n1 * sizeof(double), 0, sizeof(double), (size_t)(max_paths - done), h->stream);
                if (e != hipSuccess) rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e));
            }
            break;
        }
        if (!h->cw_ready) {
            // no pools yet for this tensor: one entry per pool from the largest marginals (k_cguess), and rounds
            // until closure has brought in what the walk really does (the serial walker costs 20 rounds at L <= 16
            // and 400 beyond: it only takes the path if the chain is still open after CW_BOOT + the resumed rounds)
            h->cw_stamp++;
            if ((rc = ensure_lt(h))) break;
            uint8_t *pth = d_paths + n1 * done;
            cw_params P = cw_make_params(h, pth, h->spin_lmsel + n1 * done);
            hipLaunchKernelGGL(k_cguess, dim3((unsigned)((h->N + 256) / 256)), dim3(256), 0, h->stream, P, pth);
            hipLaunchKernelGGL(k_cseed, dim3((cg.S + 255) / 256), dim3(256), 0, h->stream, P, (const uint8_t *)pth, 0);
            if ((rc = post_launch(h, "k_cguess/k_cseed"))) break;
            if ((rc = launch_cw_path(h, pth, h->spin_lmsel + n1 * done, CW_BOOT_ROUNDS, 0))) break;
            if ((rc = finish_path(pth, d_recs + done, done, false))) break;
            h->cw_stat[2] += CW_BOOT_ROUNDS;
            h->cw_ready = true;
        } else {
            const int upto = done + CHUNK < max_paths ? done + CHUNK : max_paths;
            // (conditional C / the marginal term: the reweight cannot keep the table current, k_lt rebuilds it in front of
            // every path -- queued like everything else; otherwise once per look, and k_cwalk checks the masks)
            if ((rc = ensure_lt(h, false, false))) break;
            const bool inc = rw_incremental_ok(h);
            for (int s = done; s < upto && rc == GH_OK; s++) {
                h->cw_stamp++;
                if (!inc && s > done && (rc = ensure_lt(h))) break;
                if ((rc = launch_cw_path(h, d_paths + n1 * s, h->spin_lmsel + n1 * s, h->cw_rounds, (inc && s > done) ? 1 : 0))) break;
                rc = finish_path(d_paths + n1 * s, d_recs + s, s, s > done);
                h->cw_stat[2] += h->cw_rounds;
            }
            if (rc) break;
        }
        e = hipMemcpyAsync(&hs, h->dstate, sizeof hs, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e)); break; }
        if (getenv("GH_PRINT_STATE"))
            fprintf(stderr, "gh_spin(cw): n_done %d stop %d lt_stale %d cw_unres %d open_at %d rounds %d  closed in round 0/1/2/3/4+: %llu %llu %llu %llu %llu\n", hs.n_done, hs.stop, hs.lt_stale,
                    hs.cw_unres, hs.cw_open_at, h->cw_rounds, hs.dbg8[0], hs.dbg8[1], hs.dbg8[2], hs.dbg8[3], hs.dbg8[4]);
        h->cw_stat[0] += hs.n_done - done;
        done = hs.n_done;
        if (hs.stop) break;
        if (!hs.lt_stale && !hs.cw_unres) {
            // (a look costs ~25 us of idle GPU; an open chain costs the ~37 us of idle launches of every path queued
            // behind it: at one open chain in 50..100 paths the optimum is 8..16 paths per look)
            if (CHUNK < 16) CHUNK *= 2;
            // idle rounds cost three launches each: one fewer when, three looks in a row, no path needed as many as are queued
            if (hs.cw_need > 0 && hs.cw_need < h->cw_rounds) { if (++clean >= 3) { h->cw_rounds--; clean = 0; } }
            else clean = 0;
            if (hs.cw_need > 0) {
                const int zero = 0;
                e = hipMemcpyAsync(&h->dstate->cw_need, &zero, sizeof zero, hipMemcpyHostToDevice, h->stream);
                if (e != hipSuccess) { rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e)); break; }
            }
        } else { CHUNK = 8; clean = 0; }
        if (hs.lt_stale || hs.cw_unres) {
            e = hipMemcpyAsync(&h->dstate->lt_stale, zero2, sizeof zero2, hipMemcpyHostToDevice, h->stream);
            if (e != hipSuccess) { rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e)); break; }
            h->spin_requeues++;
            h->cw_stat[3]++;
            // a moved candidate mask or a window for the serial walkers: the next ensure_lt rebuilds the table in
            // full.  Behind a chain that merely stayed open the table is current: every kernel queued behind it
            // idled, and the reweights that ran kept their rows.
            h->lt_inc_path = nullptr;
            h->dirty_lt = hs.lt_stale != 0;
            // an open chain costs a look, the resumed rounds and the idle launches of everything queued behind it -- far
            // more than an idle round (three empty launches) on every path: queue one round more from now on
            if (hs.cw_unres == 1 && h->cw_rounds < 8) h->cw_rounds++;
            if (hs.cw_unres == 2) look_at_layout(hs.ranked);        // a rebuild changed the table's layout: the other instantiation, new pools
            else if (hs.cw_unres == 1 && done < max_paths) {
                // the queued rounds did not close this path's chain.  Its pools keep what has been walked (the tensor
                // has not changed): more rounds first; if the chain is still open, the serial walker takes the path
                // and its states join the pools.
                bool closed = false;
                if (!hs.lt_stale) {
                    if ((rc = ensure_lt(h))) break;
                    // Every round carries the verified chain at least one segment further (the state it stopped at joins
                    // the next pool and is walked), so S rounds always close it.  Beyond 16 lags the serial walker is
                    // k_walk_global, 3 ms per thousand SNPs: there the rounds go on until the chain is closed; up to 16
                    // lags the depth-2 walker costs about as much as a dozen rounds, and takes over after those.
                    const int more = h->L > 16 ? 48 : 12;
                    const int tries = h->L > 16 ? (cg.S + 16) / more + 1 : 1;
                    for (int a = 0; a < tries && !closed; a++) {
                        if ((rc = launch_cw_path(h, d_paths + n1 * done, h->spin_lmsel + n1 * done, more, 0, true))) break;
                        if ((rc = finish_path(d_paths + n1 * done, d_recs + done, done, false))) break;
                        h->cw_stat[2] += more;
                        e = hipMemcpyAsync(&hs, h->dstate, sizeof hs, hipMemcpyDeviceToHost, h->stream);
                        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
                        if (e != hipSuccess) { rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e)); break; }
                        closed = !hs.cw_unres && hs.n_done > done;
                        if (hs.cw_unres) {
                            e = hipMemcpyAsync(&h->dstate->lt_stale, zero2, sizeof zero2, hipMemcpyHostToDevice, h->stream);
                            if (e != hipSuccess) { rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e)); break; }
                        }
                        if (hs.stop) break;
                    }
                    if (rc != GH_OK) break;
                }
                if (!closed) {
                    h->cw_stamp++;
                    if ((rc = cw_serial_path(h, d_paths + n1 * done, d_recs + done, h->spin_lmsel + n1 * done, min_remove, done, 1, !io.no_reweight))) break;
                    e = hipMemcpyAsync(&hs, h->dstate, sizeof hs, hipMemcpyDeviceToHost, h->stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
                    if (e != hipSuccess) { rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e)); break; }
                }
                done = hs.n_done;
                if (hs.stop) break;
            }
        }
    }
    // the serial walkers' tables were not kept between pool paths: rebuild before anybody walks serially again
    // (a lone path without a reweight leaves the tensor, and the table, as they were)
    if (!io.no_reweight) {
        h->lt_inc_path = nullptr;
        h->dirty_lt = true;
    }
    if (rc == GH_OK && done > 0 && !cw_gave_up) {
        hipLaunchKernelGGL(k_hp, dim3(done, 2), dim3(64), 0, h->stream, (const double *)h->spin_lmsel, n1, (const uint8_t *)d_paths, n1,
                           (const double *)h->minfo, h->N, (const dev_state *)h->dstate, d_recs, h->sm);
        if (!io.no_reweight) hipLaunchKernelGGL(k_reweight_finish_all, dim3(done), dim3(256), 0, h->stream, h->partial, nb, h->dstate, d_recs);
        rc = post_launch(h, "k_hp/k_reweight_finish_all");
        if (rc == GH_OK) {
            e = hipMemcpyAsync(&hs, h->dstate, sizeof hs, hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e));
            else if (hs.n_done > 0) rc = results_to_host(h, paths_out, d_paths, n1 * hs.n_done, recs, d_recs, sizeof(gh_path_rec) * hs.n_done);
        }
    }
    *first_out = first;
    *gave_up = cw_gave_up;
    return rc;
}

extern "C" int gh_spin(gh_t *h, int max_paths, double min_remove, uint8_t *paths_out, gh_path_rec *recs,
                       int *n_out, int *hole_at)
{
    if (!h || !paths_out || !recs || !n_out || !hole_at) return fail(GH_ERR_ARG, "null argument");
    if (max_paths < 0) return fail(GH_ERR_ARG, "max_paths < 0");
    if (set_dev(h)) return GH_ERR_HIP;
    *n_out = 0; *hole_at = 0;
    if (max_paths == 0) return GH_OK;
    int rc;
    if (!h->have_orig && (rc = gh_snapshot_original(h))) return rc;
    const size_t n1 = (size_t)h->N + 1;
    if ((rc = ensure_spin_buffers(h, max_paths))) return rc;
    uint8_t *d_paths = h->spin_paths;
    gh_path_rec *d_recs = h->spin_recs;
    hipError_t e = hipSuccess;
    rc = GH_OK;
    bool seg = seg_ok(h->wmode, h->L);
    h->seg6 = false;
    if (rc == GH_OK && !seg && h->wmode == WM_SEG && h->L == SEG_MAX_L_NARROW && !(getenv("GH_SEG6") && atoi(getenv("GH_SEG6")) == 0)) {
        // 4^6 states can still be enumerated -- 5^6 cannot: look at the layout k_lt chose for this tensor (a window that is
        // narrow stays narrow: candidates only ever disappear)
        rc = ensure_lt(h);
        dev_state look;
        if (rc == GH_OK) {
            e = hipMemcpyAsync(&look, h->dstate, sizeof look, hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e));
            else if (look.ranked) { seg = true; h->seg6 = true; }
        }
    }
    // Spins over the enumerated states run without k_emit where the group maps fit k_rw's LDS beside everything else
    // (1024 states: 32 KB; the five-symbol radix from 5^5 states on, and 4^6, keep k_emit) and the band is not wider than
    // the halo k_rw resolves: which radix applies is the table's layout, decided on the device -- one look per spin.
    h->fuse = false;
    // MEASURED (C3, MI355X): three launches are bit-identical but NOT faster than four -- k_emit and its
    // boundary (6.8 us) go, the prologue k_rw needs instead (group maps to LDS, the chain, one more round trip) costs 6.5 us,
    // and carrying the minima through k_seg / k_scan another 2.9 us: 40.8 against 38.7 us per path.  So it is opt-in:
    // GH_FUSE=1 (large windows) or 2 (every window the maps fit; the tests).
    if (rc == GH_OK && seg && h->need_rinfo && h->W <= RW_FUSE_HALO && rw_lanes(h) == 8 && getenv("GH_FUSE") && atoi(getenv("GH_FUSE")) >= 1) {
        rc = ensure_lt(h);
        dev_state look;
        if (rc == GH_OK) {
            e = hipMemcpyAsync(&look, h->dstate, sizeof look, hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e));
            else {
                const int R = look.ranked ? 4 : 5;
                const int ppb = 256 / rw_lanes(h);
                // (windows whose segment maps all fit one workgroup's LDS keep k_emit_small: three launches as well, and a lighter k_rw)
                const bool small = emit_small_lds_bytes(h->N, h->L, R) <= 64 * 1024 && !(getenv("GH_EMIT_SMALL") && atoi(getenv("GH_EMIT_SMALL")) == 0);
                const bool force = getenv("GH_FUSE") && atoi(getenv("GH_FUSE")) == 2;      // (tests: fused whatever the size)
                if ((!small || force) && seg_radix_ok(R, h->L) && rw_fuse_lds_bytes(h->N, h->L, R, ppb, h->W) <= 64 * 1024 && (rc = alloc_seg(h)) == GH_OK) {
                    h->fuse = true;
                    h->fuse_lds = rw_fuse_lds_bytes(h->N, h->L, R, ppb, h->W);
                    // (this spin walks the enumerated states: whatever k_classify found before the flow was chosen does not apply)
                    if (hipMemsetAsync(&h->dstate->maxstates, 0, sizeof(int), h->stream) != hipSuccess) rc = fail(GH_ERR_HIP, "gh_spin failed: hipMemsetAsync");
                }
            }
        }
    }
    // k_rwseg (segwalk.hpp): the reweight of a path rides in the k_seg launch of the next one.  Lane groups of 8 (bands up to 8),
    // at most 128 positions per workgroup with the halo; behind it k_scan + k_emit, or k_emit_small alone in small windows.
    h->rws = false;
    int rws_S = 0;
    if (rc == GH_OK && seg && !h->fuse && !h->lt_full && rw_lanes(h) == 8 &&
        !(getenv("GH_RWSEG") && atoi(getenv("GH_RWSEG")) == 0)) {
        const bool five = seg_radix_ok(5, h->L);
        const seg_geom g4 = seg_geometry(h->N, h->L, 4), g5 = five ? seg_geometry(h->N, h->L, 5) : g4;
        const bool small = (five ? max2(emit_small_lds_bytes(h->N, h->L, 4), emit_small_lds_bytes(h->N, h->L, 5)) : emit_small_lds_bytes(h->N, h->L, 4)) <= 64 * 1024 &&
                           !(getenv("GH_EMIT_SMALL") && atoi(getenv("GH_EMIT_SMALL")) == 0);
        const int longest = g4.seglen > g5.seglen ? g4.seglen : g5.seglen;
        // (... and the kernel's LDS must fit: the column conditionals stage the band blocks of all those positions -- long
        // segments of a wide band in binary64 do not)
        // (small windows -- k_emit_small behind k_rwseg, two launches per path -- unless GH_RWSEG_SMALL=0)
        const bool small_ok = !(getenv("GH_RWSEG_SMALL") && atoi(getenv("GH_RWSEG_SMALL")) == 0);
        if ((!small || small_ok) && longest + h->L + 1 <= SEG_THREADS / 8 && rws_patch_off(h, h->L) + sizeof(seg_patch) <= RWS_LDS_MAX &&
            (rc = alloc_seg(h)) == GH_OK) {
            rws_S = g4.S > g5.S ? g4.S : g5.S;
            const size_t need = (size_t)rws_S * h->L * NSYM * h->W * NSYM * esize(h);
            if (need > h->seg_halo_bytes) {
                // (no early return from here on: the cleanup at the end of this function resets fuse / rws / the stride)
                if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e));
                else {
                    hipFree(h->seg_halo); h->seg_halo = nullptr; h->seg_halo_bytes = 0;
                    if (hipMalloc(&h->seg_halo, need) != hipSuccess) rc = fail(GH_ERR_NOMEM, "hipMalloc failed");
                    else h->seg_halo_bytes = need;
                }
            }
            h->rws = rc == GH_OK;
        }
    }
    int nb = rw_blocks(h, seg || cw_ok(h->wmode, h->L));       // (the widest reweight kernel this spin may launch)
    if (h->rws && rws_S > nb) nb = rws_S;                       // ... k_rwseg leaves one partial sum per segment
    h->spin_partial_stride = nb;
    if (rc == GH_OK) rc = ensure_partial(h, nb, max_paths > h->spin_cap ? max_paths : h->spin_cap);       // L only changes through gh_set_L / gh_fill, never inside a spin
    // (k_rwseg: every slot is summed over the whole stride: what a kernel with fewer workgroups leaves untouched must read 0)
    if (rc == GH_OK) rc = reset_spin_state(h, h->rws ? h->partial : nullptr, h->rws ? (size_t)nb * max_paths : 0);
    // Segment-parallel walks with a conditional table that the fused reweight keeps current (conditional A/B, no marginal
    // term): no k_lt between two paths.  Its only job there is to notice that a candidate mask moved (a count reached
    // zero; rare) and rebuild the table; instead the next k_seg sees the flag k_marg left, marks the table stale and the
    // rest of the queue does nothing.  The host then rebuilds and queues the remaining paths again.
    const bool optimistic = seg && !h->lt_full;
    dev_state hs;
    memset(&hs, 0, sizeof hs);
    int first = 0;
    h->spin_requeues = 0;
    // lag counts 6 .. 24: segments walked from candidate pools (spin_candidate_pools above)
    const bool cw = rc == GH_OK && !seg && cw_ok(h->wmode, h->L) && !h->cw_off;
    bool cw_gave_up = false;
    if (cw) {
        spin_io io;
        io.max_paths = max_paths; io.min_remove = min_remove; io.n1 = n1; io.nb = nb;
        io.d_paths = d_paths; io.d_recs = d_recs; io.paths_out = paths_out; io.recs = recs; io.no_reweight = false;
        rc = spin_candidate_pools(h, io, hs, &first, &cw_gave_up);
    }
    while ((!cw || cw_gave_up) && rc == GH_OK) {
        int launched = first;
        for (int s = first; s < max_paths && rc == GH_OK; s++) {
            if (!optimistic || s == first) { if ((rc = ensure_lt(h, !seg))) break; }
            const int cmk = !(optimistic && s > first) ? 0 : (s == h->force_stale_at && h->spin_requeues == 0 ? 2 : 1);
            if (h->rws) {
                // path s is walked by the launch that reweights path s - 1; the last path's reweight is a plain k_rw
                if (s == first) rc = launch_walk(h, d_paths + n1 * s, d_recs + s, min_remove, 1, h->spin_lmsel + n1 * s, cmk);
                else rc = launch_rwseg(h, d_paths + n1 * (s - 1), d_recs + (s - 1), s - 1, min_remove, d_paths + n1 * s, h->spin_lmsel + n1 * s, cmk);
                if (rc == GH_OK && s == max_paths - 1)
                    rc = launch_reweight_marg(h, d_paths + n1 * s, min_remove, 1, d_recs + s, s, true, s > first, 0, h->spin_lmsel + n1 * s);
                if (rc) break;
                launched = s + 1;
                continue;
            }
            if ((rc = launch_walk(h, d_paths + n1 * s, d_recs + s, min_remove, 1, h->spin_lmsel + n1 * s, cmk))) break;
            if ((rc = launch_reweight_marg(h, d_paths + n1 * s, seg ? min_remove : 0.0, 1, d_recs + s, s, seg, optimistic && s > first, 0,
                                           h->spin_lmsel + n1 * s))) break;
            launched = s + 1;
        }
        // the likelihood sums of every path in one launch (strictly sequential additions, one wavefront per sum)
        if (rc == GH_OK && launched > 0 && (seg || cw_gave_up)) {
            hipLaunchKernelGGL(k_hp, dim3(launched, 2), dim3(64), 0, h->stream, (const double *)h->spin_lmsel, n1, (const uint8_t *)d_paths, n1,
                               (const double *)h->minfo, h->N, (const dev_state *)h->dstate, d_recs, h->sm);
            rc = post_launch(h, "k_hp");
        }
        // bring the table in step with the last reweight while its path is still allocated (the next call would
        // otherwise refresh it from a freed buffer); rebuilds it in full when a candidate mask moved
        if (rc == GH_OK && launched > 0) rc = ensure_lt(h, !seg);
        // the removed mass of every path in one launch (it is only ever read by the host)
        if (rc == GH_OK && launched > 0) {
            hipLaunchKernelGGL(k_reweight_finish_all, dim3(launched), dim3(256), 0, h->stream, h->partial, nb, h->dstate, d_recs);
            rc = post_launch(h, "k_reweight_finish_all");
        }
        if (rc != GH_OK) break;
        // state and results in one wait (the results of a queue that has to be taken up again are fetched in vain: rare)
        bool staged = false;
        if (launched > 0) {
            const int sr = state_and_results_to_stage(h, &hs, d_paths, n1, d_recs, launched);
            if (sr < 0) { rc = sr; break; }
            staged = sr == GH_OK;
        }
        if (!staged) {
            e = hipMemcpyAsync(&hs, h->dstate, sizeof hs, hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        }
        if (getenv("GH_PRINT_STATE"))
            fprintf(stderr, "gh_spin: stop %d hole_at %d n_done %d first_hole %d cur_hole %d lt_stale %d cm_same %d narrow %d ranked %d\n", hs.stop, hs.hole_at,
                    hs.n_done, hs.first_hole, hs.cur_hole, hs.lt_stale, hs.cm_same, hs.narrow, hs.ranked);
        if (e == hipSuccess && optimistic && hs.lt_stale && !hs.stop && hs.n_done < max_paths) {
            // paths 0 .. n_done-1 are complete; the table has just been rebuilt by the ensure_lt above
            const int zero = 0;
            e = hipMemcpyAsync(&h->dstate->lt_stale, &zero, sizeof zero, hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) { first = hs.n_done; h->spin_requeues++; continue; }
        }
        if (e != hipSuccess) rc = fail(GH_ERR_HIP, "gh_spin failed: %s", hipGetErrorString(e));
        else if (hs.n_done > 0) {
            if (staged && hs.n_done <= launched) stage_deliver(h, paths_out, n1, recs, launched, hs.n_done);
            else rc = results_to_host(h, paths_out, d_paths, n1 * hs.n_done, recs, d_recs, sizeof(gh_path_rec) * hs.n_done);
        }
        break;
    }
    h->spin_partial_stride = 0;
    h->seg6 = false;
    h->fuse = false;
    h->rws = false;
    if (rc) return rc;
    *n_out = hs.n_done;
    *hole_at = hs.stop ? hs.hole_at : 0;
    return GH_OK;
}

// batched recovery: many windows of one shape, every kernel launched over all of them ---------
struct gh_batch {
    int n, dev, N, W, L;
    std::vector<gh_handle *> hs;
    hipStream_t stream;
    hipStream_t gstream[3];     // window groups of the batched kernels (gh_batch_spin): created on first use
    hipEvent_t gevent[3];
    win_desc *d_wd;
    dev_state *d_states;        // every window's dev_state, gathered by k_batch_states for one copy to the host
    uint8_t *d_paths;
    gh_path_rec *d_recs;
    double *d_partial;
    int cap_paths, nb;
    int prof_every;             // gh_batch_profile_enable
    int pipe_windows, pipe_aborted;     // last gh_batch_spin: windows the pipeline (wpipe.hpp) carried / handed back to gh_spin
    std::vector<hipEvent_t> pev[2];     // [0] extension, [1] reweight: start/stop pairs of the last spin (group 0's stream)
    size_t pused[2];
    int prof_windows;
    double prof_bytes[2];
};

extern "C" int gh_batch_destroy(gh_batch_t *b)
{
    if (!b) return GH_OK;
    hipSetDevice(b->dev);
    if (b->stream) hipStreamSynchronize(b->stream);
    hipFree(b->d_wd); hipFree(b->d_states); hipFree(b->d_paths); hipFree(b->d_recs); hipFree(b->d_partial);
    for (int g = 0; g < 3; g++) {
        if (b->gstream[g]) { hipStreamSynchronize(b->gstream[g]); hipStreamDestroy(b->gstream[g]); }
        if (b->gevent[g]) hipEventDestroy(b->gevent[g]);
    }
    for (int k = 0; k < 2; k++)
        for (hipEvent_t e : b->pev[k]) hipEventDestroy(e);
    if (b->stream) hipStreamDestroy(b->stream);
    delete b;
    return GH_OK;
}

extern "C" int gh_batch_create(gh_t **handles, int n, gh_batch_t **out)
{
    if (!handles || n < 1 || !out) return fail(GH_ERR_ARG, "bad argument");
    gh_handle *h0 = handles[0];
    for (int w = 0; w < n; w++) {
        gh_handle *h = handles[w];
        if (!h) return fail(GH_ERR_ARG, "null handle at %d", w);
        if (h->N != h0->N || h->W != h0->W || h->dev != h0->dev || h->cfg.storage != h0->cfg.storage ||
            h->cfg.cond_mode != h0->cfg.cond_mode || h->cfg.marginal_term != h0->cfg.marginal_term ||
            h->cfg.offer_zero != h0->cfg.offer_zero || memcmp(h->cfg.cand_order, h0->cfg.cand_order, 5) != 0)
            return fail(GH_ERR_ARG, "window %d differs from window 0 in shape, storage, mode or device", w);
        for (int v = 0; v < w; v++)
            if (handles[v] == h) return fail(GH_ERR_ARG, "window %d is the same handle as window %d", w, v);
    }
    HIPCHK(hipSetDevice(h0->dev));
    gh_batch *b = new (std::nothrow) gh_batch();
    if (!b) return fail(GH_ERR_NOMEM, "host allocation failed");
    b->n = n; b->dev = h0->dev; b->N = h0->N; b->W = h0->W; b->L = 0;
    b->hs.assign(handles, handles + n);
    b->stream = nullptr; b->d_wd = nullptr; b->d_states = nullptr; b->d_paths = nullptr; b->d_recs = nullptr; b->d_partial = nullptr;
    for (int g = 0; g < 3; g++) { b->gstream[g] = nullptr; b->gevent[g] = nullptr; }
    b->cap_paths = 0;
    b->pipe_windows = b->pipe_aborted = 0;
    b->prof_every = 0; b->pused[0] = b->pused[1] = 0; b->prof_windows = 0; b->prof_bytes[0] = b->prof_bytes[1] = 0.0;
    b->nb = (int)(((size_t)(b->N + 1) * (b->W > 8 ? b->W : 8) + 255) / 256);   // >= blocks of k_marg<.., true>
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_wd, sizeof(win_desc) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_states, sizeof(dev_state) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_partial, sizeof(double) * (size_t)b->nb * n);
    if (e != hipSuccess) { gh_batch_destroy(b); return fail(GH_ERR_NOMEM, "batch allocation failed: %s", hipGetErrorString(e)); }
    *out = b;
    return GH_OK;
}

// gh_batch_spin's bookkeeping over all windows in one launch each: the control words a spin starts from (k_spin_reset's), and
// every window's state into one array the host fetches with one copy (256 small copies cost 4 ms)
__global__ void k_batch_reset(const win_desc *wd)
{
    dev_state *st = wd[blockIdx.x].st;
    if (threadIdx.x == 0) { st->stop = 0; st->hole_at = 0; st->n_done = 0; st->lt_stale = 0; st->cw_unres = 0; st->pipe_status = 0; }
}
__global__ void k_batch_states(const win_desc *wd, dev_state *out)
{
    const unsigned *src = reinterpret_cast<const unsigned *>(wd[blockIdx.x].st);
    unsigned *dst = reinterpret_cast<unsigned *>(out + blockIdx.x);
    for (unsigned q = threadIdx.x; q < sizeof(dev_state) / 4; q += blockDim.x) dst[q] = src[q];
}

// ---- the window pipeline (wpipe.hpp): one persistent workgroup per window -------------------------------------------------
// threads per workgroup for lag count L (what the walker's register rotation leaves of 512 registers per SIMD lane); 0 = none
static int pipe_threads(int L)
{
    const int env = getenv("GH_PIPE_NT") ? atoi(getenv("GH_PIPE_NT")) : 0;      // (read on every call: the tests switch)
    // (beyond ten lags the 512-thread form walks chunks of 11..14 positions and loses to the candidate pools on their own streams:
    // 20-25k against 31k haplotypes/s at 128 windows, scratch/pipe_lsweep.py; GH_PIPE_MAX_L=14 takes it all the same -- the tests)
    const int max_l = getenv("GH_PIPE_MAX_L") ? atoi(getenv("GH_PIPE_MAX_L")) : 10;
    if (L < 2 || L > 14 || L > max_l) return 0;
    if (env == 512 || env == 768 || env == 1024) return env;
    return L <= 6 ? 1024 : (L <= 10 ? 768 : 512);
}
// which specs the pipeline carries: every conditional -- the row conditionals A, B, D and, on the to-major copy of the band, the
// column conditionals C, E --, with or without the marginal term (the walker adds it in front of the lag-1 term from the
// log-marginals the sweep keeps by rank).  GH_PIPE_COL=0 / GH_PIPE_MT=0 leave those to the batched launches (A/B measurements).
static bool pipe_spec_ok(const gh_handle *h)
{
    const bool col = h->cfg.cond_mode == GH_COND_C || h->cfg.cond_mode == GH_COND_E;
    const bool no_mt = getenv("GH_PIPE_MT") && atoi(getenv("GH_PIPE_MT")) == 0;
    const bool no_col = getenv("GH_PIPE_COL") && atoi(getenv("GH_PIPE_COL")) == 0;
    return !h->lt_full && !(col && no_col) && !(h->cfg.marginal_term && no_mt);
}
static int pipe_sweep_threads(int nt) { return nt == 1024 ? pipe_roles<1024>::NRW * 64 : (nt == 768 ? pipe_roles<768>::NRW * 64 : pipe_roles<512>::NRW * 64); }

template <typename T, int LC, int NT>
static hipError_t launch_wpipe_t(const pipe_params &P, const win_desc *d_wd, int n, size_t lds, hipStream_t st)
{
    hipError_t e = hipFuncSetAttribute((const void *)k_wpipe<T, LC, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_wpipe<T, LC, NT>), dim3(n), dim3(NT), lds, st, P, d_wd);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_wpipe(int L, int nt, const pipe_params &P, const win_desc *d_wd, int n, size_t lds, hipStream_t st)
{
#define GH_PIPE_CASE(l, t) case l: return launch_wpipe_t<T, l, t>(P, d_wd, n, lds, st);
    if (nt == 1024) {
        switch (L) { GH_PIPE_CASE(2, 1024) GH_PIPE_CASE(3, 1024) GH_PIPE_CASE(4, 1024) GH_PIPE_CASE(5, 1024) GH_PIPE_CASE(6, 1024) }
    } else if (nt == 768) {
        switch (L) { GH_PIPE_CASE(5, 768) GH_PIPE_CASE(7, 768) GH_PIPE_CASE(8, 768) GH_PIPE_CASE(9, 768) GH_PIPE_CASE(10, 768) }
    } else if (nt == 512) {
        switch (L) { GH_PIPE_CASE(3, 512) GH_PIPE_CASE(5, 512) GH_PIPE_CASE(11, 512) GH_PIPE_CASE(12, 512) GH_PIPE_CASE(13, 512) GH_PIPE_CASE(14, 512) }
    }
#undef GH_PIPE_CASE
    return hipErrorInvalidValue;
}
// ... and for windows with a few five-candidate positions (k_wpipe<.., WIDE = true>, wpipe.hpp): the default thread counts only
template <typename T, int LC, int NT>
static hipError_t launch_wpipe_w_t(const pipe_params &P, const win_desc *d_wd, int n, size_t lds, hipStream_t st)
{
    hipError_t e = hipFuncSetAttribute((const void *)k_wpipe<T, LC, NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_wpipe<T, LC, NT, true>), dim3(n), dim3(NT), lds, st, P, d_wd);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_wpipe_w(int L, int nt, const pipe_params &P, const win_desc *d_wd, int n, size_t lds, hipStream_t st)
{
#define GH_PIPE_CASE(l, t) case l: return launch_wpipe_w_t<T, l, t>(P, d_wd, n, lds, st);
    if (nt == 1024) {
        switch (L) { GH_PIPE_CASE(2, 1024) GH_PIPE_CASE(3, 1024) GH_PIPE_CASE(4, 1024) GH_PIPE_CASE(5, 1024) GH_PIPE_CASE(6, 1024) }
    } else if (nt == 768) {
        switch (L) { GH_PIPE_CASE(7, 768) GH_PIPE_CASE(8, 768) GH_PIPE_CASE(9, 768) GH_PIPE_CASE(10, 768) }
    }
#undef GH_PIPE_CASE
    return hipErrorInvalidValue;
}
static bool pipe_wide_instantiated(int L, int nt) { return (nt == 1024 && L >= 2 && L <= 6) || (nt == 768 && L >= 7 && L <= 10); }

static bool pipe_instantiated(int L, int nt)
{
    if (nt == 1024) return L >= 2 && L <= 6;
    if (nt == 768) return L == 5 || (L >= 7 && L <= 10);
    if (nt == 512) return L == 3 || L == 5 || (L >= 11 && L <= 14);
    return false;
}

// the window pipeline over the windows `wd` describes: marginals, snapshot and the full table for every window (what the batched
// launches do in front of their first path), then ONE launch that carries every window through all its paths (wpipe.hpp)
// launch_only: the preamble has run (batch_pipe_preamble) and b->d_wd + d_off holds the n descriptors this launch takes
static int batch_run_pipe(gh_batch *b, const std::vector<win_desc> &wd, int max_paths, double min_remove, int nt, bool launch_only = false, int d_off = 0, bool no_sync = false)
{
    const int n = (int)wd.size();
    if (n == 0) return GH_OK;
    if (!launch_only) HIPCHK(hipMemcpyAsync(b->d_wd, wd.data(), sizeof(win_desc) * n, hipMemcpyHostToDevice, b->stream));
    gh_handle *h0 = b->hs[0];
    const bool f64 = h0->cfg.storage == GH_STORAGE_F64;
    const int N = b->N, W = b->W, L = b->L;
    const int bwm = h0->wmode == WM_SEG ? WM_SPEC : h0->wmode;
    const unsigned marg_gx = (unsigned)(((N + 1) * 8 + 255) / 256);
    size_t lt_nb = ((size_t)(N + LT_PAD) * L * LT_BLK + 255) / 256;
    if (lt_nb > 4096) lt_nb = 4096;
    hipStream_t st = b->stream;
    const win_desc *gwd = b->d_wd + d_off;
    if (!launch_only) {
    hipLaunchKernelGGL(k_batch_reset, dim3(n), dim3(64), 0, b->stream, (const win_desc *)b->d_wd);
    HIPCHK(hipStreamSynchronize(b->stream));        // wd is a host temporary
    hipLaunchKernelGGL(k_rearm, dim3(n), dim3(64), 0, st, (dev_state *)nullptr, gwd, 0);
    if (f64) {
        hipLaunchKernelGGL((k_marg<double, false>), dim3(marg_gx, n), dim3(256), 0, st, (double *)nullptr, N, W,
                           (double *)nullptr, (double *)nullptr, (int32_t *)nullptr, (uint32_t *)nullptr, (double *)nullptr,
                           (dev_state *)nullptr, gwd, (const uint8_t *)nullptr, 0.0, 0, (double *)nullptr, 0, (double *)nullptr, 0, 0, (const double *)nullptr, (gh_path_rec *)nullptr,
                           h0->sm, h0->cfg.offer_zero, (double *)nullptr);
        hipLaunchKernelGGL(k_snapshot, dim3(marg_gx, n), dim3(256), 0, st, (double *)nullptr, (const double *)nullptr, N, gwd);
        // (the table WITHOUT the marginal term baked into its lag-1 entries: the pipeline's walker adds it itself)
        hipLaunchKernelGGL(k_lt<double>, dim3((unsigned)lt_nb, n), dim3(256), 0, st, (const double *)nullptr, N, W, L,
                           h0->cfg.cond_mode, 0, (const double *)nullptr, (const int32_t *)nullptr,
                           (const uint32_t *)nullptr, (const double *)nullptr, (double *)nullptr, (dev_state *)nullptr,
                           (const uint8_t *)nullptr, gwd, 0, walk_depth2_ok(bwm, L), (double *)nullptr, (double *)nullptr, h0->sm, (const double *)nullptr);
    } else {
        hipLaunchKernelGGL((k_marg<float, false>), dim3(marg_gx, n), dim3(256), 0, st, (float *)nullptr, N, W,
                           (double *)nullptr, (double *)nullptr, (int32_t *)nullptr, (uint32_t *)nullptr, (double *)nullptr,
                           (dev_state *)nullptr, gwd, (const uint8_t *)nullptr, 0.0, 0, (double *)nullptr, 0, (double *)nullptr, 0, 0, (const double *)nullptr, (gh_path_rec *)nullptr,
                           h0->sm, h0->cfg.offer_zero, (double *)nullptr);
        hipLaunchKernelGGL(k_snapshot, dim3(marg_gx, n), dim3(256), 0, st, (double *)nullptr, (const double *)nullptr, N, gwd);
        hipLaunchKernelGGL(k_lt<float>, dim3((unsigned)lt_nb, n), dim3(256), 0, st, (const float *)nullptr, N, W, L,
                           h0->cfg.cond_mode, 0, (const double *)nullptr, (const int32_t *)nullptr,
                           (const uint32_t *)nullptr, (const double *)nullptr, (double *)nullptr, (dev_state *)nullptr,
                           (const uint8_t *)nullptr, gwd, 0, walk_depth2_ok(bwm, L), (double *)nullptr, (double *)nullptr, h0->sm, (const float *)nullptr);
    }
    HIPCHK(hipGetLastError());
    if (nt == 0) return GH_OK;                      // (the preamble only: batch_pipe_preamble)
    }
    const int nr = pipe_sweep_threads(nt);
    pipe_params P;
    P.N = N; P.W = W; P.L = L; P.mt = h0->cfg.marginal_term ? 1 : 0; P.col = (h0->cfg.cond_mode == GH_COND_C || h0->cfg.cond_mode == GH_COND_E) ? 1 : 0;
    P.C = pipe_chunk(N, L, nr, f64 ? 8 : 4, P.mt); P.max_paths = max_paths; P.cond_mode = h0->cfg.cond_mode;
    P.synth = getenv("GH_PIPE_SYNTH") ? (atoi(getenv("GH_PIPE_SYNTH")) != 0) : 1;          // (wpipe.hpp, the loaders: what it gains and costs)
    P.offer_zero = h0->cfg.offer_zero; P.prof = (b->prof_every > 0 || getenv("GH_PIPE_STAMPS")) ? 1 : 0; P.min_remove = min_remove; P.sm = h0->sm;
    const size_t lds = pipe_lds_bytes(N, L, P.C, nr, f64 ? 8 : 4, P.mt);
    b->pused[0] = b->pused[1] = 0;
    auto pmark = [&](hipStream_t s_) {
        if (b->pused[0] >= b->pev[0].size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            b->pev[0].push_back(e);
        }
        hipEventRecord(b->pev[0][b->pused[0]++], s_);
    };
    const bool sample = b->prof_every > 0;
    if (sample) { b->prof_windows = n; pmark(st); }
    const hipError_t le = f64 ? launch_wpipe<double>(L, nt, P, gwd, n, lds, st) : launch_wpipe<float>(L, nt, P, gwd, n, lds, st);
    if (le != hipSuccess) return fail(GH_ERR_HIP, "gh_batch_spin: the pipeline launch failed (L=%d, %d threads, %zu bytes of LDS): %s", L, nt, lds, hipGetErrorString(le));
    if (sample) pmark(st);
    if (!no_sync) HIPCHK(hipStreamSynchronize(st));
    {
        // algorithmic bytes per window over the whole launch, the pipeline's own accounting: per position
        // and path the walk reads its compact table (4 rows x L lags x 32 bytes) and the bookkeeper the counts, the original
        // log-marginals and the packed word (64 + 48 + 8) and writes a path byte; the sweep reads the W rows of the path's symbol
        // (7 elements each), the counts and two packed words, writes W cells, two counts and min(L, W) x 32 bytes of table
        const double es = f64 ? 8.0 : 4.0;
        const int Lw = (h0->cfg.cond_mode == GH_COND_B || W >= L) ? L : W;
        const double ext = (double)N * (128.0 * L + 64 + 48 + 8 + 1);
        const double rw = (double)(N + 1) * (7.0 * es * W + es * W + 64 + 16 + 16 + 32.0 * Lw + 1);
        b->prof_bytes[0] = (ext + rw) * max_paths;
        b->prof_bytes[1] = 0.0;
    }
    return GH_OK;
}

// the batched launches of rounds 1-4 over the windows `wd` describes (a path = k_lt's flag check, one serial walker per window, the
// fused reweight k_marg<T,true>, the removed mass); returns with every stream drained
static int batch_run_launches(gh_batch *b, const std::vector<win_desc> &wd, int max_paths, double min_remove, bool reset = true)
{
    const int n = (int)wd.size();
    if (n == 0) return GH_OK;
    const size_t n1 = (size_t)b->N + 1;
    (void)n1;
    HIPCHK(hipMemcpyAsync(b->d_wd, wd.data(), sizeof(win_desc) * n, hipMemcpyHostToDevice, b->stream));
    if (reset) hipLaunchKernelGGL(k_batch_reset, dim3(n), dim3(64), 0, b->stream, (const win_desc *)b->d_wd);
    HIPCHK(hipStreamSynchronize(b->stream));        // wd is a host temporary

    gh_handle *h0 = b->hs[0];
    const bool f64 = h0->cfg.storage == GH_STORAGE_F64;
    const int N = b->N, W = b->W, L = b->L;
    const int bwm = h0->wmode == WM_SEG ? WM_SPEC : h0->wmode;     // batched launches: one serial walker per window
    const unsigned marg_gx = (unsigned)(((N + 1) * 8 + 255) / 256);
    size_t lt_nb = ((size_t)(N + LT_PAD) * L * LT_BLK + 255) / 256;
    if (lt_nb > 4096) lt_nb = 4096;
    walk_params P;
    P.N = N; P.L = L; P.chunk = 0; P.rearm = 1; P.G = nullptr; P.Ht = nullptr; P.Yt = nullptr; P.minfo = nullptr; P.path_out = nullptr; P.rec = nullptr; P.st = nullptr;
    P.min_remove = min_remove; P.sm = h0->sm;
    size_t lt_nb_inc = 64;                  // batched launches keep no walker tables: the steady-state k_lt only checks flags
    if (lt_nb_inc > 4096) lt_nb_inc = 4096;
    const bool inc_mode = lt_incremental_ok(h0);
    // The windows go in up to three GROUPS, each on its own stream and one phase behind the group in front: a path is a
    // serial extension (one workgroup per window, 0.6 ms for 10k SNPs whatever the number of windows: latency) followed by
    // a reweight (HBM-bound, ~4 us per window).  In step, every group would wait out the extension with an idle memory
    // system and then share it; staggered, one group reweights while the others extend.
    static const int groups_env = getenv("GH_BATCH_GROUPS") ? atoi(getenv("GH_BATCH_GROUPS")) : 0;
    int NG = groups_env > 0 ? groups_env : (n >= 96 ? 3 : (n >= 32 ? 2 : 1));
    if (NG > 3) NG = 3;
    if (NG > n) NG = n;
    for (int g = 0; g < NG; g++) {
        if (!b->gstream[g]) HIPCHK(hipStreamCreateWithFlags(&b->gstream[g], hipStreamNonBlocking));
        if (!b->gevent[g]) HIPCHK(hipEventCreateWithFlags(&b->gevent[g], hipEventDisableTiming));
    }
    b->pused[0] = b->pused[1] = 0;
    auto pmark = [&](int k, hipStream_t st) {            // one event of a start/stop pair of kernel k
        if (b->pused[k] >= b->pev[k].size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            b->pev[k].push_back(e);
        }
        hipEventRecord(b->pev[k][b->pused[k]++], st);
    };
    for (int s = 0; s < max_paths; s++) {
        // any non-null pointer tells k_lt that the fused reweight of spin s-1 has already rewritten the rows it changed
        const uint8_t *inc = (s > 0 && inc_mode) ? b->d_paths : nullptr;
        for (int g = 0; g < NG; g++) {
            const int w0 = (int)((long long)n * g / NG), ng = (int)((long long)n * (g + 1) / NG) - w0;
            const win_desc *gwd = b->d_wd + w0;
            hipStream_t st = b->gstream[g];
            if (s == 0) {
                // (group g starts when group g-1 has finished its first extension)
                if (g > 0) hipStreamWaitEvent(st, b->gevent[g - 1], 0);
                hipLaunchKernelGGL(k_rearm, dim3(ng), dim3(64), 0, st, (dev_state *)nullptr, gwd, 0);
                if (f64)
                    hipLaunchKernelGGL((k_marg<double, false>), dim3(marg_gx, ng), dim3(256), 0, st, (double *)nullptr, N, W,
                                       (double *)nullptr, (double *)nullptr, (int32_t *)nullptr, (uint32_t *)nullptr, (double *)nullptr,
                                       (dev_state *)nullptr, gwd, (const uint8_t *)nullptr, 0.0, 0, (double *)nullptr, 0, (double *)nullptr, 0, 0, (const double *)nullptr, (gh_path_rec *)nullptr,
                                       h0->sm, h0->cfg.offer_zero, (double *)nullptr);
                else
                    hipLaunchKernelGGL((k_marg<float, false>), dim3(marg_gx, ng), dim3(256), 0, st, (float *)nullptr, N, W,
                                       (double *)nullptr, (double *)nullptr, (int32_t *)nullptr, (uint32_t *)nullptr, (double *)nullptr,
                                       (dev_state *)nullptr, gwd, (const uint8_t *)nullptr, 0.0, 0, (double *)nullptr, 0, (double *)nullptr, 0, 0, (const double *)nullptr, (gh_path_rec *)nullptr,
                                       h0->sm, h0->cfg.offer_zero, (double *)nullptr);
                hipLaunchKernelGGL(k_snapshot, dim3(marg_gx, ng), dim3(256), 0, st, (double *)nullptr, (const double *)nullptr, N, gwd);
            }
            if (f64)
                hipLaunchKernelGGL(k_lt<double>, dim3((unsigned)(inc ? lt_nb_inc : lt_nb), ng), dim3(256), 0, st, (const double *)nullptr, N, W, L,
                                   h0->cfg.cond_mode, h0->cfg.marginal_term, (const double *)nullptr, (const int32_t *)nullptr,
                                   (const uint32_t *)nullptr, (const double *)nullptr, (double *)nullptr, (dev_state *)nullptr,
                                   inc, gwd, s, walk_depth2_ok(bwm, L), (double *)nullptr, (double *)nullptr, h0->sm, (const double *)nullptr);
            else
                hipLaunchKernelGGL(k_lt<float>, dim3((unsigned)(inc ? lt_nb_inc : lt_nb), ng), dim3(256), 0, st, (const float *)nullptr, N, W, L,
                                   h0->cfg.cond_mode, h0->cfg.marginal_term, (const double *)nullptr, (const int32_t *)nullptr,
                                   (const uint32_t *)nullptr, (const double *)nullptr, (double *)nullptr, (dev_state *)nullptr,
                                   inc, gwd, s, walk_depth2_ok(bwm, L), (double *)nullptr, (double *)nullptr, h0->sm, (const float *)nullptr);
            const bool sample = g == 0 && b->prof_every > 0 && s > 0 && (s % b->prof_every) == 0;
            if (sample) { b->prof_windows = ng; pmark(0, st); }
            launch_walk_any(bwm, N, L, P, st, ng, gwd, s);
            if (sample) pmark(0, st);
            if (s == 0 && g + 1 < NG) hipEventRecord(b->gevent[g], st);
            if (sample) pmark(1, st);
            if (f64)
                hipLaunchKernelGGL((k_marg<double, true>), dim3(marg_gx, ng), dim3(256), 0, st, (double *)nullptr, N, W,
                                   (double *)nullptr, (double *)nullptr, (int32_t *)nullptr, (uint32_t *)nullptr, (double *)nullptr,
                                   (dev_state *)nullptr, gwd, (const uint8_t *)nullptr, 0.0, 1, (double *)nullptr, s,
                                   inc_mode ? (double *)b->d_paths : (double *)nullptr, L, h0->cfg.cond_mode, (const double *)nullptr, (gh_path_rec *)nullptr,
                                   h0->sm, h0->cfg.offer_zero, (double *)nullptr);   // non-null = take G from wd
            else
                hipLaunchKernelGGL((k_marg<float, true>), dim3(marg_gx, ng), dim3(256), 0, st, (float *)nullptr, N, W,
                                   (double *)nullptr, (double *)nullptr, (int32_t *)nullptr, (uint32_t *)nullptr, (double *)nullptr,
                                   (dev_state *)nullptr, gwd, (const uint8_t *)nullptr, 0.0, 1, (double *)nullptr, s,
                                   inc_mode ? (double *)b->d_paths : (double *)nullptr, L, h0->cfg.cond_mode, (const double *)nullptr, (gh_path_rec *)nullptr,
                                   h0->sm, h0->cfg.offer_zero, (double *)nullptr);   // non-null = take G from wd
            if (sample) pmark(1, st);
            hipLaunchKernelGGL(k_reweight_finish, dim3(ng), dim3(256), 0, st, (const double *)nullptr, (int)marg_gx,
                               (dev_state *)nullptr, 1, (gh_path_rec *)nullptr, gwd, s);
        }
        {
            const hipError_t le = hipGetLastError();            // (a launch that failed is named with its path, not found at the end)
            if (le != hipSuccess) return fail(GH_ERR_HIP, "gh_batch_spin: a launch of path %d failed: %s", s, hipGetErrorString(le));
        }
    }
    for (int g = 0; g < NG; g++) HIPCHK(hipStreamSynchronize(b->gstream[g]));
    HIPCHK(hipGetLastError());
    {
        // algorithmic bytes per window and launch (launch_walk / launch_reweight_marg use the same definitions)
        const double es = f64 ? 8.0 : 4.0;
        const int wl = W < L ? W : L;
        b->prof_bytes[0] = (double)N * ((1.0 + (double)L) * CELL * es + 28.0);
        b->prof_bytes[1] = (double)(N + 1) * ((double)W * 2.0 * es + 1.0 + CELL * es + 2 * 64 + 88 + 8) +
                           (inc_mode ? (double)N * ((double)wl * 7 * es + (double)L * LT_ROW * 8.0) : 0.0);
    }
    return GH_OK;
}

// the windows the narrow pipeline left untouched because a position offers five candidates: the WIDE pipeline (wpipe.hpp) over them --
// marginals, snapshot and table stand (batch_run_pipe's preamble ran over every window); a window it cannot take either (too many such
// positions) keeps PIPE_NOT_STARTED
static int batch_run_pipe_wide(gh_batch *b, const std::vector<win_desc> &wd, int max_paths, double min_remove, int nt, hipStream_t stream = nullptr, int d_off = -1)
{
    const int n = (int)wd.size();
    if (n == 0) return GH_OK;
    gh_handle *h0 = b->hs[0];
    const bool f64 = h0->cfg.storage == GH_STORAGE_F64;
    const int N = b->N, W = b->W, L = b->L;
    const bool own = d_off < 0;                     // (else: the descriptors stand at b->d_wd + d_off, the caller waits for `stream`)
    if (own) { stream = b->stream; d_off = 0; HIPCHK(hipMemcpyAsync(b->d_wd, wd.data(), sizeof(win_desc) * n, hipMemcpyHostToDevice, b->stream)); }
    const int nr = pipe_sweep_threads(nt);
    pipe_params P;
    P.N = N; P.W = W; P.L = L; P.mt = h0->cfg.marginal_term ? 1 : 0; P.col = (h0->cfg.cond_mode == GH_COND_C || h0->cfg.cond_mode == GH_COND_E) ? 1 : 0;
    P.C = pipe_chunk_w(N, L, nr, f64 ? 8 : 4, P.mt); P.max_paths = max_paths; P.cond_mode = h0->cfg.cond_mode;
    P.synth = getenv("GH_PIPE_SYNTH") ? (atoi(getenv("GH_PIPE_SYNTH")) != 0) : 1;
    P.offer_zero = h0->cfg.offer_zero; P.prof = 0; P.min_remove = min_remove; P.sm = h0->sm;
    const size_t lds = pipe_lds_bytes_w(N, L, P.C, nr, f64 ? 8 : 4, P.mt);
    const hipError_t le = f64 ? launch_wpipe_w<double>(L, nt, P, b->d_wd + d_off, n, lds, stream) : launch_wpipe_w<float>(L, nt, P, b->d_wd + d_off, n, lds, stream);
    if (le != hipSuccess) return fail(GH_ERR_HIP, "gh_batch_spin: the wide pipeline launch failed (L=%d, %d threads, %zu bytes of LDS): %s", L, nt, lds, hipGetErrorString(le));
    if (own) HIPCHK(hipStreamSynchronize(b->stream));        // wd is a host temporary
    return GH_OK;
}

// gh_spin over some windows of a batch from a few host threads, every window on its own stream (the windows' kernel chains
// interleave on the GPU).  A job is (window, paths it already has): the spin writes the window's remaining paths behind them.
static int batch_spin_on_streams(gh_batch *b, const std::vector<std::pair<int, int>> &jobs, int max_paths, double min_remove,
                                 uint8_t *paths_out, gh_path_rec *recs, int *n_out, int *hole_at)
{
    const int nj = (int)jobs.size();
    if (nj == 0) return GH_OK;
    const size_t n1 = (size_t)b->N + 1;
    static const int nthr_env = getenv("GH_BATCH_THREADS") ? atoi(getenv("GH_BATCH_THREADS")) : 8;
    const int nthr = nthr_env < 1 ? 1 : (nthr_env > nj ? nj : nthr_env);
    std::vector<int> rcs(nj, GH_OK);
    std::vector<std::string> errs(nj);
    std::atomic<int> next(0);
    auto work = [&]() {
        hipSetDevice(b->dev);
        for (;;) {
            const int q = next.fetch_add(1);
            if (q >= nj) break;
            const int w = jobs[q].first, nd = jobs[q].second;
            int n2 = 0, hole2 = 0;
            if (nd < max_paths) {
                rcs[q] = gh_spin(b->hs[w], max_paths - nd, min_remove, paths_out + n1 * ((size_t)max_paths * w + nd), recs + (size_t)max_paths * w + nd, &n2, &hole2);
                if (rcs[q]) errs[q] = gh_last_error();
            }
            n_out[w] = nd + n2;
            hole_at[w] = hole2;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthr; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    for (int q = 0; q < nj; q++)
        if (rcs[q]) return fail(rcs[q], "window %d: %s", jobs[q].first, errs[q].c_str());
    return GH_OK;
}

extern "C" int gh_batch_spin(gh_batch_t *b, int max_paths, double min_remove, uint8_t *paths_out,
                             gh_path_rec *recs, int *n_out, int *hole_at)
{
    if (!b || !paths_out || !recs || !n_out || !hole_at || max_paths < 1) return fail(GH_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(b->dev));
    const bool hprof = getenv("GH_PIPE_STAMPS") != nullptr;        // host phases of this call on stderr
    auto tnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tph = tnow();
    auto phase = [&](const char *what) { if (hprof) { const double t = tnow(); fprintf(stderr, "gh_batch_spin: %-28s %8.3f ms\n", what, t - tph); tph = t; } };
    const int n = b->n;
    const size_t n1 = (size_t)b->N + 1;
    int rc;
    for (int w = 0; w < n; w++) {
        gh_handle *h = b->hs[w];
        if (h->L != b->hs[0]->L)
            return fail(GH_ERR_STATE, "window %d has L=%d but window 0 has L=%d: set one L (gh_set_L) for the batch", w, h->L, b->hs[0]->L);
        if ((rc = alloc_lt(h))) return rc;
    }
    // whatever the handles' own streams still carry (fills) must be done before the batch's stream touches the tensors -- those
    // streams, not the device: a device-wide wait also waits for whatever else the process runs (ResultExchange's gather on its
    // side stream in a multi-rank run)
    for (int w = 0; w < n; w++) HIPCHK(hipStreamSynchronize(b->hs[w]->stream));
    b->L = b->hs[0]->L;
    phase("tables allocated, device idle");
    // (measured on C3, MI355X: 8 windows 37k haplotypes/s this way against ~16k batched; 32 windows 45k either way -- the
    // chip is then busy with k_seg; from 48 windows on the batched serial walkers, one workgroup per window, win: 114k at 256)
    // (read on every call: the tests switch between the two ways; -1 = always the batched kernels)
    const int batch_cut = getenv("GH_BATCH_STREAMS_MAX") ? atoi(getenv("GH_BATCH_STREAMS_MAX")) : 47;
    // The window pipeline (wpipe.hpp): row conditionals without the marginal term, 2..14 lags, a path that fits the LDS beside the
    // walker's tables; from GH_PIPE_MIN windows on (a window alone runs at one walker's pace, ~2 000 paths/s: below two dozen
    // windows the segment-parallel spins on their own streams are faster).  GH_PIPE=0: the batched launches of rounds 1-4.
    int pipe_nt = 0;
    {
        gh_handle *h0 = b->hs[0];
        const int pipe_env = getenv("GH_PIPE") ? atoi(getenv("GH_PIPE")) : 1;
        const int pipe_min = getenv("GH_PIPE_MIN") ? atoi(getenv("GH_PIPE_MIN")) : 24;
        const int bwm0 = h0->wmode == WM_SEG ? WM_SPEC : h0->wmode;
        const int nt = pipe_threads(b->L);
        if (pipe_env && n >= pipe_min && nt && pipe_instantiated(b->L, nt) && walk_depth2_ok(bwm0, b->L) && pipe_spec_ok(h0) &&
            (unsigned long long)(b->N + 2) * 49ull * (unsigned long long)b->W < (1ull << 31) &&      // (the sweep's 32-bit element offsets)
            pipe_chunk(b->N, b->L, pipe_sweep_threads(nt), h0->cfg.storage == GH_STORAGE_F64 ? 8 : 4, h0->cfg.marginal_term) > 0)
            pipe_nt = nt;
    }
    if (!pipe_nt && batch_cut >= 0 &&
        ((seg_ok(b->hs[0]->wmode, b->L) && n <= batch_cut) || cw_ok(b->hs[0]->wmode, b->L))) {
        // The segment-parallel extensions fill the chip poorly with ONE window (four small dependent kernels per path, most
        // of their time launch and first-touch latency) but every window has its own stream: a few host threads each
        // run gh_spin over their share of the windows, and the windows' kernel chains interleave on the GPU.
        std::vector<std::pair<int, int>> jobs;
        for (int w = 0; w < n; w++) jobs.emplace_back(w, 0);
        return batch_spin_on_streams(b, jobs, max_paths, min_remove, paths_out, recs, n_out, hole_at);
    }
    if (max_paths > b->cap_paths) {
        HIPCHK(hipStreamSynchronize(b->stream));
        hipFree(b->d_paths); hipFree(b->d_recs);
        b->d_paths = nullptr; b->d_recs = nullptr; b->cap_paths = 0;
        HIPCHK(hipMalloc((void **)&b->d_paths, n1 * max_paths * n));
        HIPCHK(hipMalloc((void **)&b->d_recs, sizeof(gh_path_rec) * (size_t)max_paths * n));
        b->cap_paths = max_paths;
    }
    if (pipe_nt)
        for (int w = 0; w < n; w++) {
            gh_handle *h = b->hs[w];
            if (!h->pipe_pk) HIPCHK(hipMalloc((void **)&h->pipe_pk, sizeof(unsigned long long) * ((size_t)b->N + 2)));
            const size_t need = sizeof(double) * ((size_t)b->N + LT_PAD) * 16 * (size_t)b->L;
            if (h->pipe_gp_bytes < need) {
                hipFree(h->pipe_gp); h->pipe_gp = nullptr; h->pipe_gp_bytes = 0;
                HIPCHK(hipMalloc((void **)&h->pipe_gp, need));
                h->pipe_gp_bytes = need;
            }
            if (h->cfg.marginal_term && !h->pipe_lm) HIPCHK(hipMalloc((void **)&h->pipe_lm, sizeof(double) * 4 * ((size_t)b->N + 2)));
            if (h->cfg.cond_mode == GH_COND_C || h->cfg.cond_mode == GH_COND_E) {
                // the to-major copy of the band, in step with it (the pipeline's sweep then keeps both, element by element)
                const size_t nel = h->n_cells * CELL;
                if (!h->tband && hipMalloc(&h->tband, nel * esize(h)) != hipSuccess) { h->tband = nullptr; return fail(GH_ERR_NOMEM, "hipMalloc for the to-major band failed"); }
                if (h->tband_epoch != h->band_epoch) {
                    const unsigned nbt = (unsigned)((nel + 255) / 256);
                    if (h->cfg.storage == GH_STORAGE_F64) hipLaunchKernelGGL(k_band_to_major<double>, dim3(nbt), dim3(256), 0, b->stream, (const double *)h->band, (double *)h->tband, nel, h->W);
                    else hipLaunchKernelGGL(k_band_to_major<float>, dim3(nbt), dim3(256), 0, b->stream, (const float *)h->band, (float *)h->tband, nel, h->W);
                    h->tband_epoch = h->band_epoch;
                }
            }
        }
    std::vector<win_desc> wd(n);
    for (int w = 0; w < n; w++) {
        gh_handle *h = b->hs[w];
        wd[w].band = h->band; wd[w].cnt = h->cnt; wd[w].marg = h->marg; wd[w].minfo = h->minfo;
        wd[w].nvalid = h->nvalid; wd[w].cmask = h->cmask; wd[w].rinfo = h->need_rinfo ? h->rinfo : nullptr; wd[w].G = h->lt; wd[w].Ht = nullptr; wd[w].Yt = nullptr; wd[w].st = h->dstate;   // no walker tables in batches (kernels.hpp)
        wd[w].partial = b->d_partial + (size_t)w * b->nb;
        wd[w].paths = b->d_paths + n1 * max_paths * w;
        wd[w].recs = b->d_recs + (size_t)max_paths * w;
        wd[w].snap = h->have_orig ? 0 : 1;       // the first batched k_marg is followed by a batched snapshot
        wd[w]._pad = 0;
        wd[w].pk = h->pipe_pk;
        wd[w].gp = h->pipe_gp;
        wd[w].lmr = h->pipe_lm;
        wd[w].tband = h->tband;
        wd[w].gw = h->pipe_gw;
        wd[w].wdir = h->pipe_gw ? reinterpret_cast<int *>(h->pipe_gw + (size_t)PIPE_WMAX * pipe_wrec_doubles(b->L)) : nullptr;
        h->have_orig = true;
    }
    std::vector<dev_state> hs(n);
    auto fetch_states = [&]() -> int {
        // (b->d_wd may describe a sub-list by now: the gather takes the full list again)
        HIPCHK(hipMemcpyAsync(b->d_wd, wd.data(), sizeof(win_desc) * n, hipMemcpyHostToDevice, b->stream));
        hipLaunchKernelGGL(k_batch_states, dim3(n), dim3(64), 0, b->stream, (const win_desc *)b->d_wd, b->d_states);
        HIPCHK(hipMemcpyAsync(hs.data(), b->d_states, sizeof(dev_state) * n, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        return GH_OK;
    };
    std::vector<int> aborted;           // windows whose pipeline stopped at a moved candidate mask: gh_spin takes their remaining paths
    std::vector<std::pair<int, int>> later;     // (window, paths done): windows gh_spin finishes on their own streams behind the batch's copy
    b->pipe_windows = 0;
    phase("descriptors");
    if (pipe_nt) {
        // Windows in which a position offers five candidates (deletion columns) go to the pipeline's WIDE launch (wpipe.hpp), the
        // others to the narrow one -- side by side on two streams: which is which is only known once the marginals stand, so the
        // preamble runs first and the host looks at the states (one wait, 30 us).  GH_PIPE_WIDE=0: the narrow launch alone, what it
        // leaves goes to the batched launches as until round 5.
        gh_handle *hp0 = b->hs[0];
        const bool wide_on = !(getenv("GH_PIPE_WIDE") && atoi(getenv("GH_PIPE_WIDE")) == 0) && pipe_wide_instantiated(b->L, pipe_nt) &&
                             !hp0->cfg.offer_zero && b->N < 65536 &&
                             pipe_chunk_w(b->N, b->L, pipe_sweep_threads(pipe_nt), hp0->cfg.storage == GH_STORAGE_F64 ? 8 : 4, hp0->cfg.marginal_term) > 0;
        std::vector<win_desc> both;             // (the descriptors of the two launches: alive until both have ended)
        std::vector<char> listed((size_t)n, 1);
        if (!wide_on) {
            if ((rc = batch_run_pipe(b, wd, max_paths, min_remove, pipe_nt))) return rc;
        } else {
            if ((rc = batch_run_pipe(b, wd, max_paths, min_remove, 0))) return rc;          // reset + marginals, snapshot, table
            if ((rc = fetch_states())) return rc;
            std::vector<win_desc> nar, wid;
            const size_t need = pipe_gw_bytes(b->N, b->L);
            for (int w = 0; w < n; w++) {
                const bool live = !hs[w].stop && hs[w].first_hole > b->N;
                if (live && hs[w].ranked != 0 && hs[w].narrow != 0) nar.push_back(wd[w]);
                else if (live && hs[w].ranked == 0) {
                    gh_handle *h = b->hs[w];
                    if (h->pipe_gw_bytes < need) {
                        hipFree(h->pipe_gw); h->pipe_gw = nullptr; h->pipe_gw_bytes = 0;
                        HIPCHK(hipMalloc((void **)&h->pipe_gw, need));
                        h->pipe_gw_bytes = need;
                    }
                    wd[w].gw = h->pipe_gw;
                    wd[w].wdir = reinterpret_cast<int *>(h->pipe_gw + (size_t)PIPE_WMAX * pipe_wrec_doubles(b->L));
                    wid.push_back(wd[w]);
                } else listed[(size_t)w] = 0;       // (a hole: neither launch takes it)
            }
            both = nar;
            both.insert(both.end(), wid.begin(), wid.end());
            if (!both.empty()) HIPCHK(hipMemcpyAsync(b->d_wd, both.data(), sizeof(win_desc) * both.size(), hipMemcpyHostToDevice, b->stream));
            HIPCHK(hipStreamSynchronize(b->stream));
            if (!wid.empty() && !b->gstream[0]) HIPCHK(hipStreamCreateWithFlags(&b->gstream[0], hipStreamNonBlocking));
            if (!nar.empty() && (rc = batch_run_pipe(b, nar, max_paths, min_remove, pipe_nt, true, 0, true))) return rc;
            if (!wid.empty() && (rc = batch_run_pipe_wide(b, wid, max_paths, min_remove, pipe_nt, b->gstream[0], (int)nar.size()))) return rc;
            HIPCHK(hipStreamSynchronize(b->stream));
            if (!wid.empty()) HIPCHK(hipStreamSynchronize(b->gstream[0]));
        }
        phase("preamble + pipeline kernel");
        if ((rc = fetch_states())) return rc;
        for (int w = 0; w < n; w++)
            if (!listed[(size_t)w]) hs[w].pipe_status = PIPE_NOT_STARTED;
        phase("states to host");
        std::vector<win_desc> rest;     // not eligible when the kernel looked (a position offers five candidates, a hole): the batched launches
        for (int w = 0; w < n; w++) {
            if (hs[w].pipe_status == PIPE_NOT_STARTED) rest.push_back(wd[w]);
            else if (hs[w].pipe_status == PIPE_ABORTED) aborted.push_back(w);
            else if (hs[w].pipe_status != PIPE_DONE) return fail(GH_ERR_STATE, "gh_batch_spin: window %d left the pipeline in state %d", w, hs[w].pipe_status);
        }
        b->pipe_windows = n - (int)rest.size();
        auto print_stamps = [&]() {
        if (getenv("GH_PIPE_STAMPS")) {         // 100 MHz ticks per path as the bookkeepers of a few windows saw them
            for (int w = 0; w < n; w += (n > 4 ? n / 4 : 1)) {
#ifdef PIPE_PROF
                fprintf(stderr, "pipe window %d, Mcycles (work, own memory, barrier): sweeper %.2f %.2f %.2f  loader %.2f %.2f %.2f  bookkeeper %.2f %.2f %.2f\n", w,
                        hs[w].dbg8[0] / 1e6, hs[w].dbg8[1] / 1e6, hs[w].dbg8[2] / 1e6, hs[w].dbg8[3] / 1e6, hs[w].dbg8[4] / 1e6, hs[w].dbg8[5] / 1e6,
                        hs[w].dbg8[6] / 1e6, hs[w].dbg8[7] / 1e6, hs[w].dbg8[8] / 1e6);
                fprintf(stderr, "walker: %.2f Mcycles walking, %.2f at its barriers; bookkeeper: %.2f per-position part, %.2f sequential sums\n", hs[w].dbg8[10] / 1e6, hs[w].dbg8[11] / 1e6, hs[w].dbg[0] / 1e6, hs[w].dbg[1] / 1e6);
                fprintf(stderr, "wide walker: %.2f Mcycles in %llu groups by the exact stepper, %.2f Mcycles inside the speculative blocks, %llu primes\n", (double)(hs[w].dbg[2] & ((1ull << 40) - 1)) / 1e6,
                        (unsigned long long)(hs[w].dbg[2] >> 40), (double)(hs[w].dbg[3] & ((1ull << 40) - 1)) / 1e6, (unsigned long long)(hs[w].dbg[3] >> 40));
                fprintf(stderr, "SIMD of waves 0..15:");
                for (int q = 0; q < 16; q++) fprintf(stderr, " %d", (int)((hs[w].dbg8[9] >> (2 * q)) & 3));
                fprintf(stderr, "\n");
                break;
#else
                fprintf(stderr, "pipe window %d, us per path:", w);
                for (int q = 0; q < 12 && q < max_paths; q++) fprintf(stderr, " %.1f", (double)hs[w].dbg8[q] / 100.0);
                fprintf(stderr, "\n");
#endif
            }
        }
        };
        print_stamps();
        // What the two launches left untouched (a hole; too many five-candidate positions for the WIDE launch's records; every
        // five-candidate window under GH_PIPE_WIDE=0).  A FEW such windows go the
        // single window's way, each on its own stream from a few host threads -- the segment-parallel / candidate-pool spins (mixed
        // radix at five lags): 25-37k haplotypes/s where the batched serial walkers, one wavefront per window, give 16k for eight
        // windows; from four dozen on the batched launches over all of them.
        if (!rest.empty() && batch_cut >= 0 && (int)rest.size() <= batch_cut && (seg_ok(b->hs[0]->wmode, b->L) || cw_ok(b->hs[0]->wmode, b->L))) {
            for (int w = 0; w < n; w++)
                if (hs[w].pipe_status == PIPE_NOT_STARTED) later.emplace_back(w, 0);
            rest.clear();
        }
        if (!rest.empty()) {
            if ((rc = batch_run_launches(b, rest, max_paths, min_remove))) return rc;
            if ((rc = fetch_states())) return rc;
        }
    } else {
        if ((rc = batch_run_launches(b, wd, max_paths, min_remove))) return rc;
        if ((rc = fetch_states())) return rc;
    }
    HIPCHK(hipMemcpyAsync(paths_out, b->d_paths, n1 * max_paths * n, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(recs, b->d_recs, sizeof(gh_path_rec) * (size_t)max_paths * n, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    phase("paths and records to host");
    for (int w = 0; w < n; w++) {
        n_out[w] = hs[w].n_done;
        hole_at[w] = hs[w].stop ? hs[w].hole_at : 0;
        const bool kept = b->hs[w]->tband && b->hs[w]->tband_epoch == b->hs[w]->band_epoch && pipe_nt &&
                          (hs[w].pipe_status == PIPE_DONE || hs[w].pipe_status == PIPE_ABORTED) &&
                          (b->hs[w]->cfg.cond_mode == GH_COND_C || b->hs[w]->cfg.cond_mode == GH_COND_E);
        b->hs[w]->dirty_marg = b->hs[w]->dirty_lt = true; b->hs[w]->band_epoch++;
        if (kept) b->hs[w]->tband_epoch = b->hs[w]->band_epoch;       // (the pipeline's sweep wrote both)
        b->hs[w]->lt_inc_path = nullptr;       // the batch reweighted many paths and maintains no walker tables: rebuild in full
    }
    // aborted windows: paths 0 .. n_done-1 are complete and reweighted; marginals and table are rebuilt from the tensor (dirty flags
    // above).  In a deep spin many windows stop once counts reach zero: they, and the few the pipeline did not take, are finished
    // side by side (one after the other on the calling thread they were the tail of the batch)
    for (int w : aborted) later.emplace_back(w, hs[w].n_done);
    if ((rc = batch_spin_on_streams(b, later, max_paths, min_remove, paths_out, recs, n_out, hole_at))) return rc;
    b->pipe_aborted = (int)aborted.size();
    return GH_OK;
}

extern "C" int gh_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0) return fail(GH_ERR_ARG, "bad argument");
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(GH_ERR_NOMEM, "hipHostMalloc of %zu bytes failed", bytes); }
    *out = p;
    return GH_OK;
}
extern "C" int gh_host_free(void *p)
{
    if (p && hipHostFree(p) != hipSuccess) return fail(GH_ERR_HIP, "hipHostFree failed");
    return GH_OK;
}

extern "C" int gh_batch_pipe_info(gh_batch_t *b, int32_t out[4])
{
    if (!b || !out) return fail(GH_ERR_ARG, "bad argument");
    const int nt = pipe_threads(b->L);
    out[0] = b->pipe_windows; out[1] = b->pipe_aborted;
    out[2] = b->pipe_windows ? nt : 0;
    out[3] = b->pipe_windows ? pipe_chunk(b->N, b->L, pipe_sweep_threads(nt), b->hs[0]->cfg.storage == GH_STORAGE_F64 ? 8 : 4, b->hs[0]->cfg.marginal_term) : 0;
    return GH_OK;
}

extern "C" int gh_batch_profile_enable(gh_batch_t *b, int every)
{
    if (!b) return fail(GH_ERR_ARG, "null batch");
    b->prof_every = every > 0 ? every : 0;
    return GH_OK;
}

extern "C" int gh_batch_profile_get(gh_batch_t *b, int kernel, double *total_ms, int64_t *launches, int32_t *windows, double *bytes_per_launch)
{
    if (!b || (kernel != GH_K_WALK && kernel != GH_K_REWEIGHT)) return fail(GH_ERR_ARG, "bad argument");
    const int k = kernel == GH_K_WALK ? 0 : 1;
    double ms = 0.0;
    int64_t n = 0;
    for (size_t q = 0; q + 1 < b->pused[k]; q += 2) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, b->pev[k][q], b->pev[k][q + 1]) == hipSuccess) { ms += t; n++; }
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (windows) *windows = b->prof_windows;
    if (bytes_per_launch) *bytes_per_launch = b->prof_bytes[k] * b->prof_windows;
    return GH_OK;
}

// export / import -----------------------------------------------------------------------------
extern "C" int gh_export_band(gh_t *h, double *out)
{
    if (!h || !out) return fail(GH_ERR_ARG, "null argument");
    if (set_dev(h)) return GH_ERR_HIP;
    const size_t n = h->n_cells * CELL;
    double *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, n * sizeof(double)));
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (h->cfg.storage == GH_STORAGE_F64)
        hipLaunchKernelGGL(k_export<double>, dim3(nb), dim3(256), 0, h->stream, (const double *)h->band, d, n, h->W);
    else
        hipLaunchKernelGGL(k_export<float>, dim3(nb), dim3(256), 0, h->stream, (const float *)h->band, d, n, h->W);
    hipError_t e = hipMemcpyAsync(out, d, n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(GH_ERR_HIP, "export failed: %s", hipGetErrorString(e));
    return GH_OK;
}

extern "C" int gh_import_band(gh_t *h, const double *in)
{
    if (!h || !in) return fail(GH_ERR_ARG, "null argument");
    if (set_dev(h)) return GH_ERR_HIP;
    const size_t n = h->n_cells * CELL;
    double *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, n * sizeof(double)));
    hipError_t e = hipMemcpyAsync(d, in, n * sizeof(double), hipMemcpyHostToDevice, h->stream);
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (h->cfg.storage == GH_STORAGE_F64)
        hipLaunchKernelGGL(k_import<double>, dim3(nb), dim3(256), 0, h->stream, (double *)h->band, d, n, h->W);
    else
        hipLaunchKernelGGL(k_import<float>, dim3(nb), dim3(256), 0, h->stream, (float *)h->band, d, n, h->W);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    h->dirty_marg = h->dirty_lt = true; h->lt_inc_path = nullptr; h->band_zero = false; h->band_epoch++;
    if (e != hipSuccess) return fail(GH_ERR_HIP, "import failed: %s", hipGetErrorString(e));
    return GH_OK;
}

extern "C" int gh_export_dense(gh_t *h, double *out)
{
    if (!h || !out) return fail(GH_ERR_ARG, "null argument");
    const size_t np = (size_t)h->N + 2;
    std::vector<double> band(h->n_cells * CELL);
    int rc = gh_export_band(h, band.data());
    if (rc) return rc;
    memset(out, 0, sizeof(double) * NSYM * NSYM * np * np);
    for (size_t i = 0; i < np; i++)
        for (int d = 1; d <= h->W; d++) {
            size_t j = i + d;
            if (j >= np) break;
            const double *cell = band.data() + (i * h->W + (d - 1)) * CELL;
            for (int a = 0; a < NSYM; a++)
                for (int b = 0; b < NSYM; b++)
                    out[(((size_t)a * NSYM + b) * np + i) * np + j] = cell[a * NSYM + b];
        }
    return GH_OK;
}

// gretel-snpper -------------------------------------------------------------------------------
extern "C" int gh_coverage_sites(int device, const int32_t *ref_start, const int64_t *off, const uint8_t *codes, int64_t n_runs,
                                 int32_t start0, int32_t len, int32_t depth, int32_t *counts_out, uint8_t *site_out)
{
    if (n_runs < 0 || len < 0 || !site_out || (n_runs > 0 && (!ref_start || !off || !codes))) return fail(GH_ERR_ARG, "bad argument");
    if (len == 0) return GH_OK;
    if (device >= 0) HIPCHK(hipSetDevice(device));
    const int64_t n_bases = n_runs ? off[n_runs] : 0;
    int32_t *d_ref = nullptr;
    int64_t *d_off = nullptr;
    uint8_t *d_codes = nullptr, *d_site = nullptr;
    unsigned *d_counts = nullptr;
    hipError_t e = hipMalloc((void **)&d_counts, sizeof(unsigned) * 4 * (size_t)len);
    if (e == hipSuccess) e = hipMalloc((void **)&d_site, (size_t)len);
    if (e == hipSuccess) e = hipMalloc((void **)&d_ref, sizeof(int32_t) * (size_t)(n_runs ? n_runs : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_off, sizeof(int64_t) * (size_t)(n_runs + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_codes, (size_t)(n_bases ? n_bases : 1));
    if (e == hipSuccess) e = hipMemset(d_counts, 0, sizeof(unsigned) * 4 * (size_t)len);
    if (e == hipSuccess && n_runs) {
        e = hipMemcpy(d_ref, ref_start, sizeof(int32_t) * (size_t)n_runs, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_off, off, sizeof(int64_t) * (size_t)(n_runs + 1), hipMemcpyHostToDevice);
        if (e == hipSuccess && n_bases) e = hipMemcpy(d_codes, codes, (size_t)n_bases, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
        if (n_runs) {
            int64_t nb = (n_runs + 3) / 4;                  // one wavefront per run, four per workgroup
            if (nb > 256 * 16) nb = 256 * 16;
            hipLaunchKernelGGL(k_cov, dim3((unsigned)nb), dim3(256), 0, 0, d_ref, d_off, d_codes, n_runs, start0, len, d_counts);
        }
        hipLaunchKernelGGL(k_sites, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, 0, d_counts, len, depth, d_site);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(site_out, d_site, (size_t)len, hipMemcpyDeviceToHost);
    if (e == hipSuccess && counts_out) e = hipMemcpy(counts_out, d_counts, sizeof(unsigned) * 4 * (size_t)len, hipMemcpyDeviceToHost);
    hipFree(d_counts); hipFree(d_site); hipFree(d_ref); hipFree(d_off); hipFree(d_codes);
    if (e != hipSuccess) return fail(GH_ERR_HIP, "gh_coverage_sites failed: %s", hipGetErrorString(e));
    return GH_OK;
}

// profiling -----------------------------------------------------------------------------------
extern "C" int gh_profile_enable(gh_t *h, int on)
{
    if (!h) return fail(GH_ERR_ARG, "null handle");
    h->prof = on > 0 ? on : 0;
    for (int k = 0; k < GH_K_COUNT; k++) { h->ps[k].seq = 0; h->ps[k].open = false; }
    return GH_OK;
}

extern "C" int gh_profile_reset(gh_t *h)
{
    if (!h) return fail(GH_ERR_ARG, "null handle");
    if (set_dev(h)) return GH_ERR_HIP;
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int k = 0; k < GH_K_COUNT; k++) {
        h->ps[k].used = 0; h->ps[k].ms = 0.0; h->ps[k].launches = 0;
    }
    return GH_OK;
}

extern "C" int gh_profile_get(gh_t *h, int kernel, double *total_ms, int64_t *launches)
{
    if (!h || kernel < 0 || kernel >= GH_K_COUNT) return fail(GH_ERR_ARG, "bad argument");
    if (set_dev(h)) return GH_ERR_HIP;
    HIPCHK(hipStreamSynchronize(h->stream));
    prof_collect(h);
    if (total_ms) *total_ms = h->ps[kernel].ms;
    if (launches) *launches = h->ps[kernel].launches;
    return GH_OK;
}

__global__ void k_nop() {}

// What a HIP-event bracket reads beyond the kernel inside it, on this handle's stream: out[0] = two events with nothing
// between them, out[1] = the same around an empty one-workgroup kernel (ms, medians of `reps`).  bench.py reports them
// next to the bracketed kernel times: for kernels of 5-15 us the bracket itself is a fifth of the reading.
extern "C" int gh_profile_overhead(gh_t *h, int reps, double out[2])
{
    if (!h || !out || reps < 1) return fail(GH_ERR_ARG, "bad argument");
    if (set_dev(h)) return GH_ERR_HIP;
    std::vector<float> a, b;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return fail(GH_ERR_HIP, "hipEventCreate failed"); }
    for (int pass = 0; pass < 2; pass++)
        for (int r = 0; r < reps; r++) {
            // a kernel in front, as in the timed region (the stream is never idle there)
            hipLaunchKernelGGL(k_nop, dim3(256), dim3(256), 0, h->stream);
            hipEventRecord(e0, h->stream);
            if (pass == 1) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, h->stream);
            hipEventRecord(e1, h->stream);
            hipLaunchKernelGGL(k_nop, dim3(256), dim3(256), 0, h->stream);
            if (hipStreamSynchronize(h->stream) != hipSuccess) {
                hipEventDestroy(e0); hipEventDestroy(e1);
                return fail(GH_ERR_HIP, "gh_profile_overhead: hipStreamSynchronize failed");
            }
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) (pass ? b : a).push_back(ms);
        }
    hipEventDestroy(e0); hipEventDestroy(e1);
    auto med = [](std::vector<float> &v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; };
    out[0] = med(a);
    out[1] = med(b);
    return GH_OK;
}

extern "C" int gh_debug_pool_geometry(int32_t n_snps, int32_t L, int32_t five, int64_t out[6])
{
    if (!out || n_snps < 1 || L < CW_MIN_L || L > CW_MAX_LG) return fail(GH_ERR_ARG, "pool geometry: n_snps >= 1, %d <= L <= %d", CW_MIN_L, CW_MAX_LG);
    const int R = five ? 5 : 4;
    const cw_geom g = cw_geometry(n_snps, L);
    const bool packed = five ? L <= CW_MAX_L5 : L <= CW_MAX_L;
    out[0] = g.S; out[1] = g.seglen;
    out[2] = packed ? cw_chunk(L, R) : cwg_chunk(L, R);
    out[3] = (int64_t)(packed ? cw_lds_bytes(L, R) : cwg_lds_bytes(L, R));
    out[4] = packed ? 1 : 0;
    out[5] = CW_K * cw_lanes(R);
    return GH_OK;
}

extern "C" int gh_debug_walk_clock(gh_t *h, uint64_t out[4])
{
    if (!h || !out) return fail(GH_ERR_ARG, "null argument");
    if (set_dev(h)) return GH_ERR_HIP;
    dev_state hs;
    HIPCHK(hipMemcpyAsync(&hs, h->dstate, sizeof hs, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    out[0] = hs.dbg[0]; out[1] = hs.dbg[1]; out[2] = hs.dbg[2]; out[3] = hs.dbg[3];
    if (out[3] == 3) {
        // out[1]: the state space the extension walked -- 4 ranks, 5 symbols, 6 mixed radix (seg_class, from the same control words);
        // out[2]: the most states entering a target as k_classify found them (0: not taken)
        out[0] = (uint64_t)h->spin_requeues;
        out[1] = hs.ranked ? 4u : ((h->L == SEGM_L && hs.maxstates > 0 && hs.maxstates <= SEGM_NS) ? (uint64_t)SEG_CLS_MIXED : 5u);
        out[2] = (uint64_t)hs.maxstates;
    }
    if (out[3] == 4) { out[0] = (uint64_t)h->spin_requeues; out[1] = (uint64_t)h->cw_stat[1]; out[2] = (uint64_t)h->cw_stat[2]; }
    if (getenv("GH_PRINT_STAMPS"))      // diagnostic builds (-DSEG_STAMPS / -DGH_STAMPS)
    {
        fprintf(stderr, "stamps:");
        for (int q = 1; q < 12 && hs.dbg8[q] >= hs.dbg8[q - 1] && hs.dbg8[q - 1]; q++) fprintf(stderr, " %llu", hs.dbg8[q] - hs.dbg8[q - 1]);
        fprintf(stderr, "\n");
    }
    return GH_OK;
}

// diagnostic builds (-DRWS_STAMPS_ALL): the stamps every k_rwseg workgroup of the last launch left, 16 doubles per workgroup
extern "C" int gh_debug_segment_stamps(gh_t *h, double *out, int n_workgroups)
{
    if (!h || !out || n_workgroups < 1) return fail(GH_ERR_ARG, "bad argument");
    if (set_dev(h)) return GH_ERR_HIP;
#if !defined(RWS_STAMPS_ALL)
    return fail(GH_ERR_STATE, "gh_debug_segment_stamps: this library was not built with -DRWS_STAMPS_ALL (there are no stamps to read)");
#else
    if (!h->seg_smin) return fail(GH_ERR_STATE, "no segment-parallel walk has run on this handle");
    if ((size_t)n_workgroups * 16 * sizeof(double) > h->seg_smin_bytes)
        return fail(GH_ERR_ARG, "gh_debug_segment_stamps: %d workgroups asked for, the buffer holds %zu", n_workgroups, h->seg_smin_bytes / (16 * sizeof(double)));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->seg_smin, (size_t)n_workgroups * 16 * sizeof(double), hipMemcpyDeviceToHost));
    return GH_OK;
#endif
}

extern "C" int gh_profile_bytes(gh_t *h, int kernel, double *bytes_per_launch)
{
    if (!h || !bytes_per_launch || kernel < 0 || kernel >= GH_K_COUNT) return fail(GH_ERR_ARG, "bad argument");
    *bytes_per_launch = h->ps[kernel].bytes;
    return GH_OK;
}

#include <type_traits>
#include <cstddef>
// kernels.hpp -- the gfx950 kernels of the Gretel hot path (included by gretel_hip.hip).
//
// Symbol indices (reference order, gretel/util.py:83):  A0 C1 G2 T3 N4 -5 _6.
// "Compact" indices used by the path-extension tables:
//     to-symbols   (candidates, the 5 valid symbols)      b5: A0 C1 G2 T3 -4
//     from-symbols (what a path can contain)              a6: A0 C1 G2 T3 -4 _5
//
// Tables rebuilt from the band before every path:
//   cnt[p][8], marg[p][8]   f64   c_s(p) (s<7), [7]=total ; c_s/total          (lookup API)
//   nvalid[p] i32, cmask[p] u32   V(p), candidate bitmask over the 7 symbols     (lookup API)
//   minfo[p][16] f64              [0..4] log10 marginal of b5, [5..9] marginal of b5,
//                                 [10] candidate bitmask over b5 (as u64 bits),
//                                 [11..15] log10 ORIGINAL marginal of b5 (written by the snapshot)
//   rinfo[p][8] f64               [0..3] log10 marginal, [4..7] marginal of the r-th candidate of p (candidates in compact order;
//                                 a rank that does not exist: 0.0 / +inf) -- what the ranked tables' readers add / track per pick
//   G[i][a6][l-1][b5] f64         source-major conditional table, see k_lt
//
// The compact order IS the order the candidates are offered in (gh_config.cand_order, default A C G T -): "first wins" over
// compact indices -- columns of G, candidate ranks -- is the tie-break of gretel/gretel.py:166-174 for that order.  Kernels
// get the two maps as a `symmap` argument.
// cmask[p] holds two masks over the SYMBOLS: bits 0..6 the valid symbols seen at p (c_s > 0: V(p) counts them), bits 8..14 the
// candidates get_edge_weights_at offers there (the same, or every valid symbol with gh_config.offer_zero).
#pragma once
#include "seg_geom.hpp"

#define NSYM 7
#define CELL 49
#define SYM_N 4
#define SYM_US 6
#define VALID_MASK 0x2Fu /* A C G T - : bits 0,1,2,3,5 */
#define LT_ROW 5
#define LT_BLK 30        /* 6 from-symbols x 5 to-symbols */
#define MINFO 16
#define RINFO 8
#define CM_SEEN(w) ((w) & 0x7Fu)
#define CM_CAND(w) (((w) >> 8) & 0x7Fu)

struct symmap {
    uint32_t fwd;   // nibble b5 (0..4): the symbol with that compact index
    uint32_t inv;   // nibble s (symbol 0..6): its compact from-index a6 (valid symbols 0..4, '_' 5, N 7 = none)
};
__host__ __device__ inline symmap make_symmap(const uint8_t order[5])
{
    symmap m;
    m.fwd = 0;
    m.inv = (7u << (4 * SYM_N)) | (5u << (4 * SYM_US));
    for (int b5 = 0; b5 < 5; b5++) {
        m.fwd |= (uint32_t)order[b5] << (4 * b5);
        m.inv |= (uint32_t)b5 << (4 * order[b5]);
    }
    return m;
}

// The first 64 bytes are the control words every kernel of a path reads: one line, one (scalar) load -- each separate
// dependent load of a word the previous kernel wrote costs a kernel about a microsecond before it can start.
struct dev_state {
    int stop;        // set at a hole: later launches of the spin become no-ops
    int hole_at;
    int n_done;
    int maxstates;   // k_classify: the most states that enter any target as a mixed-radix number (product of the candidate counts of L consecutive positions); 0 = not taken / mixed radix not allowed
    int first_hole;  // smallest snp in [1,N] without a candidate (k_marg), else INT_MAX-ish   } contiguous:
    int nodel;       // stays non-zero while no position offers the LAST symbol of the candidate order ('-' by default): the depth-1 walker then works on four lanes per group (k_marg)  } re-armed
    int cm_same;     // stays non-zero while k_marg finds every candidate mask equal to the previous one  } with one
    int narrow;      // stays non-zero while every position has at most 4 candidates (k_marg)      } memset
    int ranked;      // layout of G as k_lt last built it: 1 = rows/columns are candidate RANKS (see k_lt), 0 = symbols
    int cur_hole;    // segment-parallel walk (segwalk.hpp): first_hole as k_seg found it (k_scan re-arms the flag itself)
    int lt_stale;    // set by k_seg when a candidate mask moved under the last reweight and no k_lt ran since (gh_spin)
    int cw_unres;    // candidate-pool walk (cwalk.hpp): 1 = the queued rounds did not close the chain, 2 = table not ranked
    double ratio;    // clamped min marginal of the path just walked
    int cw_open_at;  // k_cscan: -1 = chain closed, else the segment whose entry state is still to be walked
    int cw_need;     // k_cscan: most rounds a path needed to close its chain since the host last cleared this (sizes the rounds queued per path)
    unsigned long long fill[6];   // slices, crumbs, covered, bad_symbol, out_of_band, -
    unsigned long long dbg[4];   // walker wave: s_memtime / s_memrealtime at start and end (diagnostics)
    unsigned long long dbg8[12]; // -DGH_STAMPS / -DSEG_STAMPS / -DRWS_STAMPS builds: cycles per segment of the code
    int pipe_status;             // k_wpipe (wpipe.hpp): PIPE_DONE / PIPE_ABORTED / PIPE_NOT_STARTED, written by every window of a launch
    int pipe_pad;
};

struct dev_ctl {
    int stop, hole_at, n_done, maxstates, first_hole, nodel, cm_same, narrow, ranked, cur_hole, lt_stale, cw_unres;
    double ratio;
    int cw_open_at, cw_need;
};
static_assert(sizeof(dev_ctl) == 64 && offsetof(dev_state, fill) == 64, "control words = the first line of dev_state");

// all control words at once (call before the kernel's first store)
__device__ __forceinline__ dev_ctl load_ctl(const dev_state *st) { return *reinterpret_cast<const dev_ctl *>(st); }

// which state space the segment-parallel extension walks (segwalk.hpp, segmix.hpp): 4 = candidate ranks of the ranked table
// layout (every position offers at most four), 6 = mixed radix over the five-symbol layout (few positions offer five: k_classify
// found every target within SEGM_NS states), 5 = all 5^L symbol histories
__device__ __forceinline__ int seg_class(const dev_ctl &c, int L)
{
    if (c.ranked != 0) return 4;
    return (L == SEGM_L && c.maxstates > 0 && c.maxstates <= SEGM_NS) ? SEG_CLS_MIXED : 5;
}

// Batched launches (gh_batch_*): one entry per window; a kernel launched with `wd != nullptr` takes its
// window from blockIdx.y (blockIdx.x for the walkers) and its buffers from wd[window].
struct win_desc {
    void *band;
    double *cnt, *marg, *minfo;
    int32_t *nvalid;
    uint32_t *cmask;
    double *rinfo;
    double *G;
    double *Ht, *Yt;       // depth-2 walker tables (k_lt)
    dev_state *st;
    double *partial;
    uint8_t *paths;        // [max_paths][N+1]
    gh_path_rec *recs;     // [max_paths]
    int snap;              // batched k_snapshot: freeze this window's marginals as the original ones
    int _pad;
    unsigned long long *pk; // [N+2] k_wpipe (wpipe.hpp): what a sweep needs of a position's candidates, packed by its prologue
    double *gp;             // (N+LT_PAD) sources x L x 128 bytes (pipe_gp_piece) k_wpipe: its own compact copy of the ranked table (prologue)
    double *lmr;            // [N+2][4] k_wpipe with the marginal term: log10 marginal of a position's candidates by rank
    void *tband;            // k_wpipe under the column conditionals: the to-major copy of the band (kept in step)
    double *gw;             // k_wpipe<.., WIDE>: the side table of the window's five-candidate positions (wpipe.hpp: PIPE_WMAX records)
    int *wdir;              // ... and, per chunk, the records the chunk can see: first | count << 16
};

// Band layout: band[i][a][d-1][b] -- position, FROM-symbol, distance, to-symbol.  Everything a path touches at position i
// (the elements (path[i], path[i+d]) for d = 1..W, and the rows the conditional table is rebuilt from) sits in ONE run of
// W x 7 values, a few 32-byte sectors, instead of one sector per distance in a distance-major band.
__host__ __device__ __forceinline__ size_t bidx(int W, size_t i, int d, int a, int b)
{
    return ((i * NSYM + (size_t)a) * (size_t)W + (size_t)(d - 1)) * NSYM + (size_t)b;
}

__constant__ int8_t c_sym_of_char[256];

// segwalk.hpp (segment-parallel path extension); k_marg<T,true> closes the path record behind it
__device__ __forceinline__ void seg_finish(dev_state *st, gh_path_rec *rec, int N, double minm, double min_remove);

__device__ __forceinline__ int vsym(symmap m, int b5) { return (int)((m.fwd >> (4 * b5)) & 7u); }                 // b5 -> symbol
__device__ __forceinline__ int fsym(symmap m, int a6) { return a6 < 5 ? vsym(m, a6) : SYM_US; }                        // a6 -> symbol
__device__ __forceinline__ int a6_of_sym(symmap m, int s) { return (int)((m.inv >> (4 * s)) & 7u); }                  // symbol -> a6 (N: 7)
// candidate masks: one bit per SYMBOL in cmask, cm5 one bit per compact index b5
__device__ __forceinline__ uint32_t cm5_of_cmask(symmap m, uint32_t cm)
{
    uint32_t r = 0;
#pragma unroll
    for (int b5 = 0; b5 < 5; b5++) r |= ((cm >> vsym(m, b5)) & 1u) << b5;
    return r;
}
// index of the r-th set bit of a 5-bit mask (r = 0 is the lowest), -1 if there are fewer
__device__ __forceinline__ int nth_set5(uint32_t m, int r)
{
    int out = -1;
#pragma unroll
    for (int b = 0; b < 5; b++) {
        const bool set = (m >> b) & 1u;
        if (set && r == 0) out = b;
        r -= set ? 1 : 0;
    }
    return out;
}

template <typename T>
__device__ __forceinline__ double rowsum(const T *band, int W, int i, int d, int a)
{
    const T *row = band + bidx(W, i, d, a, 0);
    T acc = (T)0;
#pragma unroll
    for (int x = 0; x < NSYM; x++) acc = acc + row[x];
    return (double)acc;
}

template <typename T>
__device__ __forceinline__ double colsum(const T *band, int W, int i, int d, int b)
{
    T acc = (T)0;
#pragma unroll
    for (int x = 0; x < NSYM; x++) acc = acc + band[bidx(W, i, d, x, b)];
    return (double)acc;
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// log10's table (include/gh_detlog.h: 128 x { 1/c, log c }, 2 KB) in LDS: the kernels that take logarithms between two
// dependent memory round trips (k_marg<T,true>, k_rw, k_rwseg) read it there -- from global memory the lookup is one more
// round trip in the chain (k_rwseg +0.8 us, the batched k_marg<T,true> +0.2 ms per launch, measured).  A barrier between
// logtab_stage() and the first logarithm.
__device__ __forceinline__ void logtab_stage(double *t)
{
    for (int q = threadIdx.x; q < 256; q += blockDim.x) t[q] = gh_logtab_dev[q];
}

// hansel conditional, shared by the table builder and the per-call lookup
template <typename T>
__device__ __forceinline__ double log_conditional(const T *__restrict__ band, int W, int cond_mode,
                                                  const double *__restrict__ cnt,
                                                  const int32_t *__restrict__ nvalid, int a, int b, int i, int j,
                                                  const T *__restrict__ tband = nullptr)
{
    // (tband: the to-major copy of the band where the host keeps one -- a column sum is then a contiguous run, same addends in
    // the same order)
    const int l = j - i;
    double obs = 0.0, sum = 0.0;
    if (l <= W) {
        obs = (double)band[bidx(W, i, l, a, b)];
        if (cond_mode == GH_COND_A || cond_mode == GH_COND_D) sum = rowsum(band, W, i, l, a);
        else if (cond_mode == GH_COND_C || cond_mode == GH_COND_E) sum = tband ? rowsum(tband, W, i, l, b) : colsum(band, W, i, l, b);
    }
    double den;
    if (cond_mode == GH_COND_A || cond_mode == GH_COND_E) den = (double)nvalid[j] + sum;
    else if (cond_mode == GH_COND_B) den = (double)nvalid[i] + cnt[(size_t)i * 8 + a];
    else den = (double)nvalid[i] + sum;          // C, D
    return gh_log10((1.0 + obs) / den);
}

// ---------------------------------------------------------------------------------------------
// k_fill: gretel/util.py:226-286, one thread per read, float atomics (+1 is exact and
// order-independent up to 2^24, where float32 += 1 saturates exactly like NumPy's).
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void add_obs(T *band, int N, int W, int a, int b, int i, int j,
                                        unsigned long long *oob)
{
    int d = j - i;
    if (d < 1 || d > W || i < 0 || j > N + 1) {
        atomicAdd(oob, 1ULL);
        return;
    }
    atomicAdd(&band[bidx(W, i, d, a, b)], (T)1);
}

template <typename T>
__global__ void __launch_bounds__(256)
k_fill(T *__restrict__ band, int N, int W, const int32_t *__restrict__ rank,
       const int64_t *__restrict__ off, const uint8_t *__restrict__ bases, int64_t n_reads,
       int use_end_sentinels, dev_state *st)
{
    __shared__ unsigned long long s_acc[3];
    if (threadIdx.x < 3) s_acc[threadIdx.x] = 0;
    __syncthreads();

    unsigned long long slices = 0, crumbs = 0, covered = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads;
         r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o0 = off[r];
        const int k = (int)(off[r + 1] - o0);
        if (!(k > 1)) continue;                                  // util.py:230
        const int rk = rank[r];
        const uint8_t *s = bases + o0;
        slices++;                                                // util.py:233
        bool bad = false;
        for (int i = 0; i < k; i++) {
            int c = s[i];
            if (c_sym_of_char[c] < 0) bad = true;
            if (c != 'N' && c != '_') covered++;                 // util.py:239
        }
        if (bad) { atomicAdd(&st->fill[3], 1ULL); continue; }
        for (int i = 0; i < k; i++) {
            const int a = c_sym_of_char[s[i]];
            if (a == SYM_US || a == SYM_N) continue;             // util.py:258
            for (int j = i + 1; j < k; j++) {
                const int b = c_sym_of_char[s[j]];
                if (i == 0 && j == 1 && rk == 0) {               // util.py:262
                    add_obs(band, N, W, SYM_US, a, 0, 1, &st->fill[4]);
                    add_obs(band, N, W, a, b, 1, 2, &st->fill[4]);
                } else if ((j + rk + 1) == N && (j - i) == 1) {  // util.py:271
                    add_obs(band, N, W, a, b, N - 1, N, &st->fill[4]);
                    add_obs(band, N, W, b, SYM_US, N, N + 1, &st->fill[4]);
                } else {                                         // util.py:279
                    add_obs(band, N, W, a, b, i + rk + 1, j + rk + 1, &st->fill[4]);
                    if (use_end_sentinels && j == k - 1 && (j - i) == 1)      // util.py:283
                        add_obs(band, N, W, b, SYM_US, j + rk + 1, j + rk + 2, &st->fill[4]);
                }
                crumbs++;
            }
        }
    }
    atomicAdd(&s_acc[0], slices);
    atomicAdd(&s_acc[1], crumbs);
    atomicAdd(&s_acc[2], covered);
    __syncthreads();
    if (threadIdx.x < 3 && s_acc[threadIdx.x]) atomicAdd(&st->fill[threadIdx.x], s_acc[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// k_fill_pairs: the same pair loop with 32 lanes per read -- lane g takes the pairs (i, i + g + 1), i ascending.  For long reads
// (tens of SNPs each, a wide band: C5) the tensor is sparse -- fewer observations than cells, nothing to privatise in LDS -- and
// one thread per read sends the 64 atomics of a wavefront to 64 positions 196 W bytes apart.  Here the atomics of one step go to
// the cells (i, i + d), d = 1 .. k - 1, of ONE from-position and from-symbol: W x 7 consecutive elements of the band.
// Reads of at most 32 SNPs (the host checks).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_fill_pairs(T *__restrict__ band, int N, int W, const int32_t *__restrict__ rank,
             const int64_t *__restrict__ off, const uint8_t *__restrict__ bases, int64_t n_reads,
             int use_end_sentinels, dev_state *st)
{
    __shared__ unsigned long long s_acc[3];
    if (threadIdx.x < 3) s_acc[threadIdx.x] = 0;
    __syncthreads();
    const int g = threadIdx.x & 31;
    const bool upper = (threadIdx.x & 32) != 0;
    const int64_t grp0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5, ngrp = ((int64_t)gridDim.x * blockDim.x) >> 5;
    unsigned long long slices = 0, crumbs = 0, covered = 0;
    for (int64_t r = grp0; r < n_reads; r += ngrp) {
        const int64_t o0 = off[r];
        const int k = (int)(off[r + 1] - o0);
        if (!(k > 1)) continue;                                  // util.py:230 (the same for the 32 lanes of the read)
        const int rk = rank[r];
        const int c = g < k ? bases[o0 + g] : 'N';
        const int sym = c_sym_of_char[c];
        const unsigned long long bad64 = __ballot(g < k && sym < 0), cov64 = __ballot(g < k && c != 'N' && c != '_');
        const unsigned bad = upper ? (unsigned)(bad64 >> 32) : (unsigned)bad64, cov = upper ? (unsigned)(cov64 >> 32) : (unsigned)cov64;
        if (g == 0) { slices++; covered += __popc(cov); }       // util.py:233, 239
        if (bad) { if (g == 0) atomicAdd(&st->fill[3], 1ULL); continue; }
        for (int i = 0; i < k - 1; i++) {
            const int a = __shfl(sym, i, 32);
            if (a == SYM_US || a == SYM_N) continue;             // util.py:258
            const int j = i + g + 1;
            const int b = __shfl(sym, j < k ? j : 0, 32);
            if (j >= k) continue;
            if (i == 0 && j == 1 && rk == 0) {                   // util.py:262
                add_obs(band, N, W, SYM_US, a, 0, 1, &st->fill[4]);
                add_obs(band, N, W, a, b, 1, 2, &st->fill[4]);
            } else if ((j + rk + 1) == N && (j - i) == 1) {      // util.py:271
                add_obs(band, N, W, a, b, N - 1, N, &st->fill[4]);
                add_obs(band, N, W, b, SYM_US, N, N + 1, &st->fill[4]);
            } else {                                             // util.py:279
                add_obs(band, N, W, a, b, i + rk + 1, j + rk + 1, &st->fill[4]);
                if (use_end_sentinels && j == k - 1 && (j - i) == 1)          // util.py:283
                    add_obs(band, N, W, b, SYM_US, j + rk + 1, j + rk + 2, &st->fill[4]);
            }
            crumbs++;
        }
    }
    if (slices) atomicAdd(&s_acc[0], slices);
    if (crumbs) atomicAdd(&s_acc[1], crumbs);
    if (covered) atomicAdd(&s_acc[2], covered);
    __syncthreads();
    if (threadIdx.x < 3 && s_acc[threadIdx.x]) atomicAdd(&st->fill[threadIdx.x], s_acc[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// k_fill_own: the pair loop for a rank-sorted table with the tensor cut by OWNER.  Workgroup g owns the from-positions
// [g P, g P + P): it visits every read that can observe something from one of them (ranks in [g P - max_k, g P + P), found by
// bisection of the rank column), counts what falls into its slice with integer LDS atomics (2-byte counters where no
// workgroup sees 65 536 reads, 4-byte otherwise) and flushes the slice ONCE with plain read-modify-writes of consecutive
// cells -- nobody else writes them.  No global atomic at all: k_fill_pairs sends one float atomic per observation into a
// tensor that is mostly zeros (C5: 10 M atomics, 0.35 ms); here the cost is what the slice's bytes cost to read and write
// (C5: 2 x 206 MB), whatever the number of observations, and a read is visited (P + max_k) / P times.  The read counters
// (slices, covered SNPs, crumbs, bad symbols, out-of-band pairs) are taken by the workgroup that owns the read's rank.
// Same integers as the other fills; cells whose count is zero are not touched.
// ---------------------------------------------------------------------------------------------
#define FILL_OWN_SYMS (16 * 1024)      /* bytes of LDS for the bases of a chunk of reads */
template <typename T, typename CT, bool ZERO, int G>
__global__ void __launch_bounds__(1024)
k_fill_own(T *__restrict__ band, int N, int W, const int32_t *__restrict__ rank, const int64_t *__restrict__ off,
           const uint8_t *__restrict__ bases, int64_t n_reads, int P, int max_k, int use_end_sentinels, dev_state *st,
           const int64_t *__restrict__ first_at, int n_first, int sym_off)
{
    extern __shared__ unsigned own_slice[];          // [n_pos][7][W][7] counters of CT, the band's layout; at sym_off bytes: the chunk's bases
    __shared__ unsigned long long s_acc[5];
    __shared__ long long s_range[2];
    constexpr bool HALF = sizeof(CT) == 2;
    const int s0 = blockIdx.x * P;
    int n_pos = N + 2 - s0;
    if (n_pos > P) n_pos = P;
    if (n_pos < 1) return;
    const int n_cells = n_pos * W * CELL;
    const int n_words = HALF ? (n_cells + 1) / 2 : n_cells;
    {
        uint4 *z4 = reinterpret_cast<uint4 *>(own_slice);          // (the allocation is rounded up to 16 bytes)
        for (int q = threadIdx.x; q < (n_words + 3) / 4; q += blockDim.x) z4[q] = make_uint4(0, 0, 0, 0);
    }
    if (threadIdx.x < 5) s_acc[threadIdx.x] = 0;
    if (threadIdx.x < 2) {
        // first read with rank >= s0 - max_k (lane 0) / first read with rank > s0 + P - 1 (lane 1): the host's table of where
        // every rank starts (a bisection of the rank column is 20 dependent loads the whole workgroup waits for)
        long long want = threadIdx.x == 0 ? (long long)s0 - max_k : (long long)s0 + P;
        if (want < 0) want = 0;
        if (want > n_first - 1) want = n_first - 1;
        s_range[threadIdx.x] = first_at[want];
    }
    __syncthreads();
    auto count = [&](int a, int b, int i, int j, bool mine, unsigned long long &oob) {
        const int d = j - i;
        if (d < 1 || d > W || i < 0 || j > N + 1) { if (mine) oob++; return; }
        const int li = i - s0;
        if (li < 0 || li >= n_pos) return;
        const unsigned c = (unsigned)bidx(W, (size_t)li, d, a, b);
        if (HALF) atomicAdd(&own_slice[c >> 1], 1u << (16 * (c & 1)));
        else atomicAdd(&own_slice[c], 1u);
    };
    unsigned long long slices = 0, crumbs = 0, covered = 0, badr = 0, oob = 0;
    // The workgroup's reads in chunks whose bases fit FILL_OWN_SYMS bytes of LDS: the bases go there first, as symbol codes
    // (one coalesced pass over a contiguous run of the table; 255 = not a symbol) -- a thread that fetches the two bases of each
    // of its read's k (k - 1) / 2 pairs from memory is a chain of several hundred dependent loads (C5: 0.3 ms of 0.3).
    // Within a chunk the lanes of a wavefront take reads far apart (row x of a 64-row matrix over the chunk's reads):
    // neighbours in a rank-sorted table start at the same positions and carry the same haplotypes -- their LDS atomics
    // would hit the same counters.
    unsigned char *syms = reinterpret_cast<unsigned char *>(own_slice) + sym_off;
    const long long chunk_reads = FILL_OWN_SYMS / (max_k > 0 ? max_k : 1);
    for (long long c_lo = s_range[0]; c_lo < s_range[1]; c_lo += chunk_reads) {
        const long long c_hi = c_lo + chunk_reads < s_range[1] ? c_lo + chunk_reads : s_range[1];
        const int64_t b_lo = off[c_lo], b_hi = off[c_hi];
        __syncthreads();                                         // (the chunk before is done with syms)
        for (int64_t q = threadIdx.x; q < b_hi - b_lo; q += blockDim.x) {
            const int sc = c_sym_of_char[bases[b_lo + q]];
            syms[q] = (unsigned char)(sc < 0 ? 255 : sc);
        }
        __syncthreads();
        // G lanes per read (1 for short reads, up to 8 for long ones): lane g takes the from-indices i = g, g + G, ... and for
        // each the pairs (i, j > i).  Every pair adds H[a, b, i + rk + 1, j + rk + 1] (util.py:279-280 -- and that IS the second
        // observation of the rank-0 case, util.py:267, and the first of the last-SNP case, util.py:274); only ADJACENT pairs
        // add a second one: (_, a, 0, 1) for the first pair of a read at rank 0 (util.py:262-266), (b, _, N, N + 1) where the
        // pair ends on the last SNP (util.py:271-275), the end sentinel (b, _, j + rk + 1, j + rk + 2) behind a read's last
        // pair when asked for (util.py:283-286) -- in that order of precedence.  A from-position outside the slice needs no
        // loop over j at all.
        const long long n_mine = c_hi - c_lo;
        const long long cols = (n_mine + 63) / 64;
#if defined(FILL_OWN_DBG) && FILL_OWN_DBG == 1
        if (n_mine >= 0) continue;
#endif
        for (long long xg = threadIdx.x; xg < cols * 64 * G; xg += blockDim.x) {
            const int g = (int)(xg % G);
            const long long x = xg / G;
            const long long rr = (x & 63) * cols + (x >> 6);
            if (rr >= n_mine) continue;
            const long long r = c_lo + rr;
            const int64_t o0 = off[r];
            const int k = (int)(off[r + 1] - o0);
            if (!(k > 1)) continue;                                  // util.py:230
            const int rk = rank[r];
            const bool mine = rk >= s0 && rk < s0 + P && g == 0;     // this lane counts the read itself
            const unsigned char *s = syms + (o0 - b_lo);
            bool bad = false;
            unsigned cov = 0;
            for (int i = 0; i < k; i++) {
                const int c = s[i];
                if (c == 255) bad = true;
                if (c != SYM_N && c != SYM_US) cov++;                // util.py:239
            }
            if (mine) { slices++; covered += cov; }                  // util.py:233
            if (bad) { if (mine) badr++; continue; }
            const bool own_read = rk >= s0 && rk < s0 + P;           // (crumbs and out-of-band pairs: by the workgroup that owns the read, any lane)
            for (int i = g; i < k - 1; i += G) {
                const int a = s[i];
                if (a == SYM_US || a == SYM_N) continue;             // util.py:258
                if (own_read) crumbs += (unsigned)(k - 1 - i);
                const int pi = i + rk + 1;                           // the from-position of every pair of this i
                // the adjacent pair's second observation
                {
                    const int j = i + 1, b = s[j];
                    if (i == 0 && rk == 0) count(SYM_US, a, 0, 1, own_read, oob);
                    else if (j + rk + 1 == N) count(b, SYM_US, N, N + 1, own_read, oob);
                    else if (use_end_sentinels && j == k - 1) count(b, SYM_US, j + rk + 1, j + rk + 2, own_read, oob);
                }
                const int li = pi - s0;
                const int jmax = k - 1 < i + W ? k - 1 : i + W;      // pairs further apart than the band: out of band
                if (own_read && k - 1 > i + W) oob += (unsigned)(k - 1 - (i + W));
                if (li < 0 || li >= n_pos) continue;
                const unsigned base = (unsigned)__mul24(__mul24(li, NSYM) + a, W) * NSYM;        // bidx(W, li, 1, a, 0)
                for (int j = i + 1; j <= jmax; j++) {
                    if (j + rk + 1 > N + 1) { if (own_read) oob++; continue; }
                    const unsigned c = base + (unsigned)(j - i - 1) * NSYM + (unsigned)s[j];
                    if (HALF) atomicAdd(&own_slice[c >> 1], 1u << (16 * (c & 1)));
                    else atomicAdd(&own_slice[c], 1u);
                }
            }
        }
    }
    {
        // the five counters: summed over the wavefront first (a thousand threads adding to five LDS words one after the other
        // cost more than the pair loop of a short-read window)
        unsigned long long v5[5] = {slices, crumbs, covered, badr, oob};
#pragma unroll
        for (int q = 0; q < 5; q++) {
            unsigned lo = (unsigned)v5[q], hi = (unsigned)(v5[q] >> 32);
            unsigned long long t = v5[q];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo = (unsigned)t; hi = (unsigned)(t >> 32);
                const unsigned lo2 = __shfl_xor(lo, o), hi2 = __shfl_xor(hi, o);
                t += ((unsigned long long)hi2 << 32) | lo2;
            }
            if ((threadIdx.x & 63) == 0 && t) atomicAdd(&s_acc[q], t);
        }
    }
    __syncthreads();
    if (threadIdx.x < 5 && s_acc[threadIdx.x]) atomicAdd(&st->fill[threadIdx.x], s_acc[threadIdx.x]);
    // flush: consecutive lanes -> consecutive cells, each owned by this workgroup alone; four cells per lane and trip (16 or 32
    // bytes), eight trips' loads in flight before the first addition -- one cell per trip behind an `if` was a chain of
    // dependent round trips (C5: 0.6 ms).  ZERO: the tensor is known to hold zeros (gh_clear / gh_create): stores only.
    T *dst = band + (size_t)s0 * W * CELL;
#if defined(FILL_OWN_DBG) && FILL_OWN_DBG == 2
    if (n_cells > 0) return;
#endif
    auto cnt_at = [&](int q) -> unsigned { return HALF ? (own_slice[q >> 1] >> (16 * (q & 1))) & 0xffffu : own_slice[q]; };
    typedef T vec4 __attribute__((ext_vector_type(4)));
    const int n_quads = n_cells / 4;                              // (n_cells = n_pos * W * 49: the tail below takes the rest)
    constexpr int UN = 8;
    for (int q0 = threadIdx.x; q0 < n_quads; q0 += blockDim.x * UN) {
        vec4 v[UN];
        unsigned c[UN][4];
        bool any[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int q = q0 + u * blockDim.x;
            any[u] = false;
            if (q < n_quads) {
#pragma unroll
                for (int e = 0; e < 4; e++) { c[u][e] = cnt_at(4 * q + e); any[u] |= c[u][e] != 0; }
                if (!ZERO && any[u]) v[u] = *reinterpret_cast<const vec4 *>(dst + 4 * (size_t)q);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int q = q0 + u * blockDim.x;
            if (q < n_quads && any[u]) {
                vec4 w;
#pragma unroll
                for (int e = 0; e < 4; e++) w[e] = ZERO ? (T)c[u][e] : v[u][e] + (T)c[u][e];
                *reinterpret_cast<vec4 *>(dst + 4 * (size_t)q) = w;
            }
        }
    }
    for (int q = 4 * n_quads + threadIdx.x; q < n_cells; q += blockDim.x) {
        const unsigned c1 = cnt_at(q);
        if (c1) dst[q] = ZERO ? (T)c1 : dst[q] + (T)c1;
    }
}

// ---------------------------------------------------------------------------------------------
// What gh_reads_upload has to know about a support table before a fill can be chosen, found on the device behind the upload (three
// passes of one host thread over a million reads were 1.5 ms of every upload): the longest read, whether off[] ascends (the first
// read where it does not), whether the ranks ascend; and for a table whose ranks ascend -- launched on that assumption, thrown
// away when it does not hold -- first_at[p] = the first read whose rank is >= p, p in [0, top) with top = rank[n - 1] + 2.
struct reads_meta {
    int max_k;                      // longest read
    int unsorted;                   // a rank below the one in front of it
    long long bad_off;              // the first read whose off[] runs backwards (LLONG_MAX: none)
    int span;                       // most positions between the first and the last rank of a block of `rpb` reads
    long long dens128;              // most reads whose ranks fall into 128 consecutive positions
};
__global__ void __launch_bounds__(256)
k_reads_meta(const int32_t *__restrict__ rank, const int64_t *__restrict__ off, int64_t n, reads_meta *m)
{
    int mk = 0;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = off[q + 1] - off[q];
        if (k < 0) atomicMin((unsigned long long *)&m->bad_off, (unsigned long long)q);
        else if (k > mk) mk = (int)(k > 0x7fffffff ? 0x7fffffff : k);
        if (q > 0 && rank[q] < rank[q - 1]) m->unsorted = 1;
    }
    for (int o = 32; o; o >>= 1) { const int v = __shfl_xor(mk, o); mk = v > mk ? v : mk; }
    if ((threadIdx.x & 63) == 0 && mk > 0) atomicMax(&m->max_k, mk);
}
// (behind k_reads_meta; nothing for a table whose ranks do not ascend -- the jumps of an unsorted table would be written out at
// length.  Read q writes the entries (rank[q - 1], rank[q]]; q = n the last one, top - 1)
__global__ void __launch_bounds__(256)
k_reads_first_at(const int32_t *__restrict__ rank, int64_t n, int64_t *__restrict__ first_at, int64_t top, const reads_meta *m)
{
    if (m->unsorted) return;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q <= n; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t prev = q > 0 ? (int64_t)rank[q - 1] : -1, cur = q < n ? (int64_t)rank[q] : top - 1;
        for (int64_t pp = prev + 1 < 0 ? 0 : prev + 1; pp <= cur && pp < top; pp++) first_at[pp] = q;
    }
}
// (behind k_reads_first_at: the span; the density where there is a first_at)
__global__ void __launch_bounds__(256)
k_reads_meta2(const int32_t *__restrict__ rank, int64_t n, const int64_t *__restrict__ first_at, int rpb, reads_meta *m)
{
    if (m->unsorted) return;
    int sp = 0;
    long long dn = 0;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        if (q % rpb == 0) {
            const int64_t q1 = q + rpb < n ? q + rpb : n;
            const int d = rank[q1 - 1] - rank[q];
            sp = d > sp ? d : sp;
        }
        if (first_at) {
            const int lo_p = rank[q] - 127 < 0 ? 0 : rank[q] - 127;
            const long long d = (long long)q - (long long)first_at[lo_p] + 1;
            dn = d > dn ? d : dn;
        }
    }
    for (int o = 32; o; o >>= 1) {
        const int v = __shfl_xor(sp, o); sp = v > sp ? v : sp;
        const long long w = __shfl_xor(dn, o); dn = w > dn ? w : dn;
    }
    if ((threadIdx.x & 63) == 0) { atomicMax(&m->span, sp); atomicMax((unsigned long long *)&m->dens128, (unsigned long long)dn); }
}

// k_fill_sorted: the same pair loop for a support table sorted by rank (what a coordinate-sorted
// BAM gives).  A workgroup owns a contiguous run of reads, so all its observations fall into a
// narrow slice of positions: it counts them in LDS (integer adds) and flushes the slice once
// with coalesced float atomics -- ~5x fewer and well-shaped global atomics than k_fill's scattered
// ones (MI355X guide: 64 lanes in 64 rows run 17x below the contiguous atomic rate).
// Identical results: the counts are integers either way.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void add_obs_lds(unsigned *slice, int i_lo, int n_pos, T *band, int N, int W, int a, int b,
                                            int i, int j, unsigned long long *oob)
{
    const int d = j - i;
    if (d < 1 || d > W || i < 0 || j > N + 1) { atomicAdd(oob, 1ULL); return; }
    const int li = i - i_lo;
    if (li >= 0 && li < n_pos) atomicAdd(&slice[bidx(W, li, d, a, b)], 1u);     // (the slice has the band's layout)
    else atomicAdd(&band[bidx(W, i, d, a, b)], (T)1);     // sentinel cells far from the slice
}

template <typename T>
__global__ void __launch_bounds__(256)
k_fill_sorted(T *__restrict__ band, int N, int W, const int32_t *__restrict__ rank,
              const int64_t *__restrict__ off, const uint8_t *__restrict__ bases, int64_t n_reads,
              int reads_per_block, int max_pos, int max_k, int use_end_sentinels, dev_state *st)
{
    extern __shared__ unsigned slice[];          // [max_pos][W][49]
    __shared__ unsigned long long s_acc[3];
    const int64_t r0 = (int64_t)blockIdx.x * reads_per_block;
    int64_t r1 = r0 + reads_per_block;
    if (r1 > n_reads) r1 = n_reads;
    if (r0 >= r1) return;
    // pos_from of this run's regular observations: rank[r0]+1 .. rank[r1-1]+max_k (ranks ascend); what falls
    // outside the slice (a run wider than the LDS budget, the far sentinel cells) goes straight to global memory
    const int i_lo = rank[r0] + 1;
    int n_pos = rank[r1 - 1] + max_k + 1 - i_lo;
    if (n_pos > max_pos) n_pos = max_pos;
    if (n_pos < 1) n_pos = 1;
    const int n_slice = n_pos * W * CELL;
    for (int q = threadIdx.x; q < n_slice; q += blockDim.x) slice[q] = 0;
    if (threadIdx.x < 3) s_acc[threadIdx.x] = 0;
    __syncthreads();

    unsigned long long slices = 0, crumbs = 0, covered = 0;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const int64_t o0 = off[r];
        const int k = (int)(off[r + 1] - o0);
        if (!(k > 1)) continue;                                  // util.py:230
        const int rk = rank[r];
        const uint8_t *s = bases + o0;
        slices++;                                                // util.py:233
        bool bad = false;
        for (int i = 0; i < k; i++) {
            int c = s[i];
            if (c_sym_of_char[c] < 0) bad = true;
            if (c != 'N' && c != '_') covered++;                 // util.py:239
        }
        if (bad) { atomicAdd(&st->fill[3], 1ULL); continue; }
        for (int i = 0; i < k; i++) {
            const int a = c_sym_of_char[s[i]];
            if (a == SYM_US || a == SYM_N) continue;             // util.py:258
            for (int j = i + 1; j < k; j++) {
                const int b = c_sym_of_char[s[j]];
                if (i == 0 && j == 1 && rk == 0) {               // util.py:262
                    add_obs_lds(slice, i_lo, n_pos, band, N, W, SYM_US, a, 0, 1, &st->fill[4]);
                    add_obs_lds(slice, i_lo, n_pos, band, N, W, a, b, 1, 2, &st->fill[4]);
                } else if ((j + rk + 1) == N && (j - i) == 1) {  // util.py:271
                    add_obs_lds(slice, i_lo, n_pos, band, N, W, a, b, N - 1, N, &st->fill[4]);
                    add_obs_lds(slice, i_lo, n_pos, band, N, W, b, SYM_US, N, N + 1, &st->fill[4]);
                } else {                                         // util.py:279
                    add_obs_lds(slice, i_lo, n_pos, band, N, W, a, b, i + rk + 1, j + rk + 1, &st->fill[4]);
                    if (use_end_sentinels && j == k - 1 && (j - i) == 1)      // util.py:283
                        add_obs_lds(slice, i_lo, n_pos, band, N, W, b, SYM_US, j + rk + 1, j + rk + 2, &st->fill[4]);
                }
                crumbs++;
            }
        }
    }
    atomicAdd(&s_acc[0], slices);
    atomicAdd(&s_acc[1], crumbs);
    atomicAdd(&s_acc[2], covered);
    __syncthreads();
    if (threadIdx.x < 3 && s_acc[threadIdx.x]) atomicAdd(&st->fill[threadIdx.x], s_acc[threadIdx.x]);
    // flush: consecutive lanes -> consecutive cells of the band (contiguous atomics), zeros skipped
    T *dst = band + (size_t)i_lo * W * CELL;
    const size_t band_end = (size_t)(N + 2) * W * CELL;
    for (int q = threadIdx.x; q < n_slice; q += blockDim.x) {
        const unsigned c = slice[q];
        if (c && (size_t)i_lo * W * CELL + q < band_end) atomicAdd(&dst[q], (T)c);
    }
}

template <typename T>
__global__ void k_add_batch(T *__restrict__ band, int N, int W, const uint8_t *a, const uint8_t *b,
                            const int32_t *i, const int32_t *j, int64_t n, dev_state *st)
{
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    if (a[q] >= NSYM || b[q] >= NSYM) { atomicAdd(&st->fill[3], 1ULL); return; }
    add_obs(band, N, W, a[q], b[q], i[q], j[q], &st->fill[4]);
}

// ---------------------------------------------------------------------------------------------
// k_marg: counts / marginals / candidate masks for every position p in [0, N]
// (hansel get_counts_at + get_marginal_of_at, call sites gretel/cmd.py:86, gretel/gretel.py:182)
// ---------------------------------------------------------------------------------------------
// RW = true fuses gretel/gretel.py:79-98 in front: the 8-lane group of position p first reweights the
// cells (p, p+d), d = 1..W, on `rw_path` (the element-wise arithmetic and the multiplicities of gretel/gretel.py:79-98,
// removed mass into partial[blockIdx.x]) and then takes the marginals of the cell (p, p+1) it has just
// updated -- one pass over the band instead of two, one launch less per path.  With G != nullptr it also rewrites
// the rows of the conditional table (k_lt below) that the path's cells feed, while those cells are still in cache.
template <typename T, bool RW>
__global__ void __launch_bounds__(256)
k_marg(T *band, int N, int W, double *cnt, double *marg,
       int32_t *nvalid, uint32_t *cmask, double *minfo, dev_state *st, const win_desc *wd,
       const uint8_t *rw_path, double ratio_arg, int use_state_ratio, double *partial, int spin,
       double *G, int L, int cond_mode, const double *segmin, gh_path_rec *seg_rec, symmap sm, int offer_zero, double *rinfo)
{
    __shared__ double s_red[256];
    __shared__ double s_logtab[256];
    logtab_stage(s_logtab);
    bool live = true;
    double seg_ratio = 0.0;
    if (RW && segmin) {
        // behind a segment-parallel walk (segwalk.hpp): the path's minimum marginal is still spread over the segments.
        // Every workgroup reduces it for itself (<= 256 values, exact in any order); workgroup 0 closes the record the
        // way the serial walkers' bookkeeper does.  ratio_arg carries the clamp (cmd.py:157-160).
        const dev_ctl c = load_ctl(st);
        const int nseg = seg_geometry(N, L, seg_class(c, L)).S;
        s_red[threadIdx.x] = (int)threadIdx.x < nseg ? segmin[threadIdx.x] : INFINITY;
        __syncthreads();
        for (int q = 128; q > 0; q >>= 1) {
            if ((int)threadIdx.x < q && s_red[threadIdx.x + q] < s_red[threadIdx.x]) s_red[threadIdx.x] = s_red[threadIdx.x + q];
            __syncthreads();
        }
        const double minm = s_red[0];
        __syncthreads();
        seg_ratio = minm < ratio_arg ? ratio_arg : minm;
        const bool dead = c.stop != 0 || c.lt_stale != 0 || c.cur_hole <= N;
        if (blockIdx.x == 0 && threadIdx.x == 0 && !c.stop && !c.lt_stale) seg_finish(st, seg_rec, N, minm, ratio_arg);
        if (dead) live = false;
    }
    if (wd) {
        const win_desc &d = wd[blockIdx.y];
        band = (T *)d.band; cnt = d.cnt; marg = d.marg; nvalid = d.nvalid; cmask = d.cmask; minfo = d.minfo; st = d.st; rinfo = d.rinfo;
        if (RW) { rw_path = d.paths + (size_t)spin * (N + 1); partial = d.partial; if (G) G = d.G; }
        live = !st->stop;
    }
    if (RW && use_state_ratio && !segmin && st->stop) live = false;
    __syncthreads();                                           // s_logtab stands
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int p = t >> 3, s = t & 7;
    const bool act = live && p <= N;
    const size_t pcell = (size_t)(act ? p : 0);                // cell (p, p+1): row s at bidx(W, p, 1, s, .)
    double removed = 0.0;
    int na = -1, nb = -1;
    T nval = (T)0;
    if (RW && act) {
        const double ratio = segmin ? seg_ratio : (use_state_ratio ? st->ratio : ratio_arg);
        for (int d = s + 1; d <= W; d += 8) {
            const int j = p + d;
            int mult = 0;
            if (j <= N - 1) mult = (d == 1) ? 2 : 1;
            else if (j == N) mult = (d == 1) ? 1 : 0;
            else if (j == N + 1) mult = (p == N) ? 1 : 0;
            if (mult) {
                const int a = rw_path[p];
                const int b = (j == N + 1) ? rw_path[0] : rw_path[j];
                T *e = band + bidx(W, p, d, a, b);
                T cur = *e;
                for (int q = 0; q < mult; q++) {
                    const double old = (double)cur;
                    const double nw = old - ratio * old;
                    cur = (T)nw;
                    removed += old - nw;
                }
                *e = cur;
                if (d == 1) { na = a; nb = b; nval = cur; }
            }
        }
    }
    if (RW) {
        na = __shfl(na, 0, 8); nb = __shfl(nb, 0, 8);
        nval = (T)__shfl((double)nval, 0, 8);
    }
    unsigned flag_bits = 0;
    int hole_p = 0x7fffffff;
    double c[NSYM];
    double tot = 0.0;
    int nv = 0;
    uint32_t cm = 0, cm5 = 0;
    // lane s of the 8-lane group sums row s once (sequentially, in the storage dtype); the group shares the sums
    T acc = (T)0;
    if (act && s < NSYM) {
#pragma unroll
        for (int x = 0; x < NSYM; x++) {
            T v = band[bidx(W, pcell, 1, s, x)];
            if (RW && s == na && x == nb) v = nval;      // the element this group has just rewritten
            acc = acc + v;
        }
    }
    const double mine = (double)acc;
#pragma unroll
    for (int x = 0; x < NSYM; x++) {
        c[x] = __shfl(mine, x, 8);
        if (c[x] > 0) {
            tot += c[x];
            if ((VALID_MASK >> x) & 1) { nv++; cm |= 1u << x; }
        }
    }
    // the candidates get_edge_weights_at offers at p: the valid symbols seen there, or every valid symbol (offer_zero)
    const uint32_t cand = offer_zero ? VALID_MASK : cm;
    const uint32_t cmw = cm | (cand << 8);
    cm5 = cm5_of_cmask(sm, cand);
    double my_m = 0.0, my_lm = 0.0;
    if (s < NSYM) {
        my_m = (c[s] > 0 && tot != 0.0) ? c[s] / tot : 0.0;
        if ((VALID_MASK >> s) & 1) my_lm = gh_log10_tab(my_m, s_logtab, GH_LOG_SERIAL);
    }
    if (act) {
        if (s < NSYM) {
            cnt[(size_t)p * 8 + s] = c[s];
            marg[(size_t)p * 8 + s] = my_m;
            if ((VALID_MASK >> s) & 1) {
                const int b5 = a6_of_sym(sm, s);
                minfo[(size_t)p * MINFO + b5] = my_lm;
                minfo[(size_t)p * MINFO + 5 + b5] = my_m;
                // the same by candidate rank (rinfo): this symbol's rank among the candidates of p
                if (rinfo && ((cand >> s) & 1u) && __popc(cm5 & ((1u << b5) - 1u)) < 4) {
                    const int r = __popc(cm5 & ((1u << b5) - 1u));
                    rinfo[(size_t)p * RINFO + r] = my_lm;
                    rinfo[(size_t)p * RINFO + 4 + r] = my_m;
                }
            }
        } else {
            cnt[(size_t)p * 8 + 7] = tot;
            marg[(size_t)p * 8 + 7] = 0.0;
            nvalid[p] = nv;
            // (the window's flags are collected per workgroup and leave with one atomic each at the end: see k_rw)
            if (cmask[p] != cmw) flag_bits |= 1u;                 // the conditional table must then be rebuilt in full
            cmask[p] = cmw;
            minfo[(size_t)p * MINFO + 10] = __longlong_as_double((long long)cm5);
            for (int r = __popc(cm5); rinfo && r < 4; r++) {      // ranks that do not exist
                rinfo[(size_t)p * RINFO + r] = 0.0;
                rinfo[(size_t)p * RINFO + 4 + r] = INFINITY;
            }
            if (p >= 1 && cand == 0) hole_p = p;
            if (p >= 1 && (cm5 & (1u << 4))) flag_bits |= 2u;      // the LAST symbol of the candidate order ('-' by default) is offered here
            if (p >= 1 && __popc(cm5) > 4) flag_bits |= 4u;
        }
    }
    if (RW && G && act && p < N && rw_path[p] != 4) {
        // the rows of the conditional table that this path's reweighting has changed: source p, symbol path[p],
        // lags 1..L (the lane that rewrote element (p, p+l) also owns lag l, so it reads its own store back).
        // nvalid / cmask of the targets are read while other groups rewrite them: k_lt trusts these rows only
        // when no candidate mask moved (st->cm_same), and then old and new values are the same.
        // (conditionals A, B, D without a baked marginal term only: the host passes G = nullptr otherwise)
        const int a = rw_path[p];
        const int a6 = a6_of_sym(sm, a);
        const double nv_i = (double)nv, ca = __shfl(mine, a, 8);
        // ranked tables (k_lt): the row of symbol a is the row of its rank among the candidates of p, the columns of
        // lag l are the candidates of p+l in ascending order; a path symbol that is no candidate has no row
        const bool ranked = st->ranked != 0;
        int row6 = a6;
        if (ranked && a6 < 5) row6 = ((cm5 >> a6) & 1u) ? __popc(cm5 & ((1u << a6) - 1u)) : -1;
        for (int l = s + 1; l <= L && row6 >= 0; l += 8) {
            double *out = G + (((size_t)p * 6 + row6) * L + (l - 1)) * LT_ROW;
            const int snp = p + l;
            if (!(snp <= N && (a6 < 5 || p == 0))) {
#pragma unroll
                for (int b5 = 0; b5 < LT_ROW; b5++) out[b5] = 0.0;
                continue;
            }
            double rowv[NSYM];
            double sum = 0.0;
            if (l <= W) {
                const T *rc = band + bidx(W, p, l, a, 0);
                T racc = (T)0;
#pragma unroll
                for (int x = 0; x < NSYM; x++) { const T v = rc[x]; rowv[x] = (double)v; racc = racc + v; }
                sum = (double)racc;
            } else {
#pragma unroll
                for (int x = 0; x < NSYM; x++) rowv[x] = 0.0;
            }
            const uint32_t cmj = CM_CAND(cmask[snp]);
            const double den = (cond_mode == GH_COND_A) ? (double)nvalid[snp] + sum : (cond_mode == GH_COND_D ? nv_i + sum : nv_i + ca);
            double xq[LT_ROW], v[LT_ROW];
            bool odd = false;
#pragma unroll
            for (int b5 = 0; b5 < LT_ROW; b5++) {
                double rv = rowv[0];
                const int sb = vsym(sm, b5);
#pragma unroll
                for (int x = 1; x < NSYM; x++) rv = (sb == x) ? rowv[x] : rv;
                xq[b5] = (1.0 + rv) / den;
                odd |= !gh_log10_is_normal(xq[b5]);
            }
#pragma unroll
            for (int b5 = 0; b5 < LT_ROW; b5++) v[b5] = gh_log10_normal_tab(xq[b5], 0, s_logtab, GH_LOG_SERIAL);
            if (odd) {
#pragma unroll
                for (int b5 = 0; b5 < LT_ROW; b5++) v[b5] = gh_log10_tab(xq[b5], s_logtab, GH_LOG_SERIAL);
            }
            if (!ranked) {
#pragma unroll
                for (int b5 = 0; b5 < LT_ROW; b5++) out[b5] = ((cmj >> vsym(sm, b5)) & 1) ? v[b5] : -INFINITY;
            } else {
                const uint32_t cj5 = cm5_of_cmask(sm, cmj);
#pragma unroll
                for (int rb = 0; rb < LT_ROW; rb++) {
                    const int b5 = nth_set5(cj5, rb);
                    double r = -INFINITY;
#pragma unroll
                    for (int q = 0; q < LT_ROW; q++) r = (b5 == q) ? v[q] : r;
                    out[rb] = r;
                }
            }
        }
    }
    {
        __shared__ unsigned s_flags;
        __shared__ int s_hole;
        if (threadIdx.x == 0) { s_flags = 0; s_hole = 0x7fffffff; }
        __syncthreads();
        if (flag_bits) atomicOr(&s_flags, flag_bits);
        if (hole_p != 0x7fffffff) atomicMin(&s_hole, hole_p);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned f = s_flags;
            if (f & 1u) atomicAnd(&st->cm_same, 0);
            if (f & 2u) atomicAnd(&st->nodel, 0);
            if (f & 4u) atomicAnd(&st->narrow, 0);
            if (s_hole != 0x7fffffff) atomicMin(&st->first_hole, s_hole);
        }
    }
    if (RW) {
        // fixed-order tree so the removed mass is run-to-run reproducible
        s_red[threadIdx.x] = removed;
        __syncthreads();
        for (int q = 128; q > 0; q >>= 1) {
            if ((int)threadIdx.x < q) s_red[threadIdx.x] += s_red[threadIdx.x + q];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0];
    }
}

// batched launches: re-arm the words k_marg min/and-reduces into (single windows use a memset)
// the most states that enter any target when a state is the last L picks as candidate ranks: max over t of the product of
// the radices of positions t-L+1 .. t (segmix.hpp: 5 where a position offers five candidates, 4 everywhere else)
__global__ void __launch_bounds__(256) k_classify(const uint32_t *__restrict__ cmask, int N, int L, dev_state *st)
{
    const int t = blockIdx.x * 256 + threadIdx.x + 1;
    int prod = 0;
    if (t <= N) {
        prod = 1;
        for (int l = 0; l < L; l++) {
            const int p = t - l;
            prod *= (p >= 1 && __popc(CM_CAND(cmask[p])) == 5) ? 5 : 4;     // (segm_radix: 5 where five candidates are offered, else 4)
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(prod, o); prod = x > prod ? x : prod; }
    if ((threadIdx.x & 63) == 0 && prod > 0) atomicMax(&st->maxstates, prod);
}

__global__ void k_rearm(dev_state *st, const win_desc *wd, int skip_if_stopped)
{
    if (wd) st = wd[blockIdx.x].st;
    // a window that hit a hole keeps its last flags: the k_marg that would refresh them is skipped as well
    if (skip_if_stopped && st->stop) return;
    if (threadIdx.x == 0) { st->first_hole = 0x7f7f7f7f; st->nodel = 0x7f7f7f7f; st->cm_same = 0x7f7f7f7f; st->narrow = 0x7f7f7f7f; }
}

// freeze the current log-marginals as the original ones (slot [11..15] of minfo)
__global__ void k_snapshot(double *dst_minfo, const double *src_minfo, int N, const win_desc *wd)
{
    if (wd) {
        if (!wd[blockIdx.y].snap) return;
        dst_minfo = wd[blockIdx.y].minfo;
        src_minfo = dst_minfo;
    }
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int p = t >> 3, q = t & 7;
    if (p > N || q >= 5) return;
    dst_minfo[(size_t)p * MINFO + 11 + q] = src_minfo[(size_t)p * MINFO + q];
}

// ---------------------------------------------------------------------------------------------
// k_lt: the conditional table, SOURCE-major:
//   G[i][a6][l-1][b5]  (i = source position 0..N-1, target snp = i+l)
//     = -inf                                   if b5 is not a candidate at snp (bakes the
//                                              candidate mask of gretel.py:166-174 into the sum)
//     = log10( (1 + H[a,b,i,snp]) / den )      otherwise.  The marginal term of the edge weight
//                                              (gh_config.marginal_term: w = log10 marginal + x1 + x2 ...;
//                                              (0.0 + lm) + x1 == lm + x1 bit for bit) is NOT in the table:
//                                              the segment-parallel readers add lm in front of x1 themselves
//                                              (k_seg, k_cwalk, k_cwalkg: from rinfo / minfo), so that a
//                                              reweight, which moves every marginal of every position, leaves
//                                              all but the path's own rows (columns) of G as they are.  Only
//                                              for the serial walkers (bake_lm: they read whole rows) the
//                                              lag-1 entry is stored as lm + x1, and rebuilt before every path.
//     = 0.0                                    for snp > N, i >= N (padding) and rows a path can
//                                              never select ('_' anywhere but position 0)
// One row G[i][a6] (L x 5 doubles) is everything position i contributes to the next L steps once
// symbol a6 has been selected there: the walker reads exactly one row per step.
// ---------------------------------------------------------------------------------------------
#define LT_PAD 16      /* zero source blocks behind N so the unrolled walker may overrun */
// doubles per position of a k_walk_spec LDS buffer: the raw G block (depth 1), or the derived tables H + Yr (depth 2)
// (depth 2: 64 doubles of H per target + 16 rows of Yr per source, each L-2 lags padded to an even count so that a
// lane reads its row with 16-byte LDS loads)
__host__ __device__ constexpr int walk_pos_doubles(int L, bool deep) { return deep ? 64 + 16 * (L > 2 ? ((L - 2 + 1) & ~1) : 0) : 6 * L * 5; }
#define WALK_LDS_MAX (160 * 1024)   /* LDS one workgroup can have on gfx950 */
#define WALK_OV 4      /* source blocks kept behind a chunk in LDS: the last body reads sources j+1 .. j+4 (k_walk_spec) */
// positions per LDS buffer of k_walk_spec for lag count L (two buffers of chunk + WALK_OV positions and the two
// word buffers must fit WALK_LDS_MAX; whole unrolled groups; at most 64 = the bookkeeper's lanes); 0 = does not fit.
// The variant is chosen on the device (it depends on the window's candidate masks), the chunk follows from it.
__host__ __device__ constexpr int walk_chunk(int L, bool deep)
{
    int c = (WALK_LDS_MAX - 2 * 64 * 8) / (2 * walk_pos_doubles(L, deep) * 8) - WALK_OV;
    if (c > 64) c = 64;
    c = (c / L) * L;
    return c >= L ? c : 0;
}
__host__ __device__ constexpr size_t walk_lds_bytes(int L, bool deep)
{
    return 2 * (size_t)(walk_chunk(L, deep) + WALK_OV) * walk_pos_doubles(L, deep) * 8 + 2 * 64 * 8;
}

template <typename T>
__device__ __forceinline__ double lt_entry(const T *band, int N, int W, int cond_mode, int bake_lm,
                                           const double *cnt, const int32_t *nvalid, const uint32_t *cmask,
                                           const double *minfo, int i, int a6, int l, int b5, symmap sm, const T *tband = nullptr)
{
    const int snp = i + l;
    if (!(i < N && snp <= N && (a6 < 5 || i == 0))) return 0.0;
    const int b = vsym(sm, b5);
    if (!((CM_CAND(cmask[snp]) >> b) & 1)) return -INFINITY;
    double v = log_conditional(band, W, cond_mode, cnt, nvalid, fsym(sm, a6), b, i, snp, tband);
    if (bake_lm && l == 1) v = minfo[(size_t)snp * MINFO + b5] + v;
    return v;
}

// The depth-2 walker (spec2_walker below) does not read G but two tables derived from the ranked G, kept in HBM in
// exactly the layout its loader waves copy into LDS (so that the loaders execute a handful of wide copies per chunk
// instead of a thousand scalar gathers beside the walker):
//   Ht[t][a2*16 + a1*4 + b] = x1 + x2 = G[t-1][a1][lag 1][b] + G[t-2][a2][lag 2][b]   (t >= 2; t = 1: x1 alone; position 0
//                             always contributes its '_' row; the first addition of the reference's lag-ascending sum)
//   Yt[i][w][b][l - 3]       = G[i][w][lag l][b], l = 3..L, rows padded to an even number of lags
// for positions 0 .. N + WALK_TPAD - 1 (zeros behind the table: the walker runs whole chunks).
#define WALK_TPAD 72
__host__ __device__ constexpr int deep_nyp(int L) { return L > 2 ? ((L - 2 + 1) & ~1) : 0; }

// what k_lt stores at G[i][row6][lag - 1][col5] in the ranked layout
template <typename T>
__device__ __forceinline__ double lt_entry_ranked(const T *band, int N, int W, int cond_mode, int bake_lm,
                                                  const double *cnt, const int32_t *nvalid, const uint32_t *cmask,
                                                  const double *minfo, int i, int row6, int lag, int col5, symmap sm, const T *tband = nullptr)
{
    const int snp = i + lag;
    if (!(i < N && snp <= N && row6 != 4)) return 0.0;
    int a6 = row6;
    if (a6 < 4) a6 = nth_set5(cm5_of_cmask(sm, CM_CAND(cmask[i])), a6);
    if (a6 < 0) return 0.0;
    const int b5 = nth_set5(cm5_of_cmask(sm, CM_CAND(cmask[snp])), col5);
    return b5 >= 0 ? lt_entry(band, N, W, cond_mode, bake_lm, cnt, nvalid, cmask, minfo, i, a6, lag, b5, sm, tband) : -INFINITY;
}

// inc_path == nullptr: rebuild every entry.  Otherwise (conditional A or B, no marginal term, and the
// tensor changed ONLY through a path reweight): every cell that changed is H[path[i], path[j], i, j], which
// only enters the rows G[i][a6(path[i])][*][*], and k_marg<T, true> rewrote those N*L*5 entries in the same
// pass that reweighted the cells.  k_lt then only refreshes the entries of Ht / Yt those rows feed -- unless k_marg saw
// a candidate mask change (V(p) or the -inf masks moved) or the path was cut short by a hole: then it rebuilds everything.
template <typename T>
__global__ void __launch_bounds__(256)
k_lt(const T *band, int N, int W, int L, int cond_mode, int marginal_term /* = bake_lm, see above */,
     const double *cnt, const int32_t *nvalid, const uint32_t *cmask,
     const double *minfo, double *G, dev_state *st, const uint8_t *inc_path, const win_desc *wd, int spin,
     int allow_ranked, double *Ht, double *Yt, symmap sm, const T *tband)
{
    if (wd) {
        const win_desc &d = wd[blockIdx.y];
        band = (const T *)d.band; cnt = d.cnt; nvalid = d.nvalid; cmask = d.cmask; minfo = d.minfo; G = d.G; st = d.st;
        Ht = d.Ht; Yt = d.Yt;
        if (st->stop) return;
        if (inc_path) inc_path = d.paths + (size_t)(spin - 1) * (N + 1);
    }
    const int nyp = deep_nyp(L), ypos = 16 * nyp;
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsize = (size_t)gridDim.x * blockDim.x;
    // inc_path != null: k_marg<T, true> has already rewritten the rows of G the last path changed; they stand when no
    // candidate mask moved and the path was complete -- otherwise rebuild everything
    if (inc_path && st->cm_same && !st->stop) {
        if (!(st->ranked && Ht)) return;
        // row r_p = rank of path[p] at p changed for every p: refresh Ht[p+1][*][r_p][*], Ht[p+2][r_p][*][*], Yt[p][r_p]
        auto Gat = [&](int i, int row, int lag, int col) { return G[(((size_t)i * 6 + row) * L + (lag - 1)) * LT_ROW + col]; };
        const size_t total = (size_t)N * 4;                     // one thread per (position, target rank b)
        for (size_t idx = gtid; idx < total; idx += gsize) {
            const int p = (int)(idx >> 2), b = (int)(idx & 3);
            if (p == 0) continue;                               // position 0 is done below
            const int a6 = a6_of_sym(sm, inc_path[p]);
            const uint32_t c5 = cm5_of_cmask(sm, CM_CAND(cmask[p]));
            if (inc_path[p] == 4 || a6 > 4 || !((c5 >> a6) & 1u)) continue;
            const int r = __popc(c5 & ((1u << a6) - 1u));
            const double g1 = Gat(p, r, 1, b), g2 = Gat(p, r, 2, b);
            double *h1 = Ht + (size_t)(p + 1) * 64 + r * 4 + b, *h2 = Ht + (size_t)(p + 2) * 64 + r * 16 + b;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                h1[q * 16] = g1 + Gat(p - 1, p - 1 == 0 ? 5 : q, 2, b);      // a2 = q
                h2[q * 4] = Gat(p + 1, q, 1, b) + g2;                         // a1 = q
            }
            for (int li = 0; li + 2 < L; li++) Yt[(size_t)p * ypos + (r * 4 + b) * nyp + li] = Gat(p, r, li + 3, b);
        }
        // position 0 carries '_' (row 5) whatever the hypothesis: targets 1 and 2 and the rows of source 0 in full
        const size_t total0 = (size_t)128 + ypos;
        for (size_t q = gtid; q < total0; q += gsize) {
            if (q < 128) {
                const int tt = 1 + (int)(q >> 6), ln = (int)(q & 63), b = ln & 3, a1 = (ln >> 2) & 3;
                double v = Gat(tt - 1, tt - 1 == 0 ? 5 : a1, 1, b);
                if (tt >= 2) v = v + Gat(0, 5, 2, b);
                Ht[(size_t)tt * 64 + ln] = v;
            } else {
                const int r = (int)(q - 128), wb = r / nyp, li = r % nyp;
                if (li + 2 < L) Yt[r] = Gat(0, 5, li + 3, wb & 3);
            }
        }
        return;
    }
    // Ranked layout (every position has at most 4 candidates and the caller's walker can use it): row r < 4 of
    // source i is the r-th candidate of i in ascending symbol order, column c < 4 of lag l the c-th candidate of
    // i+l; missing ranks get a row of zeros / a column of -inf, row 5 stays the '_' row of position 0.  First-wins
    // tie-breaking over ranks is first-wins over symbols, so the depth-2 walker (4 symbols) also serves windows
    // with '-' candidates; its bookkeeper maps ranks back through the candidate bits of minfo.
    const bool ranked = allow_ranked && st->narrow != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->ranked = ranked ? 1 : 0; st->maxstates = 0; }      // (k_classify follows a rebuild)
    const size_t total = (size_t)(N + LT_PAD) * 6 * L * LT_ROW;
    for (size_t t = gtid; t < total; t += gsize) {
        const int b5 = (int)(t % LT_ROW);
        size_t r = t / LT_ROW;
        const int l = (int)(r % L) + 1;
        r /= L;
        const int a6 = (int)(r % 6);
        const int i = (int)(r / 6);
        G[t] = ranked ? lt_entry_ranked(band, N, W, cond_mode, marginal_term, cnt, nvalid, cmask, minfo, i, a6, l, b5, sm, tband)
                      : lt_entry(band, N, W, cond_mode, marginal_term, cnt, nvalid, cmask, minfo, i, a6, l, b5, sm, tband);
    }
    if (!(ranked && Ht)) return;
    // the derived tables in full, from the band (other threads are still writing G)
    const int nsrc_all = N + LT_PAD;
    const size_t nH = (size_t)(N + WALK_TPAD) * 64;
    for (size_t q = gtid; q < nH; q += gsize) {
        const int tt = (int)(q >> 6), ln = (int)(q & 63), b = ln & 3, a1 = (ln >> 2) & 3, a2 = ln >> 4;
        double v = 0.0;
        if (tt >= 1 && tt - 1 < nsrc_all) {
            const int s1 = tt - 1, s2 = tt - 2;
            v = lt_entry_ranked(band, N, W, cond_mode, marginal_term, cnt, nvalid, cmask, minfo, s1, s1 == 0 ? 5 : a1, 1, b, sm, tband);
            if (tt >= 2 && L >= 2)
                v = v + lt_entry_ranked(band, N, W, cond_mode, marginal_term, cnt, nvalid, cmask, minfo, s2, s2 == 0 ? 5 : a2, 2, b, sm, tband);
        }
        Ht[q] = v;
    }
    if (ypos > 0) {
        const size_t nY = (size_t)(N + WALK_TPAD) * ypos;
        for (size_t q = gtid; q < nY; q += gsize) {
            const int i = (int)(q / ypos), r = (int)(q % ypos), wb = r / nyp, li = r % nyp;
            double v = 0.0;
            if (i < nsrc_all && li + 2 < L)
                v = lt_entry_ranked(band, N, W, cond_mode, marginal_term, cnt, nvalid, cmask, minfo, i, i == 0 ? 5 : (wb >> 2), li + 3, wb & 3, sm, tband);
            Yt[q] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_walk_src: gretel/gretel.py:143-189 as one workgroup of 8 wavefronts:
//   wave 0     the walker: N dependent steps, one LDS row read + (L-1) adds + a 3-level DPP
//              max + ballot per step; fully unrolled over LC steps so that the L(L-1)/2 live
//              lag terms rotate through registers by renaming, not by moves
//   wave 1     the bookkeeper: one chunk behind the walker, turns the selected symbols into
//              path bytes and the sequential log10-marginal sums (gretel.py:182-186)
//   waves 2-7  the loaders: stream the next chunk of G from L2/HBM into the other LDS buffer
// ---------------------------------------------------------------------------------------------
struct walk_params {
    int N, L;
    int chunk;                // source positions per LDS buffer (multiple of L, <= 64)
    int rearm;                // spin loops: re-arm first_hole/nodel/cm_same for the k_marg<T,true> that follows
    int depth2;               // k_walk_spec: depth-2 speculation where the window allows it (<= 4 candidates per position, L >= 2)
    int _pad;
    const double *G;          // [(N+LT_PAD)][6][L][5]
    const double *Ht, *Yt;    // depth-2 tables of k_lt: [(N+WALK_TPAD)][64], [(N+WALK_TPAD)][16*deep_nyp(L)]
    const double *minfo;      // [N+2][16]
    uint8_t *path_out;        // device [N+1]
    gh_path_rec *rec;         // device
    dev_state *st;
    double min_remove;
    symmap sm;
};

__device__ __forceinline__ void copy_to_lds(double *dst, const double *src, size_t n_dbl, int tid, int nthr)
{
    // both sides are 16-byte aligned and n_dbl is even; 8 x 16-byte loads in flight per lane
    const double2 *s = reinterpret_cast<const double2 *>(src);
    double2 *d = reinterpret_cast<double2 *>(dst);
    const unsigned n = (unsigned)(n_dbl >> 1), step = (unsigned)nthr * 8u;
    unsigned q0 = 0;
    for (; q0 + step <= n; q0 += step) {
        const unsigned q = q0 + (unsigned)tid;
        const double2 v0 = s[q], v1 = s[q + nthr], v2 = s[q + 2 * nthr], v3 = s[q + 3 * nthr];
        const double2 v4 = s[q + 4 * nthr], v5 = s[q + 5 * nthr], v6 = s[q + 6 * nthr], v7 = s[q + 7 * nthr];
        d[q] = v0; d[q + nthr] = v1; d[q + 2 * nthr] = v2; d[q + 3 * nthr] = v3;
        d[q + 4 * nthr] = v4; d[q + 5 * nthr] = v5; d[q + 6 * nthr] = v6; d[q + 7 * nthr] = v7;
    }
    for (unsigned q = q0 + (unsigned)tid; q < n; q += (unsigned)nthr) d[q] = s[q];
}

// max without the sNaN-quieting v_max(x,x) pair the compiler adds around fmax
__device__ __forceinline__ double vmax_f64(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// arg-max over lanes 0..7 of each row, lowest lane among the maxima: gretel.py:166-174
// (first candidate is the incumbent, later ones win on strict >).  Non-candidates carry -inf.
__device__ __forceinline__ int argmax8(double acc)
{
    double m = acc;
    m = vmax_f64(m, dpp_f64<0xB1>(m));      // quad_perm [1,0,3,2]
    m = vmax_f64(m, dpp_f64<0x4E>(m));      // quad_perm [2,3,0,1]
    m = vmax_f64(m, dpp_f64<0x141>(m));     // row_half_mirror
    // (a NaN weight -- log10 of a zero marginal plus an infinite conditional, zero-count candidates only -- in FIRST place is
    // the reference's incumbent and nothing compares greater than it; anywhere else it never wins: v_max ignores it)
    const unsigned long long win = __builtin_amdgcn_ballot_w64(acc == m || ((threadIdx.x & 7) == 0 && acc != acc));
    return (int)__builtin_ctzll(win);
}

struct walk_totals {
    double hp_cur, hp_orig, minm;
};

// bookkeeper: lanes = steps of one chunk; gather in parallel, accumulate strictly in step order
__device__ __forceinline__ void book_chunk(const walk_params &P, const unsigned long long *words, int LC, int s0, int ns,
                                           int lane, walk_totals &T)
{
    double lm = 0.0, lm0 = 0.0, mg = 0.0;
    if (lane < ns) {
        const int t = s0 + lane;                        // target SNP of chunk-local step `lane`
        const unsigned long long word = words[lane / LC];
        const int w = (int)((word >> (4 * (LC - 1 - lane % LC))) & 15ull);
        const double *inf = P.minfo + (size_t)t * MINFO;
        lm = inf[w];
        mg = inf[5 + w];
        lm0 = inf[11 + w];
        P.path_out[t] = (uint8_t)vsym(P.sm, w);
    }
    for (int j = 0; j < ns; j++) {
        const double m = readlane_f64(mg, j);           // gretel.py:182
        if (m < T.minm) T.minm = m;
        T.hp_cur += readlane_f64(lm, j);                // gretel.py:185
        T.hp_orig += readlane_f64(lm0, j);              // gretel.py:186
    }
}

template <int LC>
__global__ void __launch_bounds__(512) k_walk_src(walk_params P, const win_desc *wd, int spin)
{
    extern __shared__ __align__(16) double smem[];
    if (wd) {
        const win_desc &d = wd[blockIdx.x];
        P.G = d.G; P.minfo = d.minfo; P.st = d.st;
        P.path_out = d.paths + (size_t)spin * (P.N + 1); P.rec = d.recs + spin;
    }
    dev_state *st = P.st;
    if (st->stop) return;
    constexpr int L = LC;
    constexpr int ROW = L * LT_ROW;                 // doubles per (source, a6) row
    constexpr int BLK = 6 * ROW;                    // doubles per source position
    const int C = P.chunk;
    double *const g0 = smem;
    unsigned long long *const words0 = reinterpret_cast<unsigned long long *>(smem + 2 * (size_t)C * BLK);
#define LDS_G(k) (g0 + (size_t)((k) & 1) * C * BLK)
#define LDS_W(k) (words0 + ((k) & 1) * 64)

    const int first_hole = st->first_hole;
    const int Nw = first_hole <= P.N ? first_hole - 1 : P.N;      // steps that can be decided
    const int nchunks = (Nw + C - 1) / C;                         // chunk k = sources k*C .. k*C+C-1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform role

    auto load_chunk = [&](int k, int t, int nt) {
        // whole chunks: G is padded with LT_PAD zero blocks, and C <= 64 never outruns N+LT_PAD by more than
        // the walker touches -- clip to the allocation all the same
        const int i0 = k * C;
        int nsrc = P.N + LT_PAD - i0;
        if (nsrc > C) nsrc = C;
        if (nsrc < 0) nsrc = 0;
        copy_to_lds(LDS_G(k), P.G + (size_t)i0 * BLK, (size_t)nsrc * BLK, t, nt);
        for (int q = nsrc * BLK + t; q < C * BLK; q += nt) LDS_G(k)[q] = 0.0;     // never walk over stale LDS bits
    };

    if (nchunks > 0) load_chunk(0, tid, (int)blockDim.x);
    __syncthreads();

    if (wave >= 2) {
        for (int k = 0; k < nchunks; k++) {
            if (k + 1 < nchunks) load_chunk(k + 1, tid - 128, (int)blockDim.x - 128);
            __syncthreads();
        }
        return;
    }

    if (wave == 1) {
        walk_totals T = {0.0, 0.0, INFINITY};
        if (lane == 0) P.path_out[0] = SYM_US;
        for (int k = 0; k < nchunks; k++) {
            if (k > 0) {
                const int s0 = (k - 1) * C + 1;
                book_chunk(P, LDS_W(k - 1), LC, s0, C, lane, T);
            }
            __syncthreads();
        }
        if (nchunks > 0) {
            const int s0 = (nchunks - 1) * C + 1;
            book_chunk(P, LDS_W(nchunks - 1), LC, s0, Nw - s0 + 1, lane, T);
        }
        if (lane == 0) {
            if (first_hole <= P.N) {                                  // gretel.py:176-180
                st->stop = 1;
                st->hole_at = first_hole;
            } else {
                double r = T.minm;
                if (r < P.min_remove) r = P.min_remove;               // cmd.py:157-160
                P.rec->hp_current = T.hp_cur;
                P.rec->hp_original = T.hp_orig;
                P.rec->ratio = r;                                     // the ratio the reweight will use (clamped)
                P.rec->min_marginal = T.minm;
                P.rec->magnitude = 0.0;
                st->ratio = r;
                st->n_done += 1;
                if (P.rearm) { st->first_hole = 0x7f7f7f7f; st->nodel = 0x7f7f7f7f; st->cm_same = 0x7f7f7f7f; st->narrow = 0x7f7f7f7f; }
            }
        }
        return;
    }

    // ---- walker -------------------------------------------------------------------------------
    const int bb = (lane & 7) < 5 ? (lane & 7) : 0;
    double Y[LC][LC];                       // Y[slot][l-1]: lag-l term loaded at step slot (mod LC)
#pragma unroll
    for (int u = 0; u < LC; u++)
#pragma unroll
        for (int l = 0; l < LC; l++) Y[u][l] = 0.0;
    int wprev = 5;                          // path[0] = '_'

    for (int k = 0; k < nchunks; k++) {
        const double *gb = LDS_G(k) + bb;
        unsigned long long *wk = LDS_W(k);
        const int ngroups = C / LC;
        for (int g = 0; g < ngroups; g++) {
            unsigned long long word = 0;   // 4 bits per step, LC <= 16 steps per group
#pragma unroll
            for (int u = 0; u < LC; u++) {
                // source i = k*C + g*LC + u, target t = i + 1
                const double *row = gb + (size_t)(g * LC + u) * BLK + wprev * ROW;
#pragma unroll
                for (int l = 0; l < LC; l++) Y[u][l] = row[l * LT_ROW];
                double acc = Y[u][0];
#pragma unroll
                for (int l = 1; l < LC; l++) acc += Y[(u - l + LC) % LC][l];      // l ascending
                const int w = argmax8(acc);
                word = (word << 4) + (unsigned long long)w;
                wprev = w;
            }
            wk[g] = word;
        }
        __syncthreads();
    }
#undef LDS_G
#undef LDS_W
}

// ---------------------------------------------------------------------------------------------
// k_walk_spec: the walker with depth-1 speculation.  Lane = (hypothesis group ga = lane>>3,
// candidate b = lane&7).  While symbol w_j of position j is still being resolved, group ga
// already evaluates target j+1 under the hypothesis w_j == ga: its lag-1 term is read with a
// lane-constant LDS address (no dependence on the path), lags 2..L come from rows selected by
// the symbols resolved one and more steps earlier.  All groups run the same adds and the same
// DPP arg-max, so the hypotheses cost no extra instructions; one ballot holds every group's
// winner and resolving w_{j+1} is a 64-bit shift by 8*w_j and a find-first-one.
// The sums are the same IEEE additions in the same order as in k_walk_src: bit-identical.
// ---------------------------------------------------------------------------------------------
template <bool NODEL>
__device__ __forceinline__ unsigned long long group_argmax(double acc)
{
    double m = acc;
    m = vmax_f64(m, dpp_f64<0xB1>(m));      // quad_perm [1,0,3,2]
    m = vmax_f64(m, dpp_f64<0x4E>(m));      // quad_perm [2,3,0,1]
    if (!NODEL) m = vmax_f64(m, dpp_f64<0x141>(m));     // row_half_mirror
    if (!NODEL) return __builtin_amdgcn_ballot_w64(acc == m || ((threadIdx.x & 7) == 0 && acc != acc));      // (NaN in first place: see argmax8)
    return __builtin_amdgcn_ballot_w64(acc == m);
}

// bookkeeper for k_walk_spec: word g of a chunk holds the symbols of positions j0+g*LC+1 .. j0+g*LC+LC
// (wbits bits each, the oldest highest).  Two steps per chunk, run one loop iteration apart so that the latency of
// the gather never sits between two barriers: book_gather issues the loads of the selected symbols' marginals
// (lane = chunk-local position), book_fold adds them up strictly in position order one iteration later.
typedef double lds_v2d __attribute__((ext_vector_type(2), aligned(16)));     // 16-byte loads / stores

// One chunk of bookkeeping, lane = chunk-local position.  book_prefetch loads the position's whole minfo row (128 bytes:
// it does not depend on the path) while the walker is still in that chunk; book_consume, one iteration later, picks the
// selected symbol's entries out of the registers and folds them strictly in position order.  No memory latency sits
// between the walker's last step and the end of the kernel.
struct book_row {
    lds_v2d v[8];           // minfo[j][0..15]
};

__device__ __forceinline__ void book_prefetch(const walk_params &P, int j0, int ns, int Nw, int lane, book_row &R)
{
    const int j = j0 + lane + 1;
    if (lane < ns && j <= Nw) {
        const lds_v2d *src = reinterpret_cast<const lds_v2d *>(P.minfo + (size_t)j * MINFO);
#pragma unroll
        for (int q = 0; q < 8; q++) R.v[q] = src[q];
    }
}

__device__ __forceinline__ void book_consume(const walk_params &P, const unsigned long long *words, int LC, int wbits,
                                             bool ranked, int j0, int ns, int Nw, int lane, const book_row &R,
                                             walk_totals &T, double &lane_min)
{
    double lm = 0.0, lm0 = 0.0, mg = INFINITY;
    const int j = j0 + lane + 1;
    if (lane < ns && j <= Nw) {
        const unsigned long long word = words[lane / LC];
        int w = (int)((word >> (wbits * (LC - 1 - lane % LC))) & ((1ull << wbits) - 1ull));
        if (ranked) {
            w = nth_set5((uint32_t)__double_as_longlong(R.v[5].x), w);      // minfo[10]: candidate bits; rank -> symbol
            if (w < 0) w = 0;           // cannot happen for a resolved position (it has a candidate of that rank)
        }
        // minfo[w], minfo[5 + w], minfo[11 + w] out of the registers (w < 5)
        const double row[16] = {R.v[0].x, R.v[0].y, R.v[1].x, R.v[1].y, R.v[2].x, R.v[2].y, R.v[3].x, R.v[3].y,
                                R.v[4].x, R.v[4].y, R.v[5].x, R.v[5].y, R.v[6].x, R.v[6].y, R.v[7].x, R.v[7].y};
        lm = row[0]; mg = row[5]; lm0 = row[11];
#pragma unroll
        for (int q = 1; q < 5; q++) {
            lm = (w == q) ? row[q] : lm;
            mg = (w == q) ? row[5 + q] : mg;
            lm0 = (w == q) ? row[11 + q] : lm0;
        }
        P.path_out[j] = (uint8_t)vsym(P.sm, w);
    }
    if (mg < lane_min) lane_min = mg;                   // gretel.py:182, per lane; reduced over lanes at the end (min is exact)
    int s = 0;
    for (; s + 4 <= ns; s += 4) {                       // four positions per trip: less loop overhead beside the walker
#pragma unroll
        for (int q = 0; q < 4; q++) {
            T.hp_cur += readlane_f64(lm, s + q);        // gretel.py:185 (+0.0 for unused lanes)
            T.hp_orig += readlane_f64(lm0, s + q);      // gretel.py:186
        }
    }
    for (; s < ns; s++) {
        T.hp_cur += readlane_f64(lm, s);
        T.hp_orig += readlane_f64(lm0, s);
    }
}

// Body j of the walk (j = 0 .. Nw-1).  Entering it: w_j is resolved (sh = 8*w_j), B is the ballot of
// target j+1, the row of source j (Y_j) and the lag-1 hypothesis terms of target j+2 are in flight.
//   A  resolve w_{j+1} = ffs(B >> sh)                                   scalar chain
//   S  acc_{j+2} = hyp_{j+2}[ga] + Y_j[lag 2] + Y_{j-1}[lag 3] + ...     vector chain, independent of A
//   R  issue the row of source j+1 under w_{j+1} and the hypothesis terms of target j+3
//   M  B = group-wise arg-max of acc_{j+2}                               independent of R
// so the LDS latency of R is covered by M and by the next body's A and S.
typedef __attribute__((address_space(3))) const double lds_cdouble;

// diagnostic builds only (-DGH_STAMPS): s_memtime stamps between the segments of a walker body, summed per
// segment in scalar registers and stored once at the end into st->dbg[4..8]; never defined in the product build
#ifdef GH_STAMPS
#define GH_STAMP(i)                                                                                   \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        gh_seg[i] += t_ - gh_tprev;                                                                   \
        gh_tprev = t_;                                                                                \
    } while (0)
#else
#define GH_STAMP(i)
#endif

template <int LC, bool NODEL>
__device__ __forceinline__ void spec_walker(const walk_params &P, double *g0, unsigned long long *words0,
                                            int C, int RS, int nchunks, int lane)
{
    constexpr int ROW = LC * LT_ROW;
    constexpr int BLK = 6 * ROW;
    const int b = lane & 7;
    const int bb = NODEL ? (lane & 3) : (b < 5 ? b : 0);
    const int ga = (lane >> 3) < 6 ? (lane >> 3) : 5;
    double Y[LC][LC];                       // Y[slot][l]: lag-(l+1) term of the source with index == slot (mod LC)
#pragma unroll
    for (int u = 0; u < LC; u++)
#pragma unroll
        for (int l = 0; l < LC; l++) Y[u][l] = 0.0;

    // state entering body 0: w_0 = '_' (a6 = 5); target 1 has the single lag-1 term of source 0
#ifdef GH_STAMPS
    unsigned long long gh_seg[5] = {0, 0, 0, 0, 0}, gh_tprev = __builtin_amdgcn_s_memtime();
#endif
    int sh = 8 * 5;
    unsigned long long B = group_argmax<NODEL>(g0[bb + ga * ROW]);
#pragma unroll
    for (int l = 1; l < LC; l++) Y[0][l] = g0[bb + 5 * ROW + l * LT_ROW];
    double hyp = g0[bb + BLK + ga * ROW];   // lag 1 of target 2: source 1 under every hypothesis

    // LDS byte addresses are formed by hand: one v_mad_u32_u24 (row base + w * row bytes) right behind the
    // scalar resolve replaces the s_mul / s_add / v_add / v_add chain hipcc emits for the pointer form
    constexpr unsigned ROWB = ROW * 8, BLKB = BLK * 8;
    const unsigned lds0 = (unsigned)(uintptr_t)g0 + (unsigned)bb * 8u;
    unsigned rowb_v;
    asm("v_mov_b32 %0, %1" : "=v"(rowb_v) : "i"(ROWB));

    for (int k = 0; k < nchunks; k++) {
        unsigned vg = lds0 + (unsigned)(k & 1) * (unsigned)(C + WALK_OV) * (unsigned)RS * 8u;   // block of source k*C + g*LC
        unsigned long long *wk = words0 + (k & 1) * 64;
        const int ngroups = C / LC;
        asm volatile(".p2align 6");          // see spec2_walker
        for (int g = 0; g < ngroups; g++) {
            unsigned long long word = 0;
            const unsigned vgh = vg + (unsigned)ga * ROWB;
#pragma unroll
            for (int u = 0; u < LC; u++) {
                GH_STAMP(0);
                // A: resolve w_{j+1}   (body j = k*C + g*LC + u)
                const int w = (int)__builtin_ctzll(B >> sh);
                sh = 8 * w;
                word = (word << 4) | (unsigned long long)w;
                GH_STAMP(1);
                // S: finish the sum of target j+2 (lag l+1 comes from source j-(l-1)), l ascending
                double acc = hyp;
#pragma unroll
                for (int l = 1; l < LC; l++) acc += Y[(u - (l - 1) + LC) % LC][l];
                GH_STAMP(2);
                // R: row of source j+1 under its real symbol; lag-1 terms of target j+3 (source j+2)
                unsigned vrow;
                const unsigned vstep = vg + (unsigned)(u + 1) * BLKB;
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(vrow) : "s"(w), "v"(rowb_v), "v"(vstep));
                lds_cdouble *row = (lds_cdouble *)vrow;
#pragma unroll
                for (int l = 1; l < LC; l++) Y[(u + 1) % LC][l] = row[l * LT_ROW];
                hyp = *(lds_cdouble *)(vgh + (unsigned)(u + 2) * BLKB);
                GH_STAMP(3);
                // M: ballot of target j+2
                B = group_argmax<NODEL>(acc);
                GH_STAMP(4);
                // keep the next body's adds (which wait for the reads issued above) behind this arg-max:
                // an in-order wave that stalls on them early would serialise the whole chain
                __builtin_amdgcn_sched_barrier(0);
            }
            wk[g] = word;
            vg += (unsigned)LC * BLKB;
        }
        __syncthreads();
    }
#ifdef GH_STAMPS
    if (lane == 0)
        for (int q = 0; q < 5; q++) P.st->dbg8[q] = gh_seg[q];
#endif
}

// ---------------------------------------------------------------------------------------------
// spec2_walker: depth-2 speculation, for windows whose positions have at most 4 candidates each: k_lt then
// builds G over candidate ranks (2 bits each), so '-' may be among them.
// A lone wavefront issues one instruction every 5 cycles whatever the dependencies (scratch/ubench7),
// so a body costs 5 cycles x its instruction count once the dependency chain is long enough not to bind:
// this variant is about having few instructions per step.
// Lane = (a2 = lane>>4, a1 = (lane>>2)&3, b = lane&3): group (a2,a1) evaluates target t under the
// hypothesis w_{t-2} == a2, w_{t-1} == a1.  The loader waves do not stage G in LDS but two tables derived from it
// (copied from the Ht / Yt that k_lt maintains, or derived on the fly in batched launches):
//   H[t][lane]       = x1 + x2 = G[t-1][a1][lag 1][b] + G[t-2][a2][lag 2][b]   (path-independent; the first
//                      addition of the reference's lag-ascending sum; position 0 always contributes its '_' row)
//   Yr[i][w][b][l-3] = G[i][w][lag l][b], l = 3..L                              (read once w_i is resolved)
// Body j holds four stages that work on four different targets and do not depend on each other:
//   A  w_{j+1} = ffs(B_{j+1} >> 4*(4 w_{j-1} + w_j)) & 3; the shift is the low bits of the symbol history     5 SALU
//   M  B_{j+2} = group-wise arg-max of acc_{j+2}                          (acc from the previous body)         7 VALU
//   S  acc_{j+3} = H_{j+3} + Y_j[lag 3] + Y_{j-1}[lag 4] + ...            (rows read in earlier bodies)       L-2 VALU
//   R  issue the row of source j+1 under w_{j+1} and H_{j+4}                                       1 VALU + LDS reads
// The only cycles are A -> A and A -> R -> S -> M -> A, which spans three bodies.
// Same IEEE additions in the same order as k_walk_src: bit-identical.
// ---------------------------------------------------------------------------------------------
template <int LC>
struct deep_layout {
    static constexpr int NY = LC - 2;                  // lags taken from resolved rows
    static constexpr int NYP = NY > 0 ? ((NY + 1) & ~1) : 0;   // ... padded to an even count: rows are 16-byte aligned
    static constexpr int HPOS = 64;                    // doubles of H per target
    static constexpr int YPOS = 16 * NYP;              // doubles of Yr per source
    static constexpr int POS = HPOS + YPOS;
};

template <int LC>
__device__ __forceinline__ void spec2_walker(const walk_params &P, double *g0, unsigned long long *words0,
                                             int C, int RS, int nchunks, int lane)
{
    static_assert(LC >= 2, "depth-2 speculation needs two lags");
    typedef deep_layout<LC> DL;
    constexpr int NY = DL::NY;
    constexpr unsigned HB = DL::HPOS * 8, YB = DL::YPOS * 8, YWB = 4 * DL::NYP * 8;
    const int b = lane & 3;
    double Y[LC][LC];                       // Y[slot][l]: lag-(l+1) term of the source with index == slot (mod LC); l >= 2 used
#pragma unroll
    for (int u = 0; u < LC; u++)
#pragma unroll
        for (int l = 0; l < LC; l++) Y[u][l] = 0.0;

    const int npos = C + WALK_OV;
    const unsigned bufB = (unsigned)npos * (unsigned)RS * 8u;
    unsigned h0 = (unsigned)(uintptr_t)g0 + (unsigned)lane * 8u;
    unsigned y0 = (unsigned)(uintptr_t)g0 + (unsigned)npos * HB + (unsigned)b * (unsigned)DL::NYP * 8u;
    asm("" : "+v"(h0), "+v"(y0));          // opaque: otherwise hipcc rematerialises the LDS base (a null-check select) in every group

    // state entering body 0: targets 1 and 2 have no resolved lag yet; the row of source 0 is the '_' row in every slot
    unsigned long long B = group_argmax<true>(*(lds_cdouble *)(h0 + HB));
    double accP = *(lds_cdouble *)(h0 + 2 * HB);
    double H12 = *(lds_cdouble *)(h0 + 3 * HB);
#pragma unroll
    for (int l = 2; l < LC; l++) Y[0][l] = *(lds_cdouble *)(y0 + (unsigned)(l - 2) * 8u);
    unsigned hist = 0, sh = 0;
    unsigned yw_v;
    asm("v_mov_b32 %0, %1" : "=v"(yw_v) : "i"(YWB));

    for (int k = 0; k < nchunks; k++) {
        unsigned vh = h0 + (unsigned)(k & 1) * bufB, vy = y0 + (unsigned)(k & 1) * bufB;
        unsigned long long *wk = words0 + (k & 1) * 64;
        const int ngroups = C / LC;
        // the group loop starts on an instruction-fetch boundary: a lone wave at one instruction per 5 cycles has no slack
        // for a fetch that straddles two lines (unaligned: 124 cycles per step, aligned: 117)
        constexpr int UG = LC <= 8 ? 2 : 1;    // groups (of LC bodies, the register rotation period) per loop trip
        auto group = [&](int g, auto gg_) {
            constexpr int gg = decltype(gg_)::value;
#pragma unroll
            for (int u = 0; u < LC; u++) {
                // A: resolve w_{j+1}   (body j = k*C + g*LC + u); hist keeps 2 bits per symbol, newest lowest
                const unsigned w = (unsigned)__builtin_ctzll(B >> (sh & 63u)) & 3u;
                hist = (hist << 2) + w;
                sh = hist << 2;
                // M: ballot of target j+2
                B = group_argmax<true>(accP);
                // S: target j+3, lag l+1 from source j-(l-2), l ascending
                double acc = H12;
#pragma unroll
                for (int l = 2; l < LC; l++) acc += Y[(u - (l - 2) + 2 * LC) % LC][l];
                accP = acc;
                // R: row of source j+1 under its real symbol (lags 3..L); x1 + x2 of target j+4
                if constexpr (NY > 0) {
                    unsigned vrow;
                    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(vrow) : "s"(w), "v"(yw_v), "v"(vy));
                    // 16-byte reads (their offset field is 16 bits wide; ds_read2_b64 would need an address add per read
                    // once (u + 1) * YB passes 2040 bytes)
                    const unsigned rb = vrow + (unsigned)(gg * LC + u + 1) * YB;
#pragma unroll
                    for (int l = 2; l + 1 < LC; l += 2) {
                        const lds_v2d pr = *(const __attribute__((address_space(3))) lds_v2d *)(rb + (unsigned)(l - 2) * 8u);
                        Y[(u + 1) % LC][l] = pr.x;
                        Y[(u + 1) % LC][l + 1] = pr.y;
                    }
                    if constexpr (NY & 1) Y[(u + 1) % LC][LC - 1] = *(lds_cdouble *)(rb + (unsigned)(NY - 1) * 8u);
                }
                H12 = *(lds_cdouble *)(vh + (unsigned)(gg * LC + u + 4) * HB);
                        }
            wk[g] = (unsigned long long)hist;
        };
        int g = 0;
        asm volatile(".p2align 6");
        for (; g + UG <= ngroups; g += UG) {
            group(g, std::integral_constant<int, 0>{});
            if constexpr (UG > 1) group(g + 1, std::integral_constant<int, 1>{});
            vh += (unsigned)(UG * LC) * HB;
            vy += (unsigned)(UG * LC) * YB;
        }
        for (; g < ngroups; g++) {
            group(g, std::integral_constant<int, 0>{});
            vh += (unsigned)LC * HB;
            vy += (unsigned)LC * YB;
        }
        __syncthreads();
    }
}

template <int LC>
__global__ void __launch_bounds__(512) k_walk_spec(walk_params P, const win_desc *wd, int spin)
{
    extern __shared__ __align__(16) double smem[];
    if (wd) {
        const win_desc &d = wd[blockIdx.x];
        P.G = d.G; P.Ht = d.Ht; P.Yt = d.Yt; P.minfo = d.minfo; P.st = d.st;
        P.path_out = d.paths + (size_t)spin * (P.N + 1); P.rec = d.recs + spin;
    }
    dev_state *st = P.st;
    if (st->stop) return;
    constexpr int ROW = LC * LT_ROW;
    constexpr int BLK = 6 * ROW;
    double *const g0 = smem;

    // (readfirstlane: the flags are the same for every lane, but loaded from memory the kernel also writes, so the
    // compiler would treat everything derived from them -- chunk size, loop bounds, the scalar resolve -- as divergent)
    const int first_hole = __builtin_amdgcn_readfirstlane(st->first_hole);
    const bool nodel = __builtin_amdgcn_readfirstlane(st->nodel) != 0;
    // which walker, loader and word format.  G is ranked (k_lt) exactly when the host allowed it (same
    // condition as P.depth2) and no position has more than 4 candidates: then the 4-symbol depth-2 walker applies.
    const bool deep = LC >= 2 && __builtin_amdgcn_readfirstlane(st->ranked) != 0 && P.depth2 && blockDim.x == 512;
    const int RS = walk_pos_doubles(LC, deep);      // doubles per position in an LDS buffer
    const int C = walk_chunk(LC, deep);             // positions per chunk (the host sized the LDS for either variant)
    unsigned long long *const words0 = reinterpret_cast<unsigned long long *>(smem + 2 * (size_t)(C + WALK_OV) * RS);
    const int Nw = first_hole <= P.N ? first_hole - 1 : P.N;      // positions that can be decided
    const int nchunks = (Nw + C - 1) / C;                         // bodies 0..Nw-1, chunk k = k*C..k*C+C-1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // The loader waves' pipeline, whatever they load: chunk k+1 goes from registers to LDS (paced) while the walker is in
    // chunk k, then the loads of chunk k+2 are issued and stay in flight across the barrier (their latency must not sit
    // between two barriers: the walker waits there too).
    auto loader_loop = [&](auto &fetch, auto &store, auto &set) {
        fetch(0, set);
        store(0, set, std::false_type{});                    // nobody is walking yet
        if (nchunks > 1) fetch(1, set);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int k = 0; k < nchunks; k++) {
            if (k + 1 < nchunks) store(k + 1, set, std::true_type{});     // empties the registers fetch(k + 2) fills
            if (k + 2 < nchunks) fetch(k + 2, set);
            // LDS stores done, global loads still in flight: no vmcnt wait here (a fence would add one)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    };

    if (deep) {
        if constexpr (LC >= 2) {
            // depth-2 layout.  With the tables of k_lt (single windows) the 384 loader threads copy the chunk's slices of
            // Ht and Yt into the buffer: 16-byte copies, ~50 instructions per wave and chunk.  Without them (batched
            // launches, where refreshing the tables for every window costs more HBM traffic than it saves the walkers)
            // they derive H and Yr from G themselves.  Either way the loads of chunk k+2 are issued before the barrier
            // that ends chunk k and stay in flight across it (their latency must not sit between two barriers, the
            // walker waits there too); the LDS stores of chunk k+1 happen while the walker is in chunk k.
            typedef deep_layout<LC> DL;
            constexpr int NT = 384, MAXPOS = walk_chunk(LC, true) + WALK_OV;      // = C + WALK_OV
            if (wave >= 2) {
                const int t = tid - 128;
                const int npos = C + WALK_OV;
                if (P.Ht) {
                    constexpr int NVH = MAXPOS * DL::HPOS / 2, NVY = MAXPOS * DL::YPOS / 2;    // double2 per chunk
                    constexpr int MAXH = (NVH + NT - 1) / NT, MAXY = (NVY + NT - 1) / NT;
                    // s_sleep units (64 cycles) between two stores: the store phase takes about 60 % of what the
                    // walker needs for the chunk (~88 + 7 L cycles per step)
#ifdef WALK_PACE
                    constexpr int PACE = WALK_PACE;
#else
                    constexpr int PACE = (walk_chunk(LC, true) * (88 + 7 * LC) * 62 / 100) / ((MAXH + MAXY) * 64);
#endif
                    struct regs { lds_v2d vh[MAXH], vy[MAXY > 0 ? MAXY : 1]; } set;
                    auto fetch = [&](int k, regs &R) {
                        const lds_v2d *srcH = reinterpret_cast<const lds_v2d *>(P.Ht + (size_t)k * C * DL::HPOS);
#pragma unroll
                        for (int it = 0; it < MAXH; it++) {
                            const int q = t + it * NT;
                            if (q < NVH) R.vh[it] = srcH[q];
                        }
                        if constexpr (MAXY > 0) {
                            const lds_v2d *srcY = reinterpret_cast<const lds_v2d *>(P.Yt + (size_t)k * C * DL::YPOS);
#pragma unroll
                            for (int it = 0; it < MAXY; it++) {
                                const int q = t + it * NT;
                                if (q < NVY) R.vy[it] = srcY[q];
                            }
                        }
                    };
                    // the 64 KB of a chunk must not reach LDS in one burst: the walker's row reads would queue behind ~70 wide
                    // writes once per chunk (measured 5-9 cycles per step); s_sleep spreads the stores over the chunk
                    auto store = [&](int k, regs &R, auto pace) {      // pace: std::true_type / false_type (a run-time flag here costs the walker 5 cycles per step)
                        lds_v2d *dstH = reinterpret_cast<lds_v2d *>(g0 + (size_t)(k & 1) * npos * RS);
#pragma unroll
                        for (int it = 0; it < MAXH; it++) {
                            const int q = t + it * NT;
                            if (q < NVH) dstH[q] = R.vh[it];
                            if constexpr (decltype(pace)::value && PACE > 0) __builtin_amdgcn_s_sleep(PACE);
                        }
                        if constexpr (MAXY > 0) {
                            lds_v2d *dstY = dstH + NVH;
#pragma unroll
                            for (int it = 0; it < MAXY; it++) {
                                const int q = t + it * NT;
                                if (q < NVY) dstY[q] = R.vy[it];
                                if constexpr (decltype(pace)::value && PACE > 0) __builtin_amdgcn_s_sleep(PACE);
                            }
                        }
                    };
                    loader_loop(fetch, store, set);
                } else {
                    constexpr int MAXH = (MAXPOS * DL::HPOS + NT - 1) / NT;
                    constexpr int MAXY = DL::YPOS ? (MAXPOS * DL::YPOS + NT - 1) / NT : 0;
                    const int nsrc_all = P.N + LT_PAD;                    // source blocks G holds (the last LT_PAD are zeros)
                    const int nh = npos * DL::HPOS, ny = npos * DL::YPOS;
                    const int bb = t & 3, a1 = (t >> 2) & 3, a2 = (t >> 4) & 3;    // NT is a multiple of 64: lane-constant
                    struct regs { double x1[MAXH], x2[MAXH], yv[MAXY > 0 ? MAXY : 1]; } set;
                    auto fetch = [&](int k, regs &R) {
                        const int i0 = k * C;
#pragma unroll
                        for (int it = 0; it < MAXH; it++) {
                            const int q = t + it * NT;
                            const int tt = i0 + (q >> 6);
                            R.x1[it] = 0.0; R.x2[it] = 0.0;
                            if (q < nh && tt >= 1 && tt - 1 < nsrc_all) {
                                const int s1 = tt - 1, s2 = tt - 2;
                                R.x1[it] = P.G[(size_t)s1 * BLK + (s1 == 0 ? 5 : a1) * ROW + bb];
                                if (tt >= 2) R.x2[it] = P.G[(size_t)s2 * BLK + (s2 == 0 ? 5 : a2) * ROW + LT_ROW + bb];
                            }
                        }
                        if constexpr (MAXY > 0) {
#pragma unroll
                            for (int it = 0; it < MAXY; it++) {
                                const int q = t + it * NT;
                                R.yv[it] = 0.0;
                                if (q < ny) {
                                    const int p = q / DL::YPOS, r = q % DL::YPOS, wb = r / DL::NYP, l = 2 + r % DL::NYP;
                                    const int sidx = i0 + p;
                                    if (sidx < nsrc_all && l < LC)
                                        R.yv[it] = P.G[(size_t)sidx * BLK + (sidx == 0 ? 5 : (wb >> 2)) * ROW + l * LT_ROW + (wb & 3)];
                                }
                            }
                        }
                    };
                    auto store = [&](int k, regs &R, auto) {
                        const int i0 = k * C;
                        double *dst = g0 + (size_t)(k & 1) * npos * RS;
#pragma unroll
                        for (int it = 0; it < MAXH; it++) {
                            const int q = t + it * NT;
                            // target 1 has the single term x1 (not 0.0 + x1: the reference starts from the first addend)
                            if (q < nh) dst[q] = (i0 + (q >> 6) >= 2) ? R.x1[it] + R.x2[it] : R.x1[it];
                        }
                        if constexpr (MAXY > 0) {
                            double *yr = dst + (size_t)npos * DL::HPOS;
#pragma unroll
                            for (int it = 0; it < MAXY; it++) {
                                const int q = t + it * NT;
                                if (q < ny) yr[q] = R.yv[it];
                            }
                        }
                    };
                    loader_loop(fetch, store, set);
                }
                return;
            }
            __syncthreads();
        }
    } else {
        // depth-1 layout: the G blocks of positions k*C .. k*C + C + WALK_OV - 1, copied as they are (16-byte copies)
        constexpr int NT = 384, MAXPOS = walk_chunk(LC, false) + WALK_OV;
        constexpr int NV = MAXPOS * BLK / 2, MAXV = (NV + NT - 1) / NT;          // double2 per chunk / per thread
        constexpr int PACE = (walk_chunk(LC, false) * (110 + 9 * LC) * 62 / 100) / (MAXV * 64);
        if (wave >= 2 && blockDim.x == 512) {
            const int t = tid - 128;
            const int npos = C + WALK_OV;
            struct regs { lds_v2d v[MAXV]; } set;
            auto fetch = [&](int k, regs &R) {
                const int i0 = k * C;
                int nsrc = P.N + LT_PAD - i0;                // what lies behind the table must read as 0.0 (finite sums,
                if (nsrc > npos) nsrc = npos;                // symbol 0 wins): the walker always runs whole chunks
                if (nsrc < 0) nsrc = 0;
                const int nv = nsrc * BLK / 2;
                const lds_v2d *src = reinterpret_cast<const lds_v2d *>(P.G + (size_t)i0 * BLK);
#pragma unroll
                for (int it = 0; it < MAXV; it++) {
                    const int q = t + it * NT;
                    R.v[it] = lds_v2d{0.0, 0.0};
                    if (q < nv) R.v[it] = src[q];
                }
            };
            auto store = [&](int k, regs &R, auto pace) {
                lds_v2d *dst = reinterpret_cast<lds_v2d *>(g0 + (size_t)(k & 1) * npos * RS);
#pragma unroll
                for (int it = 0; it < MAXV; it++) {
                    const int q = t + it * NT;
                    if (q < NV) dst[q] = R.v[it];
                    if constexpr (decltype(pace)::value && PACE > 0) __builtin_amdgcn_s_sleep(PACE);
                }
            };
            loader_loop(fetch, store, set);
            return;
        }
        if (blockDim.x != 512) {
            // other workgroup sizes (GH_WALK_THREADS, A/B measurements): every thread copies, no pipelining
            auto load_chunk = [&](int k, int t, int nt) {
                const int i0 = k * C;
                const int npos = C + WALK_OV;
                double *dst = g0 + (size_t)(k & 1) * npos * RS;
                int nsrc = P.N + LT_PAD - i0;
                if (nsrc > npos) nsrc = npos;
                if (nsrc < 0) nsrc = 0;
                copy_to_lds(dst, P.G + (size_t)i0 * BLK, (size_t)nsrc * BLK, t, nt);
                for (int q = nsrc * BLK + t; q < npos * BLK; q += nt) dst[q] = 0.0;
            };
            load_chunk(0, tid, (int)blockDim.x);
            __syncthreads();
            if (wave >= 2) {
                for (int k = 0; k < nchunks; k++) {
                    if (k + 1 < nchunks) load_chunk(k + 1, tid - 128, (int)blockDim.x - 128);
                    __syncthreads();
                }
                return;
            }
        } else {
            __syncthreads();        // the loaders' first barrier
        }
    }
    if (wave == 1) {
        walk_totals T = {0.0, 0.0, INFINITY};
        const int wbits = deep ? 2 : 4;
        double lane_min = INFINITY;
        book_row R0, R1;                    // chunk c lives in R[c & 1]
#pragma unroll
        for (int q = 0; q < 8; q++) { R0.v[q] = lds_v2d{0.0, 0.0}; R1.v[q] = lds_v2d{0.0, 0.0}; }
        if (lane == 0) P.path_out[0] = SYM_US;
        // iteration k (while the walker is in chunk k): prefetch the rows of chunk k, consume chunk k-1
        auto consume = [&](int c, const book_row &R) {
            book_consume(P, words0 + (c & 1) * 64, LC, wbits, deep, c * C, C, Nw, lane, R, T, lane_min);
        };
        for (int k = 0; k < nchunks; k += 2) {
            book_prefetch(P, k * C, C, Nw, lane, R0);
            if (k >= 1) consume(k - 1, R1);
            // the word reads are done, the prefetch's global loads stay in flight across the barrier (no vmcnt wait)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (k + 1 < nchunks) {
                book_prefetch(P, (k + 1) * C, C, Nw, lane, R1);
                consume(k, R0);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        if (nchunks > 0) {
            if (nchunks & 1) consume(nchunks - 1, R0);
            else consume(nchunks - 1, R1);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(lane_min, off);
            if (o < lane_min) lane_min = o;
        }
        T.minm = lane_min;
        if (lane == 0) {
            if (first_hole <= P.N) {                                  // gretel.py:176-180
                st->stop = 1;
                st->hole_at = first_hole;
            } else {
                double r = T.minm;
                if (r < P.min_remove) r = P.min_remove;               // cmd.py:157-160
                P.rec->hp_current = T.hp_cur;
                P.rec->hp_original = T.hp_orig;
                P.rec->ratio = r;                                     // the ratio the reweight will use (clamped)
                P.rec->min_marginal = T.minm;
                P.rec->magnitude = 0.0;
                st->ratio = r;
                st->n_done += 1;
                if (P.rearm) { st->first_hole = 0x7f7f7f7f; st->nodel = 0x7f7f7f7f; st->cm_same = 0x7f7f7f7f; st->narrow = 0x7f7f7f7f; }
            }
        }
        return;
    }
    __builtin_amdgcn_s_setprio(3);          // the walker is the critical path; loaders and bookkeeper have slack
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (deep) {
        if constexpr (LC >= 2) spec2_walker<LC>(P, g0, words0, C, RS, nchunks, lane);
    } else if (nodel) spec_walker<LC, true>(P, g0, words0, C, RS, nchunks, lane);
    else spec_walker<LC, false>(P, g0, words0, C, RS, nchunks, lane);
    if (lane == 0) {
        st->dbg[0] = __builtin_amdgcn_s_memtime() - t0;          // shader cycles of the walk
        st->dbg[1] = __builtin_amdgcn_s_memrealtime() - r0;      // 100 MHz ticks of the walk
        st->dbg[2] = (unsigned long long)nchunks * C;            // steps executed
        st->dbg[3] = deep ? 2 : (nodel ? 1 : 0);                 // variant: speculation depth 2 / 1 without '-' / 1 with '-'
    }
}

// ---------------------------------------------------------------------------------------------
// k_walk_global: the same walk with G read straight from global memory by one wavefront.
// Fallback for L > 16 (register rotation no longer fits); not a fast path.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_walk_global(walk_params P, int hist_len, const win_desc *wd, int spin)
{
    extern __shared__ uint8_t lpath[];
    if (wd) {
        const win_desc &d = wd[blockIdx.x];
        P.G = d.G; P.minfo = d.minfo; P.st = d.st;
        P.path_out = d.paths + (size_t)spin * (P.N + 1); P.rec = d.recs + spin;
    }
    dev_state *st = P.st;
    if (st->stop) return;
    const int lane = threadIdx.x;
    const int hmask = hist_len - 1;
    const int L = P.L;
    const size_t ROW = (size_t)L * LT_ROW, BLK = 6 * ROW;
    const int first_hole = st->first_hole;
    const bool ranked = st->ranked != 0;
    const int Nw = first_hole <= P.N ? first_hole - 1 : P.N;
    double hp_cur = 0.0, hp_orig = 0.0, minm = INFINITY;
    const int bb = (lane & 7) < 5 ? (lane & 7) : 0;

    if (lane == 0) { lpath[0] = 5; P.path_out[0] = SYM_US; }
    __syncthreads();

    for (int snp = 1; snp <= Nw; snp++) {
        const int lmax = L < snp ? L : snp;
        double acc = 0.0;
        for (int l0 = 1; l0 <= lmax; l0 += 8) {
            double x[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int l = l0 + q;
                x[q] = 0.0;
                if (l <= lmax)
                    x[q] = P.G[(size_t)(snp - l) * BLK + lpath[(snp - l) & hmask] * ROW + (size_t)(l - 1) * LT_ROW + bb];
            }
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (l0 + q <= lmax) acc += x[q];
        }
        const int w = __builtin_amdgcn_readfirstlane(argmax8(acc));
        const double *inf = P.minfo + (size_t)snp * MINFO;
        // ranked table (k_lt): rows and columns are candidate ranks; the symbol comes back through the candidate bits
        int b5 = w;
        if (ranked) {
            b5 = nth_set5((uint32_t)__double_as_longlong(inf[10]), w);
            if (b5 < 0) b5 = 0;
        }
        const double mg = inf[5 + b5];
        if (mg < minm) minm = mg;
        hp_cur += inf[b5];
        hp_orig += inf[11 + b5];
        if (lane == 0) {
            lpath[snp & hmask] = (uint8_t)w;
            P.path_out[snp] = (uint8_t)vsym(P.sm, b5);
        }
        __syncthreads();
    }
    if (lane == 0) {
        if (first_hole <= P.N) {
            st->stop = 1;
            st->hole_at = first_hole;
        } else {
            double r = minm;
            if (r < P.min_remove) r = P.min_remove;
            P.rec->hp_current = hp_cur;
            P.rec->hp_original = hp_orig;
            P.rec->ratio = r;
            P.rec->min_marginal = minm;
            P.rec->magnitude = 0.0;
            st->ratio = r;
            st->n_done += 1;
            if (P.rearm) { st->first_hole = 0x7f7f7f7f; st->nodel = 0x7f7f7f7f; st->cm_same = 0x7f7f7f7f; st->narrow = 0x7f7f7f7f; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// reweighting: gretel/gretel.py:79-98 restricted to the band (all other cells are zero and stay
// zero) lives in k_marg<T, true>.  Multiplicities of the reference's pair enumeration:
//   (p,p+1), p <= N-2 : twice      (N-1,N) : once      (p,q), q-p>=2, q <= N-1 : once
//   (p,N), p < N-1    : never      (N,N+1) with symbols (path[N], path[0]) : once
// k_reweight_finish adds the per-block partial sums of the removed mass in a fixed order (a per-block
// __threadfence + ticket inside k_marg was tried: the agent-scope fences cost 2.5x in batched runs).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_reweight_finish(const double *partial, int nb, dev_state *st, int use_state,
                  gh_path_rec *rec, const win_desc *wd, int spin)
{
    __shared__ double s_red[256];
    if (wd) {
        const win_desc &d = wd[blockIdx.x];
        partial = d.partial; st = d.st; rec = d.recs + spin;
    }
    if (use_state && st->stop) return;
    double acc = 0.0;
    for (int q = threadIdx.x; q < nb; q += 256) acc += partial[q];
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) s_red[threadIdx.x] += s_red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        rec->magnitude = s_red[0];
        if (use_state) rec->ratio = st->ratio;
    }
}

// the same reduction for every path of a spin at once (block s = path s, partial sums kept per path): the removed
// mass is only read by the host, so gh_spin defers it to one launch behind the loop instead of one per path
__global__ void __launch_bounds__(256)
k_reweight_finish_all(const double *partial, int nb, const dev_state *st, gh_path_rec *recs)
{
    __shared__ double s_red[256];
    const int s = blockIdx.x;
    if (s >= st->n_done) return;            // a hole ended the recovery before this path
    const double *part = partial + (size_t)s * nb;
    double acc = 0.0;
    for (int q = threadIdx.x; q < nb; q += 256) acc += part[q];
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s_red[threadIdx.x] += s_red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) recs[s].magnitude = s_red[0];
}

// one-cell helpers ----------------------------------------------------------------------------
template <typename T>
__global__ void k_reweight_one(T *p, double ratio, double *removed)
{
    const double old = (double)*p;
    const double nw = old - ratio * old;
    *p = (T)nw;
    *removed = old - nw;
}

// the exported / imported tensor keeps the distance-major order [(N+2)][W][7][7] (cell (i, i+d) as 49 values)
// the band once more, TO-major: tb[bidx(W, p, d, b, a)] = band[bidx(W, p, d, a, b)] (per position the symbol pair transposed) --
// k_rw under the column conditionals reads a cell's COLUMN as one contiguous run there (gretel/gretel.py:79-98 under a
// conditional whose denominator is a column sum)
template <typename T>
__global__ void __launch_bounds__(256) k_band_to_major(const T *__restrict__ band, T *__restrict__ tb, size_t n, int W)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;      // element of the copy: [p][b][d][a]
    if (e >= n) return;
    const size_t pe = (size_t)NSYM * W * NSYM, p = e / pe;
    const int r = (int)(e - p * pe), a = r % NSYM, d1 = (r / NSYM) % W, b = r / (NSYM * W);
    tb[e] = band[p * pe + ((size_t)a * W + d1) * NSYM + b];
}

template <typename T>
__global__ void k_export(const T *__restrict__ band, double *__restrict__ out, size_t n, int W)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int b = (int)(t % NSYM), a = (int)((t / NSYM) % NSYM), d = (int)((t / CELL) % W) + 1;
    const size_t i = t / ((size_t)CELL * W);
    out[t] = (double)band[bidx(W, i, d, a, b)];
}

template <typename T>
__global__ void k_import(T *__restrict__ band, const double *__restrict__ in, size_t n, int W)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int b = (int)(t % NSYM), a = (int)((t / NSYM) % NSYM), d = (int)((t / CELL) % W) + 1;
    const size_t i = t / ((size_t)CELL * W);
    band[bidx(W, i, d, a, b)] = (T)in[t];
}

// hansel get_edge_weights_at for an arbitrary host-supplied history (compat API; one wave,
// straight from the band so that any symbol -- N included -- may sit in the history)
template <typename T>
__global__ void k_edge_weights(const T *__restrict__ band, int W, int cond_mode, int p, int L, int marginal_term,
                               const double *__restrict__ cnt, const double *__restrict__ marg,
                               const int32_t *__restrict__ nvalid, const uint32_t *__restrict__ cmask,
                               const uint8_t *__restrict__ hist /* hist[l-1] = path[p-l] */,
                               double *w, int *mask)
{
    const int lane = threadIdx.x;
    const uint32_t cm = CM_CAND(cmask[p]);
    if (lane == 0) *mask = (int)cm;
    if (lane >= NSYM) return;
    double acc = 0.0;
    if ((cm >> lane) & 1) {
        if (marginal_term) acc += gh_log10(marg[(size_t)p * 8 + lane]);
        const int lmax = L < p ? L : p;
        for (int l = 1; l <= lmax; l++)
            acc += log_conditional(band, W, cond_mode, cnt, nvalid, hist[l - 1], lane, p - l, p);
    }
    w[lane] = acc;
}

__global__ void k_gap(const double *cnt, int N, int *first_gap)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > N) return;
    if (cnt[(size_t)p * 8 + 7] == 0.0) atomicMin(first_gap, p);
}

// ---------------------------------------------------------------------------------------------
// gretel-snpper (gretel/snpper.py:29-50) as a histogram: k_cov counts, per position of the window, the reads showing
// A, C, G, T (one wavefront per aligned run, lanes stride over its bases, integer atomics); k_sites marks the
// positions where more than one base is seen on more than `depth` reads (snpper.py:38-40).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_cov(const int32_t *__restrict__ ref_start, const int64_t *__restrict__ off, const uint8_t *__restrict__ codes,
      int64_t n_runs, int32_t start0, int32_t len, unsigned *__restrict__ counts /* [4][len] */)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_runs; r += nw) {
        const int64_t o0 = off[r], o1 = off[r + 1];
        const int32_t p0 = ref_start[r] - start0;
        for (int64_t q = o0 + lane; q < o1; q += 64) {
            const int c = codes[q];
            const int64_t p = p0 + (q - o0);
            if (c < 4 && p >= 0 && p < len) atomicAdd(&counts[(size_t)c * len + p], 1u);
        }
    }
}

// (depth compares as the reference's `counts > depth` does, signed: a negative depth marks every position a site, snpper.py:38-40)
__global__ void k_sites(const unsigned *__restrict__ counts, int32_t len, int32_t depth, uint8_t *__restrict__ site)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= len) return;
    int n = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) n += (long long)counts[(size_t)c * len + p] > (long long)depth ? 1 : 0;
    site[p] = n > 1 ? 1 : 0;
}

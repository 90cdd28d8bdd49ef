// cwalk.hpp -- segment-parallel path extension for lag counts whose state space cannot be enumerated (included by
// gretel_hip.hip behind segwalk.hpp, which enumerates up to 5^5 states, and 4^6 when the table is ranked).
// Ranked tables (every position has at most four candidates): 2 bits per pick, L = 6..32.  Tables over the symbols
// A C G T - (a five-candidate position somewhere): 3 bits per pick, L = 6..21.  Beyond (up to 128 lags): k_cwalkg, states as
// bytes next to their hash.  Both keep the slice of the table a chunk of targets reads in LDS, target-major (round 4).
//
// Same decomposition -- cut the window into <= 512 segments (<= 256 beyond 13 lags), know for every segment what it does
// to the state that enters it, chain the segments -- but a segment is only walked from a POOL of candidate entry states
// (<= 64 per segment, kept from path to path), and the chain is then VERIFIED, never assumed:
//
//   k_cwalk   one workgroup per segment, four lanes (one per candidate rank; eight per entry over the symbols) per pool
//             entry that has not been walked under the current tensor (and, "run-on", up to CW_RUNON segments further for
//             an entry whose exit state the next pool does not hold: its walk there arrives with the request to join): the same lag-ascending binary64 sums and
//             first-wins arg-max as every other walker, from the conditional table staged in LDS; records the exit
//             state and the picks.
//   k_clink   (behind k_cwalk, the pools stand still) where every entry's exit state sits in the next segment's pool.
//   k_cscan   one workgroup: chains from the start state along the links.  The chain either reaches the end -- every
//             hop an exact pool hit onto a walked entry -- or stops at a state that still has to be walked (an exit
//             that is not in the next pool is inserted there).
//   (repeat k_cwalk / k_clink / k_cscan for the pending entries: each round walks only what is new)
//   k_cemit   one workgroup per segment: the picks of the entry that is on the verified chain -> path symbols, selected
//             log-marginals, minimum marginal (as k_emit).
//
// Nothing approximate survives: a path is emitted only when every segment's entry state equals the previous segment's
// exit state, both produced by exact walks.  When the rounds queued for a path do not close the chain, the kernels
// behind flag the path unresolved and idle; the host queues more rounds for it (every round carries the verified chain
// at least one segment further) or, up to 16 lags, hands it to the serial walker, whose states then join the pools.
// The pools start from the largest marginals (k_cguess): only where walks START -- never what is emitted -- depends on
// that.  (Round 2 also carried a kernel that started them from the states the reads show; measured, it closed no chain
// earlier -- after the first handful of paths a deep spin follows chimeras of what the reweights left -- and it is gone.)
#pragma once

#define CW_K 64                 /* pool entries per segment */
#define CW_MIN_L 6
#ifndef CW_CH_CAP
#define CW_CH_CAP 40
#endif
#define CW_MAX_L 32             /* 2 bits per pick in a 64-bit state; 32 targets x 32 lags x 4 x 4 doubles are 128 KB of LDS */
typedef unsigned long long cw_key;
#define CW_MAX_SEG 512
#define CW_MIN_LEN 32
#define CW_BOOT_ROUNDS 8        /* rounds queued for the first path of a tensor (pools started from k_cguess) */
#define CW_RUNON 8              /* segments a walker may run on into, behind its own, with an exit state the next pool does not hold */

// f(integral_constant<0>), f(integral_constant<1>), ...: a run of steps with their index as a compile-time constant
template <typename F, int... U>
__device__ __forceinline__ void cw_unrolled(F &&f, std::integer_sequence<int, U...>) { (f(std::integral_constant<int, U>{}), ...); }

// Segments: a wavefront alone on its SIMD issues an instruction every ~5 cycles, so where the table slice of a chunk
// leaves room for two workgroups per CU (L <= 13: 76 KB each) the window is cut into twice as many, half as long
// segments, and every SIMD has two walkers to interleave.
__host__ __device__ constexpr int cw_max_seg(int L) { return L <= 13 ? 512 : 256; }
struct cw_geom { int seglen, S, NW, NW5; };               // words of picks per entry: 16 per word (ranks), 8 per word (symbols)
__host__ __device__ inline cw_geom cw_geometry(int N, int L)
{
    cw_geom g;
    const int smax = cw_max_seg(L);
    int len = (N + smax - 1) / smax;
    if (len < CW_MIN_LEN) len = CW_MIN_LEN;
    g.seglen = len;
    g.S = (N + len - 1) / len;
    g.NW = (len + 15) / 16;
    g.NW5 = (len + 7) / 8;
    return g;
}
// R = 4: the ranked table (every position has at most four candidates), 2 bits per pick, four lanes per pool entry.
// R = 5: the table over the symbols A C G T - (a position with five candidates somewhere), 3 bits per pick -- 21 lags in a
// 64-bit state --, eight lanes per entry (five at work), five rows per source in LDS.
#define CW_MAX_L5 21
__host__ __device__ constexpr int cw_lanes(int R) { return R == 4 ? 4 : 8; }
__host__ __device__ constexpr int cw_bits(int R) { return R == 4 ? 2 : 3; }
// The slice of G a chunk of targets needs, in LDS TARGET-major (round 4): Gs[tl][lag - 1][row][col] = G[t - lag][row][lag - 1][col]
// for the chunk's targets t = c0 + 1 + tl -- exactly the (source, lag) pairs the chunk reads, CH x L blocks of rows x columns.
// (Rounds 2-3 copied whole source blocks as they lie in G -- (CH + L - 1) sources x L lags, of which a chunk reads the diagonal
// band only: at 24 lags 16 targets per chunk, fewer than the L steps the unrolled walk takes at a time, and nothing beyond 24.)
// rows / columns kept: the four ranks of the ranked table; the five symbols A C G T - of the symbol table
__host__ __device__ constexpr int cw_rows(int R) { return R == 4 ? 4 : 5; }
__host__ __device__ constexpr int cw_cols(int R) { return R == 4 ? 4 : 5; }
__host__ __device__ constexpr int cw_lds_budget(int L, int R)
{
    return R == 4 ? (L <= 13 ? 76 * 1024 : (L <= 17 ? 96 * 1024 : (L <= 19 ? 120 * 1024 : 150 * 1024)))
                  : (L <= 12 ? 76 * 1024 : (L <= 13 ? 96 * 1024 : (L <= 15 ? 120 * 1024 : 150 * 1024)));
}
// targets per LDS chunk of k_cwalk (+ 5 doubles per target: its log-marginals, for windows walked with the marginal term)
__host__ __device__ constexpr int cw_chunk(int L, int R)
{
    int c = cw_lds_budget(L, R) / ((L * cw_rows(R) * cw_cols(R) + 5) * 8);
    // (the next chunk's slice waits in registers under the walk: from 18 lags on no more targets than a segment of the
    // usual length has, or the walk's own registers go through the accumulator file)
    if (L >= 18 && c > CW_CH_CAP) c = CW_CH_CAP > L ? CW_CH_CAP : L;
    c = c > 64 ? 64 : (c < 2 ? 2 : c);
    // (round 6) whole blocks of L steps, at most 48 targets: a chunk is walked as ONE unrolled run of steps whose slot in the
    // register rotation and whose LDS offsets are compile-time constants -- a chunk boundary in the middle of a block would
    // need a second unrolling per phase.  (Every lag count here leaves room for at least one block.)
    if (c >= L) { c = c / L * L; while (c > 48 && c > L) c -= L; }
    return c;
}
__host__ __device__ constexpr size_t cw_lds_bytes(int L, int R) { return (size_t)cw_chunk(L, R) * (L * cw_rows(R) * cw_cols(R) + 5) * 8; }

// k_cwalkg: targets per chunk and LDS at a lag count known at run time
__host__ __device__ inline int cwg_chunk(int L, int R)
{
    int c = (140 * 1024) / ((L * cw_rows(R) * cw_cols(R) + 5) * 8);      // (16 KB of static LDS beside it: the doubled rings of picks)
    return c > 64 ? 64 : (c < 1 ? 1 : c);
}
__host__ __device__ inline size_t cwg_lds_bytes(int L, int R) { return (size_t)cwg_chunk(L, R) * (L * cw_rows(R) * cw_cols(R) + 5) * 8; }

struct cw_params {
    int N, L;
    int rearm;                // spin loops: k_cemit re-arms first_hole/nodel/cm_same/narrow for the k_rw that follows
    int check_masks;          // as seg_params
    int round;                // 0 = first round of a path; later rounds idle once the chain has closed
    int last_round;           // k_cscan: an open chain after this round flags the path unresolved
    int stamp;                // path counter of the spin (pool entries remember when they were last on a chain)
    int _pad;
    const double *G;          // ranked or over the symbols (st->ranked): k_cwalk<LC, 4> / k_cwalk<LC, 5>
    const double *minfo;
    const double *rinfo;      // [N+2][8]: log10 marginal / marginal by candidate rank
    int mt;                   // gh_config.marginal_term: log10 marginal(b, t) in front of x1 (see k_lt)
    symmap sm;
    dev_state *st;
    cw_key *keys, *exits;     // [S][CW_K]
    int32_t *last_hit;        // [S][CW_K]
    int32_t *npool;           // [S]
    cw_key *pend;             // [S][CW_K]: states waiting to join the pool (exits of the previous segment's walks)
    int32_t *npend;           // [S]
    // run-on (k_cwalk): a walker whose exit state the next pool does not hold walks on into the next segment itself and leaves
    // the RESULT with the request -- the owner then merges a walked entry, and a new track is discovered in one launch
    // instead of one segment per round
    cw_key *pend_exit;        // [S][CW_K] exit state of a pending entry that arrives walked
    int32_t *pend_ready;      // [S][CW_K] the path (P.stamp) under whose tensor pend_exit / phist were walked; 0 = a bare request.
                              // (a request may outlive its path -- the chain closed before its pool merged again: its walk is then stale)
    uint32_t *phist;          // [S][NW][CW_K] its picks
    // The request lists are double-buffered by launch: a k_cwalk APPENDS to the set above and CONSUMES (merges, then empties)
    // the set below, which the previous launch appended to and nobody touches now -- a slot is never handed out again while
    // a run-on walk that reserved it may still be writing its result.  (k_cwalkg: both sets are the same, as before.)
    cw_key *pend_c, *pend_exit_c;
    int32_t *npend_c, *pend_ready_c;
    uint32_t *phist_c;
    uint8_t *walked;          // [S][CW_K]: walked under the current tensor
    int8_t *nxt;              // [S][CW_K]: index of the entry's exit state in the next segment's pool, -1 = not there
    uint32_t *hist;           // [S][NW][CW_K]
    int32_t *true_idx;        // [S]
    double *segmin;           // [S]
    uint8_t *path_out;
    double *lmsel;
    // lag counts beyond what fits a 64-bit state (k_cwalkg): a key is then the HASH of the state, and the state itself -- one
    // byte per pick, lag 1 first -- sits next to it; equal hashes are confirmed on the bytes wherever two states are compared
    uint8_t *keys_d, *exits_d, *pend_d;       // [S][CW_K][LD], null in the packed mode
    uint8_t *pend_d_c, *pend_exit_d, *pend_exit_d_c;      // k_cwalkg's run-on: the consumed set of the requests' bytes; the exit states a request arrives with
    int LD;                   // bytes per state (L rounded up to 4)
    int runon;                // k_cwalk: segments a walker may run on into behind its own (<= CW_RUNON)
    cw_key key0;              // key of the start state (0 in the packed mode)
};

__device__ __forceinline__ cw_key cw_hash_digits(const uint8_t *d, int L)
{
    unsigned long long h = 0xcbf29ce484222325ull;
    for (int l = 0; l < L; l++) { h ^= d[l]; h *= 0x100000001b3ull; }
    h ^= h >> 32; h *= 0x9e3779b97f4a7c15ull; h ^= h >> 29;
    return h;
}
// (states lie LD = L rounded up to 4 bytes apart in 4-byte-aligned arrays: compared as words, every load on its way before the
// first comparison -- byte by byte with an early exit it was a chain of up to L dependent loads, 15-30 us of k_clink at 33-48 lags;
// the bytes of the last word beyond L are not looked at)
__device__ __forceinline__ bool cw_same_digits(const uint8_t *a, const uint8_t *b, int L)
{
    const uint32_t *x = reinterpret_cast<const uint32_t *>(a), *y = reinterpret_cast<const uint32_t *>(b);
    const int nw = (L + 3) >> 2;
    const uint32_t tail = (L & 3) ? ((1u << (8 * (L & 3))) - 1u) : 0xffffffffu;
    uint32_t diff = 0;
    for (int w = 0; w + 1 < nw; w++) diff |= x[w] ^ y[w];
    diff |= (x[nw - 1] ^ y[nw - 1]) & tail;
    return diff == 0;
}

// -------------------------------------------------------------------------------------------------------------
// k_cwalk: quad q of workgroup s walks pool entry q of segment s (if it has not been walked under this tensor), then
// looks its exit state up in the next segment's pool.
// -------------------------------------------------------------------------------------------------------------
template <int LC, int R>
__global__ void __launch_bounds__(CW_K * cw_lanes(R)) k_cwalk(cw_params P)
{
    constexpr int LPE = cw_lanes(R), BITS = cw_bits(R), NTHR = CW_K * LPE;
    constexpr unsigned DMASK = (1u << BITS) - 1u;
    constexpr int PPW = R == 4 ? 16 : 8, WB = R == 4 ? 2 : 4;     // picks per word of hist, bits each
    extern __shared__ __align__(16) unsigned char cw_smem[];
    dev_state *st = P.st;
    const dev_ctl c = load_ctl(st);
    if (c.stop || c.lt_stale || c.cw_unres) return;
    if (P.round > 0 && c.cw_open_at < 0) return;            // the chain closed in an earlier round
    if (P.check_masks == 2 || (P.check_masks && c.cm_same == 0)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st->lt_stale = 1;
        return;
    }
    if ((c.ranked != 0) != (R == 4)) {                      // the table is not in this instantiation's layout (it was rebuilt): the host looks again
        if (blockIdx.x == 0 && threadIdx.x == 0) st->cw_unres = 2;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->cur_hole = c.first_hole;      // (the flags stand until k_cemit re-arms them)
        if (P.round == 0) st->cw_open_at = 0;           // a new path: its chain is open until a k_cscan says otherwise (round 0's may be skipped)
    }
    const cw_geom g = cw_geometry(P.N, P.L);
    const int s = blockIdx.x, tid = threadIdx.x;
    if (s >= g.S) return;
    // (1) the states the previous segment's walks ended in, and that this pool does not hold yet, join it (only this
    // workgroup writes pool s).  A full pool gives up the entries that were on a chain longest ago, never one of this path.
    __shared__ int s_n;
    if (tid < 64) {
        int n0 = P.npool[s];
        const int np = P.npend_c[s] < CW_K ? P.npend_c[s] : CW_K;
        cw_key *keys = P.keys + (size_t)s * CW_K;
        int32_t *lh = P.last_hit + (size_t)s * CW_K;
        for (int k = 0; k < np; k++) {
            const cw_key x = P.pend_c[(size_t)s * CW_K + k];
            const bool dup = __builtin_amdgcn_ballot_w64(tid < n0 && keys[tid] == x) != 0;
            if (dup) continue;
            int slot = n0;
            if (n0 >= CW_K) {
                int mine = (tid == 0 && s == 0) ? 0x7fffffff : lh[tid], who = tid;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const int om = __shfl_xor(mine, o), ow = __shfl_xor(who, o);
                    if (om < mine || (om == mine && ow < who)) { mine = om; who = ow; }
                }
                if (mine >= P.stamp) continue;              // every entry was on a chain of this path (cannot happen: one per path)
                slot = who;
            } else n0++;
            // (a request that arrives with its walk -- run-on -- joins as a walked entry: exit state and picks are copied)
            const bool ready = P.pend_ready_c && P.pend_ready_c[(size_t)s * CW_K + k] == P.stamp && P.stamp != 0;
            if (tid == 0) {
                keys[slot] = x; lh[slot] = P.stamp - 1; P.walked[(size_t)s * CW_K + slot] = ready ? 1 : 0;
                if (ready) P.exits[(size_t)s * CW_K + slot] = P.pend_exit_c[(size_t)s * CW_K + k];
            }
            if (ready) {
                const int nw_m = R == 4 ? g.NW : g.NW5;
                for (int w = tid; w < nw_m; w += 64) P.hist[((size_t)s * nw_m + w) * CW_K + slot] = P.phist_c[((size_t)s * nw_m + w) * CW_K + k];
            }
            __builtin_amdgcn_s_waitcnt(0);                  // the next candidate's duplicate search reads keys[]
        }
        if (tid == 0) { P.npool[s] = n0; P.npend_c[s] = 0; s_n = n0; }
    }
    __syncthreads();
    const int n = s_n;
    const int q = tid / LPE, b = tid & (LPE - 1);
    const int bcol = b < R ? b : R - 1;                     // (R = 5: lanes 5..7 of the group read a valid column and carry -inf)
    const bool live = q < n && P.walked[(size_t)s * CW_K + q] == 0;
    if (!__syncthreads_or(live ? 1 : 0)) return;            // nothing new to walk in this segment
    constexpr int CH = cw_chunk(LC, R);
    constexpr cw_key SMASK = BITS * LC >= 64 ? ~0ull : ((1ull << (BITS * LC)) - 1ull);
    constexpr int ROWS = cw_rows(R), COLS = cw_cols(R), ENT = ROWS * COLS;      // one (target, lag) block: rows x columns doubles
    double *Gs = reinterpret_cast<double *>(cw_smem);       // [CH][LC][ROWS][COLS]
    cw_key sigma = live ? P.keys[(size_t)s * CW_K + q] : 0ull;
    int seg = s;                                            // the segment being walked: s, then (run-on) s + 1, ...
    int t0 = s * g.seglen;
    int t1 = t0 + g.seglen < P.N ? t0 + g.seglen : P.N;
    const unsigned shift = (unsigned)(tid & 63 & ~(LPE - 1));     // this lane group's bits in the wave's ballot
    const int nw_e = R == 4 ? g.NW : g.NW5;
    int word_i = 0;
    unsigned word = 0;
    bool active = live;                                     // this lane group still walks
    int pslot = 0;                                          // run-on: the pending slot of pool `seg` this walk belongs to
    uint32_t *hdst = P.hist + (size_t)s * nw_e * CW_K + q;  // where its words of picks go (stride CW_K): the entry's own, or a pending slot's
    // The slice of a chunk: for target t = c0 + 1 + tl and lag l the rows of source t - l at that lag -- a run of COLS doubles
    // per (tl, l, row) in G ([i][row][lag][col]), copied as a unit.  Position 0 carries '_' whatever the digit says (row 5),
    // positions in front of it add +0.0.  The loads of chunk k+1 are issued before chunk k is walked and stay in registers
    // under the walk: their latency is off the critical path.
    constexpr int UNITS = CH * LC * ROWS;
    constexpr int NV = (UNITS + NTHR - 1) / NTHR;
    double pre[NV][COLS];
    // marginal term: the log-marginals of the chunk's targets (column b: rank or symbol, like G's columns), fetched with the
    // slice, added in front of the lag-1 entries once the slice stands in LDS
    constexpr int NLM = (CH * LT_ROW + NTHR - 1) / NTHR;
    double *Lms = Gs + (size_t)CH * LC * ENT;
    double prelm[NLM];
    auto fetch = [&](int c0) {
        const int nc = t1 - c0 < CH ? t1 - c0 : CH;
        if (P.mt) {
#pragma unroll
            for (int k = 0; k < NLM; k++) {
                const int e = tid + k * NTHR;
                const int tl = e / LT_ROW, bb = e - tl * LT_ROW;
                const int tgt = c0 + 1 + tl;
                prelm[k] = 0.0;
                if (tl < nc && bb < R) prelm[k] = R == 4 ? P.rinfo[(size_t)tgt * RINFO + bb] : P.minfo[(size_t)tgt * MINFO + bb];
            }
        }
        const int total = nc * LC * ROWS;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const int u = tid + k * NTHR;
#pragma unroll
            for (int cc = 0; cc < COLS; cc++) pre[k][cc] = 0.0;
            if (u < total) {
                const int row = u % ROWS, l1 = (u / ROWS) % LC, tl = u / (ROWS * LC);
                const int i = c0 + tl - l1;                           // source of lag l1 + 1 at target c0 + 1 + tl
                if (i >= 0) {
                    const double *src = P.G + (((size_t)i * 6 + (i == 0 ? 5 : row)) * LC + l1) * LT_ROW;
#pragma unroll
                    for (int cc = 0; cc < COLS; cc++) pre[k][cc] = src[cc];
                }
            }
        }
    };
    auto store = [&](int c0) {
        const int nc = t1 - c0 < CH ? t1 - c0 : CH;
        const int total = nc * LC * ROWS;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const int u = tid + k * NTHR;
            if (u < total) {
#pragma unroll
                for (int cc = 0; cc < COLS; cc++) Gs[(size_t)u * COLS + cc] = pre[k][cc];
            }
        }
        if (P.mt) {
#pragma unroll
            for (int k = 0; k < NLM; k++) {
                const int e = tid + k * NTHR;
                if (e < nc * LT_ROW) Lms[e] = prelm[k];
            }
            __syncthreads();
            // lag 1 of target tl, row r, column bb:  (0.0 + lm) + x1 -- the reference's first addition
            for (int e = tid; e < nc * ENT; e += NTHR) {
                const int tl = e / ENT, rc = e - tl * ENT, bb = rc % COLS;
                double *g = Gs + (size_t)tl * LC * ENT + rc;
                *g = Lms[tl * LT_ROW + bb] + *g;
            }
        }
    };
    // the keys of the pool behind the segment being walked, for the closure search at its end: fetched while the walk has not
    // begun (one lane per key) instead of one dependent global load per key behind it
    __shared__ cw_key s_next[CW_K];
    __shared__ int s_nn;
    for (int hop = 0; ; hop++) {
    if (tid < CW_K) s_next[tid] = seg + 1 < g.S ? P.keys[(size_t)(seg + 1) * CW_K + tid] : 0ull;
    if (tid == 0) s_nn = seg + 1 < g.S ? P.npool[seg + 1] : 0;      // (entries behind the count are leftovers of earlier tensors)
    fetch(t0);
    // the row addresses of the last LC picks in front of the segment: slot (LC - l) % LC = the pick l positions back
    constexpr unsigned ROWD = COLS;                              // doubles per row of a (target, lag) block
    const unsigned lane_base = (unsigned)(uintptr_t)Gs + (unsigned)bcol * 8u;
    unsigned dig[LC];
#pragma unroll
    for (int l = 1; l <= LC; l++) dig[(LC - l) % LC] = lane_base + ((unsigned)(sigma >> (BITS * (l - 1))) & DMASK) * (ROWD * 8u);
    for (int c0 = t0; c0 < t1; c0 += CH) {
        const int nc = t1 - c0 < CH ? t1 - c0 : CH;
        __syncthreads();                                          // the previous chunk has been walked
        store(c0);
        __syncthreads();
        if (c0 + CH < t1) fetch(c0 + CH);
        // One step (round 6: no arithmetic per term).  Slot u of `dig` holds the LDS byte address of (row = the pick of the step
        // that is u (mod LC) into the segment, column = this lane's candidate) inside the FIRST (target, lag) block of the slice;
        // the term of lag l at the chunk's v-th target is one LDS read from the slot of the pick made l steps ago at the
        // compile-time offset of block (v, l) -- hipcc folds it into the instruction where it is below 64 KB and adds it where
        // not (the last targets of the larger slices).  Chunks are whole blocks of LC steps and a segment starts at slot 0, so
        // every index is a constant; the steps behind the segment's end are skipped by a uniform branch.  (Rounds 2-5: blocks of
        // LC steps in a run-time loop, an address addition per term, and the steps a chunk had beyond its last whole block --
        // 9 of 53 at C5 -- with the rows shifted out of the state word term by term.)
        static_assert(CH % LC == 0 || CH < LC, "chunks are whole blocks");
        typedef const __attribute__((address_space(3))) double lds_cd;
        cw_unrolled([&](auto v_) __attribute__((always_inline)) {
            constexpr int v = decltype(v_)::value, u = v % LC;
            if (v < nc) {
                double x[LC];
#pragma unroll
                for (int l = 1; l <= LC; l++) x[l - 1] = *(lds_cd *)(dig[((u - l) % LC + LC) % LC] + (unsigned)((v * LC + (l - 1)) * ENT * 8));
                double acc = x[0];
#pragma unroll
                for (int l = 2; l <= LC; l++) acc = acc + x[l - 1];
                if (R == 5 && b >= R) acc = -INFINITY;                   // (the idle lanes of the group)
                double m = vmax_f64(acc, dpp_f64<0xB1>(acc));            // quad_perm [1,0,3,2]
                m = vmax_f64(m, dpp_f64<0x4E>(m));                       // quad_perm [2,3,0,1]
                if (R == 5) m = vmax_f64(m, dpp_f64<0x141>(m));          // row_half_mirror: the other quad of the eight lanes
                // (R = 5: a NaN weight in first place is the reference's incumbent and stays it -- kernels.hpp, argmax8)
                const unsigned long long win = __builtin_amdgcn_ballot_w64(R == 5 ? (acc == m || (b == 0 && acc != acc)) : acc == m);
                const unsigned d = (unsigned)__builtin_ctz((unsigned)(win >> shift) & ((1u << LPE) - 1u));       // first wins (gretel.py:166-174)
                sigma = (sigma << BITS) | (cw_key)d;                     // (what is shifted beyond LC picks is masked off behind the segment)
                dig[u] = lane_base + d * (ROWD * 8u);
                const int gt = c0 - t0 + v;                              // position inside the segment
                word |= d << (WB * (gt % PPW));
                if ((gt % PPW) == PPW - 1 || gt == t1 - t0 - 1) {
                    if (active && b == 0) hdst[(size_t)word_i * CW_K] = word;
                    word = 0;
                    word_i++;
                }
            }
        }, std::make_integer_sequence<int, CH>{});
    }
    sigma &= SMASK;
    // the segment is walked.  hop 0: the entry's own walk; later hops: a walk on behalf of a pending request of pool `seg`
    int go_on = 0;
    bool there = false;
    {
        // every lane of the group looks at its share of the next pool's keys (all hold the same sigma); a barrier of the chunk
        // loop stands between the staging above and this
        const int nn = s_nn < CW_K ? s_nn : CW_K;
        bool mine = false;
        for (int k = b; k < CW_K; k += LPE) mine |= k < nn && s_next[k] == sigma;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(mine);
        there = ((bal >> shift) & ((1ull << LPE) - 1ull)) != 0ull;
    }
    if (active && b == 0) {
        if (hop == 0) {
            P.exits[(size_t)s * CW_K + q] = sigma;
            P.walked[(size_t)s * CW_K + q] = 1;
        } else {
            const size_t pe = (size_t)seg * CW_K + (size_t)pslot;
            P.pend_exit[pe] = sigma;
            P.pend_ready[pe] = P.stamp;
        }
        // (3) closure: an exit state the next pool does not hold asks to join it.  (The next workgroup may be merging
        // its own pending list right now: a missed match only costs a duplicate request, dropped at the merge; the hops
        // themselves are resolved by k_clink, after this kernel.)  Run-on: this lane group then walks the next segment
        // itself, from that state, and the request carries the walk -- the chain of a NEW track is found in one launch, not
        // one segment per round.
        if (seg + 1 < g.S) {
            if (!there) {
                const int slot = atomicAdd(&P.npend[seg + 1], 1);
                if (slot < CW_K) {
                    P.pend[(size_t)(seg + 1) * CW_K + slot] = sigma;
                    if (P.pend_ready) {
                        P.pend_ready[(size_t)(seg + 1) * CW_K + slot] = 0;
                        if (hop < P.runon) go_on = 1 + slot;
                    }
                }
            }
        }
    }
    go_on = __shfl(go_on, (int)(tid & 63 & ~(LPE - 1)));      // lane b == 0 of the group decides
    active = go_on != 0;
    if (!__syncthreads_or(active ? 1 : 0)) break;           // nobody walks on: done
    seg++;
    t0 = seg * g.seglen;
    t1 = t0 + g.seglen < P.N ? t0 + g.seglen : P.N;
    word_i = 0;
    word = 0;
    pslot = active ? go_on - 1 : 0;
    hdst = P.phist + (size_t)seg * nw_e * CW_K + pslot;
    }
}

// -------------------------------------------------------------------------------------------------------------
// k_cwalkg: k_cwalk for lag counts whose state no longer fits 64 bits
// (L > 32 over ranks, L > 21 over symbols; L <= CW_MAX_LG).  The same pools, links and chain; what differs:
//   * the slice of G a chunk of targets needs stands in LDS target-major as in k_cwalk (round 4; rounds 2-3 read the L terms
//     of every step from global memory: a round trip to L2 per step, 600 us per path at 25 lags), sized at run time: as many
//     targets per chunk as 140 KB hold at this lag count (22 at 48 lags, 8 at 128); the terms are added in lag order as
//     everywhere, in groups of CWG_CHUNK lags whose picks lie behind one another in a doubled ring;
//   * a state is L bytes (one per pick, lag 1 first) next to its 64-bit hash; the last L picks of an entry live in a
//     ring in LDS.
// -------------------------------------------------------------------------------------------------------------
#define CW_MAX_LG 128
#define CWG_CHUNK 8           /* lags per unrolled group of a step (the last group: as many as are left) */
template <int R>
__global__ void __launch_bounds__(CW_K * cw_lanes(R)) k_cwalkg(cw_params P)
{
    constexpr int LPE = cw_lanes(R);
    constexpr int PPW = R == 4 ? 16 : 8, WB = R == 4 ? 2 : 4;
    // ring[q][(t - l) & 127] = pick of position t - l, kept TWICE (slot + 128 as well): the eight picks a group of lags needs
    // then lie behind one another wherever the ring wraps, and their reads differ by a compile-time offset only
    __shared__ __align__(16) uint8_t ring[CW_K][2 * CW_MAX_LG];
    extern __shared__ __align__(16) unsigned char cwg_smem[];
    constexpr int ROWS = cw_rows(R), COLS = cw_cols(R), ENT = ROWS * COLS, NTHR = CW_K * LPE;
    dev_state *st = P.st;
    const dev_ctl c = load_ctl(st);
    if (c.stop || c.lt_stale || c.cw_unres) return;
    if (P.round > 0 && c.cw_open_at < 0) return;
    if (P.check_masks == 2 || (P.check_masks && c.cm_same == 0)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st->lt_stale = 1;
        return;
    }
    if ((c.ranked != 0) != (R == 4)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st->cw_unres = 2;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->cur_hole = c.first_hole;
        if (P.round == 0) st->cw_open_at = 0;               // a new path: its chain is open until a k_cscan says otherwise (round 0's may be skipped)
    }
    const cw_geom g = cw_geometry(P.N, P.L);
    const int s = blockIdx.x, tid = threadIdx.x, L = P.L, LD = P.LD;
    if (s >= g.S) return;
    // (1) pending states join the pool (as k_cwalk; equal hashes are confirmed on the bytes).  A request that arrives with its walk
    // (run-on, round 4: as in k_cwalk) joins as a walked entry: exit state, its bytes and the picks are copied.
    __shared__ int s_n;
    if (tid < 64) {
        int n0 = P.npool[s];
        const int np = P.npend_c[s] < CW_K ? P.npend_c[s] : CW_K;
        cw_key *keys = P.keys + (size_t)s * CW_K;
        int32_t *lh = P.last_hit + (size_t)s * CW_K;
        const int nw_m = R == 4 ? g.NW : g.NW5;
        for (int k = 0; k < np; k++) {
            const cw_key x = P.pend_c[(size_t)s * CW_K + k];
            const uint8_t *xd = P.pend_d_c + ((size_t)s * CW_K + k) * LD;
            const bool mine_dup = tid < n0 && keys[tid] == x && cw_same_digits(P.keys_d + ((size_t)s * CW_K + tid) * LD, xd, L);
            if (__builtin_amdgcn_ballot_w64(mine_dup) != 0) continue;
            int slot = n0;
            if (n0 >= CW_K) {
                int mine = (tid == 0 && s == 0) ? 0x7fffffff : lh[tid], who = tid;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const int om = __shfl_xor(mine, o), ow = __shfl_xor(who, o);
                    if (om < mine || (om == mine && ow < who)) { mine = om; who = ow; }
                }
                if (mine >= P.stamp) continue;
                slot = who;
            } else n0++;
            const bool ready = P.pend_ready_c && P.pend_ready_c[(size_t)s * CW_K + k] == P.stamp && P.stamp != 0;
            if (tid == 0) {
                keys[slot] = x; lh[slot] = P.stamp - 1; P.walked[(size_t)s * CW_K + slot] = ready ? 1 : 0;
                if (ready) P.exits[(size_t)s * CW_K + slot] = P.pend_exit_c[(size_t)s * CW_K + k];
            }
            for (int l = tid; l < L; l += 64) P.keys_d[((size_t)s * CW_K + slot) * LD + l] = xd[l];
            if (ready) {
                const uint8_t *ed = P.pend_exit_d_c + ((size_t)s * CW_K + k) * LD;
                for (int l = tid; l < L; l += 64) P.exits_d[((size_t)s * CW_K + slot) * LD + l] = ed[l];
                for (int w = tid; w < nw_m; w += 64) P.hist[((size_t)s * nw_m + w) * CW_K + slot] = P.phist_c[((size_t)s * nw_m + w) * CW_K + k];
            }
            __builtin_amdgcn_s_waitcnt(0);
        }
        if (tid == 0) { P.npool[s] = n0; P.npend_c[s] = 0; s_n = n0; }
    }
    __syncthreads();
    const int n = s_n;
    const int q = tid / LPE, b = tid & (LPE - 1);
    const int bcol = b < R ? b : R - 1;
    const bool live = q < n && P.walked[(size_t)s * CW_K + q] == 0;
    if (!__syncthreads_or(live ? 1 : 0)) return;
    const unsigned shift = (unsigned)(tid & 63 & ~(LPE - 1));
    const int nw_e = R == 4 ? g.NW : g.NW5;
    // the entry's last L picks into its ring: the pick of lag l (position t0 + 1 - l) at slot (t0 + 1 - l) & 127
    for (int w = b; w < 2 * CW_MAX_LG / 4; w += LPE) reinterpret_cast<uint32_t *>(ring[q])[w] = 0u;      // (an idle lane group reads row 0)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (live)
        for (int l = 1 + b; l <= L; l += LPE) {
            const uint8_t d = P.keys_d[((size_t)s * CW_K + q) * LD + (l - 1)];
            const int sl = (s * g.seglen + 1 - l) & (CW_MAX_LG - 1);
            ring[q][sl] = d; ring[q][sl + CW_MAX_LG] = d;
        }
    __syncthreads();
    // the slice: Gs[tl][lag - 1][row][col] = G[t - lag][row][lag - 1][col] for the chunk's targets t = c0 + 1 + tl (position 0
    // carries '_' whatever the pick says: its row 5 in every row; positions in front of it: +0.0), then the log-marginals
    // of the targets (marginal term: in front of x1)
    const int CH = cwg_chunk(L, R);
    double *Gs = reinterpret_cast<double *>(cwg_smem);
    double *Lms = Gs + (size_t)CH * L * ENT;
    const uint8_t *myring = ring[q];
    bool active = live;                                     // this lane group still walks
    int seg = s, pslot = 0;
    uint32_t *hdst = P.hist + (size_t)s * nw_e * CW_K + q;  // where its words of picks go (stride CW_K): the entry's own, or a pending slot's
    for (int hop = 0; ; hop++) {
    const int t0 = seg * g.seglen;
    const int t1 = t0 + g.seglen < P.N ? t0 + g.seglen : P.N;
    int word_i = 0;
    unsigned word = 0;
    for (int c0 = t0; c0 < t1; c0 += CH) {
        const int nc = t1 - c0 < CH ? t1 - c0 : CH;
        __syncthreads();                                        // the chunk before has been walked
        for (int u = tid; u < nc * L * ROWS; u += NTHR) {
            const int row = u % ROWS, l1 = (u / ROWS) % L, tl = u / (ROWS * L);
            const int i = c0 + tl - l1;
            double v[COLS];
#pragma unroll
            for (int cc = 0; cc < COLS; cc++) v[cc] = 0.0;
            if (i >= 0) {
                const double *src = P.G + (((size_t)i * 6 + (i == 0 ? 5 : row)) * L + l1) * LT_ROW;
#pragma unroll
                for (int cc = 0; cc < COLS; cc++) v[cc] = src[cc];
            }
#pragma unroll
            for (int cc = 0; cc < COLS; cc++) Gs[(size_t)u * COLS + cc] = v[cc];
        }
        if (P.mt)
            for (int e = tid; e < nc * LT_ROW; e += NTHR) {
                const int tl = e / LT_ROW, bb = e - tl * LT_ROW;
                Lms[e] = bb < R ? (R == 4 ? P.rinfo[(size_t)(c0 + 1 + tl) * RINFO + bb] : P.minfo[(size_t)(c0 + 1 + tl) * MINFO + bb]) : 0.0;
            }
        __syncthreads();
        for (int tl = 0; tl < nc; tl++) {
            const int t = c0 + 1 + tl;
            // lag l: source i = t - l, row = the pick made there.  Lag 1 first (the marginal term goes in front of it), then groups
            // of CWG_CHUNK lags: a group that lies wholly inside the window and the lag count takes its picks from one run of the
            // doubled ring and its terms from one run of the slice -- five instructions per lag; the group at either edge is
            // predicated term by term (a term that does not exist reads the lag-1 slot and is replaced by +0.0)
            const double *base = Gs + (size_t)tl * L * ENT + bcol;
            const double lm_t = P.mt ? Lms[tl * LT_ROW + bcol] : 0.0;       // marginal term: in front of x1
            double acc;
            {
                const int row1 = t - 1 <= 0 ? 0 : (int)myring[(t - 1) & (CW_MAX_LG - 1)];      // (position 0: its row 5 stands in every row)
                const double x1 = base[row1 * COLS];
                acc = P.mt ? lm_t + x1 : x1;
            }
            // NL lags l0 .. l0 + NL - 1, all inside the window: picks from one run of the doubled ring, terms from one run of the slice
            auto group = [&](auto nl_, int l0) __attribute__((always_inline)) {
                constexpr int NL = decltype(nl_)::value;
                const uint8_t *rg = myring + ((t - (l0 + NL - 1)) & (CW_MAX_LG - 1));      // picks of lags l0+NL-1 .. l0, ascending positions
                const double *bl = base + (size_t)(l0 - 1) * ENT;
                int rows[NL];
                double x[NL];
#pragma unroll
                for (int u = 0; u < NL; u++) rows[u] = (int)rg[NL - 1 - u];
#pragma unroll
                for (int u = 0; u < NL; u++) x[u] = bl[u * ENT + rows[u] * COLS];
#pragma unroll
                for (int u = 0; u < NL; u++) acc = acc + x[u];
            };
            for (int l0 = 2; l0 <= L; l0 += CWG_CHUNK) {
                const int nl = L - l0 + 1 < CWG_CHUNK ? L - l0 + 1 : CWG_CHUNK;
                if (t - (l0 + nl - 1) >= 0) {
                    static_assert(CWG_CHUNK == 8, "the switch below names the group sizes");
                    switch (nl) {
                        case 8: group(std::integral_constant<int, 8>{}, l0); break;
                        case 7: group(std::integral_constant<int, 7>{}, l0); break;
                        case 6: group(std::integral_constant<int, 6>{}, l0); break;
                        case 5: group(std::integral_constant<int, 5>{}, l0); break;
                        case 4: group(std::integral_constant<int, 4>{}, l0); break;
                        case 3: group(std::integral_constant<int, 3>{}, l0); break;
                        case 2: group(std::integral_constant<int, 2>{}, l0); break;
                        default: group(std::integral_constant<int, 1>{}, l0); break;
                    }
                } else {
                    // the first positions of the window: terms in front of it do not exist (read the lag-1 slot, replaced by +0.0)
                    int rows[CWG_CHUNK];
                    double x[CWG_CHUNK];
#pragma unroll
                    for (int u = 0; u < CWG_CHUNK; u++) {
                        const int l = l0 + u;
                        const int le = (l <= L && t - l >= 0) ? l : 1;
                        rows[u] = t - le <= 0 ? 0 : (int)myring[(t - le) & (CW_MAX_LG - 1)];
                    }
#pragma unroll
                    for (int u = 0; u < CWG_CHUNK; u++) {
                        const int l = l0 + u;
                        const bool have = l <= L && t - l >= 0;
                        const int le = have ? l : 1;
                        const double v = base[(le - 1) * ENT + rows[u] * COLS];
                        x[u] = have ? v : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < CWG_CHUNK; u++)
                        if (l0 + u <= L) acc = acc + x[u];
                }
            }
            if (R == 5 && b >= R) acc = -INFINITY;
            double m = vmax_f64(acc, dpp_f64<0xB1>(acc));
            m = vmax_f64(m, dpp_f64<0x4E>(m));
            if (R == 5) m = vmax_f64(m, dpp_f64<0x141>(m));
            const unsigned long long win = __builtin_amdgcn_ballot_w64(R == 5 ? (acc == m || (b == 0 && acc != acc)) : acc == m);
            const unsigned d = (unsigned)__builtin_ctz((unsigned)(win >> shift) & ((1u << LPE) - 1u));
            if (active && b == 0) { ring[q][t & (CW_MAX_LG - 1)] = (uint8_t)d; ring[q][(t & (CW_MAX_LG - 1)) + CW_MAX_LG] = (uint8_t)d; }      // (read again at the earliest one step later, by this lane group only)
            const int gt = t - t0 - 1;
            word |= d << (WB * (gt % PPW));
            if ((gt % PPW) == PPW - 1 || gt == t1 - t0 - 1) {
                if (active && b == 0) hdst[(size_t)word_i * CW_K] = word;
                word = 0;
                word_i++;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // the segment is walked.  Its exit state: the last L picks, lag 1 first (out of the ring); its hash.  hop 0: the entry's own
    // walk; later hops: a walk on behalf of the pending request `pslot` of pool `seg` (the result goes with the request)
    __syncthreads();
    auto dg = [&](int l) -> uint8_t {                       // pick of lag l behind the segment
        const int i = t1 + 1 - l;
        return i >= 1 ? ring[q][i & (CW_MAX_LG - 1)] : (uint8_t)0;
    };
    if (active) {
        uint8_t *xd = hop == 0 ? P.exits_d + ((size_t)s * CW_K + q) * LD : P.pend_exit_d + ((size_t)seg * CW_K + pslot) * LD;
        for (int l = 1 + b; l <= L; l += LPE) xd[l - 1] = dg(l);
    }
    int go_on = 0;
    if (active && b == 0) {
        unsigned long long hh = 0xcbf29ce484222325ull;      // (cw_hash_digits over the ring)
        for (int l = 1; l <= L; l++) { hh ^= dg(l); hh *= 0x100000001b3ull; }
        hh ^= hh >> 32; hh *= 0x9e3779b97f4a7c15ull; hh ^= hh >> 29;
        const cw_key sigma = hh;
        if (hop == 0) {
            P.exits[(size_t)s * CW_K + q] = sigma;
            P.walked[(size_t)s * CW_K + q] = 1;
        } else {
            const size_t pe = (size_t)seg * CW_K + (size_t)pslot;
            P.pend_exit[pe] = sigma;
            __builtin_amdgcn_s_waitcnt(0);                  // (the exit state's bytes were stored by this lane group above)
            P.pend_ready[pe] = P.stamp;
        }
        // closure, as in k_cwalk: an exit state the next pool does not hold asks to join it -- and, run-on, is walked on from here
        if (seg + 1 < g.S) {
            const cw_key *kn = P.keys + (size_t)(seg + 1) * CW_K;
            bool there = false;
            const int nn = P.npool[seg + 1];
            for (int k = 0; k < nn && k < CW_K && !there; k++) {
                if (kn[k] != sigma) continue;
                const uint8_t *kd = P.keys_d + ((size_t)(seg + 1) * CW_K + k) * LD;
                bool same = true;
                for (int l = 1; l <= L && same; l++) same = kd[l - 1] == dg(l);
                there = same;
            }
            if (!there) {
                const int slot = atomicAdd(&P.npend[seg + 1], 1);
                if (slot < CW_K) {
                    P.pend[(size_t)(seg + 1) * CW_K + slot] = sigma;
                    for (int l = 1; l <= L; l++) P.pend_d[((size_t)(seg + 1) * CW_K + slot) * LD + (l - 1)] = dg(l);
                    if (P.pend_ready) {
                        P.pend_ready[(size_t)(seg + 1) * CW_K + slot] = 0;
                        if (hop < P.runon) go_on = 1 + slot;
                    }
                }
            }
        }
    }
    go_on = __shfl(go_on, (int)(tid & 63 & ~(LPE - 1)));      // lane b == 0 of the group decides
    active = go_on != 0;
    if (!__syncthreads_or(active ? 1 : 0)) break;           // nobody walks on: done
    seg++;
    pslot = active ? go_on - 1 : 0;
    hdst = P.phist + (size_t)seg * nw_e * CW_K + pslot;
    }
}

// -------------------------------------------------------------------------------------------------------------
// k_cwalk2<LC>: 33..64 lags over ranks (round 6).  k_cwalkg's pools -- a state is its L picks as bytes next to their hash -- with
// k_cwalk's step: the row offsets of the last LC picks rotate through registers (slot u = the pick of the step that is u (mod LC)
// into a block of LC steps, every index a compile-time constant), the terms of a step are LC LDS reads at compile-time offsets
// and LC - 1 additions in lag order, and the next chunk's slice waits in registers under the walk.  A block of LC steps is two
// chunks of the slice (LC x 16 x 8 bytes per target: 24 targets are 148 KB at 48 lags), so a chunk boundary always falls on the
// same two steps of the unrolled block.  One instantiation serves every lag count L <= LC: the slice's blocks of lags beyond L
// are zeros, and x + 0.0 is x for the arg-max (as for the positions in front of the window everywhere here) -- LC = 36, 40, ..., 64
// are compiled; beyond 48 lags a block is FOUR chunks (16 targets of 8 KB at 64 lags).  Picks also go to a ring in LDS (one byte store per step, off the chain): the exit state is read from there, as
// in k_cwalkg.  k_cwalkg took 6.5 rounds of 36 us per path at L = 33; this takes k_cwalk<32>'s 2.7 with run-on.
// -------------------------------------------------------------------------------------------------------------
#define CW2_MAX_L 64            /* over ranks */
#define CW2_MAX_L5 40           /* over the symbols (512 threads: 256 registers per lane) */
#define CW2_RING 64
// a block of LC steps is K chunks of the slice.  Over ranks (16 doubles per target and lag): two up to 48 lags (24 targets of 6 KB),
// four beyond (16 of 8 KB at 64); over the symbols (25 doubles): one at 24 lags (117 KB), two up to 32, four beyond (10 of 8 KB at 40:
// with three the next chunk's slice in registers pushed the walk's own into scratch memory)
__host__ __device__ constexpr int cw2_parts(int LC, int R = 4) { return R == 4 ? (LC <= 48 ? 2 : 4) : (LC <= 24 ? 1 : (LC <= 32 ? 2 : 4)); }
__host__ __device__ constexpr int cw2_chunk(int LC, int R = 4) { return (LC + cw2_parts(LC, R) - 1) / cw2_parts(LC, R); }
__host__ __device__ constexpr size_t cw2_lds_bytes(int LC, int R = 4) { return (size_t)cw2_chunk(LC, R) * (LC * cw_rows(R) * cw_cols(R) + 5) * 8; }
__host__ __device__ constexpr int cw2_lc(int L, int R = 4) { return R == 4 ? (L <= 36 ? 36 : (L + 3) / 4 * 4) : (L <= 24 ? 24 : (L + 3) / 4 * 4); }

template <int LC, int R>
__global__ void __launch_bounds__(CW_K * cw_lanes(R)) k_cwalk2(cw_params P)
{
    constexpr int LPE = cw_lanes(R), PPW = R == 4 ? 16 : 8, WB = R == 4 ? 2 : 4;
    constexpr int ROWS = cw_rows(R), COLS = cw_cols(R), ENT = ROWS * COLS, NTHR = CW_K * LPE;
    constexpr int KP = cw2_parts(LC, R), CHA = cw2_chunk(LC, R);  // a block of LC steps: KP chunks of at most CHA targets
    static_assert(R == 4 ? (LC > CW_MAX_L && LC <= CW2_MAX_L) : (LC > CW_MAX_L5 && LC <= CW2_MAX_L5), "lag counts of k_cwalk2");
    static_assert(LC <= CW2_RING && (KP - 1) * CHA < LC, "a block's parts");
    __shared__ __align__(16) uint8_t ring[CW_K][CW2_RING];  // ring[q][t & 63] = pick at position t
    extern __shared__ __align__(16) unsigned char cw2_smem[];
    dev_state *st = P.st;
    const dev_ctl c = load_ctl(st);
    if (c.stop || c.lt_stale || c.cw_unres) return;
    if (P.round > 0 && c.cw_open_at < 0) return;
    if (P.check_masks == 2 || (P.check_masks && c.cm_same == 0)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st->lt_stale = 1;
        return;
    }
    if ((c.ranked != 0) != (R == 4)) {                      // the table is not in this instantiation's layout (it was rebuilt): the host looks again
        if (blockIdx.x == 0 && threadIdx.x == 0) st->cw_unres = 2;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->cur_hole = c.first_hole;
        if (P.round == 0) st->cw_open_at = 0;
    }
    const cw_geom g = cw_geometry(P.N, P.L);
    const int s = blockIdx.x, tid = threadIdx.x, L = P.L, LD = P.LD;
    if (s >= g.S) return;
    // (1) pending states join the pool: k_cwalkg's merge, word for word (equal hashes are confirmed on the bytes; a request that
    // arrives with its walk joins as a walked entry)
    __shared__ int s_n;
    if (tid < 64) {
        int n0 = P.npool[s];
        const int np = P.npend_c[s] < CW_K ? P.npend_c[s] : CW_K;
        cw_key *keys = P.keys + (size_t)s * CW_K;
        int32_t *lh = P.last_hit + (size_t)s * CW_K;
        const int nw_m = R == 4 ? g.NW : g.NW5;
        for (int k = 0; k < np; k++) {
            const cw_key x = P.pend_c[(size_t)s * CW_K + k];
            const uint8_t *xd = P.pend_d_c + ((size_t)s * CW_K + k) * LD;
            const bool mine_dup = tid < n0 && keys[tid] == x && cw_same_digits(P.keys_d + ((size_t)s * CW_K + tid) * LD, xd, L);
            if (__builtin_amdgcn_ballot_w64(mine_dup) != 0) continue;
            int slot = n0;
            if (n0 >= CW_K) {
                int mine = (tid == 0 && s == 0) ? 0x7fffffff : lh[tid], who = tid;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const int om = __shfl_xor(mine, o), ow = __shfl_xor(who, o);
                    if (om < mine || (om == mine && ow < who)) { mine = om; who = ow; }
                }
                if (mine >= P.stamp) continue;
                slot = who;
            } else n0++;
            const bool ready = P.pend_ready_c && P.pend_ready_c[(size_t)s * CW_K + k] == P.stamp && P.stamp != 0;
            if (tid == 0) {
                keys[slot] = x; lh[slot] = P.stamp - 1; P.walked[(size_t)s * CW_K + slot] = ready ? 1 : 0;
                if (ready) P.exits[(size_t)s * CW_K + slot] = P.pend_exit_c[(size_t)s * CW_K + k];
            }
            for (int l = tid; l < L; l += 64) P.keys_d[((size_t)s * CW_K + slot) * LD + l] = xd[l];
            if (ready) {
                const uint8_t *ed = P.pend_exit_d_c + ((size_t)s * CW_K + k) * LD;
                for (int l = tid; l < L; l += 64) P.exits_d[((size_t)s * CW_K + slot) * LD + l] = ed[l];
                for (int w = tid; w < nw_m; w += 64) P.hist[((size_t)s * nw_m + w) * CW_K + slot] = P.phist_c[((size_t)s * nw_m + w) * CW_K + k];
            }
            __builtin_amdgcn_s_waitcnt(0);
        }
        if (tid == 0) { P.npool[s] = n0; P.npend_c[s] = 0; s_n = n0; }
    }
    __syncthreads();
    const int n = s_n;
    const int q = tid / LPE, b = tid & (LPE - 1);
    const int bcol = b < R ? b : R - 1;                     // (R = 5: lanes 5..7 of the group read a valid column and carry -inf)
    const bool live = q < n && P.walked[(size_t)s * CW_K + q] == 0;
    if (!__syncthreads_or(live ? 1 : 0)) return;
    const unsigned shift = (unsigned)(tid & 63 & ~(LPE - 1));
    const int nw_e = R == 4 ? g.NW : g.NW5;
    // the entry's last L picks into its ring (an idle lane group reads row 0 everywhere)
    for (int w = b; w < CW2_RING / 4; w += LPE) reinterpret_cast<uint32_t *>(ring[q])[w] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (live)
        for (int l = 1 + b; l <= L; l += LPE)
            ring[q][(s * g.seglen + 1 - l) & (CW2_RING - 1)] = P.keys_d[((size_t)s * CW_K + q) * LD + (l - 1)];
    __syncthreads();
    double *Gs = reinterpret_cast<double *>(cw2_smem);      // [<= CHA][LC][ROWS][COLS]
    double *Lms = Gs + (size_t)CHA * LC * ENT;
    bool active = live;
    int seg = s, pslot = 0;
    uint32_t *hdst = P.hist + (size_t)s * nw_e * CW_K + q;
    // the slice of a chunk (k_cwalk's, with the lag count known at run time: blocks of lags beyond L are zeros), fetched into
    // registers while the chunk before is walked
    constexpr int UNITS = CHA * LC * ROWS;
    constexpr int NV = (UNITS + NTHR - 1) / NTHR;
    constexpr int NLM = (CHA * LT_ROW + NTHR - 1) / NTHR;
    double pre[NV][COLS];
    double prelm[NLM];
    auto fetch = [&](int c0, int nc) {
        if (P.mt) {
#pragma unroll
            for (int k = 0; k < NLM; k++) {
                const int e = tid + k * NTHR;
                const int tl = e / LT_ROW, bb = e - tl * LT_ROW;
                prelm[k] = 0.0;
                if (tl < nc && bb < R) prelm[k] = R == 4 ? P.rinfo[(size_t)(c0 + 1 + tl) * RINFO + bb] : P.minfo[(size_t)(c0 + 1 + tl) * MINFO + bb];
            }
        }
        const int total = nc * LC * ROWS;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const int u = tid + k * NTHR;
#pragma unroll
            for (int cc = 0; cc < COLS; cc++) pre[k][cc] = 0.0;
            if (u < total) {
                const int row = u % ROWS, l1 = (u / ROWS) % LC, tl = u / (ROWS * LC);
                const int i = c0 + tl - l1;                           // source of lag l1 + 1 at target c0 + 1 + tl
                if (i >= 0 && l1 < L) {
                    const double *src = P.G + (((size_t)i * 6 + (i == 0 ? 5 : row)) * L + l1) * LT_ROW;
#pragma unroll
                    for (int cc = 0; cc < COLS; cc++) pre[k][cc] = src[cc];
                }
            }
        }
    };
    auto store = [&](int nc) {
        const int total = nc * LC * ROWS;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const int u = tid + k * NTHR;
            if (u < total) {
#pragma unroll
                for (int cc = 0; cc < COLS; cc++) Gs[(size_t)u * COLS + cc] = pre[k][cc];
            }
        }
        if (P.mt) {
#pragma unroll
            for (int k = 0; k < NLM; k++) {
                const int e = tid + k * NTHR;
                if (e < nc * LT_ROW) Lms[e] = prelm[k];
            }
            __syncthreads();
            for (int e = tid; e < nc * ENT; e += NTHR) {              // lag 1: (0.0 + lm) + x1, the reference's first addition
                const int tl = e / ENT, rc = e - tl * ENT, bb = rc % COLS;
                double *gp = Gs + (size_t)tl * LC * ENT + rc;
                *gp = Lms[tl * LT_ROW + bb] + *gp;
            }
        }
    };
    // the hashes of the pool behind the segment being walked, for the closure search at its end (as k_cwalk: one lane per key while
    // the walk has not begun, instead of a chain of dependent global loads behind it)
    __shared__ cw_key s_next[CW_K];
    __shared__ int s_nn;
    constexpr int NW4 = (LC + 3) / 4;                       // words of a state
    const int nwr = LD >> 2;                                // ... at this lag count
    const uint32_t tailm = (L & 3) ? ((1u << (8 * (L & 3))) - 1u) : 0xffffffffu;
    for (int hop = 0; ; hop++) {
    const int t0 = seg * g.seglen;
    const int t1 = t0 + g.seglen < P.N ? t0 + g.seglen : P.N;
    if (tid < CW_K) s_next[tid] = seg + 1 < g.S ? P.keys[(size_t)(seg + 1) * CW_K + tid] : 0ull;
    if (tid == 0) s_nn = seg + 1 < g.S ? P.npool[seg + 1] : 0;      // (entries behind the count are leftovers of earlier tensors)
    int word_i = 0;
    unsigned word = 0;
    constexpr unsigned ROWD = COLS;
    // slot (LC - l) % LC = the pick l positions in front of the segment's first target (lags beyond L: whatever the ring holds, a
    // valid row of a block of zeros)
    unsigned dig[LC];
#pragma unroll
    for (int l = 1; l <= LC; l++) dig[(LC - l) % LC] = (unsigned)ring[q][(t0 + 1 - l) & (CW2_RING - 1)] * ROWD;
    auto step = [&](int c0, int tl, auto rowoff) __attribute__((always_inline)) {
        const double *base = Gs + (size_t)tl * LC * ENT + bcol;
        double x[LC];
#pragma unroll
        for (int l = 1; l <= LC; l++) x[l - 1] = base[(l - 1) * ENT + rowoff(l)];
        double acc = x[0];
#pragma unroll
        for (int l = 2; l <= LC; l++) acc = acc + x[l - 1];
        if (R == 5 && b >= R) acc = -INFINITY;                   // (the idle lanes of the group)
        double m = vmax_f64(acc, dpp_f64<0xB1>(acc));            // quad_perm [1,0,3,2]
        m = vmax_f64(m, dpp_f64<0x4E>(m));                       // quad_perm [2,3,0,1]
        if (R == 5) m = vmax_f64(m, dpp_f64<0x141>(m));          // row_half_mirror: the other quad of the eight lanes
        // (R = 5: a NaN weight in first place is the reference's incumbent and stays it -- kernels.hpp, argmax8)
        const unsigned long long win = __builtin_amdgcn_ballot_w64(R == 5 ? (acc == m || (b == 0 && acc != acc)) : acc == m);
        const unsigned d = (unsigned)__builtin_ctz((unsigned)(win >> shift) & ((1u << LPE) - 1u));       // first wins (gretel.py:166-174)
        const int t = c0 + 1 + tl;
        if (active && b == 0) ring[q][t & (CW2_RING - 1)] = (uint8_t)d;      // (read again behind the segment only)
        const int gt = t - t0 - 1;
        word |= d << (WB * (gt % PPW));
        if ((gt % PPW) == PPW - 1 || gt == t1 - t0 - 1) {
            if (active && b == 0) hdst[(size_t)word_i * CW_K] = word;
            word = 0;
            word_i++;
        }
        return d;
    };
    int c0 = t0;
    fetch(c0, t1 - c0 < CHA ? t1 - c0 : CHA);
    while (c0 < t1) {
        // the block's parts in turn: part p holds steps p CHA .. min((p + 1) CHA, LC) - 1, its chunk of the slice staged in front of
        // them and the next part's on its way under them
        cw_unrolled([&](auto p_) __attribute__((always_inline)) {
            constexpr int p = decltype(p_)::value, U0 = p * CHA, CNT = U0 + CHA <= LC ? CHA : LC - U0;
            constexpr int pn = (p + 1) % KP, CNTN = pn * CHA + CHA <= LC ? CHA : LC - pn * CHA;      // (the part behind it)
            if (c0 < t1) {
                const int nc = t1 - c0 < CNT ? t1 - c0 : CNT;
                __syncthreads();                                      // the chunk before has been walked
                store(nc);
                __syncthreads();
                if (c0 + nc < t1) fetch(c0 + nc, t1 - c0 - nc < CNTN ? t1 - c0 - nc : CNTN);
                cw_unrolled([&](auto u_) __attribute__((always_inline)) {
                    constexpr int u = U0 + decltype(u_)::value;
                    if (u - U0 < nc) {
                        const unsigned d = step(c0, u - U0, [&](int l) { return dig[((u - l) % LC + LC) % LC]; });
                        dig[u] = d * ROWD;
                    }
                }, std::make_integer_sequence<int, CNT>{});
                c0 += nc;
            }
        }, std::make_integer_sequence<int, KP>{});
    }
    // the segment is walked.  Its exit state -- the last L picks, lag 1 first -- out of the ring, as the words it is kept in
    // (bytes beyond L zero), and its hash; hop 0: the entry's own walk, later hops: a walk on behalf of the pending request
    // `pslot` of pool `seg`.  Every lane of the group holds the whole state: each then compares its share of the next pool.
    __syncthreads();
    uint32_t xw[NW4];
    unsigned long long hh = 0xcbf29ce484222325ull;          // (cw_hash_digits)
#pragma unroll
    for (int j = 0; j < NW4; j++) {
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int l = 4 * j + e + 1, i = t1 + 1 - l;
            const uint32_t d = (l <= L && i >= 1) ? (uint32_t)ring[q][i & (CW2_RING - 1)] : 0u;
            w |= d << (8 * e);
            if (l <= L) { hh ^= (unsigned long long)d; hh *= 0x100000001b3ull; }
        }
        xw[j] = w;
    }
    hh ^= hh >> 32; hh *= 0x9e3779b97f4a7c15ull; hh ^= hh >> 29;
    const cw_key sigma = hh;
    if (active) {
        uint32_t *xd = reinterpret_cast<uint32_t *>(hop == 0 ? P.exits_d + ((size_t)s * CW_K + q) * LD : P.pend_exit_d + ((size_t)seg * CW_K + pslot) * LD);
#pragma unroll
        for (int j = 0; j < NW4; j++)
            if ((j & (LPE - 1)) == b && j < nwr) xd[j] = xw[j];
    }
    bool there = false;
    {
        const int nn = s_nn < CW_K ? s_nn : CW_K;
        bool mine = false;
        for (int k = b; k < nn; k += LPE) {
            if (s_next[k] != sigma) continue;
            const uint32_t *kd = reinterpret_cast<const uint32_t *>(P.keys_d + ((size_t)(seg + 1) * CW_K + k) * LD);
            uint32_t diff = 0;
#pragma unroll
            for (int j = 0; j < NW4; j++)
                if (j < nwr) diff |= (kd[j] ^ xw[j]) & (j == nwr - 1 ? tailm : 0xffffffffu);
            mine |= diff == 0;
        }
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(mine);
        there = ((bal >> shift) & ((1ull << LPE) - 1ull)) != 0ull;
    }
    int go_on = 0;
    if (active && b == 0) {
        if (hop == 0) {
            P.exits[(size_t)s * CW_K + q] = sigma;
            P.walked[(size_t)s * CW_K + q] = 1;
        } else {
            const size_t pe = (size_t)seg * CW_K + (size_t)pslot;
            P.pend_exit[pe] = sigma;
            P.pend_ready[pe] = P.stamp;                     // (read by the owner in a later launch: the bytes above are there by then)
        }
        // closure, as in k_cwalk: an exit state the next pool does not hold asks to join it -- and, run-on, is walked on from here
        if (seg + 1 < g.S && !there) {
            const int slot = atomicAdd(&P.npend[seg + 1], 1);
            if (slot < CW_K) {
                P.pend[(size_t)(seg + 1) * CW_K + slot] = sigma;
                uint32_t *pd = reinterpret_cast<uint32_t *>(P.pend_d + ((size_t)(seg + 1) * CW_K + slot) * LD);
#pragma unroll
                for (int j = 0; j < NW4; j++)
                    if (j < nwr) pd[j] = xw[j];
                if (P.pend_ready) {
                    P.pend_ready[(size_t)(seg + 1) * CW_K + slot] = 0;
                    if (hop < P.runon) go_on = 1 + slot;
                }
            }
        }
    }
    go_on = __shfl(go_on, (int)(tid & 63 & ~(LPE - 1)));      // lane b == 0 of the group decides
    active = go_on != 0;
    if (!__syncthreads_or(active ? 1 : 0)) break;           // nobody walks on: done
    seg++;
    pslot = active ? go_on - 1 : 0;
    hdst = P.phist + (size_t)seg * nw_e * CW_K + pslot;
    }
}

// -------------------------------------------------------------------------------------------------------------
// k_clink: behind k_cwalk the pools stand still.  Lane q of workgroup s finds where the exit state of entry (s, q) sits
// in pool s+1:  hop >= 0: that entry, walked under this tensor;  -2: there, still to be walked;  -1: not there.
// -------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(CW_K) k_clink(cw_params P)
{
    const dev_ctl c = load_ctl(P.st);
    if (c.stop || c.lt_stale || c.cw_unres) return;
    if (P.round > 0 && c.cw_open_at < 0) return;
    const cw_geom g = cw_geometry(P.N, P.L);
    const int s = blockIdx.x, q = threadIdx.x;
    if (s + 1 >= g.S) return;
    const size_t e = (size_t)s * CW_K + q;
    const cw_key kn = P.keys[(size_t)(s + 1) * CW_K + q];             // lane q holds key q of the next pool
    const bool kw = P.walked[(size_t)(s + 1) * CW_K + q] != 0;
    const int nn = P.npool[s + 1];
    const bool mine = q < P.npool[s] && P.walked[e];
    const cw_key x = P.exits[e];
    int h = -1;
    for (int k = 0; k < nn; k++) {                                     // (uniform trip count; key k by broadcast)
        const cw_key kk = ((cw_key)(unsigned)__builtin_amdgcn_readlane((int)(kn >> 32), k) << 32) |
                          (cw_key)(unsigned)__builtin_amdgcn_readlane((int)(kn & 0xffffffffull), k);
        const bool ww = __builtin_amdgcn_readlane(kw ? 1 : 0, k) != 0;
        if (h == -1 && kk == x &&
            (!P.keys_d || cw_same_digits(P.exits_d + e * (size_t)P.LD, P.keys_d + ((size_t)(s + 1) * CW_K + k) * (size_t)P.LD, P.L)))
            h = ww ? k : -2;
    }
    P.nxt[e] = (int8_t)(mine ? h : -1);
}

// -------------------------------------------------------------------------------------------------------------
// k_cscan: follows the hops from the start state (pool 0, entry 0).  Every hop lands on the entry that HOLDS the previous
// entry's exit state and has been walked under this tensor (k_clink compared the keys).  Two levels, so that the longest
// dependent chain is 16 + 16 + 16 table reads instead of 256: (1) every (group of 16 segments, entry) is carried
// through its group -- a map per group; (2) one thread chains the groups; (3) one thread per group walks its group again
// from the entry the chain enters it with.  Where the chain cannot hop it is open: the state waits in the next pool
// for the next round, or is on its way there (pending list; put there now if it is not).
// -------------------------------------------------------------------------------------------------------------
#define CW_GRP 16
__global__ void __launch_bounds__(1024) k_cscan(cw_params P)
{
    __shared__ int8_t hop[CW_MAX_SEG * CW_K];
    __shared__ int8_t gmap[(CW_MAX_SEG / CW_GRP) * CW_K];   // entry at the group's first segment -> entry behind its last hop, -1 = stuck
    __shared__ int tru[CW_MAX_SEG + 1];
    __shared__ int gin[CW_MAX_SEG / CW_GRP + 1];
    __shared__ int s_stuck;
    dev_state *st = P.st;
    const dev_ctl c = load_ctl(st);
    if (c.stop || c.lt_stale || c.cw_unres) return;
    if (P.round > 0 && c.cw_open_at < 0) return;
    const cw_geom g = cw_geometry(P.N, P.L);
    const int S = g.S, tid = threadIdx.x;
    const int NG = (S - 1 + CW_GRP - 1) / CW_GRP;           // hops s -> s+1 for s = 0 .. S-2, in groups of 16
    {
        const int n4 = S * CW_K / 4;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(P.nxt);
        uint32_t *dst = reinterpret_cast<uint32_t *>(hop);
        for (int e = tid; e < n4; e += 1024) dst[e] = src[e];
    }
    if (tid == 0) s_stuck = 0x7fffffff;
    __syncthreads();
    for (int e = tid; e < NG * CW_K; e += 1024) {           // (1)
        const int gr = e / CW_K;
        int x = e - gr * CW_K;
        const int s1 = (gr + 1) * CW_GRP < S - 1 ? (gr + 1) * CW_GRP : S - 1;
        for (int s = gr * CW_GRP; s < s1 && x >= 0; s++) x = hop[s * CW_K + x];
        gmap[e] = (int8_t)(x >= 0 ? x : -1);
    }
    __syncthreads();
    if (tid == 0) {                                         // (2)
        const bool start_ok = P.npool[0] > 0 && P.walked[0] && P.keys[0] == P.key0;
        int x = start_ok ? 0 : -1;
        for (int gr = 0; gr <= NG; gr++) {
            gin[gr] = x;
            if (gr < NG && x >= 0) x = gmap[gr * CW_K + x];
        }
        if (!start_ok) s_stuck = -1;                        // not even the start state has been walked
    }
    __syncthreads();
    if (tid < NG && gin[tid] >= 0) {                        // (3)
        int x = gin[tid];
        const int s1 = (tid + 1) * CW_GRP < S - 1 ? (tid + 1) * CW_GRP : S - 1;
        for (int s = tid * CW_GRP; s < s1; s++) {
            tru[s] = x;
            const int h = hop[s * CW_K + x];
            if (h < 0) { atomicMin(&s_stuck, s); break; }   // the chain is open behind segment s
            x = h;
            if (s + 1 == S - 1) tru[S - 1] = x;
        }
    }
    if (S == 1 && tid == 0 && gin[0] >= 0) tru[0] = 0;
    __syncthreads();
    const int stuck = s_stuck;                              // 0x7fffffff: closed; -1: at the start; else the last segment on the chain
    const int s_end = stuck == 0x7fffffff ? S : stuck + 1;  // tru[0 .. s_end) is on the chain
    for (int e = tid; e < s_end; e += 1024) {
        P.true_idx[e] = tru[e];
        P.last_hit[(size_t)e * CW_K + tru[e]] = P.stamp;
    }
    if (tid == 0) {
        const bool open = stuck != 0x7fffffff;
        if (open && stuck >= 0 && hop[stuck * CW_K + tru[stuck]] == -1) {
            // the exit state is not in pool stuck+1: on its pending list?  (full when k_cwalk asked: first place now)
            const cw_key x = P.exits[(size_t)stuck * CW_K + tru[stuck]];
            const int np = P.npend[stuck + 1] < CW_K ? P.npend[stuck + 1] : CW_K;
            bool queued = false;
            for (int k = 0; k < np; k++)
                if (P.pend[(size_t)(stuck + 1) * CW_K + k] == x) queued = true;
            if (!queued) {
                P.pend[(size_t)(stuck + 1) * CW_K] = x;
                if (P.pend_ready) P.pend_ready[(size_t)(stuck + 1) * CW_K] = 0;      // (a bare request: to be walked)
                if (P.pend_d)
                    for (int l = 0; l < P.L; l++)
                        P.pend_d[((size_t)(stuck + 1) * CW_K) * P.LD + l] = P.exits_d[((size_t)stuck * CW_K + tru[stuck]) * P.LD + l];
                if (P.npend[stuck + 1] < 1) P.npend[stuck + 1] = 1;
            }
        }
        st->cw_open_at = open ? stuck + 1 : -1;
        if (!open && P.round + 1 > c.cw_need) st->cw_need = P.round + 1;    // (one chain at a time: no atomic needed; c = the words as this launch found them)
#ifdef CW_DIAG
        if (!open) st->dbg8[P.round < 4 ? P.round : 4] += 1;      // histogram: the round a chain closed in
#endif
        if (open && P.last_round) st->cw_unres = 1;
    }
}

// -------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cemit(cw_params P)
{
    __shared__ double s_min[4];
    dev_state *st = P.st;
    const dev_ctl c = load_ctl(st);
    if (c.stop || c.lt_stale || c.cw_unres) return;
    const cw_geom g = cw_geometry(P.N, P.L);
    const int s = blockIdx.x, tid = threadIdx.x;
    if (s >= g.S) return;
    if (s == 0 && tid == 0) {
        st->dbg[3] = 4;                                     // gh_debug_walk_clock: variant 4 = candidate-pool segments
        P.path_out[0] = SYM_US;
        P.lmsel[0] = 1.0;                                   // k_hp: this path's sums are still to be taken
        if (P.rearm && c.cur_hole > P.N) { st->first_hole = 0x7f7f7f7f; st->nodel = 0x7f7f7f7f; st->cm_same = 0x7f7f7f7f; st->narrow = 0x7f7f7f7f; }
    }
    // the tensor changes behind this path: every pool entry is walked again for the next one
    if (tid < CW_K) P.walked[(size_t)s * CW_K + tid] = 0;
    const int Nw = c.cur_hole <= P.N ? c.cur_hole - 1 : P.N;
    const int t0 = s * g.seglen;
    int t1 = t0 + g.seglen < P.N ? t0 + g.seglen : P.N;
    if (t1 > Nw) t1 = Nw;
    double mn = INFINITY;
    const bool ranked = c.ranked != 0;
    const int cand = P.true_idx[s];
    for (int tl = tid; tl < t1 - t0; tl += 256) {
        const int t = t0 + 1 + tl;
        const double *inf = P.minfo + (size_t)t * MINFO;
        int b5;
        if (ranked) {                                       // a rank: the symbol through the candidate bits
            const unsigned word = P.hist[((size_t)s * g.NW + (tl >> 4)) * CW_K + cand];
            b5 = nth_set5((uint32_t)__double_as_longlong(inf[10]), (int)((word >> (2 * (tl & 15))) & 3u));
            if (b5 < 0) b5 = 0;
        } else {                                            // the symbol itself (A C G T -), four bits per pick
            const unsigned word = P.hist[((size_t)s * g.NW5 + (tl >> 3)) * CW_K + cand];
            b5 = (int)((word >> (4 * (tl & 7))) & 7u);
            if (b5 > 4) b5 = 0;
        }
        P.path_out[t] = (uint8_t)vsym(P.sm, b5);
        P.lmsel[t] = inf[b5];
        const double m = inf[5 + b5];
        if (m < mn) mn = m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double x = __shfl_xor(mn, o);
        if (x < mn) mn = x;
    }
    if ((tid & 63) == 0) s_min[tid >> 6] = mn;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++)
            if (s_min[w] < mn) mn = s_min[w];
        P.segmin[s] = mn;
    }
}

// -------------------------------------------------------------------------------------------------------------
// k_cseed: the states of a finished path (symbols) at the segment boundaries -> pool entries.  Used behind the serial
// walker (the spin's first path, and every path the pools could not close): merge = 0 replaces the pools, 1 adds.
// -------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cseed(cw_params P, const uint8_t *path, int merge)
{
    const cw_geom g = cw_geometry(P.N, P.L);
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= g.S) return;
    const int p = s * g.seglen;                             // the state entering target p + 1: picks of p, p-1, ...
    const bool ranked = P.st->ranked != 0;                  // digits: candidate ranks (2 bits) or symbols A C G T - (3 bits)
    const int bits = ranked ? 2 : 3;
    auto digit = [&](int i) -> unsigned {
        if (i < 1) return 0u;
        const uint32_t cm5 = (uint32_t)__double_as_longlong(P.minfo[(size_t)i * MINFO + 10]);
        const int a6 = a6_of_sym(P.sm, path[i]);
        return ranked ? ((unsigned)__popc(cm5 & ((1u << a6) - 1u)) & 3u) : (unsigned)(a6 < 5 ? a6 : 0);
    };
    cw_key sigma = 0;
    if (!P.keys_d) {
        for (int l = P.L; l >= 1; l--) sigma = (sigma << bits) | (cw_key)digit(p + 1 - l);      // oldest first: the pick of lag 1 ends in the lowest bits
    } else {
        // (the hash of the bytes, lag 1 first -- as cw_hash_digits takes them)
        unsigned long long hh = 0xcbf29ce484222325ull;
        for (int l = 1; l <= P.L; l++) { hh ^= digit(p + 1 - l); hh *= 0x100000001b3ull; }
        hh ^= hh >> 32; hh *= 0x9e3779b97f4a7c15ull; hh ^= hh >> 29;
        sigma = hh;
    }
    cw_key *keys = P.keys + (size_t)s * CW_K;
    int n = merge ? P.npool[s] : 0;
    bool there = false;
    for (int k = 0; k < n; k++) {
        if (keys[k] != sigma) continue;
        bool same = true;
        if (P.keys_d)
            for (int l = 1; l <= P.L && same; l++) same = P.keys_d[((size_t)s * CW_K + k) * P.LD + (l - 1)] == digit(p + 1 - l);
        if (same) { there = true; P.last_hit[(size_t)s * CW_K + k] = P.stamp; }
    }
    if (!there) {
        int slot = n;
        if (n >= CW_K) {
            int oldest = 0x7fffffff;
            slot = s == 0 ? 1 : 0;
            for (int k = (s == 0 ? 1 : 0); k < CW_K; k++) {
                const int lh = P.last_hit[(size_t)s * CW_K + k];
                if (lh < oldest) { oldest = lh; slot = k; }
            }
        } else n++;
        keys[slot] = sigma;
        if (P.keys_d)
            for (int l = 1; l <= P.L; l++) P.keys_d[((size_t)s * CW_K + slot) * P.LD + (l - 1)] = (uint8_t)digit(p + 1 - l);
        P.last_hit[(size_t)s * CW_K + slot] = P.stamp;
    }
    P.npool[s] = n;
    P.npend[s] = 0;
    P.npend_c[s] = 0;
    for (int k = 0; k < CW_K; k++) P.walked[(size_t)s * CW_K + k] = 0;
}

// -------------------------------------------------------------------------------------------------------------
// k_cguess: a first guess for pools that hold nothing yet -- the symbol with the largest marginal at every position
// (first wins).  k_cseed turns its states at the segment boundaries into one entry per pool; wherever the guess is not
// what the walk does there, closure brings the right states in over the next rounds.  Only a guess: no result depends
// on it (a chain is emitted when every hop is an exact hit, whatever the pools were started from).
// -------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cguess(cw_params P, uint8_t *path)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > P.N) return;
    if (p == 0) { path[0] = SYM_US; return; }
    const double *inf = P.minfo + (size_t)p * MINFO;
    int best = 0;
    double bm = inf[5];
#pragma unroll
    for (int b5 = 1; b5 < 5; b5++)
        if (inf[5 + b5] > bm) { bm = inf[5 + b5]; best = b5; }
    path[p] = (uint8_t)vsym(P.sm, best);
}

// segwalk.hpp -- segment-parallel path extension (included by gretel_hip.hip behind kernels.hpp).
//
// gretel/gretel.py:143-187 is a chain of N dependent steps: step t picks the arg-max of a sum that depends on the
// last L picks.  One wavefront walking that chain (k_walk_spec) is bound by its own instruction issue and leaves
// 255 CUs idle.  But the step is a finite-state transducer: with R candidates per position (ranks 0..3 in the ranked
// layout of k_lt, symbols A C G T - otherwise) the state entering target t is the last L digits,
//     sigma = sum_{l=1..L} d_{t-l} * R^(l-1)            (digits of positions <= 0 are 0),
// R^L states in all (1 024 at L = 5), and the pick is a pure function Next[t][sigma] in [0, R).  So:
//
//   k_seg    one workgroup per segment of the window (<= 256 segments):
//            (1) builds Next[t][.] for its targets from the conditional table G -- for EVERY state the same IEEE
//                additions in the same lag-ascending order as the walkers (acc = x1; acc += x2; ...), first-wins
//                arg-max over the candidates in rank/symbol order (gretel.py:166-174) -- in LDS;
//            (2) walks ALL R^L entry states through the segment by table lookups and keeps, per entry state, the exit
//                state (segment map M_s) and the picks made on the way (hist, a few bits per position).
//   k_scan   one workgroup per group of <= 16 consecutive segments: composes the group's maps -- one map per group,
//            and inside the group the prefix maps in front of every segment.
//   k_emit   one workgroup per segment: chains the maps of the groups in front of it from the known start state
//            (<= 15 lookups) and its prefix map = its TRUE entry state, reads that state's picks from hist, and does
//            the bookkeeping of gretel.py:182-187 for its positions (path symbols, the selected symbols'
//            log-marginals, the minimum marginal).
//   Spins (gh_spin) run WITHOUT k_emit: k_seg also carries, per entry state, the minimum marginal of the picks made on the
//   way, k_scan composes (map, min) pairs -- min is associative along the chain --, and the reweight kernel k_rw chains the
//   <= 16 group maps itself: its true entry states, the picks of its own positions (and of the W behind them), and the
//   path's minimum marginal = the ratio every reweighted cell needs.  Three launches per path instead of four.
//   k_hp     the two log-likelihood sums of gretel.py:185-186, strictly left to right in binary64 (one wavefront per
//            sum); they feed nothing on the device, so gh_spin runs them for all paths at once behind the loop.
//
// Nothing is speculated: every entry state is enumerated, the composition is exact, the result is bit-identical to
// the serial walkers (which stay: L > 5, batched launches, GH_WALK=spec).  Work per path: N * R^L sums instead of N.
#pragma once

#include "seg_geom.hpp"

struct seg_params {
    int N, L;
    int rearm;                // spin loops: k_scan re-arms first_hole/nodel/cm_same/narrow for the k_marg<T,true> that follows
    int check_masks;          // spins without a k_lt between paths: a candidate mask that moved under the last reweight makes the table stale
    const double *G;          // [(N+LT_PAD)][6][L][5], ranked or not (st->ranked)
    const double *minfo;      // [N+2][16]
    const double *rinfo;      // [N+2][8]: log10 marginal / marginal by candidate rank
    int mt;                   // gh_config.marginal_term: the edge weight starts with log10 marginal(b, t) (added in front of x1)
    int nanp;                 // sums can be NaN (zero-count candidates offered AND the marginal term): seg_argmax
    symmap sm;
    dev_state *st;
    uint32_t *hist;           // [S][NW][NS] picks of every entry state, DPW per word, the first lowest
    uint16_t *maps;           // [S][NS] segment maps
    uint16_t *pmaps;          // [S][NS] prefix maps: entry state of the group -> entry state of the segment
    uint16_t *gmaps;          // [G1][NS] group maps
    double *segmin;           // [S] minimum marginal of the symbols selected in each segment
    double *smin;             // [S][NS] minimum marginal of the picks of every (segment, entry state)   (k_seg)
    double *gmin;             // [G1][NS] the same over a whole group, by the group's entry state       (k_scan)
    uint8_t *cm5snap;         // [N+2] candidate bits of every position as k_seg saw them (rank -> symbol for whoever emits)
    uint8_t *path_out;        // [N+1]
    double *lmsel;            // [N+1] log10 marginal of the selected symbol per position (for k_hp)
    // k_rwseg (the reweight of the path before, fused into this path's k_seg launch): k_emit leaves, for every segment, a copy
    // of the band blocks of the LC positions in front of it (they belong to its neighbour, who rewrites them in that launch)
    int rws;                  // 1: k_scan takes over k_seg's look at the flags (they are only final when k_rwseg has ended)
    int W, esz;               // band width, bytes per element
    const void *band;
    void *halo;               // [S][LC][7][W][7] elements
    int patch_off;            // k_rwseg: byte offset of the halo rows in its dynamic LDS (behind k_seg's regions for either radix)
    // what the reweight inside k_rwseg found, one entry per workgroup: .x = bit 0 a candidate mask moved, bit 1 the last symbol of the
    // candidate order is offered somewhere, bit 2 a position has five candidates; .y = its first position without a candidate.
    // Plain stores, overwritten by every launch: in small windows the next launch, k_emit_small, reduces them in every workgroup --
    // it is also the launch that re-arms the control words, which its workgroups therefore must not read (k_scan, alone between
    // two launches, reads the control words)
    int2 *rwflags;            // [S]
};

// (every wavefront for itself, at the start of a kernel: every thread takes part)
__device__ __forceinline__ int2 rws_flags_reduce(const int2 *fl, int S)
{
    const int lane = threadIdx.x & 63;
    int f = 0, hole = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 7; k++) {                              // (at most 400 segments: seg_geometry)
        const int q = lane + 64 * k;
        if (q < S) { const int2 v = fl[q]; f |= v.x; hole = v.y < hole ? v.y : hole; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        f |= __shfl_xor(f, o);
        const int h2 = __shfl_xor(hole, o);
        hole = h2 < hole ? h2 : hole;
    }
    return make_int2(f, hole);
}

// what k_seg's look at the flags of the last reweight is behind k_rwseg (they only stand when that launch has ended): the control
// words as the reweight's atomics would have left them, the hole, a stale table.  Returns false when the table is stale (the
// caller returns); `hole` = the first position without a candidate.  Thread 0 of workgroup 0 writes, everybody decides alike.
__device__ __forceinline__ bool rws_take_flags(const seg_params &P, dev_state *st, int S, int &hole)
{
    int2 fr = rws_flags_reduce(P.rwflags, S);
    if (fr.y == 0x7fffffff) fr.y = 0x7f7f7f7f;                 // (no hole: the value the armed control word holds)
    hole = fr.y;
    const bool first = blockIdx.x == 0 && threadIdx.x == 0;
    if (first) {
        if (fr.x & 1) st->cm_same = 0;
        if (fr.x & 2) st->nodel = 0;
        if (fr.x & 4) st->narrow = 0;
        if (fr.y < st->first_hole) st->first_hole = fr.y;
    }
    if (P.check_masks == 2 || (P.check_masks && (fr.x & 1))) {
        if (first) st->lt_stale = 1;
        return false;
    }
    if (first) st->cur_hole = fr.y;
    return true;
}

// -------------------------------------------------------------------------------------------------------------
// k_seg
// -------------------------------------------------------------------------------------------------------------
// first-wins arg-max over R sums (gretel.py:166-174: the first candidate is the incumbent, a later one wins on strict >).
// As a tournament: the incumbent of a pair is its first member unless the second is strictly greater, and a later pair
// (or the fifth value) only wins on strictly greater than the maximum so far -- the same index as the scan front to
// back for every input without NaNs (the sums are finite or -inf), with v_max_f64 where the scan needs two
// v_cndmask per value it carries along.
template <int R>
__device__ __forceinline__ unsigned seg_argmax(const double (&vin)[R], bool nanp = false)
{
    static_assert(R == 4 || R == 5, "four ranks or five symbols");
    double v[R];
#pragma unroll
    for (int b = 0; b < R; b++) v[b] = vin[b];
    if (R == 5 && nanp) {
        // a NaN (zero-count candidates WITH the marginal term: log10(0) + an infinite conditional) never wins the reference's
        // scan unless it is offered first (below); inside a pair of the tournament it would shield its partner: x > NaN is false.
        // -inf never wins either (nothing is greater than the incumbent through it), and the tournament is exact without NaNs.
#pragma unroll
        for (int b = 1; b < R; b++) v[b] = v[b] != v[b] ? -INFINITY : v[b];
    }
    const bool c01 = v[1] > v[0], c23 = v[3] > v[2];
    const double m01 = vmax_f64(v[0], v[1]), m23 = vmax_f64(v[2], v[3]);
    const bool c = m23 > m01;
    unsigned idx = c ? (c23 ? 3u : 2u) : (c01 ? 1u : 0u);
    if constexpr (R == 5) {
        const double m = vmax_f64(m01, m23);
        idx = v[4] > m ? 4u : idx;
        // (a NaN -- zero-count candidates with the marginal term: log10(0) + an infinite conditional -- offered FIRST is the
        // incumbent of the scan and nothing compares greater than it; anywhere else it never wins, as here: kernels.hpp, argmax8.
        // The ranked layout never sees one: its columns are candidates that were observed.)
        idx = v[0] != v[0] ? 0u : idx;
    }
    return idx;
}

// min without the canonicalising v_max(x,x) pair the compiler adds around fmin (no NaNs here: marginals, +inf)
__device__ __forceinline__ double vmin_f64(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// TRACK: spins without k_emit -- per entry state also the minimum marginal of its picks (smin)
// halo rows of k_rwseg: for the LC positions in front of the segment, the table row of the path's symbol as it stands AFTER the
// reweight this launch applies (the neighbour who owns those positions writes it to G in this very launch: what G holds
// there when this workgroup stages its slice is one or the other)
struct seg_patch {
    double row[SEG_MAX_L_NARROW][SEG_MAX_L_NARROW][LT_ROW];   // [halo position][lag - 1][column]   (column mode: [..][..][digit row])
    int row6[SEG_MAX_L_NARROW];                                // the row it replaces (digit), -1 = none
    // conditionals C / E: a reweighted cell moves one COLUMN of the block of (position, lag) -- the entry of every digit row
    int col[SEG_MAX_L_NARROW][SEG_MAX_L_NARROW];               // the column, -1 = none
    unsigned rmask[SEG_MAX_L_NARROW][SEG_MAX_L_NARROW];        // the digit rows that exist
    int colmode;
    // marginal term: log10 marginal of the halo positions after the reweight, by candidate rank (R = 4) and by symbol (R = 5) --
    // rinfo / minfo of those positions are being rewritten by the neighbour while this workgroup stages
    double lm4[SEG_MAX_L_NARROW][4], lm5[SEG_MAX_L_NARROW][5];
    // candidate bits (compact order) after the reweight of every position this workgroup holds -- the halo first (slots 0 .. L-1,
    // workgroups behind the first), then its own: the mixed-radix extension ranks its digits through them (segmix.hpp)
    uint8_t cmall[SEG_THREADS / 8];
};

// (NANP: sums can be NaN -- a separate instantiation: the sanitising selects in the arg-max of the 5^L loop cost the wide window a third
// of its speed when they hung on a run-time flag)
template <int R, int LC, bool TRACK, bool NANP = false>
__device__ __forceinline__ void seg_body(const seg_params &P, unsigned char *smem, const seg_patch *patch = nullptr)
{
    typedef typename seg_radix<R>::next_t next_t;
    constexpr int BITS = seg_radix<R>::BITS;
    constexpr unsigned MASK = (1u << BITS) - 1u;
    constexpr unsigned NS = seg_ipow(R, LC), NI = NS / R, RR = R * R;
    constexpr int CH = seg_chunk(R, LC);
    constexpr int SPT = (NS + SEG_THREADS - 1) / SEG_THREADS;      // states per thread
    const seg_geom g = seg_geometry(P.N, LC, R);
    const int s = blockIdx.x, tid = threadIdx.x;
    if (s >= g.S) return;
    const int t0 = s * g.seglen;                                   // targets t0+1 .. t1
    const int t1 = t0 + g.seglen < P.N ? t0 + g.seglen : P.N;
    double *Gs = reinterpret_cast<double *>(smem);                 // [(CH + LC - 1)][LC][R][R]
    next_t *Nx = reinterpret_cast<next_t *>(Gs + (size_t)(CH + LC - 1) * LC * RR);   // [CH][NI]
    double *Ms = reinterpret_cast<double *>(smem + seg_lds_bytes(R, LC));            // [CH][R] marginal of column b at every target
    constexpr bool nanp = NANP;
    double mn[SPT];                                                // minimum marginal of the picks, per entry state
#pragma unroll
    for (int q = 0; q < SPT; q++) mn[q] = INFINITY;

    unsigned sigma[SPT];                                           // (threads beyond NS walk state 0 and store nothing)
#pragma unroll
    for (int q = 0; q < SPT; q++) sigma[q] = (unsigned)(tid + q * SEG_THREADS) < NS ? (unsigned)(tid + q * SEG_THREADS) : 0u;

    SEG_STAMP(0);
    for (int c0 = t0; c0 < t1; c0 += CH) {
        const int nc = t1 - c0 < CH ? t1 - c0 : CH;                // targets c0+1 .. c0+nc
        // (1a) the slice of G this chunk needs: sources c0+1-LC .. c0+nc-1 (slot ii = i - (c0+1-LC)), every lag, the
        // rows of the R digits and the R candidate columns.  Position 0 carries '_' whatever the digit says (row 5);
        // positions < 0 do not exist: their terms are +0.0, which leaves every partial sum as it is.
        // One thread per (slot, lag, digit): the R columns are one run of G (wide loads; G rows are 40 bytes apart, so
        // the loads are 8-byte aligned only -- fine for global memory) and one aligned run of Gs.  With the marginal
        // term the lag-1 entries get log10 marginal(b, i + 1) added in front: (0.0 + lm) + x1, the reference's first
        // addition -- by candidate rank (rinfo) or by symbol (minfo), like the columns.
        const int nsrc = nc + LC - 1;
        for (int e = tid; e < nsrc * LC * R; e += SEG_THREADS) {
            const int d = e % R, l = (e / R) % LC, ii = e / (R * LC);
            const int i = c0 + 1 - LC + ii;
            double v[R];
#pragma unroll
            for (int b = 0; b < R; b++) v[b] = 0.0;
            if (i >= 0) {
                const double *src = P.G + (((size_t)i * 6 + (i == 0 ? 5 : d)) * LC + l) * LT_ROW;
#pragma unroll
                for (int b = 0; b < R; b++) v[b] = src[b];
                if (P.mt && l == 0) {
                    const double *lm = R == 4 ? P.rinfo + (size_t)(i + 1) * RINFO : P.minfo + (size_t)(i + 1) * MINFO;
                    // (k_rwseg: position i + 1 in front of the segment belongs to the neighbour, who is rewriting it: from the patch)
                    if (patch && c0 == t0 && t0 > 0 && ii + 1 < LC) lm = R == 4 ? patch->lm4[ii + 1] : patch->lm5[ii + 1];
#pragma unroll
                    for (int b = 0; b < R; b++) v[b] = lm[b] + v[b];
                }
            }
            double *dst = Gs + (size_t)e * R;
#pragma unroll
            for (int b = 0; b < R; b++) dst[b] = v[b];
        }
        // the marginals of the columns at targets c0+1 .. c0+nc (by rank or by symbol, like the columns)
        if (patch && c0 == t0 && t0 > 0) {
            // the first chunk's slots 0 .. LC-1 are the halo sources t0+1-LC .. t0: their path rows from the patch
            __syncthreads();
            for (int e = tid; e < LC * LC * R; e += SEG_THREADS) {
                const int b = e % R, l = (e / R) % LC, hp = e / (R * LC);
                if (patch->colmode) {
                    // column mode: b runs over the digit rows
                    const int col = patch->col[hp][l];
                    if (col >= 0 && col < R && ((patch->rmask[hp][l] >> b) & 1u)) {
                        double v = patch->row[hp][l][b];
                        if (P.mt && l == 0) {
                            const int i = t0 + 1 - LC + hp;
                            const double *lm = R == 4 ? P.rinfo + (size_t)(i + 1) * RINFO : P.minfo + (size_t)(i + 1) * MINFO;
                            if (hp + 1 < LC) lm = R == 4 ? patch->lm4[hp + 1] : patch->lm5[hp + 1];
                            v = lm[col] + v;
                        }
                        Gs[((size_t)(hp * LC + l) * R + b) * R + col] = v;
                    }
                    continue;
                }
                const int d = patch->row6[hp];
                if (d >= 0 && d < R) {
                    double v = patch->row[hp][l][b];
                    if (P.mt && l == 0) {
                        const int i = t0 + 1 - LC + hp;
                        const double *lm = R == 4 ? P.rinfo + (size_t)(i + 1) * RINFO : P.minfo + (size_t)(i + 1) * MINFO;
                        if (hp + 1 < LC) lm = R == 4 ? patch->lm4[hp + 1] : patch->lm5[hp + 1];
                        v = lm[b] + v;
                    }
                    Gs[((size_t)(hp * LC + l) * R + d) * R + b] = v;
                }
            }
        }
        if (TRACK) {
            for (int e = tid; e < nc * R; e += SEG_THREADS) {
                const int tl = e / R, b = e - tl * R;
                const int t = c0 + 1 + tl;
                Ms[e] = R == 4 ? P.rinfo[(size_t)t * RINFO + 4 + b] : P.minfo[(size_t)t * MINFO + 5 + b];
            }
        }
        __syncthreads();
        SEG_STAMP(1);
        // (1b) Next for every (target, state).  Entry idx of a position holds the digits d_1 .. d_{L-1} (d_1 lowest) and
        // the R picks for the R values of the oldest digit d_L.  A task takes the digits d_1 .. d_{L-1-SH} as given and
        // loops over the SH oldest but one itself (SH = 1 for L >= 4: the partial sum over the young lags and the R x R
        // terms of lag L are shared by R entries, a third of the LDS reads and a fifth fewer additions).  Lag l of
        // chunk-local target tl comes from slot tl + LC - l.  Same additions, same order, for every state.
        {
            constexpr int SH = LC >= 4 ? 1 : 0;                    // digits looped inside a task besides d_L
            constexpr unsigned NJ = NI / (SH ? R : 1);             // tasks per position
            for (unsigned task = tid; task < (unsigned)nc * NJ; task += SEG_THREADS) {
                const unsigned tl = task / NJ, j = task - tl * NJ;
                double xl[R][R];                                   // lag L, every value of d_L
#pragma unroll
                for (int dL = 0; dL < R; dL++) {
                    const double *row = Gs + ((size_t)(tl * LC + (LC - 1)) * R + dL) * R;
#pragma unroll
                    for (int b = 0; b < R; b++) xl[dL][b] = row[b];
                }
                double acc[R];
                unsigned rem = j;
                constexpr int NYOUNG = LC - 1 - SH;                // lags summed before the in-task loop
                if constexpr (NYOUNG >= 1) {
#pragma unroll
                    for (int l = 1; l <= NYOUNG; l++) {
                        const unsigned d = rem % R;
                        rem /= R;
                        const double *row = Gs + ((size_t)((tl + LC - l) * LC + (l - 1)) * R + d) * R;
#pragma unroll
                        for (int b = 0; b < R; b++) acc[b] = l == 1 ? row[b] : acc[b] + row[b];
                    }
                }
#pragma unroll
                for (int dS = 0; dS < (SH ? R : 1); dS++) {
                    double acc2[R];
                    if constexpr (SH) {                            // lag L-1 under digit dS
                        const double *row = Gs + ((size_t)((tl + 1) * LC + (LC - 2)) * R + dS) * R;
#pragma unroll
                        for (int b = 0; b < R; b++) acc2[b] = NYOUNG >= 1 ? acc[b] + row[b] : row[b];
                    } else {
#pragma unroll
                        for (int b = 0; b < R; b++) acc2[b] = acc[b];
                    }
                    unsigned packed = 0;
#pragma unroll
                    for (int dL = 0; dL < R; dL++) {
                        double v[R];
#pragma unroll
                        for (int b = 0; b < R; b++) v[b] = LC >= 2 ? acc2[b] + xl[dL][b] : xl[dL][b];
                        packed |= seg_argmax<R>(v, nanp) << (BITS * dL);
                    }
                    Nx[(size_t)tl * NI + j + (SH ? dS * NJ : 0)] = (next_t)packed;
                }
            }
        }
        __syncthreads();
        SEG_STAMP(2);
        // (2) every entry state through the chunk; its picks go to hist one word (DPW picks) at a time
        {
            constexpr int DPW = seg_radix<R>::DPW;
            const int w0 = (c0 - t0) / DPW;                        // chunks are whole words
            auto walk_word = [&](int tw, int nu, auto full_) {
                constexpr bool full = decltype(full_)::value;      // a whole word: no per-step bound checks in the chain
                unsigned word[SPT];
#pragma unroll
                for (int q = 0; q < SPT; q++) word[q] = 0;
                const next_t *rows = Nx + (size_t)tw * NI;
#pragma unroll
                for (int u = 0; u < DPW; u++) {
                    if (full || u < nu) {
#pragma unroll
                        for (int q = 0; q < SPT; q++) {
                            const unsigned sg = sigma[q];
                            const unsigned hi = sg / NI, idx = sg - hi * NI;
                            const unsigned d = ((unsigned)rows[u * NI + idx] >> (BITS * hi)) & MASK;
                            word[q] |= d << (BITS * u);
                            sigma[q] = idx * R + d;
                            if (TRACK) mn[q] = vmin_f64(mn[q], Ms[(tw + u) * R + d]);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < SPT; q++) {
                    const unsigned s0 = tid + q * SEG_THREADS;
                    if (s0 < NS) P.hist[((size_t)s * g.NW + w0 + tw / DPW) * NS + s0] = word[q];
                }
            };
            int tw = 0;
            for (; tw + DPW <= nc; tw += DPW) walk_word(tw, DPW, std::true_type{});
            if (tw < nc) walk_word(tw, nc - tw, std::false_type{});
        }
        SEG_STAMP(3);
        __syncthreads();                                           // Gs / Nx are overwritten by the next chunk
    }
#pragma unroll
    for (int q = 0; q < SPT; q++) {
        const unsigned s0 = tid + q * SEG_THREADS;
        if (s0 < NS) {
            P.maps[(size_t)s * NS + s0] = (uint16_t)sigma[q];
            if (TRACK) P.smin[(size_t)s * NS + s0] = mn[q];
        }
    }
    SEG_STAMP(4);
}

#include "segmix.hpp"

template <int LC, bool TRACK>
__global__ void __launch_bounds__(SEG_THREADS) k_seg(seg_params P)
{
    extern __shared__ __align__(16) unsigned char seg_smem[];
    dev_state *st = P.st;
    const dev_ctl c = load_ctl(st);
    if (c.stop || c.lt_stale) return;
    if (P.check_masks == 2 || (P.check_masks && c.cm_same == 0)) {      // (2: forced, tests)
        // k_marg<T,true> saw a candidate mask change: V(p) and the -inf masks in G moved, the rows it rewrote are not
        // enough.  Every kernel queued behind this one returns at once; the host rebuilds G and queues the paths again.
        if (blockIdx.x == 0 && threadIdx.x == 0) st->lt_stale = 1;
        return;
    }
    // the flags k_marg left for this path; k_scan re-arms them for the next k_marg, k_emit reads the copy
    if (blockIdx.x == 0 && threadIdx.x == 0) st->cur_hole = c.first_hole;
    const int cls = __builtin_amdgcn_readfirstlane(seg_class(c, LC));
    if (cls == 4) seg_body<4, LC, TRACK>(P, seg_smem);
    else if constexpr (seg_radix_ok(5, LC)) {                  // (beyond: the host only launches this for ranked tables)
        if constexpr (LC == SEGM_L && !TRACK) { if (cls == SEG_CLS_MIXED) { seg_body_mixed<LC, false>(P, seg_smem); return; } }
        if (P.nanp) seg_body<5, LC, TRACK, true>(P, seg_smem);
        else seg_body<5, LC, TRACK, false>(P, seg_smem);
    }
}

// -------------------------------------------------------------------------------------------------------------
// k_scan: group map = composition of the group's segment maps
// -------------------------------------------------------------------------------------------------------------
template <int R, int LC, bool TRACK>
__device__ __forceinline__ void scan_body(const seg_params &P, unsigned char *smem)
{
    constexpr int NS = seg_ns(R, LC);
    const seg_geom g = seg_geometry(P.N, LC, R);
    const int grp = blockIdx.x, tid = threadIdx.x;
    if (grp >= g.G1) return;
    const int s_lo = grp * g.G2;
    const int n = (s_lo + g.G2 <= g.S ? g.G2 : g.S - s_lo);
    uint16_t *M = reinterpret_cast<uint16_t *>(smem);              // [n][NS]
    const uint16_t *src = P.maps + (size_t)s_lo * NS;
    if constexpr (NS % 8 == 0) {
        // (16 bytes per thread and trip: a map of 2048 states two bytes at a time was most of what this kernel did)
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(M);
        for (int e = tid; e < n * (NS / 8); e += SEG_THREADS) d4[e] = s4[e];
    } else {
        for (int e = tid; e < n * NS; e += SEG_THREADS) M[e] = src[e];
    }
    // the candidate bits of this group's positions as this path's kernels see them: the reweight behind rewrites them
    // while its neighbours still map ranks to symbols (k_rw without k_emit reads this copy)
    if (TRACK) {
        const int t_lo = s_lo * g.seglen + 1;
        int t_hi = (s_lo + n) * g.seglen;
        if (t_hi > P.N) t_hi = P.N;
        for (int t = t_lo + tid; t <= t_hi; t += SEG_THREADS)
            P.cm5snap[t] = (uint8_t)__double_as_longlong(P.minfo[(size_t)t * MINFO + 10]);
    }
    __syncthreads();
    for (int s0 = tid; s0 < NS; s0 += SEG_THREADS) {
        int x = s0;
        // (map, min) pairs compose along the chain: the group's minimum for entry state s0 is the minimum over its segments of
        // what each holds for the state that enters IT.  The gathers go out as the states become known, the minimum is taken last.
        double mv[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            mv[j] = INFINITY;
            if (j < n) {
                if (j > 0) P.pmaps[(size_t)(s_lo + j) * NS + s0] = (uint16_t)x;      // what enters segment s_lo + j
                if (TRACK) mv[j] = P.smin[(size_t)(s_lo + j) * NS + x];
                x = M[(size_t)j * NS + x];
            }
        }
        P.gmaps[(size_t)grp * NS + s0] = (uint16_t)x;
        if (TRACK) {
            double m = mv[0];
#pragma unroll
            for (int j = 1; j < 16; j++) m = vmin_f64(m, mv[j]);
            P.gmin[(size_t)grp * NS + s0] = m;
        }
    }
}

template <int LC, bool TRACK>
__global__ void __launch_bounds__(SEG_THREADS) k_scan(seg_params P)
{
    extern __shared__ __align__(16) unsigned char seg_smem[];
    dev_state *st = P.st;
    const dev_ctl c = load_ctl(st);
    if (c.stop || c.lt_stale) return;
    int cur_hole = c.cur_hole;
    if (P.rws) {
        // behind k_rwseg the flags of the reweight only stand now: k_seg's look at them happens here (from the control words the
        // reweight's atomics left: reducing the workgroups' flag words as k_emit_small does costs this kernel a round trip, 0.85 us)
        if (P.check_masks == 2 || (P.check_masks && c.cm_same == 0)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) st->lt_stale = 1;
            return;
        }
        cur_hole = c.first_hole;
        if (blockIdx.x == 0 && threadIdx.x == 0) st->cur_hole = c.first_hole;
    }
    // (a path that ends in a hole is followed by no k_marg: the flags must stand, as after the serial walkers)
    if (P.rearm && cur_hole > P.N && blockIdx.x == 0 && threadIdx.x == 0) {
        st->first_hole = 0x7f7f7f7f; st->nodel = 0x7f7f7f7f; st->cm_same = 0x7f7f7f7f; st->narrow = 0x7f7f7f7f;
    }
    const int cls = __builtin_amdgcn_readfirstlane(seg_class(c, LC));
    if (cls == 4) scan_body<4, LC, TRACK>(P, seg_smem);
    else if constexpr (seg_radix_ok(5, LC)) {
        if constexpr (LC == SEGM_L && !TRACK) { if (cls == SEG_CLS_MIXED) { scan_body<SEG_CLS_MIXED, LC, false>(P, seg_smem); return; } }
        scan_body<5, LC, TRACK>(P, seg_smem);
    }
}

// -------------------------------------------------------------------------------------------------------------
// k_emit
// -------------------------------------------------------------------------------------------------------------
// SMALL: the segment maps of the whole window fit the LDS (S * NS * 2 bytes <= 64 KB: short memories, small windows):
// this kernel composes them itself -- group maps by all threads, then the chain -- and no k_scan runs in front of it.
template <int R, int LC, bool SMALL = false>
__device__ __forceinline__ void emit_body(const seg_params &P, unsigned char *smem, int first_hole)
{
    constexpr int BITS = seg_radix<R>::BITS, DPW = seg_radix<R>::DPW;
    constexpr unsigned MASK = (1u << BITS) - 1u;
    constexpr int NS = seg_ns(R, LC);
    __shared__ int s_sigma;
    __shared__ double s_min[SEG_THREADS / 64];
    const seg_geom g = seg_geometry(P.N, LC, R);
    const int s = blockIdx.x, tid = threadIdx.x;
    if (s >= g.S) return;
    const int Nw = first_hole <= P.N ? first_hole - 1 : P.N;       // positions that can be decided (gretel.py:176-180)
    const int t0 = s * g.seglen;
    int t1 = t0 + g.seglen < P.N ? t0 + g.seglen : P.N;
    if (t1 > Nw) t1 = Nw;
    if (s == 0 && tid == 0) { P.path_out[0] = SYM_US; P.lmsel[0] = 1.0; }      // gretel.py:138; k_hp: sums still to be taken
    if (P.halo && s + 1 < g.S) {
        // for k_rwseg: the band blocks of this segment's last LC positions (the halo of the next one) as they stand now
        const int te = t0 + g.seglen < P.N ? t0 + g.seglen : P.N;
        const size_t words = (size_t)LC * NSYM * P.W * NSYM * P.esz / 4;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(static_cast<const unsigned char *>(P.band) + (size_t)(te + 1 - LC) * NSYM * P.W * NSYM * P.esz);
        uint32_t *dst = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(P.halo) + (size_t)(s + 1) * LC * NSYM * P.W * NSYM * P.esz);
        for (size_t e = tid; e < words; e += SEG_THREADS) dst[e] = src[e];
    }
    if (t0 >= t1) {
        if (tid == 0) P.segmin[s] = INFINITY;
        return;
    }
    const int grp = s / g.G2, q = s - grp * g.G2;
    uint16_t *GM = reinterpret_cast<uint16_t *>(smem);             // [grp][NS] group maps in front of this group
    uint16_t *PM = GM + (size_t)grp * NS;                          // [NS] prefix map of this segment (q >= 1)
    uint16_t *MS = GM + (((size_t)g.G1 * NS + 1) & ~(size_t)1);   // SMALL: [s][NS] the segment maps in front of this one (4-byte aligned)
    if constexpr (SMALL) {
        // the maps of segments 0 .. s-1 (what lies behind this segment does not matter to it)
        const uint32_t *src = reinterpret_cast<const uint32_t *>(P.maps);
        uint32_t *dst = reinterpret_cast<uint32_t *>(MS);
        for (int e = tid; e < s * NS / 2 + 1; e += SEG_THREADS) dst[e] = src[e];
    } else if constexpr (NS % 8 == 0) {
        const uint4 *g4 = reinterpret_cast<const uint4 *>(P.gmaps);
        uint4 *d4 = reinterpret_cast<uint4 *>(GM);
        for (int e = tid; e < grp * (NS / 8); e += SEG_THREADS) d4[e] = g4[e];
        if (q > 0) {
            const uint4 *p4 = reinterpret_cast<const uint4 *>(P.pmaps + (size_t)s * NS);
            uint4 *m4 = reinterpret_cast<uint4 *>(PM);
            for (int e = tid; e < NS / 8; e += SEG_THREADS) m4[e] = p4[e];
        }
    } else {
        for (int e = tid; e < grp * NS; e += SEG_THREADS) GM[e] = P.gmaps[e];
        if (q > 0) {
            const uint16_t *src = P.pmaps + (size_t)s * NS;
            for (int e = tid; e < NS; e += SEG_THREADS) PM[e] = src[e];
        }
    }
    // the rows of minfo do not depend on the path: in flight under the chain (one position per thread and trip)
    const int npos = t1 - t0;
    lds_v2d row[8];
    if (tid < npos) {
        const lds_v2d *src = reinterpret_cast<const lds_v2d *>(P.minfo + (size_t)(t0 + 1 + tid) * MINFO);
#pragma unroll
        for (int k = 0; k < 8; k++) row[k] = src[k];
    }
    __syncthreads();
    if constexpr (SMALL) {
        // the maps of the whole groups in front of this one, every state of every such group on its own thread
        for (int e = tid; e < grp * NS; e += SEG_THREADS) {
            const int gj = e / NS;
            int x = e - gj * NS;
            for (int j = 0; j < g.G2; j++) x = MS[(size_t)(gj * g.G2 + j) * NS + x];
            GM[e] = (uint16_t)x;
        }
        __syncthreads();
        if (tid == 0) {
            int x = 0;
            for (int j = 0; j < grp; j++) x = GM[(size_t)j * NS + x];
            for (int j = 0; j < q; j++) x = MS[(size_t)(grp * g.G2 + j) * NS + x];
            s_sigma = x;
        }
    } else if (tid == 0) {
        // the true entry state: the start state 0 through the maps of the groups in front, then the prefix map
        int x = 0;
        for (int j = 0; j < grp; j++) x = GM[(size_t)j * NS + x];
        if (q > 0) x = PM[x];
        s_sigma = x;
    }
    __syncthreads();
    const int sg = s_sigma;
    // bookkeeping of gretel.py:182-187 for positions t0+1 .. t1 (the sums themselves: k_hp)
    double mn = INFINITY;
    for (int tl = tid; tl < npos; tl += SEG_THREADS) {
        const int t = t0 + 1 + tl;
        if (tl >= SEG_THREADS) {                                    // segments longer than the workgroup: later trips load here
            const lds_v2d *src = reinterpret_cast<const lds_v2d *>(P.minfo + (size_t)t * MINFO);
#pragma unroll
            for (int k = 0; k < 8; k++) row[k] = src[k];
        }
        const unsigned word = P.hist[((size_t)s * g.NW + tl / DPW) * NS + sg];
        int b5 = (int)((word >> (BITS * (tl % DPW))) & MASK);
        const double r16[16] = {row[0].x, row[0].y, row[1].x, row[1].y, row[2].x, row[2].y, row[3].x, row[3].y,
                                row[4].x, row[4].y, row[5].x, row[5].y, row[6].x, row[6].y, row[7].x, row[7].y};
        if (R != 5) {                                               // rank -> symbol through the candidate bits (ranked layout, mixed radix)
            b5 = nth_set5((uint32_t)__double_as_longlong(r16[10]), b5);
            if (b5 < 0) b5 = 0;     // cannot happen for a decided position
        }
        double lm = r16[0], m = r16[5];
#pragma unroll
        for (int k = 1; k < 5; k++) {
            lm = (b5 == k) ? r16[k] : lm;
            m = (b5 == k) ? r16[5 + k] : m;
        }
        P.path_out[t] = (uint8_t)vsym(P.sm, b5);
        P.lmsel[t] = lm;
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
        for (int w = 1; w < SEG_THREADS / 64; w++)
            if (s_min[w] < mn) mn = s_min[w];
        P.segmin[s] = mn;
    }
}

template <int LC>
__global__ void __launch_bounds__(SEG_THREADS) k_emit(seg_params P)
{
    extern __shared__ __align__(16) unsigned char seg_smem[];
    dev_state *st = P.st;
    const dev_ctl c = load_ctl(st);
    if (c.stop || c.lt_stale) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) st->dbg[3] = 3;      // gh_debug_walk_clock: variant 3 = segment-parallel
    const int cls = __builtin_amdgcn_readfirstlane(seg_class(c, LC));
    if (cls == 4) emit_body<4, LC>(P, seg_smem, c.cur_hole);
    else if constexpr (seg_radix_ok(5, LC)) {
        if constexpr (LC == SEGM_L) { if (cls == SEG_CLS_MIXED) { emit_body<SEG_CLS_MIXED, LC>(P, seg_smem, c.cur_hole); return; } }
        emit_body<5, LC>(P, seg_smem, c.cur_hole);
    }
}

// k_scan and k_emit in one launch where the window's maps fit the LDS (emit_body<.., true>)
template <int LC>
__global__ void __launch_bounds__(SEG_THREADS) k_emit_small(seg_params P)
{
    extern __shared__ __align__(16) unsigned char seg_smem[];
    dev_state *st = P.st;
    const dev_ctl c = load_ctl(st);
    if (c.stop || c.lt_stale) return;
    int cur_hole = c.cur_hole;
    if (P.rws) {
        // behind k_rwseg (two launches per path in small windows): k_scan's look at what the reweight found happens here, in
        // every workgroup for itself -- nobody reads the control words the first thread rewrites
        const int S = seg_geometry(P.N, LC, seg_class(c, LC)).S;
        if (!rws_take_flags(P, st, S, cur_hole)) return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->dbg[3] = 3;
        // (k_scan's other job: the flags for the reweight that follows -- every k_seg workgroup has read them: it ran in the launch before)
        if (P.rearm && cur_hole > P.N) { st->first_hole = 0x7f7f7f7f; st->nodel = 0x7f7f7f7f; st->cm_same = 0x7f7f7f7f; st->narrow = 0x7f7f7f7f; }
    }
    const int cls = __builtin_amdgcn_readfirstlane(seg_class(c, LC));
    if (cls == 4) emit_body<4, LC, true>(P, seg_smem, cur_hole);
    else if constexpr (seg_radix_ok(5, LC)) {
        if constexpr (LC == SEGM_L) { if (cls == SEG_CLS_MIXED) { emit_body<SEG_CLS_MIXED, LC, true>(P, seg_smem, cur_hole); return; } }
        emit_body<5, LC, true>(P, seg_smem, cur_hole);
    }
}

// what the serial walkers' bookkeeper does at the end of a walk: hole -> stop, else the record and the ratio the
// reweight will use (gretel.py:176-180,189; cmd.py:157-160).  minm = minimum over segmin.
__device__ __forceinline__ void seg_finish(dev_state *st, gh_path_rec *rec, int N, double minm, double min_remove)
{
    if (st->cur_hole <= N) {
        st->stop = 1;
        st->hole_at = st->cur_hole;
        return;
    }
    double r = minm;
    if (r < min_remove) r = min_remove;
    rec->ratio = r;
    rec->min_marginal = minm;
    rec->magnitude = 0.0;
    st->ratio = r;
    st->n_done += 1;
}

__device__ __forceinline__ int seg_count(const dev_state *st, int N, int L)
{
    return seg_geometry(N, L, seg_class(load_ctl(st), L)).S;
}

// -------------------------------------------------------------------------------------------------------------
// k_rw: what k_marg<T, true> does behind a serial walker -- reweight the path's cells (gretel/gretel.py:79-98, same
// multiplicities), the marginals of every position from the cell (p, p+1) just updated, the rows of the conditional
// table the path changed -- behind a segment-parallel walk, where it also reduces the minimum marginal over the
// segments, clamps it (cmd.py:157-160) and closes the path record.  Same thread layout (8 lanes per position), same
// arithmetic in the same order, so the same bits; what differs is the order of the MEMORY operations: a kernel this
// small is a chain of dependent round trips (each about a microsecond), so every load whose address does not depend
// on the path's symbols or on the ratio is issued up front, the band row a lane rewrites is read ONCE (the element it
// reweights, the row sum of the conditional and the row's entries all come out of those registers), and nothing is
// read back after a store.
// -------------------------------------------------------------------------------------------------------------
// LP lanes per position: 8, or 32 for bands / lag counts above 8 (every distance and every lag in one round)
// COL: conditionals C and E, whose denominator is a COLUMN sum of the cell: reweighting element (path[p], path[p+d]) of
// cell (p, p+d) then moves the entries G[p][x][d-1][col of path[p+d]] of EVERY from-row x -- one column of the table
// block instead of one row (the same number of entries); the lane reads the cell's column instead of its row.
template <typename V>
__device__ __forceinline__ V pick7(const V (&r)[NSYM], int x)
{
    V v = r[0];
#pragma unroll
    for (int q = 1; q < NSYM; q++) v = (x == q) ? r[q] : v;
    return v;
}

// k_rw without a k_emit in front of it (fz.hist != nullptr; spins over the enumerated states): what it reads instead of
// `path` and `segmin`.  Everything here was written by the k_seg / k_scan of this path and is not touched by this kernel.
struct fuse_params {
    const uint32_t *hist;     // [S][NW][NS]
    const uint16_t *pmaps;    // [S][NS]
    const uint16_t *gmaps;    // [G1][NS]
    const double *gmin;       // [G1][NS]
    const uint8_t *cm5snap;   // [N+2]
    uint8_t *path_out;        // [N+1]  this path's symbols (every workgroup writes its own positions)
    double *lmsel;            // [N+1]  log10 marginal of every selected symbol, as it was when the path was walked (k_hp)
};
#define RW_FUSE_HALO 64       /* fused: band widths up to this (the picks of the W positions behind a workgroup's own) */
#define RW_FUSE_SEGS 24       /* ... and at most this many segments under one workgroup's positions + halo */
// segments under the positions of one workgroup (ppb of them) and the W behind: their prefix maps are staged too
__host__ __device__ inline int rw_fuse_nts(int N, int L, int R, int ppb, int W)
{
    const seg_geom g = seg_geometry(N, L, R);
    int n = (ppb + W + g.seglen - 1) / g.seglen + 1;
    return n < g.S ? n : g.S;
}
__host__ __device__ inline size_t rw_fuse_lds_bytes(int N, int L, int R, int ppb, int W)
{
    const seg_geom g = seg_geometry(N, L, R);
    // the group maps; the prefix maps of the segments underneath; entry states of the groups and segments; picks
    return (((size_t)g.G1 * g.NS * 2 + 15) & ~(size_t)15) + (size_t)rw_fuse_nts(N, L, R, ppb, W) * g.NS * 2 + 512;
}

// (row mode within 72 VGPRs = seven waves per SIMD: at C5 the kernel is a stream of dependent round trips and its
// throughput follows the waves in flight -- 78 registers, six waves, cost 10 of 50 us)
#ifndef RW_COL_WAVES
#define RW_COL_WAVES 4          /* (5: 96 registers and spills, no faster; 6: 80 registers, 240 bytes of scratch per lane, 100 us at C5) */
#endif
template <typename T, int LP, bool COL, bool FUSED>
__global__ void __launch_bounds__(256, FUSED ? 4 : (COL ? RW_COL_WAVES : 7))
k_rw(T *band, int N, int W, double *cnt, double *marg, int32_t *nvalid, uint32_t *cmask, double *minfo, dev_state *st,
     const uint8_t *path, double min_remove, double *partial, double *G, int L, int cond_mode, const double *segmin,
     gh_path_rec *rec, int nseg_arg, symmap sm, int offer_zero, double *rinfo, int stage, fuse_params fz, int fuse_lds, T *tband)
{
    __shared__ double s_min4[4], s_sum4[4];
    __shared__ double s_logtab[256];
    extern __shared__ __align__(16) unsigned char rw_smem_all[];
    logtab_stage(s_logtab);                                 // (the barriers of the minimum's reduction stand between this and the first logarithm)
    const int tid = threadIdx.x;
    constexpr int PPB = 256 / LP;                           // positions per workgroup
    constexpr bool fused = FUSED;                           // (a compile-time switch: the prologue's registers cost the plain kernel a wave per SIMD)
    unsigned char *rw_smem = rw_smem_all + fuse_lds;        // (the fused prologue's LDS lies in front of the staged band block)
    RW_STAMP(0);
    // COL, stage != 0: the band block of this workgroup's positions goes to LDS as it lies in memory (one contiguous run,
    // 16-byte loads, issued before anything else): the columns are then LDS reads.  Strided 4-byte column loads straight
    // from memory measured 6-12 us for k_rw<float, 8, true> at C3 (every line of the band, cold, one sector per lane).
    T *blk = reinterpret_cast<T *>(rw_smem);
    const size_t pos_elems = (size_t)NSYM * W * NSYM;
    if (COL && (stage & 1)) {
        const int p0 = blockIdx.x * PPB;
        const int np = p0 + PPB <= N + 2 ? PPB : (N + 2 > p0 ? N + 2 - p0 : 0);
        typedef T vecT __attribute__((ext_vector_type(16 / sizeof(T))));
        const size_t nv = (size_t)np * pos_elems * sizeof(T) / 16;
        const vecT *src = reinterpret_cast<const vecT *>(band + (size_t)p0 * pos_elems);
        vecT *dst = reinterpret_cast<vecT *>(blk);
        for (size_t q = tid; q < nv; q += 256) dst[q] = src[q];
    }
    const dev_ctl c = load_ctl(st);
    const int Rr = seg_class(c, L);                         // (the fused flow never runs over the mixed radix: mixed_allowed)
    const seg_geom sg = seg_geometry(N, L, Rr);
    const int nseg = nseg_arg > 0 ? nseg_arg : sg.S;
    double my_segmin = INFINITY;                            // (<= 512 segments: two per thread at most)
    uint16_t *GM = reinterpret_cast<uint16_t *>(rw_smem_all);           // fused: [G1][NS] group maps
    const int nts = rw_fuse_nts(N, L, Rr, PPB, W);
    uint16_t *PMs = reinterpret_cast<uint16_t *>(rw_smem_all + (((size_t)sg.G1 * sg.NS * 2 + 15) & ~(size_t)15));      // [nts][NS] prefix maps
    int *xs = reinterpret_cast<int *>(PMs + (size_t)nts * sg.NS);      // [32] entry state of every group
    int *ent = xs + 32;                                     // [RW_FUSE_SEGS] entry state of the segments under this workgroup
    uint8_t *sym = reinterpret_cast<uint8_t *>(ent + RW_FUSE_SEGS);     // [PPB + halo] symbol picked at p0 + u ...
    uint8_t *pk5 = sym + 128;                               // ... and its compact index
    if (!fused) {
        for (int q = tid; q < nseg; q += 256) {
            const double v = segmin[q];
            if (v < my_segmin) my_segmin = v;
        }
    } else {
        // the group maps -> LDS (they do not depend on anything this kernel computes): 16-byte copies, the allocation has slack
        const uint4 *src = reinterpret_cast<const uint4 *>(fz.gmaps);
        uint4 *dst = reinterpret_cast<uint4 *>(GM);
        const int n128 = (sg.G1 * sg.NS * 2 + 15) / 16;
        for (int q = tid; q < n128; q += 256) dst[q] = src[q];
        // ... and the prefix maps of the segments under this workgroup's positions and halo (2 KB each at 1024 states)
        const int p0 = blockIdx.x * PPB;
        const int s_first = p0 >= 1 ? (p0 - 1) / sg.seglen : 0;
        const int ns = s_first + nts <= sg.S ? nts : sg.S - s_first;
        const uint4 *ps = reinterpret_cast<const uint4 *>(fz.pmaps + (size_t)s_first * sg.NS);      // (NS * 2 bytes: a multiple of 16 or
        uint4 *pd = reinterpret_cast<uint4 *>(PMs);                                                   //  the allocation's slack covers the tail)
        const int np128 = ns > 0 ? (ns * sg.NS * 2 + 15) / 16 : 0;
        for (int q = tid; q < np128; q += 256) pd[q] = ps[q];
    }
    if (c.stop || c.lt_stale || c.cw_unres) return;
    if (c.cur_hole <= N) {                                  // the walk ended in a hole: nothing to reweight (gretel.py:176-180)
        if (blockIdx.x == 0 && tid == 0) seg_finish(st, rec, N, 0.0, min_remove);
        return;
    }
    const int t = blockIdx.x * 256 + tid;
    const int p = t / LP, s = t % LP;
    const bool act = p <= N;
    const int pp = act ? p : 0;
    const int p0w = blockIdx.x * PPB;                       // first position of this workgroup
    // fused: what the positions of this workgroup and the W behind them need before the path's symbols are known -- the
    // candidate bits k_seg saw (rank -> symbol), this position's log-marginals as the walk saw them (lmsel)
    int cm5_u = 0;
    double lm_old = 0.0;
    if (fused) {
        const int tu = p0w + tid;                           // thread u < PPB + halo: position p0 + u
        if (tid < PPB + RW_FUSE_HALO && tu >= 1 && tu <= N) cm5_u = fz.cm5snap[tu];
        if (act && s < 5) lm_old = minfo[(size_t)pp * MINFO + s];
    }
    // ---- round 1 ---------------------------------------------------------------------------------------------
    const int d0 = s + 1;                                   // this lane's distance (reweight) and lag (table row / column)
    const int j0 = pp + d0;
    int mult0 = 0;                                          // multiplicities of the reference's pair enumeration
    if (act && d0 <= W) {
        if (j0 <= N - 1) mult0 = (d0 == 1) ? 2 : 1;
        else if (j0 == N) mult0 = (d0 == 1) ? 1 : 0;
        else if (j0 == N + 1) mult0 = (pp == N) ? 1 : 0;
    }
    int a = 0, b0 = 0;
    if (!fused) {
        a = path[pp];
        b0 = mult0 ? ((j0 == N + 1) ? path[0] : path[j0]) : 0;
    }
    // this lane owns the table entries of lag d0: a row (every lag up to L), or a column (only where the cell changed)
    const bool lag_row = G && act && pp < N && d0 <= L && j0 <= N && (!COL || (mult0 > 0 && d0 <= W));
    // the row sums of cell (p, p+1) = the counts at p.  A reweight moves one element of that cell, in the row of the path's symbol:
    // every other row's sum is the one the pass before left in `cnt` (stage & 2: the host knows cnt to be current), and that
    // row is the one lane 0 reads anyway -- one 64-byte line per position instead of seven (bands of 20: 384 of ~1000 bytes read)
    // COL with the to-major copy of the band (stage & 4; tband[bidx(W, p, d, b, a)] = band[bidx(W, p, d, a, b)]): the COLUMN a lane
    // needs is a contiguous run there, nothing is staged, and the row sums come from cnt as well -- lane 0 then also reads the row of
    // the path's symbol of cell (p, p+1) from the band itself (rowA), the one row whose sum moves
    const bool tb = COL && (stage & 4);
    const bool cnt_ok = (!COL || tb) && (stage & 2);
    T crow[NSYM];
    T rowA[NSYM];
    double cold = 0.0;
    if (cnt_ok) {
        cold = (act && s < NSYM) ? cnt[(size_t)pp * 8 + s] : 0.0;
        if (COL) {
#pragma unroll
            for (int x = 0; x < NSYM; x++) rowA[x] = (act && s == 0) ? band[bidx(W, pp, 1, a, x)] : (T)0;
        }
    } else if (!(COL && (stage & 1))) {
#pragma unroll
        for (int x = 0; x < NSYM; x++) crow[x] = (act && s < NSYM) ? band[bidx(W, pp, 1, s, x)] : (T)0;      // cell (p, p+1), row s
    }
    int nv_t = 0;
    uint32_t cm_t = 0;
    if (lag_row) { nv_t = nvalid[j0]; cm_t = CM_CAND(cmask[j0]); }
    const uint32_t cm_old = (act && s == 7) ? cmask[pp] : 0u;
    static_assert(LP == 8 || LP == 16 || LP == 32, "lane groups of 8, 16 or 32");
    // ---- the path's minimum marginal, while round 1 is in flight ----------------------------------------------
    if (fused) {
        // (1) the start state through the group maps: the state that enters every group; and, through the prefix maps, every
        // segment under this workgroup's positions and halo (a group's first segment is entered as the group is)
        const int s_first = p0w >= 1 ? (p0w - 1) / sg.seglen : 0;
        __syncthreads();
        if (tid == 0) {
            int x = 0;
            for (int j = 0; j < sg.G1; j++) { xs[j] = x; x = GM[(size_t)j * sg.NS + x]; }
            for (int k = 0; k < nts && s_first + k < sg.S; k++) {
                const int sq = s_first + k, grp = sq / sg.G2;
                ent[k] = sq > grp * sg.G2 ? (int)PMs[(size_t)k * sg.NS + xs[grp]] : xs[grp];
            }
        }
        __syncthreads();
        // (2) ONE round trip: the groups' minima for those states (their minimum is the path's), and the picks of positions
        // p0 .. p0 + PPB - 1 + halo out of hist, for their segment's true entry state
        if (tid < sg.G1) my_segmin = fz.gmin[(size_t)tid * sg.NS + xs[tid]];
        if (tid < PPB + RW_FUSE_HALO) {
            const int tu = p0w + tid;
            int sy = SYM_US, k5 = 0;
            if (tu >= 1 && tu <= N && tid < PPB + W) {
                const int sq = (tu - 1) / sg.seglen, tl = (tu - 1) - sq * sg.seglen;
                const int bits = Rr == 4 ? 2 : 3, dpw = Rr == 4 ? 16 : 10;
                const unsigned word = fz.hist[((size_t)sq * sg.NW + tl / dpw) * sg.NS + ent[sq - s_first]];
                k5 = (int)((word >> (bits * (tl % dpw))) & ((1u << bits) - 1u));
                if (Rr == 4) { k5 = nth_set5((uint32_t)cm5_u, k5); if (k5 < 0) k5 = 0; }
                sy = vsym(sm, k5);
            }
            sym[tid] = (uint8_t)sy;
            pk5[tid] = (uint8_t)k5;
        }
        __syncthreads();
        const int pl = tid / LP;                            // this lane's position inside the workgroup
        a = sym[pl];
        b0 = mult0 ? ((j0 == N + 1) ? SYM_US : sym[j0 - p0w]) : 0;
        if (blockIdx.x == 0 && tid == 0) st->dbg[3] = 3;    // gh_debug_walk_clock: variant 3 = segment-parallel (k_emit's mark)
        // this workgroup's share of the path record: symbols and selected log-marginals of its own positions
        if (act && s == 0) {
            fz.path_out[p] = p == 0 ? (uint8_t)SYM_US : (uint8_t)a;
        }
        {
            const int k5 = pk5[pl];
            const double lmv = __shfl(lm_old, k5, LP);
            if (act && s == 0) fz.lmsel[p] = p == 0 ? 1.0 : lmv;       // ([0] = 1.0: k_hp still has this path's sums to take)
        }
    }
    // (inside the wavefront by lane exchange, the four wavefronts through LDS: one barrier where a tree over 256 slots takes nine)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double w = __shfl_xor(my_segmin, o);
        my_segmin = w < my_segmin ? w : my_segmin;
    }
    if ((tid & 63) == 0) s_min4[tid >> 6] = my_segmin;
    __syncthreads();
    double minm = s_min4[0];
#pragma unroll
    for (int q = 1; q < 4; q++) minm = s_min4[q] < minm ? s_min4[q] : minm;
    RW_STAMP(1);
    const double ratio = minm < min_remove ? min_remove : minm;
    if (blockIdx.x == 0 && tid == 0) seg_finish(st, rec, N, minm, min_remove);
    // ---- round 2: the row of cell (p, p+d0) under the path's symbol at p -- or its column under the symbol at p+d0 ---
    const bool need_row = act && d0 <= W && (mult0 > 0 || lag_row);
    T *rowp = COL ? band + bidx(W, pp, d0 <= W ? d0 : 1, 0, b0) : band + bidx(W, pp, d0 <= W ? d0 : 1, a, 0);
    constexpr int one = 1;
    const int estride = COL ? W * NSYM : one;               // elements between two entries of the run this lane reads
    const int esel = COL ? a : b0;                          // the entry of the run that is reweighted
    // (no branch around the loads: a lane without a run reads the start of its position's block and drops it -- behind a branch the
    // compiler waits for each of the seven strided loads before it issues the next: 7 round trips, 15 us for k_rw<float, 8, COL>)
    const T *runp = need_row ? rowp : band + bidx(W, pp, 1, 0, 0);      // (its own position's block)
    T rrow[NSYM];
    if (COL && (stage & 1)) {
        // (the reduction above has passed barriers behind the staging stores)
        const T *lp = blk + (size_t)(tid / LP) * pos_elems;                // this position's block in LDS
#pragma unroll
        for (int x = 0; x < NSYM; x++) crow[x] = (act && s < NSYM) ? lp[((size_t)s * W) * NSYM + x] : (T)0;
        const T *lc = lp + ((size_t)(d0 <= W ? d0 : 1) - 1) * NSYM + b0;
#pragma unroll
        for (int x = 0; x < NSYM; x++) rrow[x] = lc[(size_t)x * estride];
    } else if (COL && tb) {
        const T *tcol = tband + bidx(W, pp, d0 <= W ? d0 : 1, b0, 0);      // the column of symbol b0 of cell (p, p+d0): from-symbols 0..6
#pragma unroll
        for (int x = 0; x < NSYM; x++) rrow[x] = need_row ? tcol[x] : (T)0;
    } else if (COL) {
#pragma unroll
        for (int x = 0; x < NSYM; x++) rrow[x] = runp[(size_t)x * estride];
    } else {
        // (a row is one contiguous run: the compiler fetches it with two wide loads, and a lane without one issues nothing --
        // with bands of 20 a third of the lanes have no row, and this kernel is bound by the memory instructions it issues)
#pragma unroll
        for (int x = 0; x < NSYM; x++) rrow[x] = need_row ? rowp[x] : (T)0;
    }
#pragma unroll
    for (int x = 0; x < NSYM; x++) rrow[x] = need_row ? rrow[x] : (T)0;
    RW_STAMP(2);
    // ---- reweight ---------------------------------------------------------------------------------------------
    double removed = 0.0;
    int na = -1, nb = -1;
    T nval = (T)0;
    if (mult0) {
        T cur = rrow[0];
#pragma unroll
        for (int x = 1; x < NSYM; x++) cur = (x == esel) ? rrow[x] : cur;
        for (int q = 0; q < mult0; q++) {
            const double old = (double)cur;
            const double nw = old - ratio * old;
            cur = (T)nw;
            removed += old - nw;
        }
        rowp[(size_t)esel * estride] = cur;
        if (COL && tb) tband[bidx(W, pp, d0, b0, a)] = cur;
#pragma unroll
        for (int x = 0; x < NSYM; x++) rrow[x] = (x == esel) ? cur : rrow[x];
        if (d0 == 1) { na = a; nb = b0; nval = cur; }
    }
    na = __shfl(na, 0, LP); nb = __shfl(nb, 0, LP);
    nval = (T)__shfl((double)nval, 0, LP);
    // ---- marginals of position p (k_marg, same order of operations) ---------------------------------------------
    unsigned flag_bits = 0;
    int hole_p = 0x7fffffff;
    double cs[NSYM];
    double tot = 0.0;
    int nv = 0;
    uint32_t cm = 0, cm5 = 0;
    T acc = (T)0;
    double mine;
    if (cnt_ok) {
        // lane 0 holds the row of the path's symbol (the rewritten element in place): the same left-to-right sum in the storage type
#pragma unroll
        for (int x = 0; x < NSYM; x++) acc = acc + (COL ? ((x == nb) ? nval : rowA[x]) : rrow[x]);
        const double suma = __shfl((double)acc, 0, LP);
        mine = (s == na) ? suma : cold;
    } else {
#pragma unroll
        for (int x = 0; x < NSYM; x++) {
            T v = crow[x];
            if (s == na && x == nb) v = nval;               // the element this group has just rewritten
            acc = acc + v;
        }
        mine = (double)acc;
    }
#pragma unroll
    for (int x = 0; x < NSYM; x++) {
        cs[x] = __shfl(mine, x, LP);
        if (cs[x] > 0) {
            tot += cs[x];
            if ((VALID_MASK >> x) & 1) { nv++; cm |= 1u << x; }
        }
    }
    const uint32_t cand = offer_zero ? VALID_MASK : cm;     // (k_marg)
    const uint32_t cmw = cm | (cand << 8);
    cm5 = cm5_of_cmask(sm, cand);
    if (act) {
        if (s < NSYM) {
            const double m = (cs[s] > 0 && tot != 0.0) ? cs[s] / tot : 0.0;
            cnt[(size_t)p * 8 + s] = cs[s];
            marg[(size_t)p * 8 + s] = m;
            if ((VALID_MASK >> s) & 1) {
                const int b5 = a6_of_sym(sm, s);
                const double lm = gh_log10_tab(m, s_logtab, GH_LOG_BOTH);
                minfo[(size_t)p * MINFO + b5] = lm;
                minfo[(size_t)p * MINFO + 5 + b5] = m;
                const int r = __popc(cm5 & ((1u << b5) - 1u));
                if (rinfo && ((cand >> s) & 1u) && r < 4) {
                    rinfo[(size_t)p * RINFO + r] = lm;
                    rinfo[(size_t)p * RINFO + 4 + r] = m;
                }
            }
        } else if (s == 7) {
            cnt[(size_t)p * 8 + 7] = tot;
            marg[(size_t)p * 8 + 7] = 0.0;
            nvalid[p] = nv;
            cmask[p] = cmw;
            minfo[(size_t)p * MINFO + 10] = __longlong_as_double((long long)cm5);
            for (int r = __popc(cm5); rinfo && r < 4; r++) {
                rinfo[(size_t)p * RINFO + r] = 0.0;
                rinfo[(size_t)p * RINFO + 4 + r] = INFINITY;
            }
            // the window's flags: collected per workgroup in LDS, one global atomic per flag and workgroup (thousands of
            // positions and-ing the same word one by one cost 12 ns each: 0.2 ms per path in a window full of '-')
            if (cm_old != cmw) flag_bits |= 1u;                    // the conditional table must then be rebuilt in full
            if (p >= 1 && (cm5 & (1u << 4))) flag_bits |= 2u;      // the LAST symbol of the candidate order ('-' by default) is offered here
            if (p >= 1 && __popc(cm5) > 4) flag_bits |= 4u;
            if (p >= 1 && cand == 0) hole_p = p;
        }
    }
    RW_STAMP(3);
    const bool ranked = c.ranked != 0;
    const double nv_i = (double)nv;
    // ---- bands wider than the lane group: the remaining distances, one by one (and, COL, their table columns) -------
    // COL: entries G[p][x][l-1][col(b)] for every from-row x that exists, from the column `colT` of cell (p, p+l)
    // one entry of a column: from-symbol FS[q] of lag l, value of the cell's element `cval`, into column `col` of its row
    auto col_entry = [&](int l, int q, double cval, double den, int col) __attribute__((always_inline)) {
        const int x6 = a6_of_sym(sm, q < 4 ? q : q + 1);       // the six from-symbols a path can hold: 0 1 2 3 5 6
        int row6 = x6;
        if (x6 == 5) { if (p != 0) return; }                   // the '_' row exists at position 0 only
        else if (ranked) {
            if (!((cm5 >> x6) & 1u)) return;                   // not a candidate of p: no row
            row6 = __popc(cm5 & ((1u << x6) - 1u));
            if (row6 > 3) return;
        }
        G[(((size_t)p * 6 + row6) * L + (l - 1)) * LT_ROW + col] = gh_log10_tab((1.0 + cval) / den, s_logtab, GH_LOG_BOTH);
    };
    // the denominator and the column index of lag l (col < 0: no column for this symbol -- N, '_', or no candidate: a rebuild follows)
    auto col_head = [&](const T (&colT)[NSYM], int nvt, uint32_t cmt, int b, double &den) __attribute__((always_inline)) -> int {
        T cacc = (T)0;
#pragma unroll
        for (int x = 0; x < NSYM; x++) cacc = cacc + colT[x];
        den = (cond_mode == GH_COND_C ? nv_i : (double)nvt) + (double)cacc;
        const int b5c = a6_of_sym(sm, b);
        const uint32_t cj5 = cm5_of_cmask(sm, cmt);
        if (b5c >= 5 || !((cj5 >> b5c) & 1u)) return -1;
        return ranked ? __popc(cj5 & ((1u << b5c) - 1u)) : b5c;
    };
    auto table_col = [&](int l, const T (&colT)[NSYM], int nvt, uint32_t cmt, int b) __attribute__((always_inline)) {
        T cacc = (T)0;
#pragma unroll
        for (int x = 0; x < NSYM; x++) cacc = cacc + colT[x];
        const double den = (cond_mode == GH_COND_C ? nv_i : (double)nvt) + (double)cacc;
        const int b5c = a6_of_sym(sm, b);
        const uint32_t cj5 = cm5_of_cmask(sm, cmt);
        if (b5c >= 5 || !((cj5 >> b5c) & 1u)) return;        // no column for this symbol (N, '_', or no candidate: a rebuild follows)
        const int col = ranked ? __popc(cj5 & ((1u << b5c) - 1u)) : b5c;
        // the six from-symbols a path can hold, in SYMBOL order (compile-time register indices); the order the candidates are
        // offered in only moves the row an entry is stored in
        constexpr int FS[6] = {0, 1, 2, 3, 5, 6};
        double xq[6], v[6];
        bool odd = false;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            xq[q] = (1.0 + (double)colT[FS[q]]) / den;
            odd |= !gh_log10_is_normal(xq[q]);
        }
#pragma unroll
        for (int q = 0; q < 6; q++) v[q] = gh_log10_normal_tab(xq[q], 0, s_logtab, GH_LOG_BOTH);
        if (odd) {
#pragma unroll
            for (int q = 0; q < 6; q++) v[q] = gh_log10_tab(xq[q], s_logtab, GH_LOG_BOTH);
        }
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const int x6 = a6_of_sym(sm, FS[q]);
            int row6 = x6;
            if (x6 == 5) { if (p != 0) continue; }           // the '_' row exists at position 0 only
            else if (ranked) {
                if (!((cm5 >> x6) & 1u)) continue;           // not a candidate of p: no row
                row6 = __popc(cm5 & ((1u << x6) - 1u));
                if (row6 > 3) continue;
            }
            G[(((size_t)p * 6 + row6) * L + (l - 1)) * LT_ROW + col] = v[q];
        }
    };
    for (int d = d0 + LP; act && d <= W; d += LP) {
        const int j = p + d;
        int mult = 0;
        if (j <= N - 1) mult = 1;
        else if (j == N + 1) mult = (p == N) ? 1 : 0;
        if (mult) {
            const int b = (j == N + 1) ? (fused ? SYM_US : (int)path[0]) : (fused ? (int)sym[j - p0w] : (int)path[j]);
            T *e = band + bidx(W, p, d, a, b);
            const double old = (double)*e;
            const double nw = old - ratio * old;
            *e = (T)nw;
            if (COL && tb) tband[bidx(W, p, d, b, a)] = (T)nw;
            removed += old - nw;
            if (COL && G && p < N && d <= L && j <= N) {
                T colT[NSYM];
#pragma unroll
                for (int x = 0; x < NSYM; x++) colT[x] = tb ? tband[bidx(W, p, d, b, x)] : band[bidx(W, p, d, x, b)];      // (all seven in flight; the element just rewritten is replaced)
#pragma unroll
                for (int x = 0; x < NSYM; x++) colT[x] = (x == a) ? (T)nw : colT[x];
                if constexpr (LP == 8) table_col(d, colT, nvalid[j], CM_CAND(cmask[j]), b);
                else {
                    double den;
                    const int col = col_head(colT, nvalid[j], CM_CAND(cmask[j]), b, den);
#pragma unroll 1
                    for (int q = 0; q < 6 && col >= 0; q++) col_entry(d, q, (double)pick7(colT, q < 4 ? q : q + 1), den, col);
                }
            }
        }
    }
    // ---- the table entries of lag d0: the row of path[p] (same divisions, same log10 as k_lt), or the column ---------
    if (COL) {
        if constexpr (LP >= 16) {
            // as the rows below: the columns go through LDS and their 6 L entries are dealt out over the whole lane group -- six
            // divisions and logarithms per lane held forty registers more than the kernel has at six waves per SIMD
            __shared__ T s_cols[256 / LP][LP][NSYM];
            __shared__ double s_cden[256 / LP][LP];
            __shared__ int s_ccol[256 / LP][LP];
            const int pl = tid / LP;
            const int Lr = L < LP ? L : LP;
            if (d0 <= Lr) {
                double den = 1.0;
                int col = -1;
                if (lag_row) {
                    col = col_head(rrow, nv_t, cm_t, b0, den);
#pragma unroll
                    for (int x = 0; x < NSYM; x++) s_cols[pl][s][x] = rrow[x];
                }
                s_cden[pl][s] = den;
                s_ccol[pl][s] = col;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int e = s; e < Lr * 6; e += LP) {
                const int l1 = e / 6, q = e - l1 * 6;
                const int col = s_ccol[pl][l1];
                if (col >= 0) col_entry(l1 + 1, q, (double)s_cols[pl][l1][q < 4 ? q : q + 1], s_cden[pl][l1], col);
            }
        } else if (lag_row) table_col(d0, rrow, nv_t, cm_t, b0);
    } else if (G && act && p < N && a != 4) {
        const int a6 = a6_of_sym(sm, a);
        const double ca = __shfl(mine, a, LP);
        int row6 = a6;
        if (ranked && a6 < 5) row6 = ((cm5 >> a6) & 1u) ? __popc(cm5 & ((1u << a6) - 1u)) : -1;
        auto table_row = [&](int l, const T (&rowT)[NSYM], int nvt, uint32_t cmt) __attribute__((always_inline)) {
            double *out = G + (((size_t)p * 6 + row6) * L + (l - 1)) * LT_ROW;
            if (!(p + l <= N && (a6 < 5 || p == 0))) {
#pragma unroll
                for (int b5 = 0; b5 < LT_ROW; b5++) out[b5] = 0.0;
                return;
            }
            double rowv[NSYM];
            T racc = (T)0;
#pragma unroll
            for (int x = 0; x < NSYM; x++) { rowv[x] = (double)rowT[x]; racc = racc + rowT[x]; }      // zeros beyond the band
            const double sum = (double)racc;
            const double den = (cond_mode == GH_COND_A) ? (double)nvt + sum : (cond_mode == GH_COND_D ? nv_i + sum : nv_i + ca);
            // the five candidates in SYMBOL order (compile-time register indices); the order they are offered in only moves
            // the column an entry is stored in
            constexpr int VS[LT_ROW] = {0, 1, 2, 3, 5};
            double xq[LT_ROW], v[LT_ROW];
            bool odd = false;
#pragma unroll
            for (int q = 0; q < LT_ROW; q++) xq[q] = (1.0 + rowv[VS[q]]) / den;
#pragma unroll
            for (int q = 0; q < LT_ROW; q++) odd |= !gh_log10_is_normal(xq[q]);
#pragma unroll
            for (int q = 0; q < LT_ROW; q++) v[q] = gh_log10_normal_tab(xq[q], 0, s_logtab, GH_LOG_BOTH);
            if (odd) {
#pragma unroll
                for (int q = 0; q < LT_ROW; q++) v[q] = gh_log10_tab(xq[q], s_logtab, GH_LOG_BOTH);
            }
            if (sm.fwd == 0x53210u) {
                // the default order: compact index q IS the q-th valid symbol -- the row is put together in registers and leaves
                // as one contiguous 40-byte run (C5: the kernel is bound by its stores)
                double r[LT_ROW];
                if (!ranked) {
#pragma unroll
                    for (int q = 0; q < LT_ROW; q++) r[q] = ((cmt >> VS[q]) & 1) ? v[q] : -INFINITY;
                } else {
                    const uint32_t cj5 = (cmt & 15u) | (((cmt >> 5) & 1u) << 4);
#pragma unroll
                    for (int rb = 0; rb < LT_ROW; rb++) {
                        const int b5 = nth_set5(cj5, rb);
                        double x = -INFINITY;
#pragma unroll
                        for (int q = 0; q < LT_ROW; q++) x = (b5 == q) ? v[q] : x;
                        r[rb] = x;
                    }
                }
#pragma unroll
                for (int q = 0; q < LT_ROW; q++) out[q] = r[q];
            } else if (!ranked) {
#pragma unroll
                for (int q = 0; q < LT_ROW; q++) out[a6_of_sym(sm, VS[q])] = ((cmt >> VS[q]) & 1) ? v[q] : -INFINITY;
            } else {
                // column = the candidate's rank at the target; ranks that do not exist hold -inf
                const uint32_t cj5 = cm5_of_cmask(sm, cmt);
#pragma unroll
                for (int q = 0; q < LT_ROW; q++) {
                    const int b5 = a6_of_sym(sm, VS[q]);
                    if ((cj5 >> b5) & 1u) out[__popc(cj5 & ((1u << b5) - 1u))] = v[q];
                }
                for (int rb = __popc(cj5); rb < LT_ROW; rb++) out[rb] = -INFINITY;
            }
        };
        if constexpr (LP >= 16) {
            // Lane groups of 16 / 32 (bands or lag counts above 8): one lane per lag would leave the five divisions and logarithms of a
            // row to L of 32 lanes -- at C5 (L = 11) a third of every wavefront works while this kernel is bound by exactly that
            // arithmetic.  The rows go through LDS instead (same wavefront: no barrier) and the 5 L entries are dealt out over all the group's
            // lanes; entry e = (lag - 1) * 5 + column lies at out[e], so the lanes' stores are one contiguous run.  Same division,
            // same log10 on the same operands: the bits do not depend on the lane that computes them.
            __shared__ T s_rows[256 / LP][LP][NSYM];
            __shared__ double s_den[256 / LP][LP];
            __shared__ uint32_t s_cmt[256 / LP][LP];
            const int pl = tid / LP;
            const int Lr = L < LP ? L : LP;
            if (d0 <= Lr) {
                T racc = (T)0;
#pragma unroll
                for (int x = 0; x < NSYM; x++) { s_rows[pl][s][x] = rrow[x]; racc = racc + rrow[x]; }      // zeros beyond the band
                const double sum = (double)racc;
                s_den[pl][s] = (cond_mode == GH_COND_A) ? (double)nv_t + sum : (cond_mode == GH_COND_D ? nv_i + sum : nv_i + ca);
                // the symbol behind each of the five columns of this lag, 4 bits each (15: no such candidate -> -inf)
                uint32_t colsym = 0;
#pragma unroll
                for (int col = 0; col < LT_ROW; col++) {
                    const int b5 = ranked ? nth_set5(cm5_of_cmask(sm, cm_t), col) : col;
                    const int sy = b5 >= 0 ? vsym(sm, b5) : 0;
                    colsym |= (uint32_t)((b5 >= 0 && ((cm_t >> sy) & 1u)) ? sy : 15) << (4 * col);
                }
                s_cmt[pl][s] = colsym;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (row6 >= 0) {
                double *out = G + ((size_t)p * 6 + row6) * L * LT_ROW;
                const bool row_live = a6 < 5 || p == 0;
                for (int e = s; e < Lr * LT_ROW; e += LP) {
                    const int l1 = e / LT_ROW, col = e - l1 * LT_ROW;
                    double r = 0.0;
                    if (p + l1 + 1 <= N && row_live) {
                        const int sy = (int)((s_cmt[pl][l1] >> (4 * col)) & 15u);
                        r = -INFINITY;
                        if (sy != 15) {
                            const double xq = (1.0 + (double)s_rows[pl][l1][sy]) / s_den[pl][l1];
                            r = gh_log10_is_normal(xq) ? gh_log10_normal_tab(xq, 0, s_logtab, GH_LOG_BOTH) : gh_log10_tab(xq, s_logtab, GH_LOG_BOTH);
                        }
                    }
                    out[e] = r;
                }
            }
        } else if (d0 <= L && row6 >= 0) table_row(d0, rrow, nv_t, cm_t);
        for (int l = d0 + LP; l <= L && row6 >= 0; l += LP) {    // lag counts above the lane group: the remaining lags, one by one
            T rowT[NSYM];
#pragma unroll
            for (int x = 0; x < NSYM; x++) rowT[x] = (T)0;
            int nvt = 0;
            uint32_t cmt = 0;
            if (p + l <= N) {
                if (l <= W) {
                    const T *rc = band + bidx(W, p, l, a, 0);      // (this lane's own store included)
#pragma unroll
                    for (int x = 0; x < NSYM; x++) rowT[x] = rc[x];
                }
                nvt = nvalid[p + l];
                cmt = CM_CAND(cmask[p + l]);
            }
            table_row(l, rowT, nvt, cmt);
        }
    }
    RW_STAMP(4);
    // ---- removed mass: fixed-order tree, as k_marg ----------------------------------------------------------------
    __shared__ unsigned s_flags;
    __shared__ int s_hole;
    if (tid == 0) { s_flags = 0; s_hole = 0x7fffffff; }
    // (a fixed tree -- butterflies inside the wavefront, then the four wavefronts in order -- so the sum is the same from run to run)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o);
    if ((tid & 63) == 0) s_sum4[tid >> 6] = removed;
    __syncthreads();
    if (flag_bits) atomicOr(&s_flags, flag_bits);
    if (hole_p != 0x7fffffff) atomicMin(&s_hole, hole_p);
    __syncthreads();
    if (tid == 0) {
        partial[blockIdx.x] = (s_sum4[0] + s_sum4[1]) + (s_sum4[2] + s_sum4[3]);
        const unsigned f = s_flags;
        if (f & 1u) atomicAnd(&st->cm_same, 0);
        if (f & 2u) atomicAnd(&st->nodel, 0);
        if (f & 4u) atomicAnd(&st->narrow, 0);
        if (s_hole != 0x7fffffff) atomicMin(&st->first_hole, s_hole);
    }
    RW_STAMP(5);
}

// -------------------------------------------------------------------------------------------------------------
// k_rwseg: the reweight of path k-1 and the k_seg of path k in ONE launch -- three dependent launches per path instead of
// four, and the table rows a path's reweight writes are used by the very workgroup that wrote them.
//
// Workgroup s owns the positions of segment s (t0+1 .. t1; workgroup 0 also position 0) and does for them what k_rw does
// (8 lanes per position: reweight the cells on the path, the marginals, the table row of the path's symbol) -- then walks
// segment s as k_seg does.  Its k_seg needs the table rows of the LC positions in front of the segment too, and those are
// rewritten by its neighbour in this same launch: it recomputes exactly those rows itself, from a copy of the neighbour's
// band blocks that k_emit made one launch earlier (seg_params::halo), and patches them into its LDS slice.  No workgroup
// waits for another, nothing is read that somebody else is writing (except, as in k_rw, nvalid / cmask of the targets,
// which only count while no candidate mask moves).
// The flags the reweight leaves (first hole, moved masks) are only final when the launch has ended: k_scan looks at them
// (seg_params::rws).  COL (conditionals C, E): the cell's column instead of its row, the band blocks of the workgroup's positions
// staged in LDS, the patch carries columns.  Lane groups of 8 (W <= 8), at most 128 positions per workgroup with the halo: gh_spin checks.
// -------------------------------------------------------------------------------------------------------------
// minimum over the 64 lanes of a wavefront, in every lane, without LDS: a minimum does not mind seeing a value twice, so the
// exchanges are the cheap DPP patterns (neighbour, other pair, other quad, other half of the row of 16), then the four rows
__device__ __forceinline__ double lane_f64(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double wave_min_f64(double v)
{
    double x;
    x = dpp_f64<0xB1>(v); v = x < v ? x : v;          // quad_perm [1,0,3,2]
    x = dpp_f64<0x4E>(v); v = x < v ? x : v;          // quad_perm [2,3,0,1]
    x = dpp_f64<0x141>(v); v = x < v ? x : v;         // row_half_mirror
    x = dpp_f64<0x140>(v); v = x < v ? x : v;         // row_mirror
    const double r0 = lane_f64(v, 0), r1 = lane_f64(v, 16), r2 = lane_f64(v, 32), r3 = lane_f64(v, 48);
    const double a = r1 < r0 ? r1 : r0, b = r3 < r2 ? r3 : r2;
    return b < a ? b : a;
}

struct rws_params {
    void *band;
    double *cnt, *marg, *minfo, *rinfo, *G;
    int32_t *nvalid;
    uint32_t *cmask;
    const uint8_t *path;      // the path to reweight (k - 1)
    double min_remove;
    double *partial;          // removed mass, one per workgroup
    gh_path_rec *rec;
    int cond_mode, offer_zero;
};

template <typename T, int LC, bool COL>
__device__ __forceinline__ bool rwseg_reweight(const seg_params &P, const rws_params &Q, const dev_ctl &c, unsigned char *smem, seg_patch *patch)
{
    const int N = P.N, W = P.W, L = LC, tid = threadIdx.x, sblk = blockIdx.x;
    const symmap sm = P.sm;
    dev_state *st = P.st;
    double *sred = reinterpret_cast<double *>(smem);                  // [SEG_THREADS] scratch (k_seg's regions are not in use yet)
    __shared__ double s_logtab[256];
    logtab_stage(s_logtab);
    if (tid < SEG_MAX_L_NARROW) patch->row6[tid] = -1;
    if (tid < SEG_MAX_L_NARROW * SEG_MAX_L_NARROW) { (&patch->col[0][0])[tid] = -1; (&patch->rmask[0][0])[tid] = 0u; }
    if (tid == 0) patch->colmode = COL ? 1 : 0;
    const int R = seg_class(c, L);
    const seg_geom g = seg_geometry(N, L, R);
    if (sblk >= g.S) { if (tid == 0) { Q.partial[sblk] = 0.0; P.rwflags[sblk] = make_int2(0, 0x7fffffff); } return false; }
    RWS_STAMP(0);
    // ---- the path's minimum marginal over the segments: every wavefront takes it for itself -- a few loads per lane, issued with
    // the loads of the path, the cells and the masks, then lane exchanges (wave_min_f64): no barrier, no LDS
    double vs[7];                                                     // (at most 400 segments: seg_geometry)
#pragma unroll
    for (int k = 0; k < 7; k++) { const int q = (tid & 63) + 64 * k; vs[k] = q < g.S ? P.segmin[q] : INFINITY; }
    // ---- this workgroup's positions: the halo (recomputed, nothing stored), then its own ---------------------------
    const int t0 = sblk * g.seglen;
    const int t1 = t0 + g.seglen < N ? t0 + g.seglen : N;
    const int nh = sblk > 0 ? L : 0;                                  // halo positions t0+1-L .. t0
    const int nown = (t1 - t0) + (sblk == 0 ? 1 : 0);                 // own: t0+1 .. t1, and position 0 for workgroup 0
    const int slot = tid >> 3, s = tid & 7;
    const bool in = slot < nh + nown;
    const bool halo = in && slot < nh;
    const int p = !in ? 0 : (halo ? t0 + 1 - L + slot : (sblk == 0 ? slot : t0 + 1 + (slot - nh)));
    const bool act = in && p <= N;
    const bool own = act && !halo;
    const size_t pos_elems = (size_t)NSYM * W * NSYM;
    T *bandT = static_cast<T *>(Q.band);
    const T *blk = halo ? static_cast<const T *>(P.halo) + ((size_t)sblk * L + slot) * pos_elems : bandT + (size_t)p * pos_elems;
    if (COL) {
        // column mode: the band blocks of all this workgroup's positions -> LDS (behind the reduction scratch), as they lie in
        // memory: the halo from k_emit's copy, the own ones from the band -- two contiguous runs, 16-byte loads (k_rw's staging)
        T *blkL = reinterpret_cast<T *>(smem + (size_t)SEG_THREADS * 8);
        const size_t wper = pos_elems * sizeof(T) / 4;                // 4-byte words per position (a block is 196 W bytes: 16-byte
        const uint32_t *srcH = reinterpret_cast<const uint32_t *>(static_cast<const T *>(P.halo) + (size_t)sblk * L * pos_elems);      // copies only fit W = 4, 8)
        const int p_first = sblk == 0 ? 0 : t0 + 1;
        const uint32_t *srcO = reinterpret_cast<const uint32_t *>(bandT + (size_t)p_first * pos_elems);
        uint32_t *dst = reinterpret_cast<uint32_t *>(blkL);
        for (size_t q = tid; q < (size_t)nh * wper; q += SEG_THREADS) dst[q] = srcH[q];
        for (size_t q = tid; q < (size_t)nown * wper; q += SEG_THREADS) dst[(size_t)nh * wper + q] = srcO[q];
        __syncthreads();
        blk = blkL + (size_t)(in ? slot : 0) * pos_elems;
    }
    const int d0 = s + 1, j0 = p + d0;
    int mult0 = 0;
    if (act && d0 <= W) {
        if (j0 <= N - 1) mult0 = (d0 == 1) ? 2 : 1;
        else if (j0 == N) mult0 = (d0 == 1) ? 1 : 0;
        else if (j0 == N + 1) mult0 = (p == N) ? 1 : 0;
    }
    const int a = act ? Q.path[p] : 0;
    const int b0 = mult0 ? ((j0 == N + 1) ? Q.path[0] : Q.path[j0]) : 0;
    const bool lag_row = Q.G && act && p < N && d0 <= L && j0 <= N && (!COL || (mult0 > 0 && d0 <= W));
    T crow[NSYM];
#pragma unroll
    for (int x = 0; x < NSYM; x++) crow[x] = (act && s < NSYM) ? blk[((size_t)s * W) * NSYM + x] : (T)0;     // cell (p, p+1), row s
    int nv_t = 0;
    uint32_t cm_t = 0;
    if (lag_row) { nv_t = Q.nvalid[j0]; cm_t = CM_CAND(Q.cmask[j0]); }
    const uint32_t cm_old = (own && s == 7) ? Q.cmask[p] : 0u;
    const bool need_row = act && d0 <= W && (mult0 > 0 || lag_row);
    const size_t roff = ((size_t)a * W + (size_t)((d0 <= W ? d0 : 1) - 1)) * NSYM;
    // the run this lane works on: the ROW of the path's symbol at p, or (COL) the COLUMN of the path's symbol at p + d0
    const size_t run0 = COL ? ((size_t)((d0 <= W ? d0 : 1) - 1)) * NSYM + b0 : roff;
    const size_t rstride = COL ? (size_t)W * NSYM : 1;
    const int esel = COL ? a : b0;                                  // the entry of the run that is reweighted
    T rrow[NSYM];
#pragma unroll
    for (int x = 0; x < NSYM; x++) rrow[x] = need_row ? blk[run0 + (size_t)x * rstride] : (T)0;
    double v0 = vs[0];
#pragma unroll
    for (int k = 1; k < 7; k++) v0 = vs[k] < v0 ? vs[k] : v0;
    const double minm = wave_min_f64(v0);
    RWS_STAMP(1);
    // (the patch was armed by lanes of wavefront 0 and is written by the halo lanes, which are lanes of wavefront 0 too: in order)
    static_assert(SEG_MAX_L_NARROW * 8 <= 64 && SEG_MAX_L_NARROW * SEG_MAX_L_NARROW <= 64, "halo lane groups and the patch's arming lanes in one wavefront");
    const double ratio = minm < Q.min_remove ? Q.min_remove : minm;
    if (sblk == 0 && tid == 0) seg_finish(st, Q.rec, N, minm, Q.min_remove);
    RWS_STAMP(2);
    double removed = 0.0;
    int na = -1, nb = -1;
    T nval = (T)0;
    if (mult0) {
        T cur = rrow[0];
#pragma unroll
        for (int x = 1; x < NSYM; x++) cur = (x == esel) ? rrow[x] : cur;
        for (int q = 0; q < mult0; q++) {
            const double old = (double)cur;
            const double nw = old - ratio * old;
            cur = (T)nw;
            if (own) removed += old - nw;
        }
        if (own) bandT[(size_t)p * pos_elems + roff + b0] = cur;
#pragma unroll
        for (int x = 0; x < NSYM; x++) rrow[x] = (x == esel) ? cur : rrow[x];
        if (d0 == 1) { na = a; nb = b0; nval = cur; }
    }
    na = __shfl(na, 0, 8); nb = __shfl(nb, 0, 8);
    nval = (T)__shfl((double)nval, 0, 8);
    __syncthreads();                                                  // s_logtab stands (staged at the top: nobody waits here for memory)
    // ---- marginals of position p (k_marg / k_rw: same order of operations) ----------------------------------------------
    unsigned flag_bits = 0;
    int hole_p = 0x7fffffff;
    double cs[NSYM];
    double tot = 0.0;
    int nv = 0;
    uint32_t cm = 0;
    T acc = (T)0;
#pragma unroll
    for (int x = 0; x < NSYM; x++) {
        T v = crow[x];
        if (s == na && x == nb) v = nval;
        acc = acc + v;
    }
    const double mine = (double)acc;
#pragma unroll
    for (int x = 0; x < NSYM; x++) {
        cs[x] = __shfl(mine, x, 8);
        if (cs[x] > 0) {
            tot += cs[x];
            if ((VALID_MASK >> x) & 1) { nv++; cm |= 1u << x; }
        }
    }
    const uint32_t cand = Q.offer_zero ? VALID_MASK : cm;
    const uint32_t cmw = cm | (cand << 8);
    const uint32_t cm5 = cm5_of_cmask(sm, cand);
    if (own) {
        if (s < NSYM) {
            const double m = (cs[s] > 0 && tot != 0.0) ? cs[s] / tot : 0.0;
            Q.cnt[(size_t)p * 8 + s] = cs[s];
            Q.marg[(size_t)p * 8 + s] = m;
            if ((VALID_MASK >> s) & 1) {
                const int b5 = a6_of_sym(sm, s);
                const double lm = gh_log10_tab(m, s_logtab, GH_LOG_BOTH);
                Q.minfo[(size_t)p * MINFO + b5] = lm;
                Q.minfo[(size_t)p * MINFO + 5 + b5] = m;
                const int r = __popc(cm5 & ((1u << b5) - 1u));
                if (Q.rinfo && ((cand >> s) & 1u) && r < 4) {
                    Q.rinfo[(size_t)p * RINFO + r] = lm;
                    Q.rinfo[(size_t)p * RINFO + 4 + r] = m;
                }
            }
        } else {
            Q.cnt[(size_t)p * 8 + 7] = tot;
            Q.marg[(size_t)p * 8 + 7] = 0.0;
            Q.nvalid[p] = nv;
            Q.cmask[p] = cmw;
            Q.minfo[(size_t)p * MINFO + 10] = __longlong_as_double((long long)cm5);
            for (int r = __popc(cm5); Q.rinfo && r < 4; r++) {
                Q.rinfo[(size_t)p * RINFO + r] = 0.0;
                Q.rinfo[(size_t)p * RINFO + 4 + r] = INFINITY;
            }
            if (cm_old != cmw) flag_bits |= 1u;
            if (p >= 1 && (cm5 & (1u << 4))) flag_bits |= 2u;      // the LAST symbol of the candidate order ('-' by default) is offered here
            if (p >= 1 && __popc(cm5) > 4) flag_bits |= 4u;
            if (p >= 1 && cand == 0) hole_p = p;
        }
    }
    RWS_STAMP(3);
    if (act && s == 0) patch->cmall[slot] = (uint8_t)cm5;           // (the mixed-radix extension ranks its digits through these)
    if (halo && act && P.mt) {
        // the marginal term of the halo positions, as the owner writes it to rinfo / minfo in this launch
        if (s < NSYM) {
            if ((VALID_MASK >> s) & 1) {
                const double m = (cs[s] > 0 && tot != 0.0) ? cs[s] / tot : 0.0;
                const int b5 = a6_of_sym(sm, s);
                const double lm = gh_log10_tab(m, s_logtab, GH_LOG_BOTH);
                if (b5 < 5) patch->lm5[slot][b5] = lm;
                const int r = __popc(cm5 & ((1u << b5) - 1u));
                if (((cand >> s) & 1u) && r < 4) patch->lm4[slot][r] = lm;
            }
        } else {
            for (int r = __popc(cm5); r < 4; r++) patch->lm4[slot][r] = 0.0;
        }
    }
    // ---- the table row of lag d0 (k_rw's table_row): to G for an own position, into the patch for a halo position ---------
    const double ca = __shfl(mine, a, 8);                  // c_a(p) (every lane of the group takes part: lane a may own no lag row)
    if (COL) {
        // the COLUMN of lag d0 that the cell (p, p + d0) feeds (k_rw's table_col): one entry per from-row that exists
        if (lag_row) {
            const bool ranked = c.ranked != 0;
            T cacc = (T)0;
#pragma unroll
            for (int x = 0; x < NSYM; x++) cacc = cacc + rrow[x];
            const double den = (Q.cond_mode == GH_COND_C ? (double)nv : (double)nv_t) + (double)cacc;
            const int b5c = a6_of_sym(sm, b0);
            const uint32_t cj5 = cm5_of_cmask(sm, cm_t);
            if (b5c < 5 && ((cj5 >> b5c) & 1u)) {
                const int col = ranked ? __popc(cj5 & ((1u << b5c) - 1u)) : b5c;
                constexpr int FS[6] = {0, 1, 2, 3, 5, 6};
                double xq[6], v[6];
                bool odd = false;
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    xq[q] = (1.0 + (double)rrow[FS[q]]) / den;
                    odd |= !gh_log10_is_normal(xq[q]);
                }
#pragma unroll
                for (int q = 0; q < 6; q++) v[q] = gh_log10_normal_tab(xq[q], 0, s_logtab, GH_LOG_BOTH);
                if (odd) {
#pragma unroll
                    for (int q = 0; q < 6; q++) v[q] = gh_log10_tab(xq[q], s_logtab, GH_LOG_BOTH);
                }
                unsigned rmask = 0;
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const int x6 = a6_of_sym(sm, FS[q]);
                    int row6 = x6;
                    if (x6 == 5) { if (p != 0) continue; }
                    else if (ranked) {
                        if (!((cm5 >> x6) & 1u)) continue;
                        row6 = __popc(cm5 & ((1u << x6) - 1u));
                        if (row6 > 3) continue;
                    }
                    if (own) Q.G[(((size_t)p * 6 + row6) * L + (d0 - 1)) * LT_ROW + col] = v[q];
                    else if (row6 < LT_ROW) { patch->row[slot][d0 - 1][row6] = v[q]; rmask |= 1u << row6; }
                }
                if (!own) { patch->col[slot][d0 - 1] = col; patch->rmask[slot][d0 - 1] = rmask; }
            }
        }
    } else if (lag_row && a != 4) {
        const bool ranked = c.ranked != 0;
        const int a6 = a6_of_sym(sm, a);
        const double nv_i = (double)nv;
        int row6 = a6;
        if (ranked && a6 < 5) row6 = ((cm5 >> a6) & 1u) ? __popc(cm5 & ((1u << a6) - 1u)) : -1;
        if (row6 >= 0) {
            double out[LT_ROW];
            if (!(a6 < 5 || p == 0)) {
#pragma unroll
                for (int q = 0; q < LT_ROW; q++) out[q] = 0.0;
            } else {
                double rowv[NSYM];
                T racc = (T)0;
#pragma unroll
                for (int x = 0; x < NSYM; x++) { rowv[x] = (double)rrow[x]; racc = racc + rrow[x]; }
                const double sum = (double)racc;
                const double den = (Q.cond_mode == GH_COND_A) ? (double)nv_t + sum : (Q.cond_mode == GH_COND_D ? nv_i + sum : nv_i + ca);
                constexpr int VS[LT_ROW] = {0, 1, 2, 3, 5};
                double xq[LT_ROW], v[LT_ROW];
                bool odd = false;
#pragma unroll
                for (int q = 0; q < LT_ROW; q++) xq[q] = (1.0 + rowv[VS[q]]) / den;
#pragma unroll
                for (int q = 0; q < LT_ROW; q++) odd |= !gh_log10_is_normal(xq[q]);
#pragma unroll
                for (int q = 0; q < LT_ROW; q++) v[q] = gh_log10_normal_tab(xq[q], 0, s_logtab, GH_LOG_BOTH);
                if (odd) {
#pragma unroll
                    for (int q = 0; q < LT_ROW; q++) v[q] = gh_log10_tab(xq[q], s_logtab, GH_LOG_BOTH);
                }
#pragma unroll
                for (int q = 0; q < LT_ROW; q++) out[q] = -INFINITY;
                if (!ranked) {
#pragma unroll
                    for (int q = 0; q < LT_ROW; q++) {
                        const int b5 = a6_of_sym(sm, VS[q]);
                        const double val = ((cm_t >> VS[q]) & 1) ? v[q] : -INFINITY;
#pragma unroll
                        for (int w = 0; w < LT_ROW; w++) out[w] = (b5 == w) ? val : out[w];
                    }
                } else {
                    const uint32_t cj5 = cm5_of_cmask(sm, cm_t);
#pragma unroll
                    for (int q = 0; q < LT_ROW; q++) {
                        const int b5 = a6_of_sym(sm, VS[q]);
                        const int rb = ((cj5 >> b5) & 1u) ? __popc(cj5 & ((1u << b5) - 1u)) : -1;
#pragma unroll
                        for (int w = 0; w < LT_ROW; w++) out[w] = (rb == w) ? v[q] : out[w];
                    }
                }
            }
            if (own) {
                double *dst = Q.G + (((size_t)p * 6 + row6) * L + (d0 - 1)) * LT_ROW;
#pragma unroll
                for (int q = 0; q < LT_ROW; q++) dst[q] = out[q];
            } else {
#pragma unroll
                for (int q = 0; q < LT_ROW; q++) patch->row[slot][d0 - 1][q] = out[q];
                if (s == 0) patch->row6[slot] = row6;
            }
        }
    }
    RWS_STAMP(4);
    // ---- flags, removed mass ---------------------------------------------------------------------------------------
    __shared__ unsigned s_flags;
    __shared__ int s_hole;
    __shared__ double s_part[32];
    if (tid == 0) { s_flags = 0; s_hole = 0x7fffffff; }
    sred[tid] = removed;
    __syncthreads();
    if (flag_bits) atomicOr(&s_flags, flag_bits);
    if (hole_p != 0x7fffffff) atomicMin(&s_hole, hole_p);
    // two levels in a fixed order instead of a tree of ten barriers: 32 threads sum every 32nd value of the lanes that hold
    // positions, one thread sums the 32
    if (tid < 32) {
        const int nact = 8 * (nh + nown);
        double a2 = 0.0;
        for (int q = tid; q < nact; q += 32) a2 += sred[q];
        s_part[tid] = a2;
    }
    __syncthreads();
    if (tid == 0) {
        double tot_removed = 0.0;
#pragma unroll
        for (int q = 0; q < 32; q++) tot_removed += s_part[q];
        Q.partial[sblk] = tot_removed;
        // for k_scan the control words (atomics, rare: something moved), for k_emit_small -- where the launch that reads them is
        // the launch that re-arms them -- the workgroup's flag word (rws_take_flags)
        const unsigned f = s_flags;
        if (f & 1u) atomicAnd(&st->cm_same, 0);
        if (f & 2u) atomicAnd(&st->nodel, 0);
        if (f & 4u) atomicAnd(&st->narrow, 0);
        if (s_hole != 0x7fffffff) atomicMin(&st->first_hole, s_hole);
        P.rwflags[sblk] = make_int2((int)f, s_hole);
    }
    // what this workgroup stored (G rows, marginals) is read back by its own k_seg part
    __threadfence_block();
    __syncthreads();
    RWS_STAMP(5);
    return true;
}

template <typename T, int LC, bool COL>
__global__ void __launch_bounds__(SEG_THREADS) k_rwseg(seg_params P, rws_params Q)
{
    extern __shared__ __align__(16) unsigned char seg_smem[];
    dev_state *st = P.st;
    const dev_ctl c = load_ctl(st);
    if (c.stop || c.lt_stale || c.cw_unres) return;
    if (c.cur_hole <= P.N) {                                // the path before ended in a hole: nothing to reweight, nothing more to walk
        if (blockIdx.x == 0 && threadIdx.x == 0) seg_finish(st, Q.rec, P.N, 0.0, Q.min_remove);
        return;
    }
    seg_patch *patch = reinterpret_cast<seg_patch *>(seg_smem + P.patch_off);
    if (!rwseg_reweight<T, LC, COL>(P, Q, c, seg_smem, patch)) return;
    const int cls = __builtin_amdgcn_readfirstlane(seg_class(c, LC));
    if (cls == 4) seg_body<4, LC, false>(P, seg_smem, patch);
    else if constexpr (seg_radix_ok(5, LC)) {
        if constexpr (LC == SEGM_L) { if (cls == SEG_CLS_MIXED) { seg_body_mixed<LC, true>(P, seg_smem, patch); return; } }
        if (P.nanp) seg_body<5, LC, false, true>(P, seg_smem, patch);
        else seg_body<5, LC, false, false>(P, seg_smem, patch);
    }
}

// lone gh_generate_path: no k_marg<T,true> follows, so the record is closed here
__global__ void __launch_bounds__(256) k_seg_fin(seg_params P, gh_path_rec *rec, double min_remove, int nseg_arg)
{
    __shared__ double s_red[256];
    dev_state *st = P.st;
    if (st->stop || st->lt_stale || st->cw_unres) return;
    const int nseg = nseg_arg > 0 ? nseg_arg : seg_count(st, P.N, P.L);      // (behind the candidate pools: their segment count)
    double mine = INFINITY;
    for (int q = threadIdx.x; q < nseg; q += 256) {
        const double v = P.segmin[q];
        if (v < mine) mine = v;
    }
    s_red[threadIdx.x] = mine;
    __syncthreads();
    for (int q = 128; q > 0; q >>= 1) {
        if ((int)threadIdx.x < q && s_red[threadIdx.x + q] < s_red[threadIdx.x]) s_red[threadIdx.x] = s_red[threadIdx.x + q];
        __syncthreads();
    }
    if (threadIdx.x == 0) seg_finish(st, rec, P.N, s_red[0], min_remove);
}

// -------------------------------------------------------------------------------------------------------------
// k_hp: running_prob += log10(marginal) over the SNPs, left to right (gretel.py:185-186).  blockIdx.x = path,
// blockIdx.y = 0: current marginals (kept by k_emit at walk time), 1: original marginals (minfo[11..15], fixed since
// the snapshot, looked up through the path's symbols).  A strictly sequential binary64 sum, one wavefront per sum.
// The addends come in by vector loads, 512 at a time (the next 512 in flight under the additions), go through LDS,
// and every addition reads its addend back with a broadcast ds_read (same address in all lanes; LDS data returns in
// order, so the reads run ahead of the additions): ~2 instructions per addend where moving a vector register's lanes
// through scalar registers (two v_readlane per addend, each followed by the scalar-operand hazard) took 150 us per
// 10 000 SNPs.  An unused slot adds +0.0, which changes nothing.
// -------------------------------------------------------------------------------------------------------------
#define HP_CHUNK 512
template <int WHICH>
__device__ __forceinline__ double hp_sum(const double *lm, const uint8_t *path, const double *minfo, int N, double (*buf)[HP_CHUNK], symmap sm)
{
    const int lane = threadIdx.x;
    // (no branches around the loads: a slot beyond the window reads position N and is replaced by +0.0)
    auto value = [&](int t) -> double {
        const int tt = t <= N ? t : N;
        const double v = WHICH == 0 ? lm[tt] : minfo[(size_t)tt * MINFO + 11 + a6_of_sym(sm, path[tt])];
        return t <= N ? v : 0.0;
    };
    constexpr int PER = HP_CHUNK / 64;
    double r[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) r[k] = value(1 + lane + 64 * k);
    const int nchunks = (N + HP_CHUNK - 1) / HP_CHUNK;
    double acc = 0.0;
    for (int c = 0; c < nchunks; c++) {
        double *b = buf[c & 1];
#pragma unroll
        for (int k = 0; k < PER; k++) b[lane + 64 * k] = r[k];
        __syncthreads();
        if (c + 1 < nchunks) {
#pragma unroll
            for (int k = 0; k < PER; k++) r[k] = value(1 + (c + 1) * HP_CHUNK + lane + 64 * k);
        }
#pragma unroll 32
        for (int j = 0; j < HP_CHUNK; j++) acc += b[j];
    }
    return acc;
}

__global__ void __launch_bounds__(64)
k_hp(const double *lmsel, size_t lmsel_stride, const uint8_t *paths, size_t path_stride, const double *minfo, int N,
     const dev_state *st, gh_path_rec *recs, symmap sm)
{
    __shared__ double buf[2][HP_CHUNK];
    const int s = blockIdx.x, which = blockIdx.y;
    if (s >= st->n_done) return;
    const double *lm = lmsel + (size_t)s * lmsel_stride;
    if (lm[0] != 1.0) return;                                      // walked by a serial walker, which summed for itself
    const uint8_t *path = paths + (size_t)s * path_stride;
    const double acc = which == 0 ? hp_sum<0>(lm, path, minfo, N, buf, sm) : hp_sum<1>(lm, path, minfo, N, buf, sm);
    if (threadIdx.x == 0) {
        if (which == 0) recs[s].hp_current = acc;
        else recs[s].hp_original = acc;
    }
}

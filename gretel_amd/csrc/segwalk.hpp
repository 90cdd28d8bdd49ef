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
//   k_hp     the two log-likelihood sums of gretel.py:185-186, strictly left to right in binary64 (one wavefront per
//            sum); they feed nothing on the device, so gh_spin runs them for all paths at once behind the loop.
//
// Nothing is speculated: every entry state is enumerated, the composition is exact, the result is bit-identical to
// the serial walkers (which stay: L > 5, batched launches, GH_WALK=spec).  Work per path: N * R^L sums instead of N.
#pragma once

#define SEG_THREADS 1024
// diagnostic builds only (-DSEG_STAMPS): s_memtime at the phase boundaries of k_seg's workgroup 0 into st->dbg8
#ifdef SEG_STAMPS
#define SEG_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) P.st->dbg8[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SEG_STAMP(i)
#endif
#define SEG_MIN_LEN 8          /* shortest segment (positions) */
#define SEG_MAX_L 5            /* 5^5 = 3125 states still fit; beyond that the serial walkers run */

template <int R> struct seg_radix;
template <> struct seg_radix<4> { typedef uint8_t next_t;  static constexpr int BITS = 2, DPW = 16; };   // 4 picks of 2 bits per entry
template <> struct seg_radix<5> { typedef uint16_t next_t; static constexpr int BITS = 3, DPW = 10; };   // 5 picks of 3 bits
// (DPW: picks per 32-bit word of hist)
__host__ __device__ constexpr int seg_dpw(int R) { return R == 4 ? 16 : 10; }

__host__ __device__ constexpr int seg_ipow(int b, int e) { int r = 1; for (int i = 0; i < e; i++) r *= b; return r; }

// positions per LDS chunk of k_seg: the slice of G ((c + L - 1) sources x L lags x R x R doubles) and the chunk's
// Next tables (c x R^(L-1) entries) within 96 KB, at most 64
__host__ __device__ constexpr int seg_chunk(int R, int L)
{
    const int NI = seg_ipow(R, L - 1), sz = R == 4 ? 1 : 2;
    int c = 64;
    while (c > 8 && ((c + L - 1) * L * R * R * 8 + c * NI * sz) > 96 * 1024) c -= 8;
    return c / seg_dpw(R) * seg_dpw(R);        // whole words of hist per chunk
}

struct seg_geom {
    int NS, NI;         // states, entries per position (= NS / R: one entry holds the picks of all R oldest digits)
    int seglen, S;      // positions per segment, segments
    int G1, G2;         // groups, segments per group
    int NW;             // 32-bit words of hist per (segment, entry state)
};

// the same on host and device; R is only known on the device (st->ranked), the host sizes for both
__host__ __device__ inline seg_geom seg_geometry(int N, int L, int R)
{
    seg_geom g;
    g.NS = seg_ipow(R, L);
    g.NI = g.NS / R;
    int g2 = 32768 / g.NS;                      // one group's maps (G2 x NS x 2 bytes) within 64 KB of LDS
    if (g2 > 16) g2 = 16;
    if (g2 < 1) g2 = 1;
    const int g1max = g.NS > 2048 ? 12 : 16;
    const int smax = g1max * g2;
    int len = (N + smax - 1) / smax;
    if (len < SEG_MIN_LEN) len = SEG_MIN_LEN;
    g.seglen = len;
    g.S = (N + len - 1) / len;
    g.G2 = g2;
    g.G1 = (g.S + g2 - 1) / g2;
    // whole chunks except the last, each a whole number of words
    g.NW = (len / seg_chunk(R, L)) * (seg_chunk(R, L) / seg_dpw(R)) + (len % seg_chunk(R, L) + seg_dpw(R) - 1) / seg_dpw(R);
    return g;
}

__host__ __device__ constexpr size_t seg_lds_bytes(int R, int L)
{
    return (size_t)(seg_chunk(R, L) + L - 1) * L * R * R * 8 + (size_t)seg_chunk(R, L) * seg_ipow(R, L - 1) * (R == 4 ? 1 : 2);
}
__host__ __device__ inline size_t scan_lds_bytes(int N, int L, int R)
{
    const seg_geom g = seg_geometry(N, L, R);
    return (size_t)g.G2 * g.NS * 2;
}
__host__ __device__ inline size_t emit_lds_bytes(int N, int L, int R)
{
    const seg_geom g = seg_geometry(N, L, R);
    return (size_t)g.G1 * g.NS * 2;             // the group maps in front (<= G1 - 1) and the segment's prefix map
}

struct seg_params {
    int N, L;
    int rearm;                // spin loops: k_scan re-arms first_hole/nodel/cm_same/narrow for the k_marg<T,true> that follows
    int check_masks;          // spins without a k_lt between paths: a candidate mask that moved under the last reweight makes the table stale
    const double *G;          // [(N+LT_PAD)][6][L][5], ranked or not (st->ranked)
    const double *minfo;      // [N+2][16]
    dev_state *st;
    uint32_t *hist;           // [S][NW][NS] picks of every entry state, DPW per word, the first lowest
    uint16_t *maps;           // [S][NS] segment maps
    uint16_t *pmaps;          // [S][NS] prefix maps: entry state of the group -> entry state of the segment
    uint16_t *gmaps;          // [G1][NS] group maps
    double *segmin;           // [S] minimum marginal of the symbols selected in each segment
    uint8_t *path_out;        // [N+1]
    double *lmsel;            // [N+1] log10 marginal of the selected symbol per position (for k_hp)
};

// -------------------------------------------------------------------------------------------------------------
// k_seg
// -------------------------------------------------------------------------------------------------------------
template <int R, int LC>
__device__ __forceinline__ void seg_body(const seg_params &P, unsigned char *smem)
{
    typedef typename seg_radix<R>::next_t next_t;
    constexpr int BITS = seg_radix<R>::BITS;
    constexpr unsigned MASK = (1u << BITS) - 1u;
    constexpr int NS = seg_ipow(R, LC), NI = NS / R, RR = R * R;
    constexpr int CH = seg_chunk(R, LC);
    constexpr int SPT = (NS + SEG_THREADS - 1) / SEG_THREADS;      // states per thread
    const seg_geom g = seg_geometry(P.N, LC, R);
    const int s = blockIdx.x, tid = threadIdx.x;
    if (s >= g.S) return;
    const int t0 = s * g.seglen;                                   // targets t0+1 .. t1
    const int t1 = t0 + g.seglen < P.N ? t0 + g.seglen : P.N;
    double *Gs = reinterpret_cast<double *>(smem);                 // [(CH + LC - 1)][LC][R][R]
    next_t *Nx = reinterpret_cast<next_t *>(Gs + (size_t)(CH + LC - 1) * LC * RR);   // [CH][NI]

    int sigma[SPT];
#pragma unroll
    for (int q = 0; q < SPT; q++) sigma[q] = tid + q * SEG_THREADS;

    SEG_STAMP(0);
    for (int c0 = t0; c0 < t1; c0 += CH) {
        const int nc = t1 - c0 < CH ? t1 - c0 : CH;                // targets c0+1 .. c0+nc
        // (1a) the slice of G this chunk needs: sources c0+1-LC .. c0+nc-1 (slot ii = i - (c0+1-LC)), every lag, the
        // rows of the R digits and the R candidate columns.  Position 0 carries '_' whatever the digit says (row 5);
        // positions < 0 do not exist: their terms are +0.0, which leaves every partial sum as it is.
        const int nsrc = nc + LC - 1;
        for (int e = tid; e < nsrc * LC * RR; e += SEG_THREADS) {
            const int b = e % R, d = (e / R) % R, l = (e / RR) % LC, ii = e / (RR * LC);
            const int i = c0 + 1 - LC + ii;
            double v = 0.0;
            if (i >= 0) v = P.G[(((size_t)i * 6 + (i == 0 ? 5 : d)) * LC + l) * LT_ROW + b];
            Gs[e] = v;
        }
        __syncthreads();
        SEG_STAMP(1);
        // (1b) Next for every (target, state): entry idx holds the digits d_1 .. d_{L-1} (d_1 lowest); the R picks for the
        // R values of the oldest digit d_L share the partial sum over lags 1 .. L-1.  Lag l of chunk-local target tl
        // comes from slot tl + LC - l.
        for (int task = tid; task < nc * NI; task += SEG_THREADS) {
            const int tl = task / NI, idx = task - tl * NI;
            double acc[R];
            if constexpr (LC >= 2) {
                int rem = idx;
                {
                    const int d = rem % R;
                    rem /= R;
                    const double *row = Gs + ((size_t)((tl + LC - 1) * LC + 0) * R + d) * R;
#pragma unroll
                    for (int b = 0; b < R; b++) acc[b] = row[b];
                }
#pragma unroll
                for (int l = 2; l < LC; l++) {
                    const int d = rem % R;
                    rem /= R;
                    const double *row = Gs + ((size_t)((tl + LC - l) * LC + (l - 1)) * R + d) * R;
#pragma unroll
                    for (int b = 0; b < R; b++) acc[b] = acc[b] + row[b];
                }
            }
            unsigned packed = 0;
#pragma unroll
            for (int dL = 0; dL < R; dL++) {
                const double *row = Gs + ((size_t)(tl * LC + (LC - 1)) * R + dL) * R;
                double best;
                unsigned bi = 0;
#pragma unroll
                for (int b = 0; b < R; b++) {
                    double v;
                    if constexpr (LC >= 2) v = acc[b] + row[b];
                    else v = row[b];
                    if (b == 0) best = v;
                    else if (v > best) { best = v; bi = b; }          // first wins, later only on strict > (gretel.py:166-174)
                }
                packed |= bi << (BITS * dL);
            }
            Nx[task] = (next_t)packed;
        }
        __syncthreads();
        SEG_STAMP(2);
        // (2) every entry state through the chunk; its picks go to hist one word (DPW picks) at a time
        {
            constexpr int DPW = seg_radix<R>::DPW;
            const int w0 = (c0 - t0) / DPW;                        // chunks are whole words
            for (int tw = 0; tw < nc; tw += DPW) {
                unsigned word[SPT];
#pragma unroll
                for (int q = 0; q < SPT; q++) word[q] = 0;
#pragma unroll
                for (int u = 0; u < DPW; u++) {
                    const int tl = tw + u;
                    if (tl < nc) {
                        const next_t *row = Nx + (size_t)tl * NI;
#pragma unroll
                        for (int q = 0; q < SPT; q++) {
                            const int sg = sigma[q] < NS ? sigma[q] : 0;
                            const int hi = sg / NI, idx = sg - hi * NI;
                            const unsigned d = ((unsigned)row[idx] >> (BITS * hi)) & MASK;
                            word[q] |= d << (BITS * u);
                            sigma[q] = sigma[q] < NS ? idx * R + (int)d : sigma[q];
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < SPT; q++) {
                    const int s0 = tid + q * SEG_THREADS;
                    if (s0 < NS) P.hist[((size_t)s * g.NW + w0 + tw / DPW) * NS + s0] = word[q];
                }
            }
        }
        SEG_STAMP(3);
        __syncthreads();                                           // Gs / Nx are overwritten by the next chunk
    }
#pragma unroll
    for (int q = 0; q < SPT; q++) {
        const int s0 = tid + q * SEG_THREADS;
        if (s0 < NS) P.maps[(size_t)s * NS + s0] = (uint16_t)sigma[q];
    }
    SEG_STAMP(4);
}

template <int LC>
__global__ void __launch_bounds__(SEG_THREADS) k_seg(seg_params P)
{
    extern __shared__ __align__(16) unsigned char seg_smem[];
    dev_state *st = P.st;
    if (st->stop || st->lt_stale) return;
    if (P.check_masks == 2 || (P.check_masks && __builtin_amdgcn_readfirstlane(st->cm_same) == 0)) {      // (2: forced, tests)
        // k_marg<T,true> saw a candidate mask change: V(p) and the -inf masks in G moved, the rows it rewrote are not
        // enough.  Every kernel queued behind this one returns at once; the host rebuilds G and queues the paths again.
        if (blockIdx.x == 0 && threadIdx.x == 0) st->lt_stale = 1;
        return;
    }
    // the flags k_marg left for this path; k_scan re-arms them for the next k_marg, k_emit reads the copy
    if (blockIdx.x == 0 && threadIdx.x == 0) st->cur_hole = st->first_hole;
    if (__builtin_amdgcn_readfirstlane(st->ranked) != 0) seg_body<4, LC>(P, seg_smem);
    else seg_body<5, LC>(P, seg_smem);
}

// -------------------------------------------------------------------------------------------------------------
// k_scan: group map = composition of the group's segment maps
// -------------------------------------------------------------------------------------------------------------
template <int R, int LC>
__device__ __forceinline__ void scan_body(const seg_params &P, unsigned char *smem)
{
    constexpr int NS = seg_ipow(R, LC);
    const seg_geom g = seg_geometry(P.N, LC, R);
    const int grp = blockIdx.x, tid = threadIdx.x;
    if (grp >= g.G1) return;
    const int s_lo = grp * g.G2;
    const int n = (s_lo + g.G2 <= g.S ? g.G2 : g.S - s_lo);
    uint16_t *M = reinterpret_cast<uint16_t *>(smem);              // [n][NS]
    const uint16_t *src = P.maps + (size_t)s_lo * NS;
    for (int e = tid; e < n * NS; e += SEG_THREADS) M[e] = src[e];
    __syncthreads();
    for (int s0 = tid; s0 < NS; s0 += SEG_THREADS) {
        int x = s0;
        for (int j = 0; j < n; j++) {
            if (j > 0) P.pmaps[(size_t)(s_lo + j) * NS + s0] = (uint16_t)x;      // what enters segment s_lo + j
            x = M[(size_t)j * NS + x];
        }
        P.gmaps[(size_t)grp * NS + s0] = (uint16_t)x;
    }
}

template <int LC>
__global__ void __launch_bounds__(SEG_THREADS) k_scan(seg_params P)
{
    extern __shared__ __align__(16) unsigned char seg_smem[];
    dev_state *st = P.st;
    if (st->stop || st->lt_stale) return;
    // (a path that ends in a hole is followed by no k_marg: the flags must stand, as after the serial walkers)
    if (P.rearm && st->cur_hole > P.N && blockIdx.x == 0 && threadIdx.x == 0) {
        st->first_hole = 0x7f7f7f7f; st->nodel = 0x7f7f7f7f; st->cm_same = 0x7f7f7f7f; st->narrow = 0x7f7f7f7f;
    }
    if (__builtin_amdgcn_readfirstlane(st->ranked) != 0) scan_body<4, LC>(P, seg_smem);
    else scan_body<5, LC>(P, seg_smem);
}

// -------------------------------------------------------------------------------------------------------------
// k_emit
// -------------------------------------------------------------------------------------------------------------
template <int R, int LC>
__device__ __forceinline__ void emit_body(const seg_params &P, unsigned char *smem)
{
    constexpr int BITS = seg_radix<R>::BITS, DPW = seg_radix<R>::DPW;
    constexpr unsigned MASK = (1u << BITS) - 1u;
    constexpr int NS = seg_ipow(R, LC);
    __shared__ int s_sigma;
    __shared__ double s_min[SEG_THREADS / 64];
    const seg_geom g = seg_geometry(P.N, LC, R);
    const int s = blockIdx.x, tid = threadIdx.x;
    if (s >= g.S) return;
    const int first_hole = P.st->cur_hole;
    const int Nw = first_hole <= P.N ? first_hole - 1 : P.N;       // positions that can be decided (gretel.py:176-180)
    const int t0 = s * g.seglen;
    int t1 = t0 + g.seglen < P.N ? t0 + g.seglen : P.N;
    if (t1 > Nw) t1 = Nw;
    if (s == 0 && tid == 0) P.path_out[0] = SYM_US;                // gretel.py:138
    if (t0 >= t1) {
        if (tid == 0) P.segmin[s] = INFINITY;
        return;
    }
    const int grp = s / g.G2, q = s - grp * g.G2;
    uint16_t *GM = reinterpret_cast<uint16_t *>(smem);             // [grp][NS] group maps in front of this group
    uint16_t *PM = GM + (size_t)grp * NS;                          // [NS] prefix map of this segment (q >= 1)
    for (int e = tid; e < grp * NS; e += SEG_THREADS) GM[e] = P.gmaps[e];
    if (q > 0) {
        const uint16_t *src = P.pmaps + (size_t)s * NS;
        for (int e = tid; e < NS; e += SEG_THREADS) PM[e] = src[e];
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
    // the true entry state: the start state 0 through the maps of the groups in front, then the prefix map
    if (tid == 0) {
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
        if (R == 4) {                                               // rank -> symbol through the candidate bits
            b5 = nth_set5((uint32_t)__double_as_longlong(r16[10]), b5);
            if (b5 < 0) b5 = 0;     // cannot happen for a decided position
        }
        double lm = r16[0], m = r16[5];
#pragma unroll
        for (int k = 1; k < 5; k++) {
            lm = (b5 == k) ? r16[k] : lm;
            m = (b5 == k) ? r16[5 + k] : m;
        }
        P.path_out[t] = (uint8_t)vsym(b5);
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
    if (st->stop || st->lt_stale) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) st->dbg[3] = 3;      // gh_debug_walk_clock: variant 3 = segment-parallel
    if (__builtin_amdgcn_readfirstlane(st->ranked) != 0) emit_body<4, LC>(P, seg_smem);
    else emit_body<5, LC>(P, seg_smem);
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
    return seg_geometry(N, L, st->ranked != 0 ? 4 : 5).S;
}

// lone gh_generate_path: no k_marg<T,true> follows, so the record is closed here
__global__ void __launch_bounds__(256) k_seg_fin(seg_params P, gh_path_rec *rec, double min_remove)
{
    __shared__ double s_red[256];
    dev_state *st = P.st;
    if (st->stop || st->lt_stale) return;
    const int nseg = seg_count(st, P.N, P.L);
    s_red[threadIdx.x] = (int)threadIdx.x < nseg ? P.segmin[threadIdx.x] : INFINITY;
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
// the snapshot, looked up through the path's symbols).  One wavefront each: 64 values per load, then 64 additions in
// lane order; an unused lane adds +0.0, which changes nothing.
// -------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_hp(const double *lmsel, size_t lmsel_stride, const uint8_t *paths, size_t path_stride, const double *minfo, int N,
     const dev_state *st, gh_path_rec *recs)
{
    const int s = blockIdx.x, which = blockIdx.y, lane = threadIdx.x;
    if (s >= st->n_done) return;
    const double *lm = lmsel + (size_t)s * lmsel_stride;
    const uint8_t *path = paths + (size_t)s * path_stride;
    auto value = [&](int t) -> double {
        if (t > N) return 0.0;
        if (which == 0) return lm[t];
        return minfo[(size_t)t * MINFO + 11 + a6_of_sym(path[t])];
    };
    double acc = 0.0;
    double v = value(1 + lane);
    for (int base = 1; base <= N; base += 64) {
        const double cur = v;
        v = value(base + 64 + lane);                                // next chunk's loads in flight under the additions
#pragma unroll
        for (int j = 0; j < 64; j++) acc += readlane_f64(cur, j);
    }
    if (lane == 0) {
        if (which == 0) recs[s].hp_current = acc;
        else recs[s].hp_original = acc;
    }
}

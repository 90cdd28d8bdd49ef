// wpipe.hpp -- batched recovery as ONE persistent kernel: the spin loop of gretel/cmd.py:148-179 for a whole window inside one
// workgroup, the reweight of path s-1 (gretel/gretel.py:79-98) streaming through the tensor just ahead of the walk of path s
// (gretel/gretel.py:143-189).
//
// Why.  The batched flow of rounds 1-4 ran a path as two launches over all windows of a group: the serial walker (one wavefront
// per window, bound by its own dependent-issue latency, 0.98 ms per 10k-SNP path with every other SIMD slot of the chip idle)
// and then the fused reweight (HBM-bound, 0.65 ms per 85 windows); three stream groups overlapped the two phases only in part
// and the figure sat at 130k haplotypes/s for three rounds.  But the walk of path s at position t needs nothing of the reweight
// of path s-1 beyond position t + 4 (+ the L positions the table rows reach back from there): the sweep that reweights path
// s-1 left to right can run a few chunks AHEAD of the walker of path s, in the same workgroup, and the walker never stops.  The
// window's time per path is then the walk alone and the chip's throughput is bound by the bytes the sweep and the loaders
// move -- the HBM roofline this path belongs under.
//
// One workgroup per window, NT threads in four roles (wave -> role tables below: the walker's SIMD carries no reweight wave):
//   walker      1 wave   spec2_walker<LC> of kernels.hpp, unchanged: depth-2 speculation over the ranked table, one LDS row per step
//   bookkeeper  1 wave   one chunk behind: ranks -> symbols, the path bytes (global and the LDS copy the sweep reads), the
//                        sequential log10-marginal sums and the minimum marginal (gretel.py:182-186); closes the path record
//   loaders     NL thr.  derive the walker's tables (H = x1 + x2 per hypothesis, Yr = lags 3..L) from G for chunk k+1 while the
//                        walker is in chunk k, the loads of chunk k+2 in flight across the barrier (as k_walk_spec's)
//   sweepers    NR thr.  8 lanes per position, NR/8 positions per pass: exactly k_marg<T,true>'s arithmetic for path s-1 --
//                        cells on the path (multiplicities of SURVEY section 8 a8), the marginals of the position, the table
//                        row the cells feed -- except that the row sums of the six untouched rows come from `cnt` (as in k_rw)
// Every wave executes the same barriers: E = nchunks + 3 epochs per path and one tail.  In epoch e
//   sweepers  reweight pass e (positions e*C+4 .. e*C+C+3; pass 0 also 0..3), stores drained before the barrier
//   loaders   e = 2: fetch 0, store 0, fetch 1;  e >= 3: store chunk e-2, fetch chunk e-1
//   walker    e >= 3: chunk e-3
//   bookkeeper e >= 3: prefetch the marginal rows of chunk e-3, consume the picks of chunk e-4
// so a table row is written by a sweeper at least one barrier before a loader asks for it (same CU, same L1: workgroup-scope
// visibility needs the store drain and the barrier, nothing else), and a LDS buffer is refilled one barrier after the walker
// left it.  Per path the walker idles for three epochs and the tail (~ 6 us of 500).
//
// The pipeline only carries the steady state: a ranked table (every position offers at most four candidates), no hole, and
// every candidate mask as it was when the table was built.  A sweeper that sees a mask move (a count reached zero: rare) raises
// `abort`; the sweep in flight is completed (the tensor then holds paths 0..s-1, exactly), the walk beside it is dropped, the
// kernel ends with pipe_status = PIPE_ABORTED and n_done = s, and the host hands the window's remaining paths to gh_spin, which
// rebuilds marginals and table from the tensor.  A window that is not eligible when the kernel starts is left untouched
// (PIPE_NOT_STARTED) for the batched launches of rounds 1-4.  Results are bit-identical either way (tests/test_gpu_batch.py,
// tests/test_gpu_pipe.py): same IEEE operations in the same order as k_marg<T,true> and the serial walker.
#pragma once

#ifndef PIPE_DEV_ROLES
#define PIPE_DEV_ROLES 15     /* diagnostic builds: compile only some of the roles (register accounting) */
#endif
#define PIPE_DONE 1
#define PIPE_ABORTED 2
#define PIPE_NOT_STARTED 3

struct pipe_params {
    int N, W, L;
    int C;                  // positions per chunk: multiple of L, <= NR/8 - 4
    int max_paths;
    int cond_mode;          // A, B or D (row conditionals; no marginal term)
    int offer_zero;
    int prof;               // 1: the bookkeeper leaves s_memrealtime stamps per path in st->dbg8
    double min_remove;
    symmap sm;
};

// wave -> role.  Waves go to the SIMDs round-robin (wave w on SIMD w & 3): SIMD 0 gets the walker, the bookkeeper and loader
// waves only, the sweepers (binary64 divisions and logarithms) share the other three.
enum { PR_WALK = 0, PR_BOOK = 1, PR_LOAD = 2, PR_SWEEP = 3 };
template <int NT> struct pipe_roles;
template <> struct pipe_roles<1024> {
    static constexpr int NLW = 6, NRW = 8;
    //                                   w: 0         1          2          3          4         5          6          7
    static constexpr unsigned char map[16] = {PR_WALK << 4, (PR_SWEEP << 4) | 0, (PR_SWEEP << 4) | 1, (PR_SWEEP << 4) | 2, PR_BOOK << 4, (PR_SWEEP << 4) | 3, (PR_SWEEP << 4) | 4, (PR_SWEEP << 4) | 5,
    //                                      8                 9                  10                 11                12                13                14                15
                                              (PR_LOAD << 4) | 0, (PR_SWEEP << 4) | 6, (PR_SWEEP << 4) | 7, (PR_LOAD << 4) | 5, (PR_LOAD << 4) | 1, (PR_LOAD << 4) | 2, (PR_LOAD << 4) | 3, (PR_LOAD << 4) | 4};
};
template <> struct pipe_roles<768> {
    static constexpr int NLW = 4, NRW = 6;
    static constexpr unsigned char map[16] = {PR_WALK << 4, (PR_SWEEP << 4) | 0, (PR_SWEEP << 4) | 1, (PR_SWEEP << 4) | 2, PR_BOOK << 4, (PR_SWEEP << 4) | 3, (PR_SWEEP << 4) | 4, (PR_SWEEP << 4) | 5,
                                              (PR_LOAD << 4) | 0, (PR_LOAD << 4) | 1, (PR_LOAD << 4) | 2, (PR_LOAD << 4) | 3, 0, 0, 0, 0};
};
template <> struct pipe_roles<512> {
    static constexpr int NLW = 3, NRW = 3;
    static constexpr unsigned char map[16] = {PR_WALK << 4, (PR_LOAD << 4) | 0, (PR_LOAD << 4) | 1, (PR_LOAD << 4) | 2, PR_BOOK << 4, (PR_SWEEP << 4) | 0, (PR_SWEEP << 4) | 1, (PR_SWEEP << 4) | 2,
                                              0, 0, 0, 0, 0, 0, 0, 0};
};

// LDS of one workgroup: two table buffers of C + WALK_OV positions, the walker's words, log10's table, the sweepers' partial
// sums, a line of control words, the path (N + 2 bytes)
// doubles per position of a table buffer: X1 = lag-1 terms [row][column] (16), X2 = lag-2 terms (16), Yr = lags 3..L [row][column][lag],
// rows padded to an even number of lags as in k_walk_spec's depth-2 layout
__host__ __device__ constexpr int pipe_pos_doubles(int L) { return 32 + 16 * deep_nyp(L); }
__host__ __device__ constexpr size_t pipe_fixed_bytes(int N, int nr_threads)
{
    return (size_t)2 * 64 * 8 + 256 * 8 + (size_t)nr_threads * 8 + 64 + (((size_t)N + 2 + 15) & ~(size_t)15);
}
__host__ __device__ constexpr int pipe_chunk(int N, int L, int nr_threads)
{
    const size_t fixed = pipe_fixed_bytes(N, nr_threads);
    if (L < 2 || fixed + 2 * (size_t)(L + WALK_OV) * pipe_pos_doubles(L) * 8 > WALK_LDS_MAX) return 0;
    long c = (long)((WALK_LDS_MAX - fixed) / (2 * (size_t)pipe_pos_doubles(L) * 8)) - WALK_OV;
    const long cap = nr_threads / 8 - WALK_OV;
    if (c > cap) c = cap;
    if (c > 60) c = 60;
    c = (c / L) * L;
    return c >= L ? (int)c : 0;
}
__host__ __device__ constexpr size_t pipe_lds_bytes(int N, int L, int C, int nr_threads)
{
    return 2 * (size_t)(C + WALK_OV) * pipe_pos_doubles(L) * 8 + pipe_fixed_bytes(N, nr_threads);
}

struct pipe_ctl {
    double ratio;           // clamped minimum marginal of the path just walked: what the sweep of the next epochs removes
    int abort;              // a sweeper saw a candidate mask move
    int _pad[13];
};
static_assert(sizeof(pipe_ctl) == 64, "one line");

// the bookkeeper's consume step (kernels.hpp: book_consume) with the symbol also stored into the LDS copy of the path
__device__ __forceinline__ void pipe_book_consume(const double *minfo_unused, uint8_t *path_out, uint8_t *s_path, const unsigned long long *words,
                                                  int LC, int j0, int ns, int Nw, int lane, const book_row &R, walk_totals &T,
                                                  double &lane_min, symmap sm)
{
    (void)minfo_unused;
    double lm = 0.0, lm0 = 0.0, mg = INFINITY;
    const int j = j0 + lane + 1;
    if (lane < ns && j <= Nw) {
        const unsigned long long word = words[lane / LC];
        int w = (int)((word >> (2 * (LC - 1 - lane % LC))) & 3ull);
        w = nth_set5((uint32_t)__double_as_longlong(R.v[5].x), w);      // minfo[10]: candidate bits; rank -> compact symbol index
        if (w < 0) w = 0;
        const double row[16] = {R.v[0].x, R.v[0].y, R.v[1].x, R.v[1].y, R.v[2].x, R.v[2].y, R.v[3].x, R.v[3].y,
                                R.v[4].x, R.v[4].y, R.v[5].x, R.v[5].y, R.v[6].x, R.v[6].y, R.v[7].x, R.v[7].y};
        lm = row[0]; mg = row[5]; lm0 = row[11];
#pragma unroll
        for (int q = 1; q < 5; q++) {
            lm = (w == q) ? row[q] : lm;
            mg = (w == q) ? row[5 + q] : mg;
            lm0 = (w == q) ? row[11 + q] : lm0;
        }
        const uint8_t sym = (uint8_t)vsym(sm, w);
        path_out[j] = sym;
        s_path[j] = sym;
    }
    if (mg < lane_min) lane_min = mg;                   // gretel.py:182
    int s = 0;
    for (; s + 4 <= ns; s += 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            T.hp_cur += readlane_f64(lm, s + q);        // gretel.py:185
            T.hp_orig += readlane_f64(lm0, s + q);      // gretel.py:186
        }
    }
    for (; s < ns; s++) {
        T.hp_cur += readlane_f64(lm, s);
        T.hp_orig += readlane_f64(lm0, s);
    }
}


// spec2_walker (kernels.hpp) over the RAW lag-1 / lag-2 terms: the batched loaders of rounds 1-4 expanded H = x1 + x2 for all 16
// hypotheses (64 doubles per target, every term fetched four times, 66 VGPRs of prefetch per loader lane); here a buffer holds
//   X[i][ 0..15] = x1 of SOURCE i: G[i][a1][lag 1][b]          X[i][16..31] = x2 of source i: G[i][a2][lag 2][b]
//   Yr[i][w][b][l - 3] = G[i][w][lag l][b], l = 3..L          (position 0: its '_' row in every row slot)
// and the walker takes H of target t as X1[t-1][a1][b] + X2[t-2][a2][b] itself: one more LDS read and one more addition per step,
// four bodies ahead of their use; the same IEEE addition the loaders did, so bit-identical.  Everything else is spec2_walker.
template <int LC>
__device__ __forceinline__ void spec2x_walker(double *g0, unsigned long long *words0, int C, int nchunks, int lane)
{
    static_assert(LC >= 2, "depth-2 speculation needs two lags");
    typedef deep_layout<LC> DL;
    constexpr int NY = DL::NY;
    constexpr unsigned XB = 32 * 8, YB = DL::YPOS * 8, YWB = 4 * DL::NYP * 8;
    constexpr int RS = pipe_pos_doubles(LC);
    const int b = lane & 3;
    double Y[LC][LC];
#pragma unroll
    for (int u = 0; u < LC; u++)
#pragma unroll
        for (int l = 0; l < LC; l++) Y[u][l] = 0.0;

    const int npos = C + WALK_OV;
    const unsigned bufB = (unsigned)npos * (unsigned)RS * 8u;
    // x1 of a source for this lane's (a1, b), x2 for its (a2, b)
    unsigned x10 = (unsigned)(uintptr_t)g0 + (unsigned)(lane & 15) * 8u;
    unsigned x20 = (unsigned)(uintptr_t)g0 + 128u + (unsigned)((lane >> 4) * 4 + b) * 8u;
    unsigned y0 = (unsigned)(uintptr_t)g0 + (unsigned)npos * XB + (unsigned)b * (unsigned)DL::NYP * 8u;
    asm("" : "+v"(x10), "+v"(x20), "+v"(y0));

    // state entering body 0: target 1 has the single term x1 of source 0 (not 0.0 + x1), targets 2 and 3 both terms
    unsigned long long B = group_argmax<true>(*(lds_cdouble *)(x10));
    double accP = *(lds_cdouble *)(x10 + XB) + *(lds_cdouble *)(x20);
    double H12 = *(lds_cdouble *)(x10 + 2 * XB) + *(lds_cdouble *)(x20 + XB);
#pragma unroll
    for (int l = 2; l < LC; l++) Y[0][l] = *(lds_cdouble *)(y0 + (unsigned)(l - 2) * 8u);
    unsigned hist = 0, sh = 0;
    unsigned yw_v;
    asm("v_mov_b32 %0, %1" : "=v"(yw_v) : "i"(YWB));

    for (int k = 0; k < nchunks; k++) {
        unsigned v1 = x10 + (unsigned)(k & 1) * bufB, v2 = x20 + (unsigned)(k & 1) * bufB, vy = y0 + (unsigned)(k & 1) * bufB;
        unsigned long long *wk = words0 + (k & 1) * 64;
        const int ngroups = C / LC;
        constexpr int UG = LC <= 8 ? 2 : 1;
        auto group = [&](int g, auto gg_) {
            constexpr int gg = decltype(gg_)::value;
#pragma unroll
            for (int u = 0; u < LC; u++) {
                // A: resolve w_{j+1}   (body j = k*C + g*LC + u)
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
                // R: row of source j+1 under its real symbol (lags 3..L); x1 + x2 of target j+4 (sources j+3 and j+2)
                if constexpr (NY > 0) {
                    unsigned vrow;
                    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(vrow) : "s"(w), "v"(yw_v), "v"(vy));
                    const unsigned rb = vrow + (unsigned)(gg * LC + u + 1) * YB;
#pragma unroll
                    for (int l = 2; l + 1 < LC; l += 2) {
                        const lds_v2d pr = *(const __attribute__((address_space(3))) lds_v2d *)(rb + (unsigned)(l - 2) * 8u);
                        Y[(u + 1) % LC][l] = pr.x;
                        Y[(u + 1) % LC][l + 1] = pr.y;
                    }
                    if constexpr (NY & 1) Y[(u + 1) % LC][LC - 1] = *(lds_cdouble *)(rb + (unsigned)(NY - 1) * 8u);
                }
                H12 = *(lds_cdouble *)(v1 + (unsigned)(gg * LC + u + 3) * XB) + *(lds_cdouble *)(v2 + (unsigned)(gg * LC + u + 2) * XB);
            }
            wk[g] = (unsigned long long)hist;
        };
        int g = 0;
        asm volatile(".p2align 6");
        for (; g + UG <= ngroups; g += UG) {
            group(g, std::integral_constant<int, 0>{});
            if constexpr (UG > 1) group(g + 1, std::integral_constant<int, 1>{});
            v1 += (unsigned)(UG * LC) * XB;
            v2 += (unsigned)(UG * LC) * XB;
            vy += (unsigned)(UG * LC) * YB;
        }
        for (; g < ngroups; g++) {
            group(g, std::integral_constant<int, 0>{});
            v1 += (unsigned)LC * XB;
            v2 += (unsigned)LC * XB;
            vy += (unsigned)LC * YB;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// One sweep pass: the 8-lane group `lp` reweights position p along s_path with `ratio` (gretel/gretel.py:79-98), takes the
// marginals of p and rewrites the table row G[p][rank of path[p]] -- k_marg<T, true>'s arithmetic, operation for operation.
template <typename T>
__device__ __forceinline__ void pipe_sweep_pos(const pipe_params &P, const win_desc &d, const uint8_t *s_path, const double *s_logtab,
                                               int p, int s, double ratio, double &removed, int *abort_flag)
{
    const int N = P.N, W = P.W, L = P.L;
    const symmap sm = P.sm;
    T *band = (T *)d.band;
    const bool act = p <= N;
    const int a = act ? (int)s_path[p] : 0;
    int nb = -1;
    T nval = (T)0;
    if (act) {
        for (int dd = s + 1; dd <= W; dd += 8) {
            const int j = p + dd;
            int mult = 0;
            if (j <= N - 1) mult = (dd == 1) ? 2 : 1;
            else if (j == N) mult = (dd == 1) ? 1 : 0;
            else if (j == N + 1) mult = (p == N) ? 1 : 0;
            if (mult) {
                const int b = (j == N + 1) ? (int)s_path[0] : (int)s_path[j];
                T *e = band + bidx(W, p, dd, a, b);
                T cur = *e;
                for (int q = 0; q < mult; q++) {
                    const double old = (double)cur;
                    const double nw = old - ratio * old;
                    cur = (T)nw;
                    removed += old - nw;
                }
                *e = cur;
                if (dd == 1) { nb = b; nval = cur; }
            }
        }
    }
    nb = __shfl(nb, 0, 8);
    nval = (T)__shfl((double)nval, 0, 8);
    // c_s(p): the row of the path's symbol from the cell (p, p+1) just updated (sequentially, in the storage dtype); the other
    // rows' sums are what the pass before left in cnt -- they have not changed
    double mine = 0.0;
    if (act && s < NSYM) {
        if (s == a) {
            T acc = (T)0;
#pragma unroll
            for (int x = 0; x < NSYM; x++) {
                T v = band[bidx(W, p, 1, s, x)];
                if (x == nb) v = nval;
                acc = acc + v;
            }
            mine = (double)acc;
        } else {
            mine = d.cnt[(size_t)p * 8 + s];
        }
    }
    double c[NSYM];
    double tot = 0.0;
    int nv = 0;
    uint32_t cm = 0;
#pragma unroll
    for (int x = 0; x < NSYM; x++) {
        c[x] = __shfl(mine, x, 8);
        if (c[x] > 0) {
            tot += c[x];
            if ((VALID_MASK >> x) & 1) { nv++; cm |= 1u << x; }
        }
    }
    const uint32_t cand = P.offer_zero ? VALID_MASK : cm;
    const uint32_t cmw = cm | (cand << 8);
    const uint32_t cm5 = cm5_of_cmask(sm, cand);
    double my_m = 0.0, my_lm = 0.0;
    if (s < NSYM) {
        my_m = (c[s] > 0 && tot != 0.0) ? c[s] / tot : 0.0;
        if ((VALID_MASK >> s) & 1) my_lm = gh_log10_tab(my_m, s_logtab, GH_LOG_SERIAL);
    }
    if (act) {
        if (s < NSYM) {
            if (s == a) d.cnt[(size_t)p * 8 + s] = c[s];
            if ((VALID_MASK >> s) & 1) {
                const int b5 = a6_of_sym(sm, s);
                d.minfo[(size_t)p * MINFO + b5] = my_lm;
                d.minfo[(size_t)p * MINFO + 5 + b5] = my_m;
                if (d.rinfo && ((cand >> s) & 1u) && __popc(cm5 & ((1u << b5) - 1u)) < 4) {
                    const int r = __popc(cm5 & ((1u << b5) - 1u));
                    d.rinfo[(size_t)p * RINFO + r] = my_lm;
                    d.rinfo[(size_t)p * RINFO + 4 + r] = my_m;
                }
            }
        } else {
            d.cnt[(size_t)p * 8 + 7] = tot;
            if (d.cmask[p] != cmw) atomicOr(abort_flag, 1);        // a candidate mask moved: the table is no longer the tensor's
        }
    }
    if (act && p < N && a != SYM_N) {
        // the table row this position's cells feed: source p, rank of path[p], lags 1..L (lane s takes lags s+1, s+9, ..)
        const int a6 = a6_of_sym(sm, a);
        const double nv_i = (double)nv, ca = __shfl(mine, a, 8);
        int row6 = a6;
        if (a6 < 5) row6 = ((cm5 >> a6) & 1u) ? __popc(cm5 & ((1u << a6) - 1u)) : -1;
        for (int l = s + 1; l <= L && row6 >= 0; l += 8) {
            double *out = d.G + (((size_t)p * 6 + row6) * L + (l - 1)) * LT_ROW;
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
            const uint32_t cmj = CM_CAND(d.cmask[snp]);
            const double den = (P.cond_mode == GH_COND_A) ? (double)d.nvalid[snp] + sum : (P.cond_mode == GH_COND_D ? nv_i + sum : nv_i + ca);
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
            // ranked table: the columns of lag l are the candidates of p+l in the order they are offered in
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

template <typename T, int LC, int NT>
__global__ void __launch_bounds__(NT) k_wpipe(pipe_params P, const win_desc *wd)
{
    typedef pipe_roles<NT> RL;
    typedef deep_layout<LC> DL;
    constexpr int NL = RL::NLW * 64, NR = RL::NRW * 64;
    constexpr int MAXPOS = NR / 8;                      // positions per sweep pass = the most a table buffer holds (C + WALK_OV)
    constexpr int ROW = LC * LT_ROW, BLK = 6 * ROW;
    constexpr int RS = pipe_pos_doubles(LC);
    extern __shared__ __align__(16) double smem[];
    const win_desc d = wd[blockIdx.x];
    dev_state *st = d.st;
    const int N = P.N, C = P.C;
    const int npos = C + WALK_OV;
    double *const g0 = smem;
    unsigned long long *const words0 = reinterpret_cast<unsigned long long *>(smem + 2 * (size_t)npos * RS);
    double *const s_logtab = reinterpret_cast<double *>(words0 + 128);
    double *const s_red = s_logtab + 256;
    pipe_ctl *const ctl = reinterpret_cast<pipe_ctl *>(s_red + NR);
    uint8_t *const s_path = reinterpret_cast<uint8_t *>(ctl + 1);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rmap = RL::map[wave];
    const int role = __builtin_amdgcn_readfirstlane(rmap >> 4), ridx = __builtin_amdgcn_readfirstlane(rmap & 15);

    // eligible?  (uniform: the control line as the kernels before this one left it)
    {
        const dev_ctl c0 = load_ctl(st);
        const bool ok = !c0.stop && c0.ranked != 0 && c0.first_hole > N && c0.narrow != 0;
        if (!ok) {
            if (tid == 0) st->pipe_status = PIPE_NOT_STARTED;
            return;
        }
    }
    logtab_stage(s_logtab);
    if (tid == 0) { ctl->ratio = 0.0; ctl->abort = 0; s_path[0] = SYM_US; }
    __syncthreads();

    const int nchunks = (N + C - 1) / C;
    const int npass = (N - 3 + C - 1) / C > 1 ? (N - 3 + C - 1) / C : 1;      // sweep passes that cover positions 0..N
    const int E = nchunks + 3;
#define PIPE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PIPE_BARRIER_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

    if (PIPE_DEV_ROLES & 8 ? role == PR_SWEEP : false) {
        // ---- sweepers ---------------------------------------------------------------------------------------------
        const int t = ridx * 64 + lane, lp = t >> 3, s = t & 7;
        for (int sp = 0; sp <= P.max_paths; sp++) {
            // sp < max_paths: beside the walk of path sp; sp == max_paths: the last path's reweight, nobody walks
            const bool last = sp == P.max_paths;
            const bool do_rw = sp > 0;
            const double ratio = ctl->ratio;
            double removed = 0.0;
            const int ne = last ? npass : E;
            for (int e = 0; e < ne; e++) {
                if (do_rw && e < npass) {
                    const int p = e == 0 ? lp : e * C + WALK_OV + lp;
                    const bool mine = e == 0 ? lp < C + WALK_OV : lp < C;
                    pipe_sweep_pos<T>(P, d, s_path, s_logtab, mine ? p : N + 1, s, ratio, removed, &ctl->abort);
                }
                if (!last) PIPE_BARRIER_DRAIN();
            }
            s_red[t] = removed;
            PIPE_BARRIER_DRAIN();                   // tail: the bookkeeper sums s_red and closes the records
            if (last || ctl->abort != 0) break;     // (read between the tail and the barrier behind it: no sweeper is at work)
            PIPE_BARRIER();                         // (the bookkeeper's ratio stands)
        }
        return;
    }
    if (PIPE_DEV_ROLES & 4 ? role == PR_LOAD : false) {
        // ---- loaders: the raw terms of chunk k from G, 32 bytes (columns 0..3 of one lag of one row) per task --------------
        const int t = ridx * 64 + lane;
        constexpr int TPP = 4 * LC;                                  // tasks per position: (row, lag)
        constexpr int MAXT = (MAXPOS * TPP + NL - 1) / NL;
        const int nsrc_all = N + LT_PAD;
        const int ntask = npos * TPP;
        typedef double ld_v2d __attribute__((ext_vector_type(2), aligned(8)));      // G rows are 8-byte aligned
        struct regs { ld_v2d lo[MAXT], hi[MAXT]; } R;
        auto fetch = [&](int k) {
            const int i0 = k * C;
#pragma unroll
            for (int it = 0; it < MAXT; it++) {
                const int q = t + it * NL;
                const int pp = q / TPP, r = q % TPP, row = r / LC, l = r % LC;
                const int sidx = i0 + pp;
                R.lo[it] = ld_v2d{0.0, 0.0}; R.hi[it] = ld_v2d{0.0, 0.0};
                if (q < ntask && sidx < nsrc_all) {
                    const double *src = d.G + (size_t)sidx * BLK + (sidx == 0 ? 5 : row) * ROW + l * LT_ROW;
                    R.lo[it] = *reinterpret_cast<const ld_v2d *>(src);
                    R.hi[it] = *reinterpret_cast<const ld_v2d *>(src + 2);
                }
            }
        };
        auto store = [&](int k) {
            double *dst = g0 + (size_t)(k & 1) * npos * RS;
            double *yr = dst + (size_t)npos * 32;
#pragma unroll
            for (int it = 0; it < MAXT; it++) {
                const int q = t + it * NL;
                const int pp = q / TPP, r = q % TPP, row = r / LC, l = r % LC;
                if (q < ntask) {
                    if (l < 2) {
                        lds_v2d *o = reinterpret_cast<lds_v2d *>(dst + (size_t)pp * 32 + l * 16 + row * 4);
                        o[0] = lds_v2d{R.lo[it].x, R.lo[it].y};
                        o[1] = lds_v2d{R.hi[it].x, R.hi[it].y};
                    } else {
                        double *o = yr + (size_t)pp * DL::YPOS + (size_t)row * 4 * DL::NYP + (l - 2);
                        o[0] = R.lo[it].x; o[DL::NYP] = R.lo[it].y; o[2 * DL::NYP] = R.hi[it].x; o[3 * DL::NYP] = R.hi[it].y;
                    }
                }
            }
        };
        bool aborted = false;
        for (int sp = 0; sp < P.max_paths; sp++) {
            PIPE_BARRIER();                                         // epochs 0, 1: the sweep's first two passes
            PIPE_BARRIER();
            fetch(0);
            store(0);
            if (nchunks > 1) fetch(1);
            PIPE_BARRIER();                                         // epoch 2 (the loads of chunk 1 stay in flight)
            for (int k = 0; k < nchunks; k++) {                     // epoch k + 3: the walker is in chunk k
                if (k + 1 < nchunks) store(k + 1);
                if (k + 2 < nchunks) fetch(k + 2);
                PIPE_BARRIER();
            }
            PIPE_BARRIER();                                         // tail
            if ((aborted = ctl->abort != 0)) break;                 // (read between the tail and the barrier behind it: no sweeper is at work)
            PIPE_BARRIER();
        }
        if (!aborted) PIPE_BARRIER();                               // behind the last sweep: its partial sums
        return;
    }
    if (PIPE_DEV_ROLES & 2 ? role == PR_BOOK : false) {
        // ---- bookkeeper -----------------------------------------------------------------------------------------------
        walk_params BP;
        BP.minfo = d.minfo;
        unsigned long long t_prev = 0;
        if (P.prof) t_prev = __builtin_amdgcn_s_memrealtime();
        auto reduce_removed = [&]() {               // fixed order: NRW values per lane, then the wavefront's tree
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < RL::NRW; q++) acc += s_red[q * 64 + lane];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
            return acc;
        };
        int sp = 0;
        bool aborted = false;
        for (; sp < P.max_paths; sp++) {
            uint8_t *path_out = d.paths + (size_t)sp * (N + 1);
            walk_totals Tt = {0.0, 0.0, INFINITY};
            double lane_min = INFINITY;
            book_row R0, R1;
#pragma unroll
            for (int q = 0; q < 8; q++) { R0.v[q] = lds_v2d{0.0, 0.0}; R1.v[q] = lds_v2d{0.0, 0.0}; }
            if (lane == 0) path_out[0] = SYM_US;
            PIPE_BARRIER(); PIPE_BARRIER(); PIPE_BARRIER();         // epochs 0..2
            auto consume = [&](int c, const book_row &R) {
                pipe_book_consume(d.minfo, path_out, s_path, words0 + (c & 1) * 64, LC, c * C, C, N, lane, R, Tt, lane_min, P.sm);
            };
            for (int k = 0; k < nchunks; k += 2) {
                book_prefetch(BP, k * C, C, N, lane, R0);
                if (k >= 1) consume(k - 1, R1);
                PIPE_BARRIER();
                if (k + 1 < nchunks) {
                    book_prefetch(BP, (k + 1) * C, C, N, lane, R1);
                    consume(k, R0);
                    PIPE_BARRIER();
                }
            }
            if (nchunks & 1) consume(nchunks - 1, R0);
            else consume(nchunks - 1, R1);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double o = __shfl_xor(lane_min, off);
                if (o < lane_min) lane_min = o;
            }
            PIPE_BARRIER_DRAIN();                                   // tail: s_red of the sweep beside this walk stands
            aborted = ctl->abort != 0;
            if (sp > 0) {
                const double mag = reduce_removed();
                if (lane == 0) d.recs[sp - 1].magnitude = mag;
            }
            if (aborted) break;
            if (lane == 0) {
                double r = lane_min;
                if (r < P.min_remove) r = P.min_remove;             // cmd.py:157-160
                gh_path_rec *rec = d.recs + sp;
                rec->hp_current = Tt.hp_cur;
                rec->hp_original = Tt.hp_orig;
                rec->ratio = r;
                rec->min_marginal = lane_min;
                rec->magnitude = 0.0;
                ctl->ratio = r;
                if (P.prof && sp < 12) {
                    const unsigned long long t_now = __builtin_amdgcn_s_memrealtime();
                    st->dbg8[sp] = t_now - t_prev;
                    t_prev = t_now;
                }
            }
            PIPE_BARRIER();
        }
        if (!aborted) {
            PIPE_BARRIER_DRAIN();                                   // the last sweep has ended
            const double mag = reduce_removed();
            if (lane == 0) d.recs[P.max_paths - 1].magnitude = mag;
        }
        if (lane == 0) {
            st->n_done = sp;
            st->ratio = ctl->ratio;
            st->pipe_status = aborted ? PIPE_ABORTED : PIPE_DONE;
        }
        return;
    }
    if (!(PIPE_DEV_ROLES & 1)) return;
    // ---- walker --------------------------------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(3);
    bool aborted = false;
    for (int sp = 0; sp < P.max_paths; sp++) {
        PIPE_BARRIER(); PIPE_BARRIER(); PIPE_BARRIER();             // epochs 0..2
        spec2x_walker<LC>(g0, words0, C, nchunks, lane);            // one barrier behind every chunk
        PIPE_BARRIER();                                             // tail
        if ((aborted = ctl->abort != 0)) break;
        PIPE_BARRIER();
    }
    if (!aborted) PIPE_BARRIER();
#undef PIPE_BARRIER
#undef PIPE_BARRIER_DRAIN
}

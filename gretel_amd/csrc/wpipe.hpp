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

#ifdef PIPE_PROF
#define PIPE_PROF_STAMPS 0
#else
#define PIPE_PROF_STAMPS 1
#endif
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
    int cond_mode;          // GH_COND_*: A, B, D rewrite a ROW of the table per position and lag, C and E a COLUMN (col = 1)
    int offer_zero;
    int prof;               // 1: the bookkeeper leaves s_memrealtime stamps per path in st->dbg8
    int mt;                 // marginal term: the walker adds log10 marginal of the candidate in front of the lag-1 term
    int col;                // column conditionals (C, E): the sweep works on the to-major copy of the band as well
    int synth;              // the loaders make the table entries of lags beyond the band instead of reading them (see there)
    double min_remove;
    symmap sm;
};

// wave -> role.  Waves go to the SIMDs round-robin (wave w on SIMD w & 3): SIMD 0 gets the walker, the bookkeeper and loader
// waves only, the sweepers (binary64 divisions and logarithms) share the other three.
enum { PR_WALK = 0, PR_BOOK = 1, PR_LOAD = 2, PR_SWEEP = 3 };
template <int NT> struct pipe_roles;
template <> struct pipe_roles<1024> {
    static constexpr int NLW = 6, NRW = 8;
    // SIMD class of wave w = w & 3.  Class 0: the walker, the bookkeeper (issue priority 2 below the walker's 3; with its sums
    // through LDS and its marginal per lane it is light enough to sit here: beside two sweepers it cost the default spec 1.5 % and
    // the column conditionals 10-20 %, same-call A/B) and two loader waves; classes 1, 2: three sweepers and a loader; class 3:
    // two sweepers, two loaders.
    //                                   w: 0         1          2          3          4         5          6          7
    static constexpr unsigned char map[16] = {PR_WALK << 4, (PR_SWEEP << 4) | 0, (PR_SWEEP << 4) | 1, (PR_SWEEP << 4) | 2, (PR_BOOK << 4), (PR_SWEEP << 4) | 3, (PR_SWEEP << 4) | 4, (PR_SWEEP << 4) | 5,
                                              (PR_LOAD << 4) | 1, (PR_SWEEP << 4) | 6, (PR_SWEEP << 4) | 7, (PR_LOAD << 4) | 0, (PR_LOAD << 4) | 2, (PR_LOAD << 4) | 3, (PR_LOAD << 4) | 4, (PR_LOAD << 4) | 5};
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

// windows with a few five-candidate positions (WIDE launches, see below): the side table's geometry
#define PIPE_WREC 8             /* records per LDS buffer: wide positions among the L + C positions a chunk can see */
#define PIPE_WMAX 4096          /* wide positions per window (the side table's size) */
__host__ __device__ constexpr int pipe_wrec_doubles(int L) { return 16 * L + 8; }
__host__ __device__ constexpr size_t pipe_gw_bytes(int N, int L) { return (size_t)PIPE_WMAX * pipe_wrec_doubles(L) * 8 + ((size_t)N / (L > 0 ? L : 1) + 4) * 4 + 64; }
// LDS of one workgroup: two table buffers of C + WALK_OV positions, the walker's words, log10's table, the sweepers' partial
// sums and slots (pipe_group_doubles per lane group), a line of control words, the path (N + 2 bytes)
// doubles per position of a table buffer: X1 = lag-1 terms [row][column] (16), X2 = lag-2 terms (16), Yr = lags 3..L [row][column][lag],
// rows padded to an even number of lags as in k_walk_spec's depth-2 layout
__host__ __device__ constexpr int pipe_pos_doubles(int L, int mt = 0) { return 32 + (mt ? 4 : 0) + 16 * deep_nyp(L); }
// The pipeline's own table gp: 32-byte pieces -- the four columns of one row at one lag (column conditionals: the four rows of one
// column).  Inside a source (L x 128 bytes, always whole 64-byte lines) a row's lags lie together, [row][lag][4], up to the last EVEN
// lag count: a row is then whole lines (128 bytes at four or five lags), and the piece of an odd last lag sits behind the rows,
// [row][4].  What this is for: the sweep rewrites one row per position and path, and the memory system charges for partial-line
// writes, not for bytes (see the loaders' comment on the far lags) -- with five lags a row was 160 bytes, two lines and a half; at C3 (band 4: the fifth lag is
// never rewritten) it now is two whole lines: 92 -> 81 ms per 100 paths x 256 windows, same call.  (The position's line of cnt written
// whole instead of its two values that change: no difference -- one line either way.  What counts is the number of LINES written.)
// Offset in doubles:
// (-DPIPE_GP_ROWMAJOR / -DPIPE_GP_LAGMAJOR: the two layouts this one was measured against -- a row's lags contiguous; lag-major inside
// a source -- for profiles/r5_table_layout.txt; diagnostic builds only)
__host__ __device__ constexpr unsigned pipe_gp_piece(unsigned src, unsigned rc, unsigned lag0, unsigned L)
{
#if defined(PIPE_GP_ROWMAJOR)
    return ((src * 4u + rc) * L + lag0) * 4u;
#elif defined(PIPE_GP_LAGMAJOR)
    return ((src * L + lag0) * 4u + rc) * 4u;
#else
    const unsigned Le = L & ~1u;
    return src * (L * 16u) + (lag0 < Le ? (rc * Le + lag0) * 4u : (4u * Le + rc) * 4u);
#endif
}
__host__ __device__ constexpr size_t pipe_fixed_bytes(int N, int nr_threads, int esize)
{
    return (size_t)2 * 64 * 8 + 256 * 8 + (size_t)nr_threads * 8 + (size_t)(nr_threads / 8) * 8 * (esize + 2) * 8 + 128 + 1024 + (((size_t)N + 2 + 15) & ~(size_t)15);
}
__host__ __device__ constexpr int pipe_chunk(int N, int L, int nr_threads, int esize, int mt = 0)
{
    const size_t fixed = pipe_fixed_bytes(N, nr_threads, esize);
    if (L < 2 || fixed + 2 * (size_t)(L + WALK_OV) * pipe_pos_doubles(L, mt) * 8 > WALK_LDS_MAX) return 0;
    long c = (long)((WALK_LDS_MAX - fixed) / (2 * (size_t)pipe_pos_doubles(L, mt) * 8)) - WALK_OV;
    const long cap = nr_threads / 8 - WALK_OV;
    if (c > cap) c = cap;
    if (c > 60) c = 60;
    c = (c / L) * L;
    return c >= L ? (int)c : 0;
}
__host__ __device__ constexpr size_t pipe_lds_bytes(int N, int L, int C, int nr_threads, int esize, int mt = 0)
{
    return 2 * (size_t)(C + WALK_OV) * pipe_pos_doubles(L, mt) * 8 + pipe_fixed_bytes(N, nr_threads, esize);
}
// WIDE launches: a buffer holds L more sources in front of the chunk, and behind the two buffers stand the two record areas
// (PIPE_WREC records each) and, per buffer, one byte per position of the buffer: its record's slot (0xff: not wide)
#define PIPE_WSLOT_BYTES 128
__host__ __device__ constexpr size_t pipe_wide_extra_bytes(int L) { return 2 * ((size_t)PIPE_WREC * pipe_wrec_doubles(L) * 8 + PIPE_WSLOT_BYTES); }
__host__ __device__ constexpr int pipe_chunk_w(int N, int L, int nr_threads, int esize, int mt = 0)
{
    const size_t fixed = pipe_fixed_bytes(N, nr_threads, esize) + pipe_wide_extra_bytes(L);
    if (L < 2 || fixed + 2 * (size_t)(2 * L + WALK_OV) * pipe_pos_doubles(L, mt) * 8 > WALK_LDS_MAX) return 0;
    long c = (long)((WALK_LDS_MAX - fixed) / (2 * (size_t)pipe_pos_doubles(L, mt) * 8)) - WALK_OV - L;
    const long cap = nr_threads / 8 - WALK_OV;      // (the sweep's positions per pass, as in the narrow launch; the buffer holds L more)
    if (c > cap) c = cap;
    if (c > 60) c = 60;
    if (c > 127 - L) c = 127 - L;                   // (the walker's mask of the buffer's wide positions: two 64-bit words)
    c = (c / L) * L;
    return c >= L ? (int)c : 0;
}
__host__ __device__ constexpr size_t pipe_lds_bytes_w(int N, int L, int C, int nr_threads, int esize, int mt = 0)
{
    return 2 * (size_t)(C + L + WALK_OV) * pipe_pos_doubles(L, mt) * 8 + pipe_fixed_bytes(N, nr_threads, esize) + pipe_wide_extra_bytes(L);
}

// What a sweep needs to know about a position and never changes while the pipeline runs (the candidate masks stand, or it stops):
// one 64-bit word per position, made in the kernel's prologue from cmask / nvalid (win_desc::pk):
//   bits 0..2   V(p): valid symbols seen              bits 3..5   candidates offered (<= 4 in a ranked window)
//   bits 6..17  the symbol of the candidate of rank 0..3, 3 bits each (the order get_edge_weights_at offers them in)
//   bits 18..38 per SYMBOL 0..6 the row of G a path through it rewrites: its rank; 5 for '_' at position 0; 7 = none
//   WIDE windows (below: a few positions offer five candidates):
//   bits 39..41 the symbol of the candidate of rank 4          bits 42..57 1 + the position's index among the window's wide positions (0: not wide)
#define PK_NVALID(w) ((int)((w) & 7u))
#define PK_NCAND(w) ((int)(((w) >> 3) & 7u))
#define PK_SYM(w, rb) ((int)(((w) >> (6 + 3 * (rb))) & 7u))
#define PK_ROW6(w, sym) ((int)(((w) >> (18 + 3 * (sym))) & 7u))
#define PK_SYM4(w) ((int)(((w) >> 39) & 7u))
#define PK_WIDX(w) ((int)(((w) >> 42) & 0xffffu) - 1)
// ---- windows with a FEW five-candidate positions (round 6: k_wpipe<.., WIDE = true>) ----------------------------------------
// A deletion column here and there (gretel/util.py:178-190: '-' is an ordinary symbol) used to send the whole window to the batched
// launches of rounds 1-4.  The pipeline's tables stay what they are -- ranks 0..3 of every position, 4 x 4 entries per (source,
// lag) -- and what a fifth candidate adds lives in a SIDE table, one record per wide position w:
//     S[l][c]   the row of w's candidate of rank 4 as a SOURCE: lag l + 1, column c = 0..4                      (l = 0..L-1)
//     T[l][r]   the column of w's candidate of rank 4 as a TARGET: lag l + 1, row r = 0..4 of source w - l - 1
//     LM4       log10 marginal of that candidate (marginal term)
// (entry (row 4, column 4) of two wide positions l + 1 apart stands in both records).  The sweepers keep the records current like the
// table (the entries a reweighted cell feeds, a division and a log10 each), the loaders bring the records of the wide positions a chunk
// can see into LDS beside the chunk's tables (at most PIPE_WREC per chunk; records are numbered in position order, so they are one
// contiguous run), and the walker is an exact stepper over five lanes (no speculation: pipe_wide_walker) that takes an entry from the
// table or from a record by the ranks involved.  Buffers of a WIDE launch hold L more sources in FRONT of the chunk: the stepper
// reads the rows of the last L picks from the buffer (the narrow walker carries them in registers).
template <bool WIDE = false>
__device__ __forceinline__ unsigned long long pipe_pack(uint32_t cmw, int nvalid, int p, symmap sm)
{
    const uint32_t cm5 = cm5_of_cmask(sm, CM_CAND(cmw));
    int nc = __popc(cm5);
    if (!WIDE && nc > 4) nc = 4;
    unsigned long long w = (unsigned long long)(nvalid & 7) | ((unsigned long long)nc << 3);
    if (WIDE) {
        const int b4 = nth_set5(cm5, 4);
        w |= (unsigned long long)(b4 >= 0 ? vsym(sm, b4) : 0) << 39;
    }
    for (int rb = 0; rb < 4; rb++) {
        const int b5 = nth_set5(cm5, rb);
        w |= (unsigned long long)(b5 >= 0 ? vsym(sm, b5) : 0) << (6 + 3 * rb);
    }
    for (int sym = 0; sym < NSYM; sym++) {
        const int a6 = a6_of_sym(sm, sym);
        int row6 = 7;
        if (a6 < 5) row6 = ((cm5 >> a6) & 1u) ? __popc(cm5 & ((1u << a6) - 1u)) : 7;
        else if (a6 == 5 && p == 0) row6 = 5;
        w |= (unsigned long long)row6 << (18 + 3 * sym);
    }
    return w;
}

struct pipe_ctl {
    double ratio;           // clamped minimum marginal of the path just walked: what the sweep of the next epochs removes
    int abort;              // a sweeper saw a candidate mask move
    int _pad[13];
    double lt_beyond[8];    // the table entry of a lag beyond the band by V: log10((1 + 0) / (V + 0)) as k_lt takes it
};
static_assert(sizeof(pipe_ctl) == 128, "two lines");

// The bookkeeper's two steps (kernels.hpp: book_prefetch / book_consume), lane = chunk-local position.  It takes the marginal of
// the ONE symbol the walker selected at a position from the counts itself -- m = c_s / total, log10 m: k_marg's expressions on
// the values the sweep keeps in cnt -- instead of reading a row of minfo that the sweep would have to refresh for all four
// candidates of every position and path (a division and a log10 each: a third of the sweep's entries; the bookkeeper's lanes
// take one per position and chunk).  minfo is only read for the ORIGINAL log-marginals (gretel.py:186), which never change.
struct pipe_book_row {
    lds_v2d c[4];               // cnt[j][0..7]: c_s(j) by symbol, [7] the total
    lds_v2d o[3];               // minfo[j][10..15]: candidate bits, log10 original marginal by compact symbol index
    unsigned long long pk;      // pk[j]
};

__device__ __forceinline__ void pipe_book_prefetch(const win_desc &d, int j0, int ns, int Nw, int lane, pipe_book_row &R)
{
    const int j = j0 + lane + 1;
    const int jj = (lane < ns && j <= Nw) ? j : 0;              // (no branch around the loads: see pipe_sweep_load)
    typedef __attribute__((address_space(1))) const lds_v2d gv2;
    gv2 *cs = (gv2 *)(uintptr_t)(d.cnt + (size_t)jj * 8);
    gv2 *os = (gv2 *)(uintptr_t)(d.minfo + (size_t)jj * MINFO + 10);
#pragma unroll
    for (int q = 0; q < 4; q++) R.c[q] = cs[q];
#pragma unroll
    for (int q = 0; q < 3; q++) R.o[q] = os[q];
    R.pk = ((__attribute__((address_space(1))) const unsigned long long *)(uintptr_t)d.pk)[jj];
}

// The two sums of a path (gretel.py:185-186) strictly in position order, one chunk's addends per call, by broadcast reads (same
// address in all lanes) -- as k_hp does; moving a register's lanes through scalar registers took four v_readlane per position.
// Lane 0 adds up the first sum, lane 1 the second (the other lanes one of the two, unused): ONE addition per position, and all 64
// slots of the chunk -- slots behind the chunk's last position hold +0.0, which leaves a sum that starts at +0.0 as it is.
// PIPE_SUM_DEPTH reads in flight, waited for one by one (LDS data returns in order).  Before: a batch of eight read, awaited, added
// -- every batch waited out an LDS round trip, 2 900 cycles per chunk, a third of the bookkeeper's time and on some boxes what
// the workgroup waited for; now 2 400 beside the walker (1 600 on a SIMD without it: the chain of dependent additions itself).  Left to hipcc, 26 reads went out at once into 52
// registers, every s_waitcnt it writes is lgkmcnt(0), and an addition outside the asm statement is moved away from its wait.
// (Tried: the last loader wavefront adding up one epoch behind the bookkeeper -- it became the slowest wavefront instead.)
#ifndef PIPE_SUM_DEPTH
#define PIPE_SUM_DEPTH 16
#endif
template <int J>
__device__ __forceinline__ void pipe_sum_step(unsigned addr, double &acc, double (&r)[PIPE_SUM_DEPTH])
{
    constexpr int D = PIPE_SUM_DEPTH;
    if constexpr (J < 64) {
        if constexpr (J >= D) {
            constexpr int I = J - D;                            // wait for read I, add it, put read J into its register
            asm volatile("s_waitcnt lgkmcnt(%2)\n\tv_add_f64 %0, %0, %1" : "+v"(acc) : "v"(r[I % D]), "n"(D - 1));      // gretel.py:185 (lane 0), :186 (lane 1)
        }
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[J % D]) : "v"(addr), "n"(16 * J));
        pipe_sum_step<J + 1>(addr, acc, r);
    } else if constexpr (J < 64 + D) {
        constexpr int I = J - D;
        asm volatile("s_waitcnt lgkmcnt(%2)\n\tv_add_f64 %0, %0, %1" : "+v"(acc) : "v"(r[I % D]), "n"(63 - I));
        pipe_sum_step<J + 1>(addr, acc, r);
    }
}
__device__ __forceinline__ void pipe_sum_chunk(const lds_v2d *s_bk, int lane, double &acc)
{
    const unsigned addr = (unsigned)(uintptr_t)s_bk + (unsigned)(lane & 1) * 8u;
    double r[PIPE_SUM_DEPTH];
    pipe_sum_step<0>(addr, acc, r);
}

template <bool WIDE = false>
__device__ __forceinline__ void pipe_book_consume(uint8_t *path_out, uint8_t *s_path, const unsigned long long *words, const double *s_logtab,
                                                  lds_v2d *s_bk /* 64 pairs of addends */,
                                                  int LC, int j0, int ns, int Nw, int lane, const pipe_book_row &R, double &hp_acc /* lane 0: hp_current, lane 1: hp_original */,
                                                  double &lane_min, symmap sm, unsigned long long *prof = nullptr)
{
#ifdef PIPE_PROF
    const unsigned long long pb0 = __builtin_amdgcn_s_memtime();
#endif
    double lm = 0.0, lm0 = 0.0, mg = INFINITY;
    const int j = j0 + lane + 1;
    if (lane < ns && j <= Nw) {
        const unsigned long long word = words[lane / LC];
        // (WIDE: a group walked by the exact stepper has three bits per pick -- a rank may be 4 -- and bit 63 set)
        const int rank = (WIDE && (word >> 63)) ? (int)((word >> (3 * (LC - 1 - lane % LC))) & 7ull) : (int)((word >> (2 * (LC - 1 - lane % LC))) & 3ull);
        const int sym = (WIDE && rank == 4) ? PK_SYM4(R.pk) : PK_SYM(R.pk, rank);      // the symbol of that rank at j (pipe_pack)
        const int b5 = a6_of_sym(sm, sym);
        const double cs[8] = {R.c[0].x, R.c[0].y, R.c[1].x, R.c[1].y, R.c[2].x, R.c[2].y, R.c[3].x, R.c[3].y};
        double c = cs[0];
#pragma unroll
        for (int q = 1; q < NSYM; q++) c = (sym == q) ? cs[q] : c;
        const double tot = cs[7];
        mg = (c > 0 && tot != 0.0) ? c / tot : 0.0;            // k_marg: marg[p][s]
        lm = gh_log10_tab(mg, s_logtab, GH_LOG_SERIAL);         //         minfo[p][b5]
        const double l0[5] = {R.o[0].y, R.o[1].x, R.o[1].y, R.o[2].x, R.o[2].y};
        lm0 = l0[0];
#pragma unroll
        for (int q = 1; q < 5; q++) lm0 = (b5 == q) ? l0[q] : lm0;
        path_out[j] = (uint8_t)sym;
        s_path[j] = (uint8_t)sym;
    }
    if (mg < lane_min) lane_min = mg;                   // gretel.py:182
#ifdef PIPE_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long pb1 = __builtin_amdgcn_s_memtime();
#endif
    // the two sums strictly in position order: the addends go through LDS and come back by broadcast reads
    s_bk[lane] = lds_v2d{lm, lm0};                      // (+0.0 for unused lanes)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    pipe_sum_chunk(s_bk, lane, hp_acc);
#ifdef PIPE_PROF
    asm volatile("" :: "v"(hp_acc));
    const unsigned long long pb2 = __builtin_amdgcn_s_memtime();
    if (prof) { prof[0] += pb1 - pb0; prof[1] += pb2 - pb1; }
#endif
}

// spec2_walker (kernels.hpp) over the RAW lag-1 / lag-2 terms: the batched loaders of rounds 1-4 expanded H = x1 + x2 for all 16
// hypotheses (64 doubles per target, every term fetched four times, 66 VGPRs of prefetch per loader lane); here a buffer holds
//   X[i][ 0..15] = x1 of SOURCE i: G[i][a1][lag 1][b]          X[i][16..31] = x2 of source i: G[i][a2][lag 2][b]
//   Yr[i][w][b][l - 3] = G[i][w][lag l][b], l = 3..L          (position 0: its '_' row in every row slot)
// and the walker takes H of target t as X1[t-1][a1][b] + X2[t-2][a2][b] itself: one more LDS read and one more addition per step,
// four bodies ahead of their use; the same IEEE addition the loaders did, so bit-identical.  Everything else is spec2_walker.
// MT (the marginal term): a position's record also holds LM[b] = log10 marginal of the candidate of rank b of the TARGET the
// source's lag-1 terms belong to, and the walker starts the sum with it: (LM + x1) + x2 -- the reference's (0.0 + lm) + x1, then x2.
template <int LC, bool MT>
__device__ __forceinline__ void spec2x_walker(double *g0, unsigned long long *words0, int C, int nchunks, int lane,
                                              unsigned long long *prof = nullptr /* diagnostic builds: [0] += cycles walking, [1] += cycles at the barriers */)
{
    static_assert(LC >= 2, "depth-2 speculation needs two lags");
    typedef deep_layout<LC> DL;
    constexpr int NY = DL::NY;
    constexpr unsigned XB = (MT ? 36 : 32) * 8, YB = DL::YPOS * 8, YWB = 4 * DL::NYP * 8;
    constexpr int RS = pipe_pos_doubles(LC, MT ? 1 : 0);
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
    unsigned lm0 = (unsigned)(uintptr_t)g0 + 256u + (unsigned)b * 8u;
    asm("" : "+v"(lm0));
    auto h_of = [&](unsigned a1, unsigned a2, unsigned al) __attribute__((always_inline)) {     // x1 (+ LM in front) + x2
        if constexpr (MT) return (*(lds_cdouble *)al + *(lds_cdouble *)a1) + *(lds_cdouble *)a2;
        else return *(lds_cdouble *)a1 + *(lds_cdouble *)a2;
    };
    unsigned long long B;
    if constexpr (MT) B = group_argmax<true>(*(lds_cdouble *)(lm0) + *(lds_cdouble *)(x10));
    else B = group_argmax<true>(*(lds_cdouble *)(x10));
    double accP = h_of(x10 + XB, x20, lm0 + XB);
    double H12 = h_of(x10 + 2 * XB, x20 + XB, lm0 + 2 * XB);
#pragma unroll
    for (int l = 2; l < LC; l++) Y[0][l] = *(lds_cdouble *)(y0 + (unsigned)(l - 2) * 8u);
    unsigned hist = 0, sh = 0;
    unsigned yw_v;
    asm("v_mov_b32 %0, %1" : "=v"(yw_v) : "i"(YWB));

    for (int k = 0; k < nchunks; k++) {
        unsigned v1 = x10 + (unsigned)(k & 1) * bufB, v2 = x20 + (unsigned)(k & 1) * bufB, vy = y0 + (unsigned)(k & 1) * bufB;
        unsigned vl = lm0 + (unsigned)(k & 1) * bufB;
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
                H12 = h_of(v1 + (unsigned)(gg * LC + u + 3) * XB, v2 + (unsigned)(gg * LC + u + 2) * XB, vl + (unsigned)(gg * LC + u + 3) * XB);
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
            vl += (unsigned)(UG * LC) * XB;
            vy += (unsigned)(UG * LC) * YB;
        }
        for (; g < ngroups; g++) {
            group(g, std::integral_constant<int, 0>{});
            v1 += (unsigned)LC * XB;
            v2 += (unsigned)LC * XB;
            vl += (unsigned)LC * XB;
            vy += (unsigned)LC * YB;
        }
#ifdef PIPE_PROF
        const unsigned long long tw = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef PIPE_PROF
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
        prof[0] += tw - prof[2]; prof[1] += tb - tw; prof[2] = tb;
#endif
    }
}

// The walker of a WIDE launch (windows in which a few positions offer five candidates): an exact stepper, no speculation.  Lane
// b = 0..4 of every group of eight holds candidate b of the target (all eight groups compute the same: the pick comes out of group 0);
// the term of lag l is the entry (row = the pick made l positions ago, column b) of source t - l, read from the chunk's tables where
// both ranks are below 4, from the S record of the source where its pick was its fifth candidate, from the T record of the target for
// lane 4 -- same lag-ascending IEEE additions as everywhere ((lm + x1) + x2 + ...; lags in front of the window are not added:
// gretel/gretel.py:155, the reference's own loop bound), first-wins arg-max over the lanes (gretel.py:166-174).  Buffer position
// bi holds source k C - L + bi: target k C + q sits at bi = q + L, its sources at q + L - l.
// f(integral_constant<0>), f(integral_constant<1>), ... : the steps of a group with their index as a compile-time constant
template <typename F, int... U>
__device__ __forceinline__ void pipe_unrolled(F &&f, std::integer_sequence<int, U...>) { (f(std::integral_constant<int, U>{}), ...); }

template <int LC, bool MT>
__device__ __forceinline__ void pipe_wide_walker(const double *g0, unsigned long long *words0, const double *wide0, const uint8_t *wslot0,
                                                 int C, int nchunks, int N, int lane,
                                                 unsigned long long *prof = nullptr /* diagnostic builds: [0] += cycles walking, [1] += cycles at the barriers */)
{
    // Two walkers in one: groups of LC targets that have no five-candidate position in sight (none among the group's targets and the
    // L positions in front of them: every rank involved is below 4) are walked by spec2x_walker's body -- depth-2 speculation, 16
    // hypothesis groups x 4 lanes -- and the others by the exact stepper (slow_step: lane b = 0..4 of every eight holds candidate b; the
    // term of lag l is the entry (row = the pick made l positions ago, column b) of source t - l, from the chunk's tables where both
    // ranks are below 4, from the S record of the source where its pick was its fifth candidate, from the T record of the target for
    // lane 4).  Between the two the speculative state is made afresh from the picks (prime).  Same lag-ascending IEEE additions in both
    // ((lm + x1) + x2 + ...), first-wins arg-max (gretel.py:166-174).  Buffer position bi holds source k C - L + bi.
    typedef deep_layout<LC> DL;
    constexpr int NY = DL::NY;
    constexpr unsigned XD = MT ? 36 : 32, YPOS = DL::YPOS, NYP = DL::NYP, RS = XD + YPOS;
    constexpr unsigned XB = XD * 8u, YB = YPOS * 8u, YWB = 4u * NYP * 8u;
    constexpr unsigned RECD = (unsigned)pipe_wrec_doubles(LC);
    static_assert(3 * LC < 63, "a group's picks, three bits each, in one word beside the format bit");
    typedef __attribute__((address_space(3))) const unsigned long long lds_cu64;
    const unsigned npos = (unsigned)(C + LC + WALK_OV);
    // (LDS addresses as 32-bit numbers, reads through address space 3: through generic pointers hipcc emits flat loads)
    const unsigned g0a = (unsigned)(uintptr_t)g0, wa0 = (unsigned)(uintptr_t)wide0, ws0 = (unsigned)(uintptr_t)wslot0;
    const unsigned bufB = npos * RS * 8u;
    // the exact stepper's lanes
    const unsigned b = (unsigned)lane & 7u, b3 = (unsigned)lane & 3u;
    const bool lane4 = b >= 4u;
    // the speculative walker's lanes: hypothesis (a1, a2) = (lane >> 2 & 3, lane >> 4), candidate b3
    const unsigned x10l = (unsigned)(lane & 15) * 8u, x20l = 128u + (unsigned)((lane >> 4) * 4 + (lane & 3)) * 8u, yl = b3 * (NYP * 8u), lml = 256u + b3 * 8u;
    double Y[LC][LC];
#pragma unroll
    for (int u = 0; u < LC; u++)
#pragma unroll
        for (int l = 0; l < LC; l++) Y[u][l] = 0.0;
    unsigned long long B = 0;
    double accP = 0.0, H12 = 0.0;
    unsigned hist = 0, sh = 0;                              // the speculative walker's picks, two bits each
    unsigned long long h3 = 0;                              // the exact stepper's, three bits each, the latest lowest (uniform)
    // any of the last LC picks a fifth candidate?  (bit 2 of a pick is set only for rank 4)
    constexpr unsigned long long M4 = []() { unsigned long long m = 0; for (int l = 0; l < LC; l++) m |= 4ull << (3 * l); return m; }();
    bool primed = false, in_slow = false;                   // (uniform)
    unsigned yw_v;
    asm("v_mov_b32 %0, %1" : "=v"(yw_v) : "i"(YWB));
    for (int k = 0; k < nchunks; k++) {
        const unsigned Xb = g0a + (unsigned)(k & 1) * bufB, Yb = Xb + npos * XB;
        const unsigned Wb = wa0 + (unsigned)(k & 1) * (PIPE_WREC * RECD * 8u);
        // which positions of the buffer have their record here (bit bi; the loaders' ballot): a record's slot is its rank among them
        unsigned long long wmask, wmask2;                   // (bit bi of the pair: positions 0..63, 64..127)
        {
            const unsigned long long wm = *(lds_cu64 *)(ws0 + (unsigned)(k & 1) * PIPE_WSLOT_BYTES), wm2 = *(lds_cu64 *)(ws0 + (unsigned)(k & 1) * PIPE_WSLOT_BYTES + 8u);
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wm), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wm >> 32));
            const unsigned lo2 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wm2), hi2 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wm2 >> 32));
            wmask = ((unsigned long long)hi << 32) | lo;
            wmask2 = ((unsigned long long)hi2 << 32) | lo2;
        }
        auto slot_of = [&](int bi) __attribute__((always_inline)) -> unsigned {
            return bi < 64 ? (unsigned)__builtin_popcountll(wmask & ((1ull << bi) - 1ull))
                           : (unsigned)(__builtin_popcountll(wmask) + __builtin_popcountll(wmask2 & ((1ull << (bi - 64)) - 1ull)));
        };
        auto wide_at = [&](int bi) __attribute__((always_inline)) -> bool { return ((bi < 64 ? wmask >> bi : wmask2 >> (bi - 64)) & 1ull) != 0; };
        unsigned long long *wk = words0 + (k & 1) * 64;
        const int ngroups = C / LC;
        int nw = -1;                                        // the next group (>= g) with a five-candidate target; ngroups: none
        for (int g = 0; g < ngroups;) {
            if (nw < g) {
                const int sft = g * LC + LC + 1;                                  // (target q of the chunk: buffer position q + LC; 1 <= sft < 64)
                const unsigned long long rem = (wmask >> sft) | (wmask2 << (64 - sft)), rem2 = wmask2 >> sft;
                nw = rem != 0ull ? g + (int)((unsigned)__builtin_ctzll(rem) / (unsigned)LC)
                                 : (rem2 != 0ull ? g + (int)((64u + (unsigned)__builtin_ctzll(rem2)) / (unsigned)LC) : ngroups);
                if (nw > ngroups) nw = ngroups;
            }
            // a five-candidate position among the group's targets, or a fifth candidate among the last LC picks (only the exact stepper
            // makes such picks: h3 is then current)?  Else the groups up to nw go to the speculative walker, two at a time, with nothing
            // between them but the loop (a lone wavefront issues an instruction every five cycles: what stands between two groups counts)
            const bool slowg = g == nw || (in_slow && (h3 & M4) != 0ull);
            if (slowg) {
#ifdef PIPE_PROF
                const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
                if (!in_slow) {
                    // the picks the speculative walker made, as the stepper keeps them
                    h3 = 0;
#pragma unroll
                    for (int l = 0; l < LC; l++) h3 |= (unsigned long long)((hist >> (2 * l)) & 3u) << (3 * l);
                    in_slow = true;
                }
#pragma unroll
                for (int u = 0; u < LC; u++) {
                    const int q = g * LC + u + 1;           // chunk-local target 1 .. C (buffer position q + LC)
                    const int t = k * C + q;
                    unsigned pick = 0;
                    const bool wide_t = wide_at(q + LC);
                    if (t <= N && t > LC && !wide_t && (h3 & M4) == 0ull) {
                        // four candidates here and no fifth among the last LC picks (most steps of such a group): the tables only --
                        // lanes 4..7 of every eight repeat lanes 0..3, the first maximum among the four is the pick
                        double acc;
                        {
                            const unsigned r1 = (unsigned)h3 & 7u;
                            const double x1 = *(lds_cdouble *)(Xb + (unsigned)(q + LC - 1) * XB + r1 * 32u + b3 * 8u);
                            if constexpr (MT) acc = *(lds_cdouble *)(Xb + (unsigned)(q + LC - 1) * XB + (32u + b3) * 8u) + x1;
                            else acc = x1;
                        }
                        if constexpr (LC >= 2) acc = acc + *(lds_cdouble *)(Xb + (unsigned)(q + LC - 2) * XB + 128u + ((unsigned)(h3 >> 3) & 7u) * 32u + b3 * 8u);
#pragma unroll
                        for (int l = 3; l <= LC; l++)
                            acc = acc + *(lds_cdouble *)(Yb + (unsigned)(q + LC - l) * YB + ((unsigned)(h3 >> (3 * (l - 1))) & 7u) * YWB + yl + (unsigned)(l - 3) * 8u);
                        double m = acc;
                        m = vmax_f64(m, dpp_f64<0xB1>(m));      // quad_perm [1,0,3,2]
                        m = vmax_f64(m, dpp_f64<0x4E>(m));      // quad_perm [2,3,0,1]
                        const unsigned long long win = __builtin_amdgcn_ballot_w64(acc == m);
                        pick = (unsigned)__builtin_ctz(((unsigned)win & 0xfu) | 0x10u) & 3u;
                    } else if (t <= N && t > LC && !wide_t) {
                        // four candidates here, a fifth among the last LC picks: as above, but the term of a source whose pick was its
                        // fifth candidate comes from that source's S record (one lag in L, as a rule: the steps behind a wide position
                        // where '-' was picked)
                        auto term = [&](int l) __attribute__((always_inline)) -> double {
                            const unsigned r = (unsigned)(h3 >> (3 * (l - 1))) & 7u;
                            const int bi = q + LC - l;
                            unsigned a;
                            if (r == 4u) a = Wb + slot_of(bi) * (RECD * 8u) + (unsigned)(l - 1) * 64u + b3 * 8u;
                            else if (l <= 2) a = Xb + (unsigned)bi * XB + (unsigned)(l - 1) * 128u + r * 32u + b3 * 8u;
                            else a = Yb + (unsigned)bi * YB + r * YWB + yl + (unsigned)(l - 3) * 8u;
                            return *(lds_cdouble *)a;
                        };
                        double x[LC];
#pragma unroll
                        for (int l = 1; l <= LC; l++) x[l - 1] = term(l);
                        double acc = x[0];
                        if constexpr (MT) acc = *(lds_cdouble *)(Xb + (unsigned)(q + LC - 1) * XB + (32u + b3) * 8u) + acc;
#pragma unroll
                        for (int l = 2; l <= LC; l++) acc = acc + x[l - 1];
                        double m = acc;
                        m = vmax_f64(m, dpp_f64<0xB1>(m));      // quad_perm [1,0,3,2]
                        m = vmax_f64(m, dpp_f64<0x4E>(m));      // quad_perm [2,3,0,1]
                        const unsigned long long win = __builtin_amdgcn_ballot_w64(acc == m);
                        pick = (unsigned)__builtin_ctz(((unsigned)win & 0xfu) | 0x10u) & 3u;
                    } else if (t <= N) {
                        const unsigned recT = Wb + (wide_t ? slot_of(q + LC) : 0u) * (RECD * 8u);
                        double x[LC];
#pragma unroll
                        for (int l = 1; l <= LC; l++) {
                            x[l - 1] = 0.0;
                            if (l <= t) {
                                const unsigned r = (unsigned)(h3 >> (3 * (l - 1))) & 7u;
                                const int bi = q + LC - l;
                                const unsigned aT = recT + (8u * LC + (unsigned)(l - 1) * 8u + r) * 8u;            // (column 4: the target's record)
                                unsigned aR, stride = 8u;
                                if (r == 4u) aR = Wb + slot_of(bi) * (RECD * 8u) + (unsigned)(l - 1) * 64u;         // (row 4: the source's record)
                                else if (l <= 2) aR = Xb + (unsigned)bi * XB + ((unsigned)(l - 1) * 16u + r * 4u) * 8u;
                                else { aR = Yb + (unsigned)bi * YB + (r * 4u * NYP + (unsigned)(l - 3)) * 8u; stride = NYP * 8u; }
                                const unsigned addr = lane4 ? aT : aR + b3 * stride;
                                x[l - 1] = *(lds_cdouble *)addr;
                            }
                        }
                        double acc = x[0];
                        if constexpr (MT) {
                            // the marginal term in front of the first addition: (0.0 + lm) + x1 -- it rides with the lag-1 source's record,
                            // the fifth candidate's in the target's record
                            const unsigned al = lane4 ? recT + 16u * LC * 8u : Xb + (unsigned)(q + LC - 1) * XB + (32u + b3) * 8u;
                            acc = *(lds_cdouble *)al + acc;
                        }
#pragma unroll
                        for (int l = 2; l <= LC; l++) acc = (l <= t) ? acc + x[l - 1] : acc;
                        if (b > 4u || (b == 4u && !wide_t)) acc = -INFINITY;
                        double m = acc;
                        m = vmax_f64(m, dpp_f64<0xB1>(m));      // quad_perm [1,0,3,2]
                        m = vmax_f64(m, dpp_f64<0x4E>(m));      // quad_perm [2,3,0,1]
                        m = vmax_f64(m, dpp_f64<0x141>(m));     // row_half_mirror
                        const unsigned long long win = __builtin_amdgcn_ballot_w64(acc == m);
                        pick = (unsigned)__builtin_ctz(((unsigned)win & 0xffu) | 0x100u) & 7u;      // first wins (gretel.py:166-174)
                    }
                    h3 = (h3 << 3) | (unsigned long long)pick;
                }
                wk[g] = (1ull << 63) | (h3 & ((1ull << (3 * LC)) - 1ull));       // (bit 63: three bits per pick)
                primed = false;
#ifdef PIPE_PROF
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                prof[3] += __builtin_amdgcn_s_memtime() - ts0; prof[5] += 1;
#endif
                g++;
                continue;
            }
            // ---- a group without a fifth candidate in sight: spec2x_walker's bodies
            const unsigned pb = (unsigned)(LC + g * LC);        // buffer position of source j0 = k C + g LC (the group's first body)
            if (in_slow) {
                hist = 0;
#pragma unroll
                for (int l = 0; l < LC && l < 16; l++) hist |= ((unsigned)(h3 >> (3 * l)) & 3u) << (2 * l);
                sh = hist << 2;
                in_slow = false;
            }
#ifdef PIPE_PROF
            const unsigned long long tp0 = __builtin_amdgcn_s_memtime();
            const bool was_primed = primed;
#endif
            if (!primed) {
                if (k == 0 && g == 0) {
                    // the start of a path (spec2x_walker's prologue): target 1 has the single term x1 of source 0, targets 2 and 3 both
                    const unsigned p0 = Xb + pb * XB;
                    auto h_at = [&](unsigned a1, unsigned a2, unsigned al) __attribute__((always_inline)) {
                        if constexpr (MT) return (*(lds_cdouble *)al + *(lds_cdouble *)a1) + *(lds_cdouble *)a2;
                        else return *(lds_cdouble *)a1 + *(lds_cdouble *)a2;
                    };
                    if constexpr (MT) B = group_argmax<true>(*(lds_cdouble *)(p0 + lml) + *(lds_cdouble *)(p0 + x10l));
                    else B = group_argmax<true>(*(lds_cdouble *)(p0 + x10l));
                    accP = h_at(p0 + XB + x10l, p0 + x20l, p0 + XB + lml);
                    H12 = h_at(p0 + 2 * XB + x10l, p0 + XB + x20l, p0 + 2 * XB + lml);
#pragma unroll
                    for (int u = 0; u < LC; u++)
#pragma unroll
                        for (int l = 0; l < LC; l++) Y[u][l] = 0.0;
#pragma unroll
                    for (int l = 2; l < LC; l++) Y[0][l] = *(lds_cdouble *)(Yb + pb * YB + yl + (unsigned)(l - 2) * 8u);
                    hist = 0; sh = 0;
                } else {
                    // the speculative state in front of body j0 from the picks: every source below is at least L positions behind a
                    // five-candidate position's reach, its pick a rank below 4
                    auto pick_of = [&](int back) __attribute__((always_inline)) -> unsigned { return (hist >> (2 * back)) & 3u; };     // pick of source j0 - back
                    auto yrow = [&](int back, int lag) __attribute__((always_inline)) -> double {      // row of source j0 - back under its pick, lag `lag` >= 3
                        return *(lds_cdouble *)(Yb + (pb - (unsigned)back) * YB + pick_of(back) * YWB + yl + (unsigned)(lag - 3) * 8u);
                    };
                    auto x1_at = [&](int fwd) __attribute__((always_inline)) -> double { return *(lds_cdouble *)(Xb + (pb + fwd) * XB + x10l); };   // source j0 + fwd (fwd may be -1)
                    auto x2_at = [&](int fwd) __attribute__((always_inline)) -> double { return *(lds_cdouble *)(Xb + (pb + fwd) * XB + x20l); };
                    auto lm_at = [&](int fwd) __attribute__((always_inline)) -> double { return *(lds_cdouble *)(Xb + (pb + fwd) * XB + lml); };
                    // target j0 + 1: x1 of source j0 (hypothesis a1), x2 of source j0 - 1 (a2), lag l of source j0 + 1 - l
                    double a0;
                    if constexpr (MT) a0 = (lm_at(0) + x1_at(0)) + x2_at(-1);
                    else a0 = x1_at(0) + x2_at(-1);
#pragma unroll
                    for (int l = 3; l <= LC; l++) a0 = a0 + yrow(l - 1, l);
                    B = group_argmax<true>(a0);
                    // target j0 + 2
                    double a1_;
                    if constexpr (MT) a1_ = (lm_at(1) + x1_at(1)) + x2_at(0);
                    else a1_ = x1_at(1) + x2_at(0);
#pragma unroll
                    for (int l = 3; l <= LC; l++) a1_ = a1_ + yrow(l - 2, l);
                    accP = a1_;
                    if constexpr (MT) H12 = (lm_at(2) + x1_at(2)) + x2_at(1);
                    else H12 = x1_at(2) + x2_at(1);
                    // the rows of sources j0 + 3 - LC .. j0 (slot = source mod LC; j0 is a multiple of LC)
#pragma unroll
                    for (int back = 0; back <= LC - 3; back++)
#pragma unroll
                        for (int l = 2; l < LC; l++) Y[(LC - back) % LC][l] = yrow(back, l + 1);
                    sh = hist << 2;
                }
                primed = true;
            }
#ifdef PIPE_PROF
            if (!was_primed) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); prof[4] += __builtin_amdgcn_s_memtime() - tp0; prof[6] += 1; }
#endif
            // (two groups in one straight-line block where the next one is of this kind too -- nearly always: hipcc then issues the
            // x1 / x2 reads of all ten bodies up front, as in spec2x_walker; one group at a time the fast bodies took 230 cycles
            // per step instead of 150)
            auto fast = [&](auto ng_, int g) __attribute__((always_inline)) {
                constexpr int NG = decltype(ng_)::value;
                const unsigned pb = (unsigned)(LC + g * LC);
                const unsigned v1 = Xb + pb * XB + x10l, v2 = Xb + pb * XB + x20l, vl = Xb + pb * XB + lml, vy = Yb + pb * YB + yl;
#pragma unroll
                for (int gg = 0; gg < NG; gg++) {
                    // the x1 / x2 (/ log-marginal) terms of the group's bodies, all on their way before the first body: left to itself
                    // hipcc issues them two bodies ahead of their use here (in spec2x_walker, with registers to spare, ten) and every
                    // other body waits out an LDS round trip
                    double hx1[LC], hx2[LC], hlm[LC];
#pragma unroll
                    for (int u = 0; u < LC; u++) {
                        hx1[u] = *(lds_cdouble *)(v1 + (unsigned)(gg * LC + u + 3) * XB);
                        hx2[u] = *(lds_cdouble *)(v2 + (unsigned)(gg * LC + u + 2) * XB);
                        if constexpr (MT) hlm[u] = *(lds_cdouble *)(vl + (unsigned)(gg * LC + u + 3) * XB);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < LC; u++) {
                        // A: resolve w_{j+1}   (body j = k C + (g + gg) LC + u)
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
                        if constexpr (MT) H12 = (hlm[u] + hx1[u]) + hx2[u];
                        else H12 = hx1[u] + hx2[u];
                    }
                    wk[g + gg] = (unsigned long long)hist;     // (two bits per pick)
                }
            };
#ifdef PIPE_PROF
            const unsigned long long tf0 = __builtin_amdgcn_s_memtime();
#endif
            for (; g + 2 <= nw; g += 2) fast(std::integral_constant<int, 2>{}, g);
            if (g < nw) { fast(std::integral_constant<int, 1>{}, g); g++; }
#ifdef PIPE_PROF
            prof[7] += __builtin_amdgcn_s_memtime() - tf0;
#endif
        }
#ifdef PIPE_PROF
        const unsigned long long tw = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef PIPE_PROF
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
        prof[0] += tw - prof[2]; prof[1] += tb - tw; prof[2] = tb;
#endif
    }
}

// seven values by NAME (a row of a cell): selects over the elements of a local array make hipcc keep the array in scratch memory and
// select the address instead (measured: 64-176 bytes of scratch per lane and a scratch load per table entry)
template <typename T>
struct row7 {
    T v0, v1, v2, v3, v4, v5, v6;
    // (every value passes through an empty asm before it is selected: hipcc simplifies this function on its own before it inlines
    // it, folds "select between two loads" into "load from a selected address", and the struct then has to live in scratch memory)
    __device__ __forceinline__ T get(int k) const
    {
        T a0 = v0, a1 = v1, a2 = v2, a3 = v3, a4 = v4, a5 = v5, a6 = v6;
        asm("" : "+v"(a0)); asm("" : "+v"(a1)); asm("" : "+v"(a2)); asm("" : "+v"(a3)); asm("" : "+v"(a4)); asm("" : "+v"(a5)); asm("" : "+v"(a6));
        T r = a0;
        r = (k == 1) ? a1 : r; r = (k == 2) ? a2 : r; r = (k == 3) ? a3 : r;
        r = (k == 4) ? a4 : r; r = (k == 5) ? a5 : r; r = (k == 6) ? a6 : r;
        return r;
    }
    __device__ __forceinline__ T sum() const { return (((((((T)0 + v0) + v1) + v2) + v3) + v4) + v5) + v6; }   // sequential, storage dtype
    __device__ __forceinline__ void load(const T *p) { v0 = p[0]; v1 = p[1]; v2 = p[2]; v3 = p[3]; v4 = p[4]; v5 = p[5]; v6 = p[6]; }
    __device__ __forceinline__ void zero() { v0 = v1 = v2 = v3 = v4 = v5 = v6 = (T)0; }
};

// One sweep pass in two stages, an epoch apart.  The 8-lane group of position p first LOADS what the position needs -- lane s the
// row (p, p+s+1) of the path's symbol (7 values: the cell on the path, the row sum of lag s+1's conditional, and for s = 0 the new
// c_a(p)), the old c_s(p), the packed words of p and of the target p+s+1 -- and, one barrier later, reweights along s_path with
// `ratio` (gretel/gretel.py:79-98), takes the marginals of p and rewrites the table row G[p][rank of path[p]]: k_marg<T, true>'s
// arithmetic, operation for operation, but
//   * no load between two dependent steps (bands and lag counts above 8 take their further rounds the slow way, loads in place);
//   * only what can change is recomputed: the marginals of the CANDIDATES of p (a symbol never seen keeps m = 0, log10 m = -inf)
//     and the table entries of candidate columns (missing ranks keep the -inf, rows behind the window the 0.0 k_lt wrote);
//   * those entries -- a binary64 division and a log10 each, 4 per lag and one per candidate of p: 24 at five lags -- are DEALT
//     over the eight lanes of the group through LDS slots (one lag per lane left three lanes idle for five rounds; now three
//     rounds), and what is selected by a run-time symbol is read from the slot by address instead of through select chains over
//     registers: the first version spent 1 300 vector instructions per lane and pass, a third of them binary64, and eight
//     sweeper waves saturated three SIMDs.
// A lane's LDS operations complete in order and a group sits in one wavefront: no barrier inside.
template <typename T>
struct sweep_regs {
    row7<T> row;            // band[p][a][s+1][.]
    double cnt;             // cnt[p][s]
    unsigned long long pkp; // pk[p]
    unsigned long long pkt; // pk[p + s + 1]
    unsigned rowoff;        // element offset of that row in the band (32-bit: the host checks the tensor is smaller than 2^31 elements)
    int a, b;               // path[p]; the to-symbol of the lane's cell: path[p + s + 1], '_' behind position N
    T xe[3];                // bands wider than 8: the elements of the lane's further cells (p, p + s + 1 + 8 r), r = 1..3, on the path
    unsigned xoff[3];       // ... and where they lie (0xffffffff: no such cell)
    // column conditionals (C, E): `row` holds the COLUMN of the lane's cell through the path's to-symbol -- tband[p][b][s+1][.], the
    // to-major copy -- because the reweighted element changes its column's sum and with it the entries of ALL rows in that column;
    row7<T> xrow;           // ... and the row of the path's symbol in the cell (p, p+1), for c_a(p)
    unsigned coloff;        // element offset of that column in tband
};

// doubles per lag slot (row as 8 x T, denominator, packed word of the target) and per lane group (8 slots)
template <typename T> __host__ __device__ constexpr int pipe_slot_doubles() { return (int)(8 * sizeof(T) / 8) + 2; }
__host__ __device__ constexpr int pipe_group_doubles(int esize) { return 8 * (esize + 2); }

// (global address space spelled out: through the generic pointers of win_desc hipcc emits flat loads, which also count as LDS
// operations; and NO branch around a load -- a load under a condition is merged with its zero default right behind the branch, and
// the merge waits for the load: the first version of this stage prefetched nothing.  Lanes without work load a valid address.)
#define PIPE_GLOBAL(T_) __attribute__((address_space(1))) T_
template <typename T_> __device__ __forceinline__ PIPE_GLOBAL(T_) *pipe_gptr(T_ *p) { return (PIPE_GLOBAL(T_) *)(uintptr_t)p; }
template <typename T_> __device__ __forceinline__ const PIPE_GLOBAL(T_) *pipe_gptr(const T_ *p) { return (const PIPE_GLOBAL(T_) *)(uintptr_t)p; }

template <typename T>
__device__ __forceinline__ void pipe_sweep_load(const pipe_params &P, const win_desc &d, const uint8_t *s_path, int p, int s, sweep_regs<T> &R)
{
    const int N = P.N, W = P.W;
    const int pp = p <= N ? p : 0;                              // (a lane group without a position reads position 0 and uses nothing)
    const int a = (int)s_path[pp];
    const int dd = s + 1 <= W ? s + 1 : 1;                      // (beyond the band: the compute stage takes zeros)
    const int j = pp + s + 1;
    R.a = a;
    R.b = (int)s_path[j <= N ? j : 0];                          // (j = N + 1: the end sentinel's partner is path[0] = '_'; beyond: unused)
    R.rowoff = (((unsigned)pp * 7u + (unsigned)a) * (unsigned)W + (unsigned)(dd - 1)) * 7u;
    R.coloff = (((unsigned)pp * 7u + (unsigned)(j <= N ? R.b : SYM_US)) * (unsigned)W + (unsigned)(dd - 1)) * 7u;
    const PIPE_GLOBAL(T) *rc = P.col ? pipe_gptr((const T *)d.tband) + R.coloff : pipe_gptr((const T *)d.band) + R.rowoff;
    R.row.v0 = rc[0]; R.row.v1 = rc[1]; R.row.v2 = rc[2]; R.row.v3 = rc[3]; R.row.v4 = rc[4]; R.row.v5 = rc[5]; R.row.v6 = rc[6];
    {
        // (column conditionals; the row conditionals read the same seven values once more and use none of them)
        const PIPE_GLOBAL(T) *xr = pipe_gptr((const T *)d.band) + ((unsigned)pp * 7u + (unsigned)a) * (unsigned)W * 7u;
        R.xrow.v0 = xr[0]; R.xrow.v1 = xr[1]; R.xrow.v2 = xr[2]; R.xrow.v3 = xr[3]; R.xrow.v4 = xr[4]; R.xrow.v5 = xr[5]; R.xrow.v6 = xr[6];
    }
    R.cnt = pipe_gptr((const double *)d.cnt)[(unsigned)pp * 8u + (unsigned)s];
    R.pkp = pipe_gptr((const unsigned long long *)d.pk)[pp];
    // bands up to 32: the path's element of each further cell of this lane (one value, not the row: no lag beyond the eighth
    // takes its row from here).  No branch around the loads: a cell that does not exist reads the first one's address.
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int dx = s + 9 + 8 * r, jx = pp + dx;
        const bool ex = W > 8 && dx <= W && p <= N && jx <= N + 1;
        const int bx = (int)s_path[(ex && jx <= N) ? jx : 0];
        R.xoff[r] = ex ? (((unsigned)pp * 7u + (unsigned)a) * (unsigned)W + (unsigned)(dx - 1)) * 7u + (unsigned)bx : 0xffffffffu;
        R.xe[r] = pipe_gptr((const T *)d.band)[ex ? R.xoff[r] : R.rowoff];
    }
    const int snp = p + s + 1;
    R.pkt = pipe_gptr((const unsigned long long *)d.pk)[(snp <= N && s + 1 <= P.L) ? snp : N + 1];     // (pk[N + 1] = 0: no candidates, no entries)
}

// `prefetch` is called as soon as the registers of R have been consumed (the row is in its slot, the three scalars are copied):
// the caller issues the loads of the NEXT pass into R there, and they are in flight under everything that follows.
template <typename T, int LC, bool WIDE, typename PF>
__device__ __forceinline__ void pipe_sweep_compute(const pipe_params &P, const win_desc &d, const uint8_t *s_path, const double *s_logtab,
                                                   double *s_deal /* this lane group's slots */,
                                                   int p, int s, double ratio, sweep_regs<T> &R, double &removed, int *abort_flag, PF &&prefetch)
{
    const int N = P.N, W = P.W;
    constexpr int L = LC;
    constexpr int Lr = L < 8 ? L : 8;
    constexpr int SD = pipe_slot_doubles<T>();
    PIPE_GLOBAL(T) *band = pipe_gptr((T *)d.band);
    PIPE_GLOBAL(double) *g_cnt = pipe_gptr(d.cnt), *g_G = pipe_gptr(d.gp);      // (the pipeline's own table: k_wpipe's prologue)
    PIPE_GLOBAL(double) *g_W = pipe_gptr(d.gw);                                 // (WIDE: the side table of the five-candidate positions)
    constexpr unsigned RECD = (unsigned)pipe_wrec_doubles(LC);
    const bool act = p <= N;
    const int a = R.a;
    auto mult_of = [&](int dd) __attribute__((always_inline)) {     // how often reweight_hansel_from_path visits the cell (p, p + dd): SURVEY section 8 a8
        const int j = p + dd;
        if (j <= N - 1) return (dd == 1) ? 2 : 1;
        if (j == N) return (dd == 1) ? 1 : 0;
        if (j == N + 1) return (p == N) ? 1 : 0;
        return 0;
    };
    auto reweight = [&](T cur, int mult) __attribute__((always_inline)) {
        for (int q = 0; q < mult; q++) {
            const double old = (double)cur;
            const double nw = old - ratio * old;
            cur = (T)nw;
            removed += old - nw;
        }
        return cur;
    };
    // this lane's slot: the row it loaded, as it lies in the band
    double *slot = s_deal + s * SD;
    T *srow = reinterpret_cast<T *>(slot);
    typedef T rowvec __attribute__((ext_vector_type(4), aligned(16)));
    {
        const bool inb = s + 1 <= W;                 // beyond the band the row is zeros (the load stage fetched a valid address instead)
        *reinterpret_cast<rowvec *>(srow) = rowvec{inb ? R.row.v0 : (T)0, inb ? R.row.v1 : (T)0, inb ? R.row.v2 : (T)0, inb ? R.row.v3 : (T)0};
        *reinterpret_cast<rowvec *>(srow + 4) = rowvec{inb ? R.row.v4 : (T)0, inb ? R.row.v5 : (T)0, inb ? R.row.v6 : (T)0, (T)0};
    }
    const double cnt_old = R.cnt;
    const unsigned long long pkp = R.pkp, pkt_in = R.pkt;
    const unsigned rowoff = R.rowoff, coloff = R.coloff;
    const int b_cell = R.b;
    const row7<T> xrow = R.xrow;
    PIPE_GLOBAL(T) *tband = pipe_gptr((T *)d.tband);
    const bool COL = P.col != 0;
    const T xe0 = R.xe[0], xe1 = R.xe[1], xe2 = R.xe[2];
    const unsigned xo0 = R.xoff[0], xo1 = R.xoff[1], xo2 = R.xoff[2];
    prefetch();
    // the first round out of the slot: lane s owns the cell (p, p + s + 1)
    // (the slot holds the row of the path's symbol -- the element on the path is [b] -- or, under a column conditional, the column
    // of the path's to-symbol -- the element is [a]; the tensor's element is the same one either way, and the to-major copy follows)
    T cur0 = (T)0;
    if (act && s + 1 <= W) {
        const int mult = mult_of(s + 1);
        if (mult) {
            const int k_el = COL ? a : b_cell;
            const T cur = reweight(srow[k_el], mult);
            band[rowoff + (unsigned)b_cell] = cur;
            if (COL) tband[coloff + (unsigned)a] = cur;
            srow[k_el] = cur;
            cur0 = cur;
        }
    }
    // bands wider than 8: the lane's further cells, their elements prefetched with the row (up to a band of 32), beyond that
    // with a load in place
    if (act && W > 8) {
        auto further = [&](T e, unsigned off, int dd) __attribute__((always_inline)) {
            if (off != 0xffffffffu) {
                const int mult = mult_of(dd);
                if (mult) {
                    const T cur = reweight(e, mult);
                    band[off] = cur;
                    if (COL) {          // (off = ((p 7 + a) W + dd - 1) 7 + b: the to-symbol back out of it)
                        const unsigned base = (((unsigned)p * 7u + (unsigned)a) * (unsigned)W + (unsigned)(dd - 1)) * 7u;
                        tband[(((unsigned)p * 7u + (off - base)) * (unsigned)W + (unsigned)(dd - 1)) * 7u + (unsigned)a] = cur;
                    }
                }
            }
        };
        further(xe0, xo0, s + 9);
        further(xe1, xo1, s + 17);
        further(xe2, xo2, s + 25);
        for (int dd = s + 33; dd <= W; dd += 8) {
            const int j = p + dd;
            const int mult = mult_of(dd);
            if (mult) {
                const int b = (j == N + 1) ? (int)s_path[0] : (int)s_path[j];
                PIPE_GLOBAL(T) *e = band + bidx(W, p, dd, a, b);
                const T cur = reweight(*e, mult);
                *e = cur;
                if (COL) tband[bidx(W, p, dd, b, a)] = cur;
            }
        }
    }
    // the slot as it now stands: its sum (sequentially, in the storage dtype) is the row (column) sum of lag s+1's conditional.
    // The new c_a(p) is the sum of the row of the path's symbol in the cell (p, p+1): lane 0's slot under a row conditional, the
    // extra row with lane 0's new element in place under a column conditional; the other c_s(p) are what the pass before left in
    // cnt: they have not changed
    const rowvec r03 = *reinterpret_cast<const rowvec *>(srow), r46 = *reinterpret_cast<const rowvec *>(srow + 4);
    const double rowsum0 = (double)((((((((T)0 + r03.x) + r03.y) + r03.z) + r03.w) + r46.x) + r46.y) + r46.z);
    double ca_src = rowsum0;
    if (COL) {
        const int bq = b_cell;
        const T e0 = bq == 0 ? cur0 : xrow.v0, e1 = bq == 1 ? cur0 : xrow.v1, e2 = bq == 2 ? cur0 : xrow.v2, e3 = bq == 3 ? cur0 : xrow.v3,
                e4 = bq == 4 ? cur0 : xrow.v4, e5 = bq == 5 ? cur0 : xrow.v5, e6 = bq == 6 ? cur0 : xrow.v6;
        ca_src = (double)((((((((T)0 + e0) + e1) + e2) + e3) + e4) + e5) + e6);
    }
    const double ca_new = __shfl(ca_src, 0, 8);
    const double mine = (act && s < NSYM) ? (s == a ? ca_new : cnt_old) : 0.0;
    double tot = 0.0;
#pragma unroll
    for (int x = 0; x < NSYM; x++) {
        const double cx = __shfl(mine, x, 8);
        if (cx > 0) tot += cx;
    }
    const unsigned p8 = (unsigned)p * 8u;
    if (act) {
        // (the position's whole line -- the seven counts as they stand and the total --, not just the two values that changed)
        g_cnt[p8 + (unsigned)s] = s == 7 ? tot : mine;
        // the masks stood when the pass before ended and only c_a(p) has changed since: they still stand iff it is still positive
        if (s == 0 && ((VALID_MASK >> a) & 1) && !(ca_new > 0)) atomicOr(abort_flag, 1);
    }
    const int row6 = act && p < N ? PK_ROW6(pkp, a) : 7;        // the table row this position's cells feed (7: none)
    // Which lags have entries that can change: under conditionals A and D a lag beyond the band has an all-zero row and the
    // denominator V + 0 -- constants; under B the denominator holds c_a(p), which the reweight has just changed, at every lag.
    const int Lw = (P.cond_mode == GH_COND_B || W >= Lr) ? Lr : W;
    // lag s + 1: the denominator and the target's word beside the row (column)
    const double nv_i = (double)PK_NVALID(pkp);
    if (s < Lw) {
        double den;
        unsigned long long word;
        if (!COL) {
            den = (P.cond_mode == GH_COND_A) ? (double)PK_NVALID(pkt_in) + rowsum0 : (P.cond_mode == GH_COND_D ? nv_i + rowsum0 : nv_i + ca_new);
            word = row6 < 6 ? pkt_in : 0ull;
        } else {
            // C: V(p) + column sum, E: V(target) + column sum; the entries this cell feeds sit in the column of the path's
            // to-symbol -- its rank among the target's candidates -- of every row of the source (pkt_in = 0: no target)
            den = (P.cond_mode == GH_COND_C ? nv_i : (double)PK_NVALID(pkt_in)) + rowsum0;
            const int rbs = PK_ROW6(pkt_in, b_cell);
            word = (act && p < N && pkt_in != 0ull && rbs < (WIDE ? 5 : 4)) ? (unsigned long long)(8 + rbs) : 0ull;      // (bit 3: live)
            if constexpr (WIDE) word |= ((pkt_in >> 42) & 0xffffull) << 8;      // (the target's record, should the column be its fifth)
        }
        slot[SD - 2] = den;
        reinterpret_cast<unsigned long long *>(slot)[SD - 1] = word;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned row4 = (unsigned)(row6 < 4 ? row6 : 0);                                                  // ('_' at position 0: row 0)
    // the entries, eight per round: a quotient and its log10 each (the marginals of the position are the bookkeeper's:
    // pipe_book_consume).  Row conditionals: entry (lag, column rb of the path's row); column conditionals: entry (lag, row ra,
    // the column of the path's to-symbol).
    const int NT4 = 4 * Lw;
    const int nrows = p == 0 ? 1 : PK_NCAND(pkp);              // (column conditionals: position 0 has its '_' row only)
    // WIDE: where an entry goes.  Row conditionals: entry (row of the path's symbol at p, lag li + 1, column rb of the target whose word
    // stands in the slot) -- the table where both ranks are below 4; the S record of p where the path's symbol is p's fifth candidate;
    // the T record of the target for its fifth column (and, where both are fifth, both records).  Column conditionals: entry (row rb of
    // p, lag, the column cb of the path's to-symbol at the target) likewise.
    const unsigned rowT = (unsigned)(row6 <= 4 ? row6 : 0);    // (the '_' row of position 0 stands in row slot 0)
    auto row_dst = [&](unsigned li, unsigned rb, unsigned long long tword, PIPE_GLOBAL(double) *&dst, PIPE_GLOBAL(double) *&dst2) __attribute__((always_inline)) {
        dst2 = nullptr;
        if (!WIDE || (rb < 4u && row6 != 4)) dst = g_G + pipe_gp_piece((unsigned)p, row4, li, (unsigned)L) + rb;
        else if (rb < 4u) dst = g_W + (unsigned)PK_WIDX(pkp) * RECD + li * 8u + rb;
        else {
            dst = g_W + (unsigned)PK_WIDX(tword) * RECD + 8u * (unsigned)L + li * 8u + rowT;
            if (row6 == 4) dst2 = g_W + (unsigned)PK_WIDX(pkp) * RECD + li * 8u + 4u;
        }
    };
    auto col_dst = [&](unsigned li, unsigned rb, unsigned cb, unsigned widx_t, PIPE_GLOBAL(double) *&dst, PIPE_GLOBAL(double) *&dst2) __attribute__((always_inline)) {
        dst2 = nullptr;
        if (!WIDE || (cb < 4u && rb < 4u)) dst = g_G + pipe_gp_piece((unsigned)p, cb, li, (unsigned)L) + rb;       // (transposed copy)
        else if (cb < 4u) dst = g_W + (unsigned)PK_WIDX(pkp) * RECD + li * 8u + cb;
        else {
            dst = g_W + widx_t * RECD + 8u * (unsigned)L + li * 8u + rb;
            if (rb == 4u) dst2 = g_W + (unsigned)PK_WIDX(pkp) * RECD + li * 8u + 4u;
        }
    };
    auto entry = [&](int li, int rb) __attribute__((always_inline)) {
        const double *sl = s_deal + li * SD;
        const unsigned long long word = reinterpret_cast<const unsigned long long *>(sl)[SD - 1];
        bool live;
        int k_el;
        PIPE_GLOBAL(double) *dst, *dst2;
        if (!COL) {
            live = rb < PK_NCAND(word);
            k_el = (WIDE && rb == 4) ? PK_SYM4(word) : PK_SYM(word, rb);
            row_dst((unsigned)li, (unsigned)rb, word, dst, dst2);
        } else {
            live = (word & 8ull) != 0 && rb < nrows;
            k_el = p == 0 ? SYM_US : ((WIDE && rb == 4) ? PK_SYM4(pkp) : PK_SYM(pkp, rb));
            col_dst((unsigned)li, (unsigned)rb, (unsigned)(word & 7ull), (unsigned)((word >> 8) & 0xffffull) - 1u, dst, dst2);
        }
        if (live) {
            const double num = 1.0 + (double)reinterpret_cast<const T *>(sl)[k_el];
            const double xq = num / sl[SD - 2];
            // (k_marg takes the straight-line logarithm where the arguments are normal, the general one otherwise: same values)
            const double v = gh_log10_is_normal(xq) ? gh_log10_normal_tab(xq, 0, s_logtab, GH_LOG_SERIAL) : gh_log10_tab(xq, s_logtab, GH_LOG_SERIAL);
            *dst = v;
            if constexpr (WIDE) { if (dst2) *dst2 = v; }
        }
    };
#pragma unroll 1
    for (int t = s; t < NT4; t += 8) entry(t >> 2, t & 3);
    if constexpr (WIDE) {
        // the fifth column of a lag whose target offers five candidates (row conditionals) / the fifth row of a position that does
        // (column conditionals): one lag per lane, and nothing but the look at the slot's word where there is none -- nearly everywhere
#pragma unroll 1
        for (int li = s; li < Lw; li += 8) {
            const unsigned long long word = reinterpret_cast<const unsigned long long *>(s_deal + li * SD)[SD - 1];
            if (!COL ? PK_NCAND(word) == 5 : ((word & 8ull) != 0 && nrows == 5)) entry(li, 4);
        }
    }
    if (P.mt) {
        // the marginal term: the walker of the next path adds log10 marginal of the CANDIDATE in front of its lag-1 term, so the
        // log-marginals of all candidates of p are due again after every reweight (lmr, by rank): lanes 0..3, one each (WIDE: lane 4
        // the fifth candidate's, into the position's record)
        const int sym_m = (WIDE && s == 4) ? PK_SYM4(pkp) : PK_SYM(pkp, s & 3);
        const double c_m = __shfl(mine, sym_m, 8);
        if (act && s < PK_NCAND(pkp)) {
            const double m = (c_m > 0 && tot != 0.0) ? c_m / tot : 0.0;            // k_marg: marg[p][s], minfo[p][b5]
            const double lmv = gh_log10_tab(m, s_logtab, GH_LOG_SERIAL);
            if (!WIDE || s < 4) pipe_gptr(d.lmr)[(unsigned)p * 4u + (unsigned)s] = lmv;
            else g_W[(unsigned)PK_WIDX(pkp) * RECD + 16u * (unsigned)L] = lmv;
        }
    }
    if constexpr (L > 8) {
        // lag counts above 8: the further lags one per lane, with loads in place
        if (!COL && row6 < 6) {
            for (int l = s + 9; l <= L; l += 8) {
                const int snp = p + l;
                if (snp > N || (l > W && P.cond_mode != GH_COND_B)) continue;        // (behind the window / beyond the band: constants)
                row7<T> rw;
                rw.zero();
                if (l <= W) rw.load((const T *)d.band + bidx(W, p, l, a, 0));
                const unsigned long long pkt = d.pk[snp];
                const double sum = (double)rw.sum();
                const double den = (P.cond_mode == GH_COND_A) ? (double)PK_NVALID(pkt) + sum : (P.cond_mode == GH_COND_D ? nv_i + sum : nv_i + ca_new);
                for (int rb = 0; rb < PK_NCAND(pkt); rb++) {
                    const double xq = (1.0 + (double)rw.get((WIDE && rb == 4) ? PK_SYM4(pkt) : PK_SYM(pkt, rb))) / den;
                    const double v = gh_log10_is_normal(xq) ? gh_log10_normal_tab(xq, 0, s_logtab, GH_LOG_SERIAL) : gh_log10_tab(xq, s_logtab, GH_LOG_SERIAL);
                    PIPE_GLOBAL(double) *dst, *dst2;
                    row_dst((unsigned)(l - 1), (unsigned)rb, pkt, dst, dst2);
                    *dst = v;
                    if constexpr (WIDE) { if (dst2) *dst2 = v; }
                }
            }
        }
        if (COL && act && p < N) {
            for (int l = s + 9; l <= L; l += 8) {
                const int snp = p + l;
                if (snp > N || l > W) continue;                     // (behind the window / beyond the band: constants)
                const int b = (int)s_path[snp];
                const unsigned long long pkt = d.pk[snp];
                const int rbs = PK_ROW6(pkt, b);
                if (rbs >= (WIDE ? 5 : 4)) continue;
                row7<T> cw;
                cw.load((const T *)d.tband + bidx(W, p, l, b, 0));  // (the column as this lane's reweight of the cell left it)
                const double den = (P.cond_mode == GH_COND_C ? nv_i : (double)PK_NVALID(pkt)) + (double)cw.sum();
                for (int ra = 0; ra < nrows; ra++) {
                    const double xq = (1.0 + (double)cw.get(p == 0 ? SYM_US : ((WIDE && ra == 4) ? PK_SYM4(pkp) : PK_SYM(pkp, ra)))) / den;
                    const double v = gh_log10_is_normal(xq) ? gh_log10_normal_tab(xq, 0, s_logtab, GH_LOG_SERIAL) : gh_log10_tab(xq, s_logtab, GH_LOG_SERIAL);
                    PIPE_GLOBAL(double) *dst, *dst2;
                    col_dst((unsigned)(l - 1), (unsigned)ra, (unsigned)rbs, (unsigned)PK_WIDX(pkt), dst, dst2);
                    *dst = v;
                    if constexpr (WIDE) { if (dst2) *dst2 = v; }
                }
            }
        }
    }
}

// (second launch bound = waves per SIMD: 512 threads at up to six lags are held to 128 registers so that TWO workgroups share a CU)
template <typename T, int LC, int NT, bool WIDE = false>
__global__ void __launch_bounds__(NT, (NT == 512 && LC <= 6) ? 4 : (NT == 1024 ? 4 : (NT == 768 ? 3 : 2))) k_wpipe(pipe_params P, const win_desc *wd)
{
    typedef pipe_roles<NT> RL;
    typedef deep_layout<LC> DL;
    constexpr int NL = RL::NLW * 64, NR = RL::NRW * 64;
    constexpr int MAXPOS = NR / 8;                      // positions per sweep pass = the most a table buffer holds (C + WALK_OV)
    const int XD = P.mt ? 36 : 32;                       // doubles of a position's x1 / x2 (/ LM) record
    const int RS = XD + DL::YPOS;
    extern __shared__ __align__(16) double smem[];
    const win_desc d = wd[blockIdx.x];
    dev_state *st = d.st;
    const int N = P.N, C = P.C;
    const int npos = C + WALK_OV + (WIDE ? LC : 0);     // (WIDE: L more sources in front of the chunk -- pipe_wide_walker)
    double *const g0 = smem;
    constexpr int RECD = pipe_wrec_doubles(LC);
    double *const wide0 = smem + 2 * (size_t)npos * RS;                                    // WIDE: [2][PIPE_WREC][RECD] the chunk's records
    uint8_t *const wslot0 = reinterpret_cast<uint8_t *>(wide0 + 2 * (size_t)PIPE_WREC * RECD);   // WIDE: [2][PIPE_WSLOT_BYTES] buffer position -> slot
    unsigned long long *const words0 = WIDE ? reinterpret_cast<unsigned long long *>(wslot0 + 2 * PIPE_WSLOT_BYTES)
                                            : reinterpret_cast<unsigned long long *>(smem + 2 * (size_t)npos * RS);
    double *const s_logtab = reinterpret_cast<double *>(words0 + 128);
    double *const s_red = s_logtab + 256;
    constexpr int GD = pipe_group_doubles((int)sizeof(T));     // the sweepers' slots: doubles per lane group
    double *const s_deal = s_red + NR;
    pipe_ctl *const ctl = reinterpret_cast<pipe_ctl *>(s_deal + (size_t)(NR / 8) * GD);
    lds_v2d *const s_bk = reinterpret_cast<lds_v2d *>(ctl + 1);      // the bookkeeper's addends: 64 x 16 bytes
    uint8_t *const s_path = reinterpret_cast<uint8_t *>(s_bk + 64);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rmap = RL::map[wave];
    const int role = __builtin_amdgcn_readfirstlane(rmap >> 4), ridx = __builtin_amdgcn_readfirstlane(rmap & 15);

    // eligible?  (uniform: the control line as the kernels before this one left it)
    {
        const dev_ctl c0 = load_ctl(st);
        // (WIDE: the windows the narrow launch left -- a table over the symbols, i.e. a position with five candidates somewhere)
        const bool ok = WIDE ? (!c0.stop && c0.ranked == 0 && c0.first_hole > N && N < 65536)
                             : (!c0.stop && c0.ranked != 0 && c0.first_hole > N && c0.narrow != 0);
        if (!ok) {
            if (tid == 0) st->pipe_status = PIPE_NOT_STARTED;
            return;
        }
    }
    logtab_stage(s_logtab);
    const int nchunks = (N + C - 1) / C;
    int *const s_cnt = reinterpret_cast<int *>(g0);                                        // WIDE prologue: [nchunks + 2] (the buffers are not in use yet)
    uint16_t *const s_wpos = reinterpret_cast<uint16_t *>(s_cnt + nchunks + 2);            // WIDE prologue: [PIPE_WMAX] position of every record
    // (no static LDS in this kernel: it would move the dynamic region off its 16-byte alignment)
    volatile int *const s_badp = &ctl->_pad[0];
    if constexpr (WIDE) {
        for (int q = tid; q <= nchunks; q += NT) s_cnt[q] = 0;
        if (tid == 0) *s_badp = 0;
        __syncthreads();
    }
    for (int q = tid; q <= N + 1; q += NT) {
        const unsigned long long w = q <= N ? pipe_pack<WIDE>(d.cmask[q], d.nvalid[q], q, P.sm) : 0ull;
        d.pk[q] = w;
        if constexpr (WIDE) { if (q >= 1 && q <= N && PK_NCAND(w) == 5) atomicAdd(&s_cnt[(q - 1) / C], 1); }
        if (P.mt) {             // log10 marginal of the candidates of q by RANK (k_marg left them by compact symbol index in minfo)
#pragma unroll
            for (int r = 0; r < 4; r++)
                d.lmr[(size_t)q * 4 + r] = (q <= N && r < PK_NCAND(w)) ? d.minfo[(size_t)q * MINFO + a6_of_sym(P.sm, PK_SYM(w, r))] : 0.0;
        }
    }
    if constexpr (WIDE) {
        // the records: numbered in POSITION order (a chunk's records are then one contiguous run of the side table), at most
        // PIPE_WMAX per window and PIPE_WREC among the L + C positions a chunk can see -- else the window is left to the launches
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int k = 0; k < nchunks; k++) { const int c = s_cnt[k]; s_cnt[k] = acc; acc += c; }
            s_cnt[nchunks] = acc;
            if (acc > PIPE_WMAX) *s_badp = 1;
        }
        __syncthreads();
        if (!*s_badp) {
            for (int k = tid; k < nchunks; k += NT) {
                int idx = s_cnt[k], nprev = 0, nown = 0;
                for (int p = k * C + 1 - LC; p <= k * C; p++)
                    if (p >= 1 && PK_NCAND(d.pk[p]) == 5) nprev++;
                const int pe = k * C + C < N ? k * C + C : N;
                for (int p = k * C + 1; p <= pe; p++) {
                    const unsigned long long w = d.pk[p];
                    if (PK_NCAND(w) == 5) {
                        d.pk[p] = w | ((unsigned long long)(idx + 1) << 42);
                        s_wpos[idx] = (uint16_t)p;
                        idx++; nown++;
                    }
                }
                if (nprev + nown > PIPE_WREC) *s_badp = 1;
                d.wdir[k] = (s_cnt[k] - nprev) | ((nprev + nown) << 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (*s_badp) {
            if (tid == 0) st->pipe_status = PIPE_NOT_STARTED;
            return;
        }
    }
    // the pipeline's own copy of the conditional table: of the ranked G[source][6 rows][lag][5 columns] only what a ranked window
    // uses -- rows and columns of the ranks 0..3, position 0's '_' row as its row 0 -- as [source][4][lag][4]: 32-byte pieces on
    // 32-byte boundaries (a row of G is 40 bytes: every piece straddled two sectors), 640 instead of 800 of 1 200 bytes per
    // position at five lags.  The sweeps write it, the loaders read it; G itself is rebuilt by whoever needs it next (dirty_lt).
    // Under a column conditional the copy is TRANSPOSED, [source][column][lag][row]: what a reweighted cell changes -- the entries
    // of every row in one column -- is then one 32-byte piece per lag as well (four pieces in four rows otherwise: 169k against
    // 267k haplotypes/s for E against A before), and the loaders transpose back while they stage.
    // WIDE: G is laid out over the SYMBOLS (k_lt: a position offers five candidates); entry (rank r of source i, lag l + 1, rank c of
    // the target) through the packed words -- what k_lt's ranked layout holds for the same window: a rank the source does not have is a
    // row of zeros, a rank the target does not have -inf, behind the window zeros; position 0 has its '_' row only (row slot 0)
    auto tab5 = [&](int i, int r, int l, int c) __attribute__((always_inline)) -> double {
        const int snp = i + l + 1;
        if (i < 0 || !(i < N && snp <= N)) return 0.0;
        const unsigned long long wi = d.pk[i], wt = d.pk[snp];
        int a6 = 5;
        if (i != 0) {
            if (r >= PK_NCAND(wi)) return 0.0;
            a6 = a6_of_sym(P.sm, r == 4 ? PK_SYM4(wi) : PK_SYM(wi, r));
        } else if (r != 0) return 0.0;
        if (c >= PK_NCAND(wt)) return -INFINITY;
        const int b5 = a6_of_sym(P.sm, c == 4 ? PK_SYM4(wt) : PK_SYM(wt, c));
        return d.G[(((size_t)i * 6 + a6) * LC + l) * LT_ROW + b5];
    };
    if constexpr (WIDE) {
        for (int q = tid; q < (N + LT_PAD) * 4 * LC; q += NT) {
            const int l = q % LC, row = (q / LC) & 3, i = q / (4 * LC);
            double *o = d.gp + pipe_gp_piece((unsigned)i, (unsigned)row, (unsigned)l, (unsigned)LC);
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = !P.col ? tab5(i, row, l, e) : tab5(i, e, l, row);      // (column conditionals: the transposed copy)
        }
        // the records: S = the row of the fifth candidate as a source, T = its column as a target, LM4 = its log10 marginal
        const int nwide = s_cnt[nchunks];
        for (int q = tid; q < nwide * LC * 10; q += NT) {
            const int e = q % 5, part = (q / 5) & 1, l = (q / 10) % LC, wi = q / (10 * LC);
            const int w = (int)s_wpos[wi];
            d.gw[(size_t)wi * RECD + (part ? 8 * LC : 0) + l * 8 + e] = part ? tab5(w - l - 1, (w - l - 1) == 0 ? 0 : e, l, 4) : tab5(w, 4, l, e);
        }
        if (P.mt)
            for (int wi = tid; wi < nwide; wi += NT) {
                const int w = (int)s_wpos[wi];
                d.gw[(size_t)wi * RECD + 16 * LC] = d.minfo[(size_t)w * MINFO + a6_of_sym(P.sm, PK_SYM4(d.pk[w]))];
            }
    }
    for (int q = tid; q < (WIDE ? 0 : (N + LT_PAD) * 4 * LC); q += NT) {
        const int l = q % LC, row = (q / LC) & 3, i = q / (4 * LC);
        double *o = d.gp + pipe_gp_piece((unsigned)i, (unsigned)row, (unsigned)l, (unsigned)LC);
        if (!P.col) {
            const double *g = d.G + (((size_t)i * 6 + (i == 0 ? 5 : row)) * LC + l) * LT_ROW;
            const bool keep = i != 0 || row == 0;
            o[0] = keep ? g[0] : 0.0; o[1] = keep ? g[1] : 0.0; o[2] = keep ? g[2] : 0.0; o[3] = keep ? g[3] : 0.0;
        } else {
            // (`row` of this piece is a COLUMN c: o[r] = the entry of row r -- position 0: its '_' row as row 0, nothing else)
#pragma unroll
            for (int r = 0; r < 4; r++)
                o[r] = (i != 0 || r == 0) ? d.G[(((size_t)i * 6 + (i == 0 ? 5 : r)) * LC + l) * LT_ROW + row] : 0.0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PIPE_PROF
    if (blockIdx.x == 0 && tid == 0) st->dbg8[9] = 0;
#endif
    if (tid == 0) { ctl->ratio = 0.0; ctl->abort = 0; s_path[0] = SYM_US; }
    if (tid < 8) ctl->lt_beyond[tid] = gh_log10((1.0 + 0.0) / ((double)tid + 0.0));
    __syncthreads();

#ifdef PIPE_PROF
    if (blockIdx.x == 0 && lane == 0) {          // which SIMD each wave sits on: HW_ID bits 5:4, two bits per wave
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        atomicOr((unsigned long long *)&st->dbg8[9], (unsigned long long)((hwid >> 4) & 3u) << (2 * wave));
    }
#endif
    const int npass = (N - 3 + C - 1) / C > 1 ? (N - 3 + C - 1) / C : 1;      // sweep passes that cover positions 0..N
    const int E = nchunks + 3;
#define PIPE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PIPE_BARRIER_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // diagnostic builds (-DPIPE_PROF): cycles a role spends at work (issue to the end of its instructions), waiting for its own
    // memory operations, and waiting at the barrier, summed over the epochs of a launch; window 0 leaves them in st->dbg8
#ifdef PIPE_PROF
    unsigned long long pf_t0 = 0, pf_t1 = 0, pf_work = 0, pf_drain = 0, pf_wait = 0;
#define PIPE_PROF_BEGIN() do { pf_t0 = __builtin_amdgcn_s_memtime(); } while (0)
#define PIPE_PROF_MID() do { pf_t1 = __builtin_amdgcn_s_memtime(); pf_work += pf_t1 - pf_t0; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); pf_t0 = __builtin_amdgcn_s_memtime(); pf_drain += pf_t0 - pf_t1; } while (0)
#define PIPE_PROF_END() do { pf_wait += __builtin_amdgcn_s_memtime() - pf_t0; } while (0)
#else
#define PIPE_PROF_BEGIN()
#define PIPE_PROF_MID()
#define PIPE_PROF_END()
#endif

    if (PIPE_DEV_ROLES & 8 ? role == PR_SWEEP : false) {
        // ---- sweepers ---------------------------------------------------------------------------------------------
        const int t = ridx * 64 + lane, lp = t >> 3, s = t & 7;
        auto pos_of = [&](int e) {                  // the position this lane group takes in pass e (N + 1: none)
            if (e >= npass) return N + 1;
            if (e == 0) return lp < C + WALK_OV ? lp : N + 1;
            return lp < C ? e * C + WALK_OV + lp : N + 1;
        };
        for (int sp = 0; sp <= P.max_paths; sp++) {
            // sp < max_paths: beside the walk of path sp; sp == max_paths: the last path's reweight, nobody walks
            const bool last = sp == P.max_paths;
            const bool do_rw = sp > 0;
            const double ratio = ctl->ratio;
            double removed = 0.0;
            const int ne = last ? npass : E;
            // pass e takes R apart first thing (the row into its LDS slot, three scalars copied) and the loads of pass e + 1 are
            // issued into the same registers right there: in flight under the whole computation.  (Two register sets, the next
            // one loaded in front of the pass: hipcc parks the arriving set in other registers in the middle of the pass and
            // waits for it there; one set loaded behind the pass: the latency stands in front of every barrier.  Both measured.)
            sweep_regs<T> R;
            if (do_rw) pipe_sweep_load<T>(P, d, s_path, pos_of(0), s, R);
            for (int e = 0; e < ne; e++) {
                PIPE_PROF_BEGIN();
                if (do_rw && e < npass)
                    pipe_sweep_compute<T, LC, WIDE>(P, d, s_path, s_logtab, s_deal + lp * GD, pos_of(e), s, ratio, R, removed, &ctl->abort,
                                              [&]() __attribute__((always_inline)) { pipe_sweep_load<T>(P, d, s_path, pos_of(e + 1), s, R); });
                PIPE_PROF_MID();
                if (!last) PIPE_BARRIER_DRAIN();
                PIPE_PROF_END();
            }
            s_red[t] = removed;
            PIPE_BARRIER_DRAIN();                   // tail: the bookkeeper sums s_red and closes the records
            if (last || ctl->abort != 0) break;     // (read between the tail and the barrier behind it: no sweeper is at work)
            PIPE_BARRIER();                         // (the bookkeeper's ratio stands)
        }
#ifdef PIPE_PROF
        if (blockIdx.x == 0 && t == 0) { st->dbg8[0] = pf_work; st->dbg8[1] = pf_drain; st->dbg8[2] = pf_wait; }
#endif
        return;
    }
    if (PIPE_DEV_ROLES & 4 ? role == PR_LOAD : false) {
        // ---- loaders: the raw terms of chunk k from G, 32 bytes (columns 0..3 of one lag of one row) per task --------------
        const int t = ridx * 64 + lane;
        // tasks per position: (row, lag), and with the marginal term one more: the log10 marginals of the candidates of the
        // position's lag-1 TARGET by rank (lmr, kept by the sweep), which the walker adds in front of x1
        const int TPP = 4 * LC + (P.mt ? 1 : 0);
        constexpr int MAXT = ((MAXPOS + (WIDE ? LC : 0)) * (4 * LC + 1) + NL - 1) / NL;     // (WIDE: L more sources in front of the chunk)
        const int nsrc_all = N + LT_PAD;
        const int ntask = npos * TPP;
        const PIPE_GLOBAL(double) *gLM = pipe_gptr((const double *)d.lmr);
        typedef double ld_v2d __attribute__((ext_vector_type(2), aligned(16)));     // (32 bytes per (row, lag) of the pipeline's table)
        struct regs { ld_v2d lo[MAXT], hi[MAXT]; } R;
        unsigned w_dir = 0;                     // WIDE: the directory word of the chunk being fetched (first record | records << 16)
        unsigned long long w_pk = 0, w_pk2 = 0; // ... and the packed words of buffer positions `lane` and 64 + `lane`
        const PIPE_GLOBAL(double) *gG = pipe_gptr((const double *)d.gp);
        const PIPE_GLOBAL(double) *gPK = pipe_gptr((const double *)(const void *)d.pk);      // (a made entry's two packed words ride in lo.x / hi.x)
        // Lags beyond the band are not read but MADE (conditionals other than B): their rows of the tensor are zeros, so an entry
        // is log10((1 + 0) / (V + 0)) for a candidate column of an existing row, -inf for a missing column, 0.0 for a missing row or
        // behind the window -- k_lt's values, from the packed words of source and target.  This only saves anything where the pieces
        // not read are whole lines: with a row's lags contiguous (160 bytes at five lags, the first layout) the counters showed the
        // same bytes fetched; with the table lag-major inside a source the fetch fell by 10 % and the kernel took 19 % LONGER (the
        // sweep then wrote five half lines per position instead of one run); with pipe_gp_piece's layout the odd last lag of the four
        // rows is two whole lines behind the rows -- at C3 (band 4, five lags) two of a source's ten.
        // Measured with that layout (profiles/r5_table_layout.txt, 256 windows x 100 paths, same call): fetch - 11 %; on a box where the
        // memory system binds A 86.7 -> 75.3 ms and E + marginal term 154.1 -> 146.9; on boxes where the walker's pace nearly binds
        // E + marginal term 114.5 -> 109.3 ms and A 71.3 -> 72.3 (the loaders' arithmetic shares the walker's SIMD).  On by default;
        // GH_PIPE_SYNTH=0 reads every lag.
        const bool synth_on = P.synth != 0 && P.cond_mode != GH_COND_B && P.W < LC;
        // A lane's tasks are the same in every chunk: one word each, taken apart where it is used.  (Left to itself hipcc keeps every
        // address of every branch of every task in a register across the path loop and spills them.)
        unsigned desc[MAXT];
#pragma unroll
        for (int it = 0; it < MAXT; it++) {
            const int q = t + it * NL;
            const int pp = q / TPP, r = q % TPP;
            desc[it] = (unsigned)pp | (unsigned)r << 12 | (q < ntask ? 1u << 20 : 0u);
        }
        struct task_of {
            int pp, r, row, l; bool live;
            __device__ __forceinline__ explicit task_of(unsigned w) {
                asm volatile("" : "+v"(w));
                pp = w & 0xfff; r = (w >> 12) & 0xff; live = (w >> 20) != 0;
                row = r / LC; l = r - row * LC;
            }
        };
        auto fetch = [&](int k) {
            // (no branch around a load: see pipe_sweep_load; tasks beyond the buffer or the table read source 0 and are dropped /
            // zeroed when the chunk is stored)
            const int i0 = k * C - (WIDE ? LC : 0);               // (WIDE: the buffer begins L sources in front of the chunk)
#pragma unroll
            for (int it = 0; it < MAXT; it++) {
                const task_of q_(desc[it]);
                const int pp = q_.pp, r = q_.r, row = q_.row, l = q_.l;
                const int sidx = i0 + pp;
                const bool ok = q_.live && sidx >= 0 && sidx < nsrc_all;
                const int si = ok ? sidx : 0;
                const PIPE_GLOBAL(double) *src = gG + pipe_gp_piece((unsigned)si, (unsigned)((si == 0 && !P.col) ? 0 : row), (unsigned)l, (unsigned)LC);
                if (r == 4 * LC) src = gLM + (unsigned)(si + 1 <= N ? si + 1 : N + 1) * 4u;      // (marginal term only)
                const bool syn = synth_on && r < 4 * LC && l >= P.W;
                const int tg = si + l + 1;
                const PIPE_GLOBAL(double) *src2 = src + 2;
                if (syn) {                                            // made, not read: the packed words of source and target instead
                    src = gPK + (si <= N ? si : N + 1);
                    src2 = gPK + (tg <= N ? tg : N + 1);
                }
                typedef double ld_v2d8 __attribute__((ext_vector_type(2), aligned(8)));
                typedef PIPE_GLOBAL(ld_v2d8) gv2d;
                const ld_v2d8 a_ = *reinterpret_cast<const gv2d *>(src), b_ = *reinterpret_cast<const gv2d *>(src2);
                R.lo[it] = ld_v2d{a_.x, a_.y};
                R.hi[it] = ld_v2d{b_.x, b_.y};
            }
            if constexpr (WIDE) {
                // the chunk's directory word and the packed word of this lane's buffer position: on their way with the table (in place, in
                // store(), the two were a dependent round trip in front of every chunk: +28 % on the loaders' time)
                w_dir = (unsigned)pipe_gptr((const int *)d.wdir)[k < nchunks ? k : 0];
                const int pq = i0 + lane;
                w_pk = pipe_gptr((const unsigned long long *)d.pk)[(pq >= 1 && pq <= N) ? pq : N + 1];
                const int pq2 = pq + 64;
                w_pk2 = pipe_gptr((const unsigned long long *)d.pk)[(pq2 >= 1 && pq2 <= N) ? pq2 : N + 1];
            }
        };
        auto store = [&](int k) {
            const int i0 = k * C - (WIDE ? LC : 0);
            double *dst = g0 + (size_t)(k & 1) * npos * RS;
            double *yr = dst + (size_t)npos * XD;
            // WIDE: the records of the wide positions this chunk can see (one contiguous run of the side table: they are numbered in
            // position order) are requested first and written to LDS last: their round trip passes under the table's stores
            constexpr int MAXW = WIDE ? (PIPE_WREC * RECD + NL - 1) / NL : 1;
            double wrec[MAXW];
            int w_cnt = 0;
            if constexpr (WIDE) {
                const int first = (int)(w_dir & 0xffffu);
                w_cnt = (int)(w_dir >> 16);
                const PIPE_GLOBAL(double) *gW = pipe_gptr((const double *)d.gw) + (size_t)first * RECD;
#pragma unroll
                for (int it = 0; it < MAXW; it++) {
                    const int e = t + it * NL;
                    wrec[it] = gW[e < w_cnt * RECD ? e : 0];
                }
                if (ridx == 0) {
                    // which positions of the buffer have their record here: one bit per position (two words: C + L may reach 65), by ballot -- the records are in
                    // position order, so a record's slot is its rank among the bits below its own
                    const int pp = lane;
                    const int wi = PK_WIDX(w_pk) - first, wi2 = PK_WIDX(w_pk2) - first;
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(PK_WIDX(w_pk) >= 0 && wi >= 0 && wi < w_cnt && pp >= 1 && pp <= C + LC);
                    const unsigned long long mask2 = __builtin_amdgcn_ballot_w64(PK_WIDX(w_pk2) >= 0 && wi2 >= 0 && wi2 < w_cnt && pp + 64 <= C + LC);
                    if (lane == 0) {
                        unsigned long long *wm = reinterpret_cast<unsigned long long *>(wslot0 + (k & 1) * PIPE_WSLOT_BYTES);
                        wm[0] = mask; wm[1] = mask2;
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < MAXT; it++) {
                const task_of q_(desc[it]);
                const int pp = q_.pp, r = q_.r, row = q_.row, l = q_.l;
                if (q_.live) {
                    const bool z = i0 + pp >= nsrc_all || i0 + pp < 0;     // behind (in front of) the table: zeros (the walker runs whole chunks)
                    double x0 = z ? 0.0 : R.lo[it].x, x1 = z ? 0.0 : R.lo[it].y, x2 = z ? 0.0 : R.hi[it].x, x3 = z ? 0.0 : R.hi[it].y;
                    if (synth_on && r < 4 * LC && l >= P.W) {
                        const int sa = i0 + pp, tg = sa + l + 1;
                        const unsigned pka = (unsigned)__double_as_longlong(R.lo[it].x), pkb = (unsigned)__double_as_longlong(R.hi[it].x);
                        const bool dead = sa >= N || tg > N || sa < 0;
                        const double lt = ctl->lt_beyond[(P.cond_mode == GH_COND_A || P.cond_mode == GH_COND_E) ? PK_NVALID(pkb) : PK_NVALID(pka)];
                        const int nrow = sa == 0 ? 4 : PK_NCAND(pka), ncol = PK_NCAND(pkb);      // (position 0: its '_' row in every slot)
                        auto val = [&](int rr, int cc) __attribute__((always_inline)) { return (dead || rr >= nrow) ? 0.0 : (cc < ncol ? lt : -INFINITY); };
                        if (!P.col) { x0 = val(row, 0); x1 = val(row, 1); x2 = val(row, 2); x3 = val(row, 3); }
                        else { x0 = val(0, row); x1 = val(1, row); x2 = val(2, row); x3 = val(3, row); }
                    }
                    if (r == 4 * LC) {
                        lds_v2d *o = reinterpret_cast<lds_v2d *>(dst + (size_t)pp * XD + 32);
                        o[0] = lds_v2d{x0, x1};
                        o[1] = lds_v2d{x2, x3};
                    } else if (!P.col) {
                        if (l < 2) {
                            lds_v2d *o = reinterpret_cast<lds_v2d *>(dst + (size_t)pp * XD + l * 16 + row * 4);
                            o[0] = lds_v2d{x0, x1};
                            o[1] = lds_v2d{x2, x3};
                        } else {
                            double *o = yr + (size_t)pp * DL::YPOS + (size_t)row * 4 * DL::NYP + (l - 2);
                            o[0] = x0; o[DL::NYP] = x1; o[2 * DL::NYP] = x2; o[3 * DL::NYP] = x3;
                        }
                    } else {
                        // transposed copy: the piece holds rows 0..3 of COLUMN `row`; position 0's '_' row stands in every row slot
                        const bool p0 = i0 + pp == 0;
                        const double r1 = p0 ? x0 : x1, r2 = p0 ? x0 : x2, r3 = p0 ? x0 : x3;
                        if (l < 2) {
                            double *o = dst + (size_t)pp * XD + l * 16 + row;
                            o[0] = x0; o[4] = r1; o[8] = r2; o[12] = r3;
                        } else {
                            double *o = yr + (size_t)pp * DL::YPOS + (size_t)row * DL::NYP + (l - 2);
                            o[0] = x0; o[4 * DL::NYP] = r1; o[8 * DL::NYP] = r2; o[12 * DL::NYP] = r3;
                        }
                    }
                }
                // (pacing these stores with s_sleep, as k_walk_spec's loaders do, changed nothing here: 66.7 ms per 100 paths either
                // way at 64 windows, 70.1 with 24 units of sleep per round -- scratch/README.md)
            }
            if constexpr (WIDE) {
                double *WA = wide0 + (size_t)(k & 1) * PIPE_WREC * RECD;
#pragma unroll
                for (int it = 0; it < MAXW; it++) {
                    const int e = t + it * NL;
                    if (e < w_cnt * RECD) WA[e] = wrec[it];
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
                PIPE_PROF_BEGIN();
                if (k + 1 < nchunks) store(k + 1);
                if (k + 2 < nchunks) fetch(k + 2);
                PIPE_PROF_MID();
                PIPE_BARRIER();
                PIPE_PROF_END();
            }
            PIPE_BARRIER();                                         // tail
            if ((aborted = ctl->abort != 0)) break;                 // (read between the tail and the barrier behind it: no sweeper is at work)
            PIPE_BARRIER();
        }
        if (!aborted) PIPE_BARRIER();                               // behind the last sweep: its partial sums
#ifdef PIPE_PROF
        if (blockIdx.x == 0 && t == 0) { st->dbg8[3] = pf_work; st->dbg8[4] = pf_drain; st->dbg8[5] = pf_wait; }
#endif
        return;
    }
    if (PIPE_DEV_ROLES & 2 ? role == PR_BOOK : false) {
        // ---- bookkeeper -----------------------------------------------------------------------------------------------
        // (a chain of dependent additions like the walker's steps: without priority every instruction of it queues behind the
        // sweepers of its SIMD -- 14 cycles per instruction measured, 6 100 cycles per chunk for the two sums alone)
        __builtin_amdgcn_s_setprio(2);
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
#ifdef PIPE_PROF
        unsigned long long bprof[2] = {0, 0};
#endif
        for (; sp < P.max_paths; sp++) {
            uint8_t *path_out = d.paths + (size_t)sp * (N + 1);
            double lane_min = INFINITY, hp_acc = 0.0;
            pipe_book_row R0, R1;
            if (lane == 0) path_out[0] = SYM_US;
            PIPE_BARRIER(); PIPE_BARRIER(); PIPE_BARRIER();         // epochs 0..2
            auto consume = [&](int c, const pipe_book_row &R) {
#ifdef PIPE_PROF
                pipe_book_consume<WIDE>(path_out, s_path, words0 + (c & 1) * 64, s_logtab, s_bk, LC, c * C, C, N, lane, R, hp_acc, lane_min, P.sm, bprof);
#else
                pipe_book_consume<WIDE>(path_out, s_path, words0 + (c & 1) * 64, s_logtab, s_bk, LC, c * C, C, N, lane, R, hp_acc, lane_min, P.sm);
#endif
            };
            for (int k = 0; k < nchunks; k += 2) {
                // (consume first: the loads it waits for were issued an epoch ago; behind a fresh prefetch hipcc waits for both)
                PIPE_PROF_BEGIN();
                if (k >= 1) consume(k - 1, R1);
                pipe_book_prefetch(d, k * C, C, N, lane, R0);
                PIPE_PROF_MID();
                PIPE_BARRIER();
                PIPE_PROF_END();
                if (k + 1 < nchunks) {
                    PIPE_PROF_BEGIN();
                    consume(k, R0);
                    pipe_book_prefetch(d, (k + 1) * C, C, N, lane, R1);
                    PIPE_PROF_MID();
                    PIPE_BARRIER();
                    PIPE_PROF_END();
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
                rec->hp_current = hp_acc;
                rec->ratio = r;
                rec->min_marginal = lane_min;
                rec->magnitude = 0.0;
                ctl->ratio = r;
                if (PIPE_PROF_STAMPS && P.prof && sp < 12) {
                    const unsigned long long t_now = __builtin_amdgcn_s_memrealtime();
                    st->dbg8[sp] = t_now - t_prev;
                    t_prev = t_now;
                }
            }
            if (lane == 1) d.recs[sp].hp_original = hp_acc;
            PIPE_BARRIER();
        }
        if (!aborted) {
            PIPE_BARRIER_DRAIN();                                   // the last sweep has ended
            const double mag = reduce_removed();
            if (lane == 0) d.recs[P.max_paths - 1].magnitude = mag;
        }
#ifdef PIPE_PROF
        if (blockIdx.x == 0 && lane == 0) { st->dbg8[6] = pf_work; st->dbg8[7] = pf_drain; st->dbg8[8] = pf_wait; st->dbg[0] = bprof[0]; st->dbg[1] = bprof[1]; }
#endif
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
#ifdef PIPE_PROF
    unsigned long long wprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int sp = 0; sp < P.max_paths; sp++) {
        PIPE_BARRIER(); PIPE_BARRIER(); PIPE_BARRIER();             // epochs 0..2
#ifdef PIPE_PROF
        wprof[2] = __builtin_amdgcn_s_memtime();
        if constexpr (WIDE) {
            if (P.mt) pipe_wide_walker<LC, true>(g0, words0, wide0, wslot0, C, nchunks, N, lane, wprof);
            else pipe_wide_walker<LC, false>(g0, words0, wide0, wslot0, C, nchunks, N, lane, wprof);
        } else {
            if (P.mt) spec2x_walker<LC, true>(g0, words0, C, nchunks, lane, wprof);
            else spec2x_walker<LC, false>(g0, words0, C, nchunks, lane, wprof);
        }
#else
        if constexpr (WIDE) {
            if (P.mt) pipe_wide_walker<LC, true>(g0, words0, wide0, wslot0, C, nchunks, N, lane);
            else pipe_wide_walker<LC, false>(g0, words0, wide0, wslot0, C, nchunks, N, lane);
        } else {
            if (P.mt) spec2x_walker<LC, true>(g0, words0, C, nchunks, lane);      // one barrier behind every chunk
            else spec2x_walker<LC, false>(g0, words0, C, nchunks, lane);
        }
#endif
        PIPE_BARRIER();                                             // tail
        if ((aborted = ctl->abort != 0)) break;
        PIPE_BARRIER();
    }
    if (!aborted) PIPE_BARRIER();
#ifdef PIPE_PROF
    if (blockIdx.x == 0 && lane == 0) {
        st->dbg8[10] = wprof[0]; st->dbg8[11] = wprof[1];
        if (WIDE) { st->dbg[2] = wprof[3] | (wprof[5] << 40); st->dbg[3] = wprof[7] | (wprof[6] << 40); }     // (cycles | count << 40: groups by the exact stepper, primes; cycles inside the speculative blocks)
    }
#endif
#undef PIPE_BARRIER
#undef PIPE_PROF_BEGIN
#undef PIPE_PROF_MID
#undef PIPE_PROF_END
#undef PIPE_BARRIER_DRAIN
}

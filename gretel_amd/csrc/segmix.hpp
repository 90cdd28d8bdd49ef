// segmix.hpp -- the segment-parallel extension over a MIXED-RADIX state space (included by segwalk.hpp behind seg_body).
//
// seg_body<5, ...> walks all 5^L symbol histories through a segment as soon as ONE position of the window offers five
// candidates -- 3 125 states at L = 5 where the window without that position has 1 024: 2.3x per path for one '-' anywhere.
// Here a state is the last L picks as candidate RANKS (rank r at position p = the r-th candidate of p in the order they are
// offered in, gretel/gretel.py:166-174), written as a mixed-radix number whose digit for position p has radix
//     R_p = 5 where p offers five candidates, 4 everywhere else (fewer candidates: the missing ranks are -inf columns),
// so the states entering target t number C_t = prod_{l=1..L} R_{t-l}: 1 024 where the L positions behind t offer four, 1 280
// behind one that offers five.  With d_l the digit of position t-l and M_1 = 1, M_{l+1} = M_l R_{t-l}:
//     sigma_t = sum_l d_l M_l,      NI_t = M_L (the entries of target t: digits d_1 .. d_{L-1}; one entry holds the pick for
//     every value of the oldest digit d_L),      sigma_{t+1} = (sigma_t mod NI_t) R_t + pick.
// The exit states of a segment are numbers in the radix system of the next segment's first target, so maps compose as
// before (k_scan, k_emit: class 6 = the ranked-pick code over SEGM_NS states).
//
// The conditional table stays in the five-symbol layout k_lt / k_rw / k_rwseg maintain: the workgroup gathers its slice of G
// into RANKED form while staging it (row d of a source = its d-th candidate, column b of a lag = the b-th candidate of the
// target; a rank that does not exist: -inf), so the Next tables and the state walk see ranks only.  Same IEEE additions in the
// same lag-ascending order for every state, first-wins arg-max over ranks = over symbols in the order they are offered in:
// bit-identical to seg_body<5> and to the serial walkers (tests/test_gpu_mixed.py, the fuzz).
//
// Work is handed out per (target, block of 64 tasks): a wavefront's target is uniform, so its radices sit in scalar
// registers, loops over digits stop at the radix, and a target whose candidates number four takes the four-candidate arg-max.
#pragma once

// candidate bits of position p as the workgroup sees them, in the form the staging uses: bits 0..4 the candidates (compact
// order); bit 7: position 0 (the '_' row 5, whatever the digit); bit 6: in front of the window (its terms are +0.0)
#define SEGM_ROW_US 0x80u
#define SEGM_NOPOS 0x40u

// the radix of a position's digit: 5 where it offers five candidates, 4 everywhere else -- also where it offers fewer, at
// position 0 and in front of the window (ranks that do not exist are columns of -inf and rows nobody reaches, as in the
// four-rank layout): every target out of sight of a five-candidate position then has the SAME radices and takes the fast
// body of (1b), and its states are the 1 024 of the four-rank layout
__device__ __forceinline__ int segm_radix(unsigned cm) { return __popc(cm & 31u) == 5 ? 5 : 4; }

// first-wins arg-max over the first BR of five sums (BR = the target's candidate count, 4 or 5; fewer: the rest is -inf)
template <int BR>
__device__ __forceinline__ unsigned segm_argmax(const double (&v)[5])
{
    if constexpr (BR == 5) return seg_argmax<5>(v, false);
    else { const double w[4] = {v[0], v[1], v[2], v[3]}; return seg_argmax<4>(w, false); }
}

// inclusive prefix sum over the lanes of a wavefront, by lane exchanges inside the register file (DPP: shifts within rows of 16
// lanes, then the last lane of a row broadcast to the rows behind it) -- __shfl_up goes through the LDS crossbar, 100 cycles a
// step, and the workgroup waits for these sums
__device__ __forceinline__ unsigned segm_scan(unsigned x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);     // row_bcast:31 into rows 2 and 3
    return (unsigned)v;
}

// a workgroup barrier that only orders LDS traffic: __syncthreads() is also a fence, and on this architecture (one counter for
// vector loads and stores) a fence waits for every load in flight -- the rows of G requested in front of step (0) would have to
// arrive before the first barrier of the step, and their round trip would hide nothing
__device__ __forceinline__ void segm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// HASP: called from k_rwseg (the patch of the reweight phase is there: candidate bits in LDS, the halo's rows and log-marginals)
template <int LC, bool HASP>
__device__ __forceinline__ void seg_body_mixed(const seg_params &P, unsigned char *smem, const seg_patch *patch = nullptr)
{
    static_assert(LC == SEGM_L, "the mixed-radix extension is instantiated for L = 5");
    constexpr int CH = SEGM_CH, DPW = 10, BITS = 3, NS = SEGM_NS;
    constexpr int RS = SEGM_ROW;                                   // doubles per staged row: five columns + one of padding (16-byte reads)
    constexpr int SPT = NS / SEG_THREADS;                          // states per thread (2)
    const seg_geom g = seg_geometry(P.N, LC, SEG_CLS_MIXED);
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (s >= g.S) return;
    const int t0 = s * g.seglen;                                   // targets t0+1 .. t1
    const int t1 = t0 + g.seglen < P.N ? t0 + g.seglen : P.N;
    double *Gs = reinterpret_cast<double *>(smem);                 // [(CH + LC - 1)][LC][5][RS]  ranked rows x ranked columns
    uint16_t *Nx = reinterpret_cast<uint16_t *>(Gs + (size_t)(CH + LC - 1) * LC * 5 * RS);      // [SEGM_NXCAP]
    uint32_t *rad = reinterpret_cast<uint32_t *>(Nx + SEGM_NXCAP); // [CH] radices of target tl: bits 3l..3l+2 = R_{t-l}, l = 0..L
    uint32_t *nxo = rad + CH;                                      // [CH + 1] where target tl's entries start in Nx
    uint32_t *nit = nxo + CH + 1;                                  // [CH] NI of target tl
    uint32_t *mgc = nit + CH;                                      // [CH] 1.0f / NI as bits (the walk's division: exact, see there)
    uint32_t *ito = mgc + CH;                                      // [CH + 1] first work item of target tl
    uint8_t *cmc = reinterpret_cast<uint8_t *>(ito + CH + 1);                            // [CH + LC] candidate bits of positions c0+1-LC .. c0+CH
    int *ctl = reinterpret_cast<int *>(cmc + 64);                  // [0] targets of this chunk, [1] its work items
    uint2 *winf = reinterpret_cast<uint2 *>(ctl + 2);
    // [2 CH] work items of the Next build, everything a wavefront needs of one in a single read: .x = target | block of 64 tasks << 8,
    // .y = the target's radices, .z = reciprocal multipliers of the two youngest radices, .w = where the target's entries start
    uint4 *iti = reinterpret_cast<uint4 *>(winf + CH);
    uint32_t *wfast = reinterpret_cast<uint32_t *>(iti + 2 * CH);     // [CH / 10] 1: every target of the word has four candidates and four-candidate predecessors

    __shared__ int s_next_item;                                    // the next work item of (1b) to be drawn

    unsigned sigma[SPT];
    bool live[SPT];                                                // (threads beyond the entry states walk state 0 and store nothing)
    unsigned n_entry = 0;

    SEG_STAMP(0);
    for (int c0 = t0; c0 < t1;) {
        const int avail = t1 - c0 < CH ? t1 - c0 : CH;
        // candidate bits of position p as this workgroup sees them (k_rwseg: its reweight phase has just worked them out -- its own
        // positions and, on a copy of their cells, the halo's, which belong to the neighbour who is rewriting them -- and left them
        // in LDS: no round trip to memory in front of the staging)
        auto cm_of = [&](int p) __attribute__((always_inline)) -> unsigned {
            const int pc = p < 1 ? 1 : (p > t1 ? t1 : p);          // (an address that exists, whatever p: no branch around the read)
            unsigned cm;
            if constexpr (HASP) cm = patch->cmall[pc - (s == 0 ? 0 : t0 + 1 - LC)] & 31u;
            else cm = (unsigned)__double_as_longlong(P.minfo[(size_t)pc * MINFO + 10]) & 31u;
            cm = (p < 0 || p > t1) ? (1u | SEGM_NOPOS) : cm;
            return p == 0 ? (1u | SEGM_ROW_US) : cm;
        };
        // (1a, first half) the rows of G this chunk can need go on their way before anything else: one thread per (source slot,
        // lag, row rank 0..3) -- 880 items for a whole chunk, one trip -- whose row is the d-th candidate of the source.  Nothing
        // but loads here, from addresses that always exist (a row that does not is dropped when it is stored): the code in front
        // of the loads has no branch, and nothing waits for them before step (0) is through.  What the chunk really holds (it
        // may end early where the Next entries do not fit) is only known behind that step; the rows are patched (k_rwseg: the
        // halo's), given the marginal term, ranked by column and stored then.  The fifth row only exists at the few sources with
        // five candidates: a second pass.
        const bool halo_chunk = HASP && c0 == t0 && t0 > 0;
        constexpr int TRIPS = ((CH + LC - 1) * LC * 4 + SEG_THREADS - 1) / SEG_THREADS;
        double wst[TRIPS][5];
        int a6st[TRIPS];
        auto load_row = [&](int ii, int l, int d, double (&w)[5]) __attribute__((always_inline)) -> int {
            const int i = c0 + 1 - LC + ii;
            const unsigned cmi = cm_of(i);
            int a6 = (cmi & SEGM_ROW_US) ? 5 : nth_set5(cmi & 31u, d);
            if ((cmi & SEGM_NOPOS) || i + l + 1 > t1) a6 = -1;
            const int ic = i < 0 ? 0 : i, ac = a6 < 0 ? 0 : a6;
            const double *src = P.G + (((size_t)ic * 6 + ac) * LC + l) * LT_ROW;
#pragma unroll
            for (int b = 0; b < 5; b++) w[b] = src[b];
            return a6;
        };
#pragma unroll
        for (int k = 0; k < TRIPS; k++) {
            const int e = tid + k * SEG_THREADS;
            const int ec = e < (avail + LC - 1) * LC * 4 ? e : 0;
            a6st[k] = load_row(ec / (4 * LC), (ec >> 2) % LC, ec & 3, wst[k]);
            if (e != ec) a6st[k] = -1;
        }
        // (0) the candidate bits of the chunk's positions and, per target, radices / entries / work items (wavefront 0)
        if (tid < CH + LC) cmc[tid] = (uint8_t)cm_of(c0 + 1 - LC + tid);
        segm_lds_barrier();
        if (wave == 0) {
            unsigned rd = 0, NI = 1, NJ = 1;
            if (lane < avail) {
#pragma unroll
                for (int l = 0; l <= LC; l++) {
                    const unsigned r = (unsigned)segm_radix(cmc[lane + LC - l]);
                    rd |= r << (3 * l);
                    if (l >= 1 && l <= LC - 1) NI *= r;
                    if (l >= 1 && l <= LC - 2) NJ *= r;
                }
            }
            // entries in front of target `lane` (inclusive scan over the lanes that hold a target)
            const unsigned inc = segm_scan(lane < avail ? NI : 0u);
            // the chunk: as many targets as fit the Next region, whole words of ten unless it is the segment's tail
            const unsigned long long fits = __ballot(lane < avail && inc <= (unsigned)SEGM_NXCAP);
            int nc = __popcll(fits);                               // (the sums ascend: the lanes that fit are a prefix)
            if (nc < avail) nc = nc / DPW * DPW;
            // work items of (1b): the general ones (a target that sees a five-candidate position) FIRST -- they cost more than twice a
            // fast one, and handed out in order every wavefront draws at most one of them before the fast ones fill the rounds
            constexpr unsigned RD_ALL4_ = 4u | (4u << 3) | (4u << 6) | (4u << 9) | (4u << 12) | (4u << 15);
            const bool general = lane < nc && rd != RD_ALL4_;
            const unsigned items = lane < nc ? (NJ + 63u) / 64u : 0u;
            const unsigned ginc = segm_scan(general ? items : 0u), finc = segm_scan(general ? 0u : items);
            const unsigned gtot = (unsigned)__builtin_amdgcn_readlane((int)ginc, 63), ftot = (unsigned)__builtin_amdgcn_readlane((int)finc, 63);
            const unsigned iinc = general ? ginc : gtot + finc;      // (inclusive end of this target's items in the list)
            // words of the state walk whose ten targets are all of the fast kind: the walk takes seg_body<4>'s steps there
            {
                const unsigned long long gen64 = __ballot(general);
                if (lane < (nc + DPW - 1) / DPW) {
                    const unsigned long long m = ((1ull << DPW) - 1ull) << (lane * DPW);
                    wfast[lane] = (gen64 & m) == 0ull && (lane + 1) * DPW <= nc ? 1u : 0u;
                }
            }
            if (lane < nc) {
                rad[lane] = rd; nit[lane] = NI; nxo[lane] = inc - NI; ito[lane] = iinc - items;
                winf[lane] = make_uint2(NI | ((rd & 7u) << 12) | ((inc - NI) << 15), __float_as_uint(1.0f / (float)NI));
                // ceil(2^15 / r) of the two youngest radices without a division (everybody waits for this wavefront)
                const unsigned r1 = (rd >> 3) & 7u, r2 = (rd >> 6) & 7u;
                auto rcp15 = [](unsigned r) { return r == 1u ? 32768u : (r == 2u ? 16384u : (r == 3u ? 10923u : (r == 4u ? 8192u : 6554u))); };
                const unsigned mg = rcp15(r1) | (rcp15(r2) << 16);
                for (unsigned q = 0; q < items; q++) iti[iinc - items + q] = make_uint4((unsigned)lane | (q << 8), rd, mg, inc - NI);
            }
            if (lane == nc - 1) { nxo[nc] = inc; ctl[0] = nc; ctl[1] = (int)(gtot + ftot); s_next_item = 0; }
        }
        segm_lds_barrier();
        const int nc = ctl[0], nitems = ctl[1];
        if (nc < 1) return;                                        // (cannot happen: ten targets always fit -- rather no path than a hang)
        if (c0 == t0) {
            // the states that enter the segment: every value of the L digits in front of its first target
            const unsigned rd0 = rad[0];
            n_entry = nit[0] * ((rd0 >> (3 * LC)) & 7u);
#pragma unroll
            for (int q = 0; q < SPT; q++) {
                const unsigned s0 = (unsigned)(tid + q * SEG_THREADS);
                live[q] = s0 < n_entry;
                sigma[q] = live[q] ? s0 : 0u;
            }
        }
        // (1a, second half) columns by rank: symbol column b goes to the slot of its rank among the target's candidates; ranks
        // beyond: -inf.  Sources c0+1-LC .. c0+nc-1 (slot ii), every lag; a row that does not exist, or whose target lies behind
        // the chunk, is never read: zeros (a source in front of the window: its terms ARE +0.0).
        const int nsrc = nc + LC - 1;
        auto store_row = [&](int ii, int l, int d, int a6, double (&w)[5]) __attribute__((always_inline)) {
            const int jj = ii + l + 1;                             // the target position i + l + 1 in cmc
            double *dst = Gs + ((size_t)(ii * LC + l) * 5 + d) * RS;
            if (a6 < 0 || jj >= nc + LC) {
#pragma unroll
                for (int b = 0; b < 5; b++) dst[b] = 0.0;
                return;
            }
            if constexpr (HASP) {
                if (halo_chunk && ii < LC) {
                    // the halo's rows as they stand AFTER the reweight this launch applies (the neighbour who owns them writes them to G
                    // in this very launch: what was loaded is one or the other): the row of the path's symbol, or -- conditionals C / E --
                    // one column of the block of (position, lag), the entry of every row
                    if (!patch->colmode) {
                        if (patch->row6[ii] == a6) {
#pragma unroll
                            for (int b = 0; b < 5; b++) w[b] = patch->row[ii][l][b];
                        }
                    } else {
                        const int col = patch->col[ii][l];
                        if (col >= 0 && ((patch->rmask[ii][l] >> a6) & 1u)) {
                            const double pv = patch->row[ii][l][a6];
#pragma unroll
                            for (int b = 0; b < 5; b++) w[b] = (b == col) ? pv : w[b];
                        }
                    }
                }
            }
            if (P.mt && l == 0) {
                // the marginal term in front of the first addition: (0.0 + lm) + x1 -- log10 marginal of the target's symbols, by symbol
                // like the columns (fetched here: held across step (0) it cost ten registers of a kernel that has none to spare)
                const double *lmp = P.minfo + (size_t)(c0 + 1 - LC + ii + 1) * MINFO;
#pragma unroll
                for (int b = 0; b < 5; b++) {
                    double lm = lmp[b];
                    if constexpr (HASP) { if (halo_chunk && ii + 1 < LC) lm = patch->lm5[ii + 1][b]; }
                    w[b] = lm + w[b];
                }
            }
            // columns by rank: symbol column b goes to the slot of its rank among the target's candidates; ranks beyond: -inf
            const unsigned cmj = cmc[jj] & 31u;
            const int rj = __popc(cmj);
#pragma unroll
            for (int b = 0; b < 5; b++)
                if ((cmj >> b) & 1u) dst[__popc(cmj & ((1u << b) - 1u))] = w[b];
#pragma unroll
            for (int r = 0; r < 5; r++)
                if (r >= rj) dst[r] = -INFINITY;
        };
#pragma unroll
        for (int k = 0; k < TRIPS; k++) {
            const int e = tid + k * SEG_THREADS;
            if (e < nsrc * LC * 4) store_row(e / (4 * LC), (e >> 2) % LC, e & 3, a6st[k], wst[k]);
        }
        for (int e = tid; e < nsrc * LC; e += SEG_THREADS) {
            if (__popc(cmc[e / LC] & 31u) == 5) {
                double w5[5];
                const int a6 = load_row(e / LC, e % LC, 4, w5);
                store_row(e / LC, e % LC, 4, a6, w5);
            }
        }
        __syncthreads();
        SEG_STAMP(1);
        // (1b) Next for every (target, valid state).  A work item is (target, 64 tasks): a task takes the digits d_1 .. d_{L-2} as
        // given and loops over d_{L-1} and the oldest, d_L, itself, as seg_body does.  Nearly every target of such a window has
        // four candidates and four-candidate predecessors: those take seg_body<4>'s task body as it is -- digits by shifts, sixteen
        // (d_{L-1}, d_L) pairs straight-line, the rows of lag L in registers.  A target that sees a five-candidate position takes
        // the general body: radices from scalar registers, digits up to their radix, every row read from LDS where it is used
        // (the registers the fast body fills are all this kernel has: held there as well, the rows of the general body spilled to
        // scratch memory -- 176 bytes per lane, 8 us per launch, the first version of this file).  Either way the sum of a state is
        // built lag ascending -- acc = x_1; acc += x_2; ... -- like everywhere.  Lag l of chunk-local target tl comes from slot
        // tl + LC - l.
        // (items are DRAWN, not dealt: a wavefront that is through with its item takes the next of the list -- the general ones stand
        // first, so whoever holds one is passed by while the others work the fast ones off; dealt round-robin, a chunk with more
        // than sixteen general items left two of them on one wavefront and the other fifteen waiting)
        for (;;) {
            int it = 0;
            if (lane == 0) it = atomicAdd(&s_next_item, 1);
            it = __builtin_amdgcn_readfirstlane(it);
            if (it >= nitems) break;
            const uint4 info = iti[it];
            const int tl = __builtin_amdgcn_readfirstlane((int)(info.x & 0xffu));
            const int jb = __builtin_amdgcn_readfirstlane((int)((info.x >> 8) & 0xffu));
            const unsigned rd = (unsigned)__builtin_amdgcn_readfirstlane((int)info.y);
            const unsigned nx0 = (unsigned)__builtin_amdgcn_readfirstlane((int)info.w);
            auto row_of = [&](int l, unsigned d) { return Gs + ((size_t)((tl + LC - l) * LC + (l - 1)) * 5 + d) * RS; };
            constexpr unsigned RD_ALL4 = 4u | (4u << 3) | (4u << 6) | (4u << 9) | (4u << 12) | (4u << 15);
            if (rd == RD_ALL4) {
                // ---- four candidates everywhere in sight: 64 tasks, one item
                const unsigned j = (unsigned)lane;
                const unsigned d1 = j & 3u, d2 = (j >> 2) & 3u, d3 = j >> 4;
                double acc[4];
                {
                    const double *p1 = row_of(1, d1), *p2 = row_of(2, d2), *p3 = row_of(3, d3);
#pragma unroll
                    for (int b = 0; b < 4; b++) acc[b] = (p1[b] + p2[b]) + p3[b];
                }
                double xl[4][4];
#pragma unroll
                for (int dL = 0; dL < 4; dL++) {
                    const double *row = row_of(LC, (unsigned)dL);
#pragma unroll
                    for (int b = 0; b < 4; b++) xl[dL][b] = row[b];
                }
                uint16_t *out = Nx + nx0 + j;
#pragma unroll
                for (int dS = 0; dS < 4; dS++) {
                    const double *row = row_of(LC - 1, (unsigned)dS);
                    double acc2[4];
#pragma unroll
                    for (int b = 0; b < 4; b++) acc2[b] = acc[b] + row[b];
                    unsigned packed = 0;
#pragma unroll
                    for (int dL = 0; dL < 4; dL++) {
                        double v[4];
#pragma unroll
                        for (int b = 0; b < 4; b++) v[b] = acc2[b] + xl[dL][b];
                        packed |= seg_argmax<4>(v, false) << (BITS * dL);
                    }
                    out[dS * 64] = (uint16_t)packed;
                }
                continue;
            }
            // ---- the general body
            const unsigned mg = (unsigned)__builtin_amdgcn_readfirstlane((int)info.z);
            const unsigned r1 = (rd >> 3) & 7u, r2 = (rd >> 6) & 7u, r3 = (rd >> 9) & 7u, r4 = (rd >> 12) & 7u, r5 = (rd >> 15) & 7u;
            const unsigned NJ = r1 * r2 * r3;
            const unsigned j = (unsigned)jb * 64u + (unsigned)lane;
            if (j < NJ) {
                // (radices are 1..5, j < 125: the quotients through exact reciprocal multiplies, ceil(2^15 / r) made by wavefront 0)
                const unsigned q1 = __umul24(j, mg & 0xffffu) >> 15;
                const unsigned d1 = j - __umul24(q1, r1);
                const unsigned d3 = __umul24(q1, mg >> 16) >> 15;
                const unsigned d2 = q1 - __umul24(d3, r2);
                uint16_t *out = Nx + nx0 + j;
                // (by the target's own candidate count: only the five-candidate position itself needs the fifth column)
                auto body = [&](auto br_) __attribute__((always_inline)) {
                    constexpr int BR = decltype(br_)::value;
                    double acc[BR];
                    {
                        const double *p1 = row_of(1, d1), *p2 = row_of(2, d2), *p3 = row_of(3, d3);
#pragma unroll
                        for (int b = 0; b < BR; b++) acc[b] = (p1[b] + p2[b]) + p3[b];
                    }
                    // (four columns: the rows of lag L for digits 0..3 in registers, as in the fast body -- 32 registers, which five
                    // columns would not leave; the fifth value of a digit is read where it is used)
                    constexpr int XL = BR == 4 ? 4 : 1;
                    double xl[XL][BR];
                    if constexpr (BR == 4) {
#pragma unroll
                        for (int dL = 0; dL < 4; dL++) {
                            const double *rowL = row_of(LC, (unsigned)dL);
#pragma unroll
                            for (int b = 0; b < BR; b++) xl[dL][b] = rowL[b];
                        }
                    }
                    auto one_dS = [&](unsigned dS) __attribute__((always_inline)) {
                        const double *row = row_of(LC - 1, dS);
                        double acc2[BR];
#pragma unroll
                        for (int b = 0; b < BR; b++) acc2[b] = acc[b] + row[b];
                        unsigned packed = 0;
                        auto one_dL = [&](unsigned dL, auto fromreg_) __attribute__((always_inline)) {
                            double v[BR];
                            if constexpr (decltype(fromreg_)::value) {
#pragma unroll
                                for (int b = 0; b < BR; b++) v[b] = acc2[b] + xl[dL][b];
                            } else {
                                const double *rowL = row_of(LC, dL);
#pragma unroll
                                for (int b = 0; b < BR; b++) v[b] = acc2[b] + rowL[b];
                            }
                            packed |= seg_argmax<BR>(v, false) << (BITS * dL);
                        };
#pragma unroll
                        for (int dL = 0; dL < 4; dL++) one_dL((unsigned)dL, std::integral_constant<bool, BR == 4>{});       // (radices are 4 or 5: digits 0..3 always exist)
                        if (r5 == 5u) one_dL(4u, std::false_type{});
                        out[dS * NJ] = (uint16_t)packed;
                    };
#pragma unroll
                    for (int dS = 0; dS < 4; dS++) one_dS((unsigned)dS);
                    if (r4 == 5u) one_dS(4u);
                };
                if ((rd & 7u) == 5u) body(std::integral_constant<int, 5>{});
                else body(std::integral_constant<int, 4>{});
            }
        }
        __syncthreads();
        SEG_STAMP(2);
        // (2) every entry state through the chunk; its picks go to hist one word (ten picks) at a time.  The oldest digit is
        // sigma / NI: NI is no power of two here, and the integer routes (v_mul_hi_u32, v_mul_lo_u32) issue at a quarter of the
        // rate -- (float(sigma) + 0.5) * (1.0f / NI), truncated, is that quotient exactly for sigma < 5 NI, NI <= 2048 (the
        // half keeps the product 0.5 / NI away from an integer, the roundings move it by less than 1e-6; checked exhaustively in
        // tests/test_mixed_radix_host.py), in three full-rate instructions.  What a step needs of its target (NI, its reciprocal,
        // R_t, where the entries lie) is fetched for the ten steps of a word at once, one lane per step, and handed round through
        // scalar registers: nothing but the entry read sits in the chain of a state.  A wavefront whose second state does not
        // exist (most, behind one five-candidate position) walks one.
        {
            const int w0 = (c0 - t0) / DPW;                        // chunks are whole words
            const bool second = __builtin_amdgcn_readfirstlane((int)(n_entry > (unsigned)(SEG_THREADS + wave * 64))) != 0;
            auto walk_words = [&](auto slots_) __attribute__((always_inline)) {
                constexpr int SL = decltype(slots_)::value;
                for (int tw = 0; tw < nc; tw += DPW) {
                    const int nu = nc - tw < DPW ? nc - tw : DPW;
                    unsigned word[SL];
#pragma unroll
                    for (int q = 0; q < SL; q++) word[q] = 0;
                    if (__builtin_amdgcn_readfirstlane((int)wfast[tw / DPW]) != 0) {
                        // ten targets with four candidates and four-candidate predecessors: NI = 256, R_t = 4, their entries lie
                        // behind one another -- seg_body<4>'s step (shift, mask, one table read), nothing fetched per target
                        const uint16_t *rows = Nx + __builtin_amdgcn_readfirstlane((int)nxo[tw]);
#pragma unroll
                        for (int u = 0; u < DPW; u++) {
#pragma unroll
                            for (int q = 0; q < SL; q++) {
                                const unsigned sg = sigma[q];
                                const unsigned hi = sg >> 8, idx = sg & 255u;
                                const unsigned d = ((unsigned)rows[u * 256 + idx] >> (BITS * hi)) & 7u;
                                word[q] |= d << (BITS * u);
                                sigma[q] = idx * 4u + d;
                            }
                        }
                    } else {
                        const uint2 wi = winf[tw + (lane < nu ? lane : 0)];
#pragma unroll
                        for (int u = 0; u < DPW; u++) {
                            if (u < nu) {
                                const unsigned ix = (unsigned)__builtin_amdgcn_readlane((int)wi.x, u);
                                const float rcp = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)wi.y, u));
                                const int negNI = -(int)(ix & 0xfffu);
                                const unsigned r0 = (ix >> 12) & 7u;
                                const uint16_t *rows = Nx + (ix >> 15);
                                const float half = 0.5f * rcp;
#pragma unroll
                                for (int q = 0; q < SL; q++) {
                                    const unsigned sg = sigma[q];
                                    const unsigned hi = (unsigned)__builtin_fmaf((float)sg, rcp, half);
                                    const unsigned idx = (unsigned)__mul24((int)hi, negNI) + sg;
                                    const unsigned d = ((unsigned)rows[idx] >> (BITS * hi)) & 7u;
                                    word[q] |= d << (BITS * u);
                                    sigma[q] = __umul24(idx, r0) + d;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < SL; q++)
                        if (live[q]) P.hist[((size_t)s * g.NW + w0 + tw / DPW) * NS + (unsigned)(tid + q * SEG_THREADS)] = word[q];
                }
            };
            if (second) walk_words(std::integral_constant<int, 2>{});
            else walk_words(std::integral_constant<int, 1>{});
        }
        SEG_STAMP(3);
        __syncthreads();                                           // the chunk's tables are overwritten by the next one
        c0 += nc;
    }
    // every state of the budget gets a map entry (k_scan composes whole maps): what cannot enter the segment maps to state 0
#pragma unroll
    for (int q = 0; q < SPT; q++) P.maps[(size_t)s * NS + (unsigned)(tid + q * SEG_THREADS)] = live[q] ? (uint16_t)sigma[q] : (uint16_t)0;
    SEG_STAMP(4);
}

/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  The product (gretel_amd/, libgretel_hip.so)
 * never links, loads or calls this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker / the CPU baseline.
 *
 * Plain-C restatement of the Gretel hot path, scalar, single thread:
 *   orc_fill            gretel/util.py:226-286, 329-333  (pair loop, sentinels, L)
 *   orc_counts_at       hansel get_counts_at      (call sites gretel/cmd.py:86,127)
 *   orc_marginal        hansel get_marginal_of_at (gretel/gretel.py:182,186)
 *   orc_edge_weights    hansel get_edge_weights_at(gretel/gretel.py:155)
 *   orc_generate_path   gretel/gretel.py:102-189
 *   orc_reweight_path   gretel/gretel.py:79-98    (+ hansel reweight_observation)
 *   orc_spin            gretel/cmd.py:148-179     (generate, 1% clamp, reweight)
 *
 * PARITY STATUS: the fill is pinned by the reference's own known answers
 * (tests/test_test.py:36-52, see tests/test_oracle_golden.py).  The Hansel
 * lookups live in hanselx==0.0.92 (reference setup.py:8) which is absent from
 * /root/reference: **parity unpinned**; the arithmetic below is the frozen spec
 * of oracle/hansel_ref.py, and this file is checked
 * bit-for-bit against that Python restatement (tests/test_oracle_c_vs_py.py).
 *
 * Storage: the reference allocates a dense [7][7][N+2][N+2] tensor
 * (gretel/util.py:83).  Here only cells with 1 <= pos_to-pos_from <= band are
 * backed by memory; all other cells are identically zero in the reference too
 * whenever no read carries SNPs further apart than `band` (zero is a fixed point
 * of reweight), so results are identical.
 *
 * log10: `use_libm=1` (the default of oracle/c_oracle.py) calls libm's log10 -- what Python's
 * math.log10 calls, i.e. the reference's (gretel/gretel.py:2); `use_libm=0` uses include/gh_detlog.h,
 * the restatement of glibc's log10 the HIP kernels evaluate.  The two are the same function bit for bit
 * (tests/test_detlog.py); orc_audit_* below evaluates them side by side over whole spins.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/gh_detlog.h"

#define NSYM 7
#define CELL 49
#define SYM_N 4
#define SYM_US 6       /* '_' */
#define MIN_REMOVE 0.01 /* gretel/cmd.py:157 */

typedef struct orc {
    int n, band, storage, cond_mode, marginal_term, use_libm;
    int L;
    int full_enum;              /* reweight walks all N(N+3)/2+1 cells like the reference */
    int order[5];               /* the order get_edge_weights_at offers the candidates in (symbol indices; spec.cand_order) */
    int offer_zero;             /* spec.offer_zero: every valid symbol is a candidate, a zero count included */
    int64_t n_slices, n_crumbs, covered;
    void *h;                    /* live matrix    [(n+2)*band*49] */
    void *h0;                   /* hansel.copy()  (gretel/cmd.py:79) */
    int64_t reweight_calls;
    struct orc_audit *audit;    /* orc_audit_begin(): both log10s in lock-step (below) */
} orc_t;

/* What a spin looks like when libm's log10 (the reference's: math.log10, gretel/gretel.py:2,185-186) and the
 * one of include/gh_detlog.h (the HIP kernels') are evaluated side by side on the SAME state.
 * The spin follows the handle's own log (use_libm); a step where the other log would have picked another symbol
 * is a FLIP.  No flip in a spin => both logs recover the same paths, ratios and tensor (induction over the steps:
 * same tensor and same prefix => same candidates; no log enters the reweight).  The census counts how close the
 * steps came: margin = best - second-best edge weight, in ulps of |best|. */
#define ORC_AUDIT_BINS 8        /* exact tie | <4 | <16 | <64 | <256 | <1024 | <2^20 | rest  (ulps of |best|) */
typedef struct orc_audit {
    int64_t steps;              /* steps with >= 2 candidates */
    int64_t flips;              /* the two logs disagree on the pick */
    int64_t order_diffs;        /* ... of which: weights tie exactly under one log and not under the other */
    int64_t nan_steps;          /* a NaN weight among the candidates */
    int64_t margin_bins[ORC_AUDIT_BINS];
    int64_t first_flip_path, first_flip_snp;
    double min_margin_ulps;     /* smallest non-zero margin seen */
    double max_abs_dhp_cur, max_abs_dhp_orig;   /* per path: |hp(libm) - hp(det)| */
    double max_abs_dw;          /* largest |w_libm - w_det| over all finite candidate weights */
    int64_t paths;
} orc_audit_t;

static const int VALID[5] = {0, 1, 2, 3, 5};   /* A C G T -  (unsymbols N,_ excluded) */

static int sym_of_char(int c)
{
    switch (c) {
    case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
    case 'N': return 4; case '-': return 5; case '_': return 6;
    default: return -1;
    }
}

static double lg(const orc_t *o, double x)
{
    if (o->use_libm) {
        if (x == 0.0) return -INFINITY;
        return log10(x);
    }
    return gh_log10(x);
}

static size_t cells(const orc_t *o) { return (size_t)(o->n + 2) * o->band * CELL; }

static int in_band(const orc_t *o, int i, int j)
{
    int d = j - i;
    return d >= 1 && d <= o->band && i >= 0 && j <= o->n + 1;
}

static size_t idx(const orc_t *o, int a, int b, int i, int j)
{
    return ((size_t)i * o->band + (j - i - 1)) * CELL + a * NSYM + b;
}

static double getm(const orc_t *o, const void *m, int a, int b, int i, int j)
{
    if (!in_band(o, i, j)) return 0.0;
    return o->storage ? ((const double *)m)[idx(o, a, b, i, j)]
                      : (double)((const float *)m)[idx(o, a, b, i, j)];
}

static int setm(orc_t *o, void *m, int a, int b, int i, int j, double v)
{
    if (!in_band(o, i, j)) return v == 0.0 ? 0 : -1;
    if (o->storage) ((double *)m)[idx(o, a, b, i, j)] = v;
    else ((float *)m)[idx(o, a, b, i, j)] = (float)v;
    return 0;
}

/* sequential sum in the storage dtype, x ascending (np.sum on a 7-slice) */
static double row_sum(const orc_t *o, const void *m, int a, int i, int j)
{
    if (!in_band(o, i, j)) return 0.0;
    if (o->storage) {
        const double *p = (const double *)m + idx(o, a, 0, i, j);
        double acc = 0.0;
        for (int x = 0; x < NSYM; x++) acc = acc + p[x];
        return acc;
    } else {
        const float *p = (const float *)m + idx(o, a, 0, i, j);
        float acc = 0.0f;
        for (int x = 0; x < NSYM; x++) acc = acc + p[x];
        return (double)acc;
    }
}

static double col_sum(const orc_t *o, const void *m, int b, int i, int j)
{
    if (!in_band(o, i, j)) return 0.0;
    if (o->storage) {
        const double *p = (const double *)m + idx(o, 0, b, i, j);
        double acc = 0.0;
        for (int x = 0; x < NSYM; x++) acc = acc + p[x * NSYM];
        return acc;
    } else {
        const float *p = (const float *)m + idx(o, 0, b, i, j);
        float acc = 0.0f;
        for (int x = 0; x < NSYM; x++) acc = acc + p[x * NSYM];
        return (double)acc;
    }
}

orc_t *orc_create(int n, int band, int storage, int cond_mode, int marginal_term, int use_libm)
{
    orc_t *o = (orc_t *)calloc(1, sizeof(orc_t));
    if (!o) return NULL;
    if (band < 1) band = 1;
    o->n = n; o->band = band; o->storage = storage; o->cond_mode = cond_mode;
    o->marginal_term = marginal_term; o->use_libm = use_libm; o->L = 1;
    for (int q = 0; q < 5; q++) o->order[q] = VALID[q];
    size_t bytes = cells(o) * (storage ? 8 : 4);
    o->h = calloc(1, bytes);
    if (!o->h) { free(o); return NULL; }
    return o;
}

void orc_destroy(orc_t *o)
{
    if (!o) return;
    free(o->h); free(o->h0); free(o->audit); free(o);
}

int orc_audit_begin(orc_t *o)
{
    if (!o->audit) o->audit = (orc_audit_t *)malloc(sizeof(orc_audit_t));
    if (!o->audit) return -1;
    memset(o->audit, 0, sizeof(orc_audit_t));
    o->audit->first_flip_path = o->audit->first_flip_snp = -1;
    o->audit->min_margin_ulps = INFINITY;
    return 0;
}

int orc_audit_get(const orc_t *o, orc_audit_t *out)
{
    if (!o->audit) return -1;
    *out = *o->audit;
    return 0;
}

void orc_set_L(orc_t *o, int L) { o->L = L; }
int orc_get_L(const orc_t *o) { return o->L; }
void orc_set_full_enum(orc_t *o, int v) { o->full_enum = v; }
/* spec.cand_order (a permutation of the five valid symbol indices 0 1 2 3 5) and spec.offer_zero */
int orc_set_candidates(orc_t *o, const int32_t order[5], int offer_zero)
{
    int seen = 0;
    for (int q = 0; q < 5; q++) {
        if (order[q] < 0 || order[q] > 5 || order[q] == SYM_N) return -1;
        seen |= 1 << order[q];
    }
    if (seen != 0x2F) return -1;
    for (int q = 0; q < 5; q++) o->order[q] = order[q];
    o->offer_zero = offer_zero;
    return 0;
}
int64_t orc_reweight_calls(const orc_t *o) { return o->reweight_calls; }
void orc_get_stats(const orc_t *o, int64_t out[3])
{
    out[0] = o->n_slices; out[1] = o->n_crumbs; out[2] = o->covered;
}

/* hansel.copy() -- gretel/cmd.py:79 */
int orc_snapshot_original(orc_t *o)
{
    size_t bytes = cells(o) * (o->storage ? 8 : 4);
    if (!o->h0) o->h0 = malloc(bytes);
    if (!o->h0) return -1;
    memcpy(o->h0, o->h, bytes);
    return 0;
}

/* hansel.add_observation: storage-dtype += 1 */
int orc_add(orc_t *o, int a, int b, int i, int j)
{
    if (!in_band(o, i, j)) return -1;
    if (o->storage) ((double *)o->h)[idx(o, a, b, i, j)] += 1.0;
    else ((float *)o->h)[idx(o, a, b, i, j)] += 1.0f;
    return 0;
}

double orc_get(const orc_t *o, int a, int b, int i, int j) { return getm(o, o->h, a, b, i, j); }

/* hansel.reweight_observation (call sites gretel/gretel.py:84,96) */
double orc_reweight_obs(orc_t *o, int a, int b, int i, int j, double ratio)
{
    o->reweight_calls++;
    double old = getm(o, o->h, a, b, i, j);
    double nw = old - ratio * old;
    setm(o, o->h, a, b, i, j, nw);
    return old - nw;
}

/*
 * gretel/util.py:226-286 over a support table:
 *   rank[r]                      -- util.py:198 (SNPs left of the read's first SNP)
 *   bases[off[r] .. off[r+1])    -- util.py:238 support_seq, ASCII
 */
int orc_fill(orc_t *o, const int32_t *rank, const int64_t *off, const uint8_t *bases,
             int64_t n_reads, int use_end_sentinels)
{
    int N = o->n;
    int64_t slices = 0, crumbs = 0, covered = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        const uint8_t *s = bases + off[r];
        int k = (int)(off[r + 1] - off[r]);
        int rk = rank[r];
        if (!(k > 1)) continue;                                /* util.py:230 */
        slices++;                                              /* util.py:233 */
        for (int i = 0; i < k; i++)                            /* util.py:239 */
            if (s[i] != 'N' && s[i] != '_') covered++;
        for (int i = 0; i < k; i++) {
            int a = sym_of_char(s[i]);
            if (a < 0) return -2;
            for (int j = i + 1; j < k; j++) {
                int b = sym_of_char(s[j]);
                if (b < 0) return -2;
                if (a == SYM_US || a == SYM_N) continue;       /* util.py:258 */
                int rc = 0;
                if (i == 0 && j == 1 && rk == 0) {             /* util.py:262 */
                    rc |= orc_add(o, SYM_US, a, 0, 1);
                    rc |= orc_add(o, a, b, 1, 2);
                    crumbs++;
                } else if ((j + rk + 1) == N && (j - i) == 1) { /* util.py:271 */
                    rc |= orc_add(o, a, b, N - 1, N);
                    rc |= orc_add(o, b, SYM_US, N, N + 1);
                    crumbs++;
                } else {                                       /* util.py:279 */
                    rc |= orc_add(o, a, b, i + rk + 1, j + rk + 1);
                    crumbs++;
                    if (use_end_sentinels && j == k - 1 && (j - i) == 1)   /* util.py:283 */
                        rc |= orc_add(o, b, SYM_US, j + rk + 1, j + rk + 2);
                }
                if (rc) return -1;                             /* outside the band */
            }
        }
    }
    o->n_slices += slices; o->n_crumbs += crumbs; o->covered += covered;
    if (o->n_slices > 0)                                       /* util.py:333 */
        o->L = (int)ceil((double)o->covered / (double)o->n_slices);
    return 0;
}

/* out[0..6] = c_s(p), out[7] = total */
static void counts_m(const orc_t *o, const void *m, int p, double out[8])
{
    double tot = 0.0;
    for (int s = 0; s < NSYM; s++) {
        double c = row_sum(o, m, s, p, p + 1);
        out[s] = c;
        if (c > 0) tot += c;
    }
    out[7] = tot;
}

void orc_counts_at(const orc_t *o, int p, double out[8]) { counts_m(o, o->h, p, out); }

static double marginal_m(const orc_t *o, const void *m, int s, int p)
{
    double c[8];
    counts_m(o, m, p, c);
    if (!(c[s] > 0) || c[7] == 0.0) return 0.0;
    return c[s] / c[7];
}

double orc_marginal(const orc_t *o, int s, int p) { return marginal_m(o, o->h, s, p); }

static int n_valid_at(const orc_t *o, int p)
{
    double c[8];
    counts_m(o, o->h, p, c);
    int v = 0;
    for (int q = 0; q < 5; q++) if (c[VALID[q]] > 0) v++;
    return v;
}

double orc_conditional(const orc_t *o, int a, int b, int i, int j)
{
    double obs = getm(o, o->h, a, b, i, j);
    double den;
    if (o->cond_mode == 0) den = (double)n_valid_at(o, j) + row_sum(o, o->h, a, i, j);
    else if (o->cond_mode == 1) den = (double)n_valid_at(o, i) + row_sum(o, o->h, a, i, i + 1);
    else if (o->cond_mode == 2) den = (double)n_valid_at(o, i) + col_sum(o, o->h, b, i, j);
    else if (o->cond_mode == 4) den = (double)n_valid_at(o, j) + col_sum(o, o->h, b, i, j);   /* E: V(pos_to) + column sum */
    else den = (double)n_valid_at(o, i) + row_sum(o, o->h, a, i, j);          /* D: V(pos_from) + row sum */
    return (1.0 + obs) / den;
}

/* w[s] for s in 0..6; returns bitmask of candidates (valid symbols with c_s(p) > 0; every valid symbol with offer_zero) */
int orc_edge_weights(const orc_t *o, int p, const uint8_t *path, double w[NSYM])
{
    double c[8];
    counts_m(o, o->h, p, c);
    int mask = 0;
    int lmax = o->L < p ? o->L : p;
    for (int q = 0; q < 5; q++) {
        int b = VALID[q];
        w[b] = 0.0;
        if (!o->offer_zero && !(c[b] > 0)) continue;
        mask |= 1 << b;
        double acc = 0.0;
        if (o->marginal_term) acc += lg(o, (c[b] > 0 && c[7] != 0.0) ? c[b] / c[7] : 0.0);
        for (int l = 1; l <= lmax; l++)
            acc += lg(o, orc_conditional(o, path[p - l], b, p - l, p));
        w[b] = acc;
    }
    w[SYM_N] = 0.0; w[SYM_US] = 0.0;
    return mask;
}

static double lg_with(double x, int use_libm)
{
    if (use_libm) return x == 0.0 ? -INFINITY : log10(x);
    return gh_log10(x);
}

/* gretel.py:166-174 over weights w[] */
static int pick_first_wins(const orc_t *o, int mask, const double w[NSYM])
{
    int next_m = -1;
    double next_v = 0.0;
    for (int q = 0; q < 5; q++) {
        int b = o->order[q];
        if (!(mask & (1 << b))) continue;
        if (next_m < 0) { next_v = w[b]; next_m = b; }
        else if (w[b] > next_v) { next_v = w[b]; next_m = b; }
    }
    return next_m;
}

/* one step of the audit: the edge weights of orc_edge_weights() once more with the OTHER log10 (same conditionals, same
 * order of additions), the two picks compared, the margin of the handle's own weights binned */
static void audit_step(orc_t *o, int p, const uint8_t *path, int mask, const double w[NSYM], int pick, int path_no)
{
    orc_audit_t *a = o->audit;
    int ncand = 0;
    for (int q = 0; q < 5; q++) if (mask & (1 << VALID[q])) ncand++;
    if (ncand < 2) return;
    a->steps++;
    double c[8], w2[NSYM];
    counts_m(o, o->h, p, c);
    int lmax = o->L < p ? o->L : p;
    int other = !o->use_libm, any_nan = 0;
    for (int q = 0; q < 5; q++) {
        int b = VALID[q];
        w2[b] = 0.0;
        if (!(mask & (1 << b))) continue;
        double acc = 0.0;
        if (o->marginal_term) acc += lg_with((c[b] > 0 && c[7] != 0.0) ? c[b] / c[7] : 0.0, other);
        for (int l = 1; l <= lmax; l++)
            acc += lg_with(orc_conditional(o, path[p - l], b, p - l, p), other);
        w2[b] = acc;
        if (w[b] != w[b] || w2[b] != w2[b]) any_nan = 1;
        else if (isfinite(w[b]) && isfinite(w2[b])) {
            double d = fabs(w[b] - w2[b]);
            if (d > a->max_abs_dw) a->max_abs_dw = d;
        }
    }
    if (any_nan) a->nan_steps++;
    int pick2 = pick_first_wins(o, mask, w2);
    if (pick2 != pick) {
        a->flips++;
        if (a->first_flip_path < 0) { a->first_flip_path = path_no; a->first_flip_snp = p; }
        /* an exact tie under one log only? */
        double b1 = w[pick], b2 = w2[pick2];
        if (w[pick2] == b1 || w2[pick] == b2) a->order_diffs++;
    }
    /* margin of the handle's own weights */
    double best = w[pick], second = -INFINITY;
    for (int q = 0; q < 5; q++) {
        int b = VALID[q];
        if (!(mask & (1 << b)) || b == pick) continue;
        if (w[b] > second) second = w[b];      /* a NaN never compares greater: it is no runner-up */
    }
    if (best != best || second != second || !isfinite(best)) return;
    double m = best - second;            /* >= 0 */
    double ulp = nextafter(fabs(best), INFINITY) - fabs(best);
    double mu = isfinite(second) ? m / ulp : INFINITY;
    int bin;
    if (mu == 0.0) bin = 0;
    else if (mu < 4) bin = 1; else if (mu < 16) bin = 2; else if (mu < 64) bin = 3; else if (mu < 256) bin = 4;
    else if (mu < 1024) bin = 5; else if (mu < 1048576.0) bin = 6; else bin = 7;
    a->margin_bins[bin]++;
    if (mu > 0.0 && mu < a->min_margin_ulps) a->min_margin_ulps = mu;
}

/* gretel/gretel.py:102-189.  path has n+1 entries (symbol indices), path[0] = '_'.
 * Returns 0, or the SNP (>=1) at which no branch could be selected. */
int orc_generate_path(orc_t *o, uint8_t *path, double *hp_cur, double *hp_orig, double *min_marg)
{
    double running = 0.0, running_uw = 0.0, mn = INFINITY;
    double running2 = 0.0, running_uw2 = 0.0;         /* audit: the same sums under the other log */
    const void *m0 = o->h0 ? o->h0 : o->h;
    path[0] = SYM_US;
    for (int snp = 1; snp <= o->n; snp++) {
        double w[NSYM];
        int mask = orc_edge_weights(o, snp, path, w);
        int next_m = -1;
        double next_v = 0.0;
        for (int q = 0; q < 5; q++) {                 /* gretel.py:166-174, keys in the order the dict was filled */
            int b = o->order[q];
            if (!(mask & (1 << b))) continue;
            if (next_m < 0) { next_v = w[b]; next_m = b; }
            else if (w[b] > next_v) { next_v = w[b]; next_m = b; }
        }
        if (next_m < 0) return snp;                   /* gretel.py:176-180 */
        double m = marginal_m(o, o->h, next_m, snp);  /* gretel.py:182 */
        if (m < mn) mn = m;
        running += lg(o, m);                          /* gretel.py:185 */
        running_uw += lg(o, marginal_m(o, m0, next_m, snp));   /* gretel.py:186 */
        if (o->audit) {
            audit_step(o, snp, path, mask, w, next_m, (int)o->audit->paths);
            running2 += lg_with(m, !o->use_libm);
            running_uw2 += lg_with(marginal_m(o, m0, next_m, snp), !o->use_libm);
        }
        path[snp] = (uint8_t)next_m;
    }
    if (o->audit) {
        orc_audit_t *a = o->audit;
        double d1 = fabs(running - running2), d2 = fabs(running_uw - running_uw2);
        if (d1 > a->max_abs_dhp_cur) a->max_abs_dhp_cur = d1;
        if (d2 > a->max_abs_dhp_orig) a->max_abs_dhp_orig = d2;
        a->paths++;
    }
    *hp_cur = running; *hp_orig = running_uw; *min_marg = mn;
    return 0;
}

/* gretel/gretel.py:79-98 */
double orc_reweight_path(orc_t *o, const uint8_t *path, double ratio)
{
    double size = 0;
    int len = o->n + 1;
    for (int i = 0; i < len; i++) {
        if (i >= len - 1) {                           /* gretel.py:83-85 (j == 0, then break) */
            size += orc_reweight_obs(o, path[i], path[0], i, i + 1, ratio);
            continue;
        }
        int j0 = 0;
        if (!o->full_enum) { j0 = i - o->band; if (j0 < 0) j0 = 0; }
        for (int j = j0; j <= i + 1; j++) {
            int ti = j < i ? j : i;
            int tj = j < i ? i : j;
            size += orc_reweight_obs(o, path[ti], path[tj], ti, tj, ratio);
        }
    }
    return size;
}

/* gretel/cmd.py:148-179 without the dedupe table (host-side bookkeeping).
 * paths: [max_paths][n+1]; returns number of completed paths; *hole_at = SNP of
 * the hole that ended recovery (0 if max_paths reached). */
int orc_spin(orc_t *o, int max_paths, uint8_t *paths, double *hp_cur, double *hp_orig,
             double *ratio, double *magnitude, int *hole_at)
{
    int done = 0;
    *hole_at = 0;
    if (!o->h0 && orc_snapshot_original(o)) return -1;
    for (int s = 0; s < max_paths; s++) {
        uint8_t *p = paths + (size_t)s * (o->n + 1);
        double hc, ho, mn;
        int hole = orc_generate_path(o, p, &hc, &ho, &mn);
        if (hole) { *hole_at = hole; break; }          /* cmd.py:153 */
        if (mn < MIN_REMOVE) mn = MIN_REMOVE;          /* cmd.py:158-160 */
        double mag = orc_reweight_path(o, p, mn);      /* cmd.py:161 */
        hp_cur[s] = hc; hp_orig[s] = ho; ratio[s] = mn; magnitude[s] = mag;
        done++;
    }
    return done;
}

/* cmd.py:85-92: first position in [0,N] with total == 0, else -1 */
int orc_gap_check(const orc_t *o)
{
    double c[8];
    for (int i = 0; i <= o->n; i++) {
        counts_m(o, o->h, i, c);
        if (c[7] == 0.0) return i;
    }
    return -1;
}

/* export the band as doubles: out[(n+2)*band*49] */
void orc_export_band(const orc_t *o, double *out)
{
    size_t n = cells(o);
    for (size_t q = 0; q < n; q++)
        out[q] = o->storage ? ((const double *)o->h)[q] : (double)((const float *)o->h)[q];
}

double orc_log10(double x, int use_libm)
{
    orc_t t; memset(&t, 0, sizeof t); t.use_libm = use_libm;
    return lg(&t, x);
}

void orc_log10_many(const double *x, double *y, int64_t n, int use_libm)
{
    orc_t t; memset(&t, 0, sizeof t); t.use_libm = use_libm;
    for (int64_t q = 0; q < n; q++) y[q] = lg(&t, x[q]);
}

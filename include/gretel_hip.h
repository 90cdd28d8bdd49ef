/*
 * gretel_hip.h -- C ABI of libgretel_hip.so, the MI355X (gfx950) implementation of
 * Gretel's hot path: the Hansel SNP-pair co-observation tensor and the three
 * loops that run over it (BAM->Hansel fill, L'th-order Markov path extension,
 * per-path reweighting).
 *
 * The reference has no FFI for this path: its seam is the Python object protocol
 * of `hansel.Hansel` (third-party hanselx==0.0.92, reference setup.py:8) plus
 * two functions of gretel/gretel.py.  Each entry point below names the
 * reference interface it replaces (file:line relative to the reference tree);
 * gretel_amd/hansel.py, gretel_amd/gretel.py and gretel_amd/util.py bind them
 * with ctypes and re-expose the reference's Python names.
 *
 * Conventions
 *   - every function returns GH_OK (0) or a negative gh_status; the message of
 *     the last failure on the calling thread is gh_last_error().
 *   - "no branch at SNP k" (reference gretel/gretel.py:176-180 returns a None
 *     triple) is a VALUE (hole_at >= 1), not an error.
 *   - symbols are indices in the reference's order (gretel/util.py:83):
 *       A=0 C=1 G=2 T=3 N=4 -=5 _=6 ; unsymbols are N and _.
 *   - positions are 0..n_snps+1 (0 and n_snps+1 are the sentinels).
 *   - a handle owns its device buffers and one HIP stream; it is not
 *     thread-safe; one handle per (contig,start,end) window.
 *   - all output buffers are caller-allocated host memory unless stated.
 */
#ifndef GRETEL_HIP_H
#define GRETEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gh_handle gh_t;          /* one Hansel tensor resident in HBM */
typedef struct gh_reads gh_reads_t;     /* one support table resident in HBM */

typedef enum {
    GH_OK = 0,
    GH_ERR_ARG = -1,        /* bad argument */
    GH_ERR_HIP = -2,        /* HIP runtime failure (no device, launch error, ...) */
    GH_ERR_BAND = -3,       /* observation with pos_to-pos_from outside [1, band] */
    GH_ERR_SYMBOL = -4,     /* byte that is not one of "ACGTN-_" in a support_seq */
    GH_ERR_NOMEM = -5,
    GH_ERR_STATE = -6       /* call order (e.g. generate_path before any fill) */
} gh_status;

#define GH_NSYM 7
#define GH_STORAGE_F32 0
#define GH_STORAGE_F64 1
#define GH_COND_A 0   /* (1+H[a,b,i,j]) / (V(j) + sum_x H[a,x,i,j])   frozen default */
#define GH_COND_B 1   /* (1+H[a,b,i,j]) / (V(i) + c_a(i))                            */
#define GH_COND_C 2   /* (1+H[a,b,i,j]) / (V(i) + sum_x H[x,b,i,j])                  */
#define GH_COND_D 3   /* (1+H[a,b,i,j]) / (V(i) + sum_x H[a,x,i,j])   the reading of gretel.py:10's TODO with V at pos_from */
#define GH_COND_E 4   /* (1+H[a,b,i,j]) / (V(j) + sum_x H[x,b,i,j])   C with the "unique variants" term at pos_to       */
/* (C and E read a cell as P(a at i | b at j): the earlier variant given the candidate -- the naive-Bayes form of the
 * published method, reference README.md:79-94; A and D read it as P(b at j | a at i); B conditions on c_a(i).) */

typedef struct {
    int32_t n_snps;         /* N: number of SNPs of the window (VCF_h["N"], gretel/util.py:409) */
    int32_t band;           /* W: largest pos_to-pos_from stored (= max SNPs on a read - 1, >= 1) */
    int32_t storage;        /* GH_STORAGE_F32 | GH_STORAGE_F64 */
    int32_t cond_mode;      /* GH_COND_*  */
    int32_t marginal_term;  /* 1: edge weight also adds log10(marginal) (App. A-7) */
    int32_t device;         /* HIP device ordinal, -1 = current */
    int32_t offer_zero;     /* 1: get_edge_weights_at offers every valid symbol, also those with a zero count at the position (App. A-4/5) */
    uint8_t cand_order[8];  /* [0..4]: the order the candidates are offered in = dict insertion order of get_edge_weights_at =
                             * the tie-break of gretel/gretel.py:166-174 (first key wins); a permutation of the valid symbol
                             * indices {0,1,2,3,5}.  All zero = the default A C G T - (symbol index order). */
} gh_config;

typedef struct {
    int64_t n_slices;       /* reads with >1 SNP            gretel/util.py:233,329 */
    int64_t n_crumbs;       /* pair observations            gretel/util.py:268,276,281,330 */
    int64_t covered_snps;   /* informative SNPs on them     gretel/util.py:239 */
    int32_t L;              /* ceil(covered/slices)         gretel/util.py:333 */
    int32_t _pad;
} gh_fill_stats;

typedef struct {
    double hp_current;      /* gretel/gretel.py:185,189 */
    double hp_original;     /* gretel/gretel.py:186,189 */
    double ratio;           /* min marginal after the 1% clamp, gretel/cmd.py:157-160 */
    double magnitude;       /* reweight_hansel_from_path return, gretel/cmd.py:161 */
    double min_marginal;    /* generate_path's third return value before the clamp (the "%.10f too small" note, cmd.py:158-160) */
} gh_path_rec;

const char *gh_last_error(void);
int gh_device_count(int *n);
/* peak shader clock of a device in kHz (bench.py prices its instruction-issue floors with it) */
int gh_device_clock_khz(int device, int *khz);

/* log10 as every kernel evaluates it: include/gh_detlog.h, glibc's log10 restated so that the arg-max of
 * gretel/gretel.py:159-174 and the sums of gretel/gretel.py:185-186 see the very doubles math.log10 (gretel/gretel.py:2)
 * gives the reference.  gh_log10_device runs it on the GPU over an array, gh_log10_host is the same source compiled
 * for the host (touches no device): the tests hold both to the running libm. */
int gh_log10_device(int device, const double *x, double *y, int64_t n);
int gh_log10_host(const double *x, double *y, int64_t n);

/* Hansel.init_matrix(['A','C','G','T','N','-','_'], ['N','_'], N) -- gretel/util.py:83.
 * Allocates the zeroed banded tensor [(N+2)][band][7][7] in HBM. */
int gh_create(const gh_config *cfg, gh_t **out);
int gh_destroy(gh_t *h);
/* hansel.copy() -- gretel/cmd.py:79 (deep copy incl. L, n_slices, n_crumbs). */
int gh_copy(const gh_t *src, gh_t **out);
/* zero the tensor and the fill counters (re-use a handle for another window of the same shape) */
int gh_clear(gh_t *h);
int gh_sync(gh_t *h);

/* hansel.L / n_slices / n_crumbs attributes -- gretel/util.py:329-333, gretel/cmd.py:227-229 */
int gh_set_L(gh_t *h, int32_t L);
int gh_get_L(const gh_t *h, int32_t *L);
int gh_get_fill_stats(const gh_t *h, gh_fill_stats *out);
int gh_set_fill_stats(gh_t *h, const gh_fill_stats *in);

/* Support table (per read: rank of gretel/util.py:198, support_seq of util.py:238 as ASCII)
 * copied to HBM once; off has n_reads+1 entries.  The arrays may lie in page-locked memory (gh_host_alloc): they are then read
 * by DMA instead of through a staging copy.  GH_ERR_ARG when off[] runs backwards somewhere. */
int gh_reads_upload(const gh_t *h, const int32_t *rank, const int64_t *off, const uint8_t *bases,
                    int64_t n_reads, gh_reads_t **out);
int gh_reads_free(gh_reads_t *r);
/* the longest read of an uploaded table (max of off[q + 1] - off[q]; found on the device behind the upload): a matrix needs a band
 * of at least max_k - 1 to take it */
int gh_reads_max_k(const gh_reads_t *r, int32_t *max_k);
/* What the upload found out about the table (for the tests; the fills choose their kernel and their counter width by it):
 * info[0] longest read, [1] 1 when the ranks ascend, [2] the widest run of positions a block of 2048 reads covers + longest read + 1,
 * [3] the most reads whose ranks fall into 128 consecutive positions, [4] entries of first_at (0: none); the last three are 0 for
 * a table whose ranks do not ascend.  first_at (may be NULL): room for info[4] entries, first_at[p] = the first read whose
 * rank is >= p -- call once with NULL for the count. */
int gh_reads_info(const gh_reads_t *r, int64_t info[5], int64_t *first_at);

/* The per-read pair loop of load_from_bam -- gretel/util.py:226-286 -- and the
 * counters/L of util.py:329-333, over a device-resident support table.
 * Accumulates into the tensor (call gh_clear first for a fresh fill). */
int gh_fill(gh_t *h, const gh_reads_t *reads, int use_end_sentinels, gh_fill_stats *out);

/* hansel.add_observation / get_observation / reweight_observation --
 * gretel/util.py:266-286, tests/test_test.py:41-52, gretel/gretel.py:84,96.
 * One-cell compatibility API (a host round trip each): not the fast path. */
int gh_add(gh_t *h, int a, int b, int i, int j);
int gh_add_batch(gh_t *h, const uint8_t *a, const uint8_t *b, const int32_t *i, const int32_t *j, int64_t n);
int gh_get(gh_t *h, int a, int b, int i, int j, double *out);
int gh_reweight_obs(gh_t *h, int a, int b, int i, int j, double ratio, double *removed);

/* hansel.get_counts_at(p) -- gretel/cmd.py:86,127: out[s] = c_s(p), out[7] = total. */
int gh_counts_at(gh_t *h, int p, double out[8]);
/* hansel.get_marginal_of_at(s, p) -- gretel/gretel.py:182,186 */
int gh_marginal_of_at(gh_t *h, int s, int p, double *out);
/* hansel.get_edge_weights_at(p, current_path) -- gretel/gretel.py:155.
 * path[0..p-1] are the selected symbol indices (path[0] = '_'); w[s] is set for
 * the candidate symbols, *cand_mask has bit s set for each candidate. */
int gh_edge_weights_at(gh_t *h, int p, const uint8_t *path, double w[GH_NSYM], int *cand_mask);
/* the gap check of gretel/cmd.py:85-118: first position in [0,N] whose total is 0, else -1 */
int gh_gap_check(gh_t *h, int *first_gap);

/* candidate bitmask per position (bit s set <=> valid symbol s has c_s(p) > 0), out[0..N] */
int gh_export_cmask(gh_t *h, uint32_t *out);

/* Freeze the current marginals as the "original" ones: what gretel/cmd.py:79's
 * hansel.copy() is used for at gretel/gretel.py:186 (M0[p][s] replaces a second tensor). */
int gh_snapshot_original(gh_t *h);

/* gretel.generate_path(n_snps, hansel, original_hansel) -- gretel/gretel.py:102-189.
 * `original` may be NULL (then h's snapshot, or h itself if none was taken).
 * path_out: N+1 symbol indices, path_out[0] = '_'.  *hole_at = 0 on success, else the
 * SNP at which no branch could be selected (gretel.py:176-180); then path_out[0..hole_at-1]
 * holds the prefix walked so far and the three doubles are not written. */
int gh_generate_path(gh_t *h, const gh_t *original, uint8_t *path_out,
                     double *hp_current, double *hp_original, double *min_marginal, int *hole_at);

/* gretel.reweight_hansel_from_path(hansel, path, ratio) -- gretel/gretel.py:79-98. */
int gh_reweight_path(gh_t *h, const uint8_t *path, double ratio, double *removed);

/* The spin loop of gretel/cmd.py:148-179 without host round trips: up to max_paths x
 * { generate_path, clamp ratio to >= min_remove (cmd.py:157-160), reweight }.
 * Stops at the first hole (cmd.py:153).  paths_out: [max_paths][N+1]. */
int gh_spin(gh_t *h, int max_paths, double min_remove, uint8_t *paths_out, gh_path_rec *recs,
            int *n_out, int *hole_at);

/* Batched recovery: the spin loop of gretel/cmd.py:148-179 over MANY windows of one shape (same n_snps, band, storage,
 * modes, device and L).  From two dozen windows on and at 2..10 lags -- under every conditional, with or without the marginal
 * term -- every window is carried through ALL its paths by one persistent workgroup (gretel_amd/csrc/wpipe.hpp):
 * the reweight of path s-1 (gretel/gretel.py:79-98) sweeps through the tensor a few chunks ahead of the walk of path s
 * (gretel/gretel.py:143-189), one launch per batch, 256 windows at a time on one MI355X.  Otherwise (and for windows the
 * pipeline cannot carry: a position with five candidates, a hole) every kernel of a path is launched over all windows --
 * one path-extension workgroup per window.  The handles stay usable on their own; fill them first.
 * paths_out: [n][max_paths][N+1], recs: [n][max_paths], n_out/hole_at: [n]. */
/* Page-locked host memory for result buffers (paths_out / recs of gh_spin and gh_batch_spin accept any host memory; into pinned
 * memory the copies run at the link's rate and without a staging pass -- 256 windows x 100 paths are 256 MB).  No reference
 * counterpart: the reference keeps its paths in Python lists (gretel/cmd.py:148-179). */
int gh_host_alloc(size_t bytes, void **out);
int gh_host_free(void *p);

typedef struct gh_batch gh_batch_t;
int gh_batch_create(gh_t **handles, int n, gh_batch_t **out);
int gh_batch_destroy(gh_batch_t *b);
int gh_batch_spin(gh_batch_t *b, int max_paths, double min_remove, uint8_t *paths_out, gh_path_rec *recs,
                  int *n_out, int *hole_at);
/* HIP-event timing of the batched kernels (bench.py's roofline of the throughput mode): every = k > 0 brackets the batched
 * launches of every k-th path of the FIRST window group on its stream (0 = off); the pipeline's one launch is bracketed as
 * a whole and reported under GH_K_WALK (bytes: extension + reweight of all its paths).  gh_batch_profile_get: kernel GH_K_WALK
 * (the batched serial extension) or GH_K_REWEIGHT (the fused reweight k_marg<T,true>) of the last gh_batch_spin --
 * milliseconds summed over the sampled launches, their number, the windows one such launch covers, and the algorithmic
 * bytes per launch (per window x windows; the definitions of DESIGN.md section 3). */
/* the last gh_batch_spin: out[0] = windows the pipeline carried, out[1] = of those, windows it handed back to gh_spin before
 * their last path (a candidate mask moved under a reweight), out[2] = threads per pipeline workgroup, out[3] = positions per
 * chunk of its walker tables (0, 0 when it did not run) */
int gh_batch_pipe_info(gh_batch_t *b, int32_t out[4]);
int gh_batch_profile_enable(gh_batch_t *b, int every);
int gh_batch_profile_get(gh_batch_t *b, int kernel, double *total_ms, int64_t *launches, int32_t *windows, double *bytes_per_launch);

/* tensor export/import for --dumpmatrix (gretel/cmd.py:81-82) and tests:
 * band layout [(N+2)][band][7][7] as doubles; dense layout [7][7][N+2][N+2] (gretel/cmd.py:76-77). */
int gh_export_band(gh_t *h, double *out);
int gh_import_band(gh_t *h, const double *in);
int gh_export_dense(gh_t *h, double *out);

/* gretel-snpper's site calling -- gretel/snpper.py:29-50 -- as a GPU histogram: the aligned runs of the BAM (gio_match_runs,
 * include/gretel_io.h: per run its 0-based reference start, bases as codes A0 C1 G2 T3, 4 = other) are counted per
 * position of the window [start0, start0+len) and base; site_out[p] = 1 where more than one base is seen on more than
 * `depth` reads (snpper.py:38-40).  counts_out (optional): int32[4][len], the array pysam's count_coverage returns. */
int gh_coverage_sites(int device, const int32_t *ref_start, const int64_t *off, const uint8_t *codes, int64_t n_runs,
                      int32_t start0, int32_t len, int32_t depth, int32_t *counts_out, uint8_t *site_out);

/* Per-kernel HIP-event timing on the handle's own stream (bench.py's roofline leg).
 * gh_profile_enable(h, k): k = 0 off; k >= 1 brackets every k-th launch of each kernel with a pair of events
 * (an event between two kernels costs the stream a ~10 us bubble, so a timed region samples with k > 1). */
/* GH_K_WALK: the whole path extension of one path (one serial walker launch, or k_seg + k_scan + k_emit);
 * GH_K_SEG: k_seg alone, the largest kernel of the segment-parallel extension (inside gh_spin: the first path only);
 * GH_K_RWSEG: k_rwseg alone -- the reweight of path k-1 and the k_seg of path k in one launch (every later path of a gh_spin) */
enum { GH_K_FILL = 0, GH_K_MARG = 1, GH_K_LT = 2, GH_K_WALK = 3, GH_K_REWEIGHT = 4, GH_K_SEG = 5, GH_K_RWSEG = 6, GH_K_COUNT = 7 };
int gh_profile_enable(gh_t *h, int on);
int gh_profile_reset(gh_t *h);
int gh_profile_get(gh_t *h, int kernel, double *total_ms, int64_t *launches);
/* what a bracket reads beyond its kernel: out[0] = two events back to back, out[1] = around an empty kernel (ms, medians) */
int gh_profile_overhead(gh_t *h, int reps, double out[2]);
/* diagnostics of the last path-extension launch: out[0] = shader cycles (s_memtime) the walker wave spent,
 * out[1] = the same interval in 100 MHz ticks (s_memrealtime), out[2] = steps it executed, out[3] = the variant
 * that ran (2 = depth-2 speculation, 1 = depth 1 without '-' candidates, 0 = depth 1 with them; 3 = segment-parallel
 * walk, which has no single walker wave: then out[0] = how often the last gh_spin rebuilt the conditional table and
 * queued its remaining paths again because a candidate mask moved, out[1] = the state space it enumerated (4: candidate
 * ranks of a window whose positions offer at most four; 5: all symbol histories; 6: mixed radix -- few positions offer five),
 * out[2] = the most states that enter a target as a mixed-radix number (0: not taken); 4 = candidate-pool segments
 * (L = 6..128): out[0] = re-queues of the last gh_spin, out[1] = paths this handle handed to the serial walker so far,
 * out[2] = walk/scan rounds it queued so far) */
int gh_debug_walk_clock(gh_t *h, uint64_t out[4]);
/* diagnostic builds only (-DRWS_STAMPS_ALL): the s_memtime stamps every k_rwseg workgroup of the last launch left at its phase
 * boundaries, 16 doubles per workgroup (scratch/wg_stamps.py) -- which workgroup a launch waits for, and in which phase */
int gh_debug_segment_stamps(gh_t *h, double *out, int n_workgroups);
/* how the candidate-pool extension (gretel/gretel.py:143-189 for 6 <= L <= 128) would cut a window of n_snps positions at lag count
 * L, over candidate ranks (five = 0) or over the symbols A C G T - (five = 1); no handle, no GPU: out[0] = segments, out[1] = positions
 * per segment, out[2] = targets per LDS chunk of the walker, out[3] = bytes of dynamic LDS it is launched with, out[4] = 1 if a state
 * is a packed 64-bit word (k_cwalk), 0 if bytes next to a hash (k_cwalkg), out[5] = threads per workgroup */
int gh_debug_pool_geometry(int32_t n_snps, int32_t L, int32_t five, int64_t out[6]);
/* algorithmic bytes of the last launch of each kernel */
int gh_profile_bytes(gh_t *h, int kernel, double *bytes_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* GRETEL_HIP_H */

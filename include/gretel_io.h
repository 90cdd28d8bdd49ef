/*
 * gretel_io.h -- C ABI of libgretel_io.so: native BAM decoding for the ingest half of
 * load_from_bam (reference gretel/util.py:120-209, which goes through pysam / htslib's pileup).
 *
 * It turns a coordinate-sorted BAM into the *support table* the GPU fill consumes
 * (include/gretel_hip.h, gh_reads_upload): per read the rank of util.py:198 and the
 * support_seq of util.py:238.  Host code only (zlib), no GPU.
 *
 * What it reproduces of the reference's per-read logic:
 *   read key  "<qname>_<flag>_<1|2|0>"                       util.py:149-160
 *   LEFTMOST / reads starting before start_pos               util.py:162-171
 *   deletion / ref-skip columns -> "-"                       util.py:180-182
 *   first base of every captured allele                      util.py:184-190,238
 *   rank = number of SNPs in [1, LEFTMOST)                   util.py:198
 *   stepper "samtools": UNMAP/SECONDARY/QCFAIL/DUP reads and paired-but-not-proper reads are
 *   dropped; stepper "all" (--pepper, cmd.py:39,78) keeps the latter.
 *   pysam's max_depth: bam.pileup keeps at most 8000 reads in its buffer by default and the reference passes no other value
 *   (util.py:137) -- on amplicon-depth data reads are silently dropped.  gio_support_table_from_bam applies that default;
 *   gio_support_table_from_bam_depth takes the cap as an argument (<= 0: none).  The rule is htslib's (bam_plp_push): a read is
 *   dropped when it starts at the position the pileup iterator stands on while the buffer holds more than max_depth nodes --
 *   so the first read of a position always enters, and a later one is dropped when the reads that entered and end behind
 *   position - 1, plus the list's sentinel node, number more than max_depth.  (One pileup over the whole window, as the
 *   reference runs with its default of one BAM worker; with -@ > 1 it runs one pileup per block, util.py:288-326.)
 *
 * The BAM is streamed (BGZF blocks inflated in parallel batches; libdeflate when its runtime library is present,
 * zlib otherwise) and, like the reference's pysam fetch, uses the index next to it when there is one (<bam>.bai or
 * <stem>.bai): the scan starts at the first block that can hold an alignment overlapping the window and ends at the
 * first record behind it.  Without an index every record of the file is visited.  Record fields are validated
 * against the record size (-4 on a malformed or truncated file); long-read CIGARs kept in the CG:B,I tag are resolved.
 * Environment: GIO_THREADS (decoder threads, default = cores, <= 24; an explicit value up to 64), GIO_NO_INDEX=1, GIO_ZLIB=1,
 * GIO_NO_READ_AHEAD=1 (gio_prefetch reads the file batch by batch, as a decode without it does, instead of in one piece on all threads).
 * The path may name a FIFO or another non-regular file: it is then read front to back with plain reads (no index).
 */
#ifndef GRETEL_IO_H
#define GRETEL_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t *rank;      /* [n_reads] */
    int64_t *off;       /* [n_reads + 1] */
    uint8_t *bases;     /* [n_bases] ASCII */
    int64_t n_reads;
    int64_t n_bases;
} gio_table;

typedef struct {
    int64_t compressed_bytes;   /* read from the BAM file */
    int64_t blocks;             /* BGZF blocks inflated */
    int64_t records;            /* BAM records parsed */
    int64_t reads_kept;         /* rows of the support table */
    int32_t used_index;         /* 1: started from a .bai offset and stopped behind the window */
    int32_t libdeflate;         /* 1: libdeflate, 0: zlib */
    int32_t threads;
    int32_t reframed;           /* batches done again front to back because a thread's guessed first record was none */
    double seconds;             /* wall time of the call */
    int64_t depth_dropped;      /* records the max_depth cap dropped */
    int32_t max_row_len;        /* characters of the longest row (max of off[q + 1] - off[q]): what sizes the matrix's band */
    int32_t prefetched;         /* 1: the call took over what gio_prefetch had read and inflated */
} gio_stats;

const char *gio_last_error(void);
/* what the calling thread's last gio_support_table_from_bam did */
void gio_last_stats(gio_stats *out);

/* bam.lengths[bam.get_tid(contig)] -- gretel/util.py:27-29 */
int gio_ref_len(const char *bam_path, const char *contig, int64_t *len);

/* region: uint8[end_pos + 1], region[p] != 0 <=> 1-based position p is a SNP (VCF_h["region"],
 * util.py:393-403).  Reads appear in the order the pileup first meets them (file order).
 * Returns 0 or a negative error code; the table is released with gio_table_free. */
#define GIO_PYSAM_MAX_DEPTH 8000
int gio_support_table_from_bam(const char *bam_path, const char *contig, int32_t start_pos, int32_t end_pos,
                               const uint8_t *region, int stepper_all, gio_table *out);
/* the same with the pileup's read-buffer cap as an argument (max_depth <= 0: no cap) */
int gio_support_table_from_bam_depth(const char *bam_path, const char *contig, int32_t start_pos, int32_t end_pos,
                                     const uint8_t *region, int stepper_all, int32_t max_depth, gio_table *out);
void gio_table_free(gio_table *t);
/* Everything of a decode that does not need the SNP positions -- header, index seek, the window's BGZF blocks read and inflated (up
 * to the inflated window's cap) -- started on a thread of the library; returns at once.  The reference's CLI parses the VCF and
 * then loads the BAM (gretel/cmd.py:69-78): a caller that knows the window before it has the SNPs calls this first and parses the
 * VCF meanwhile.  The next gio_support_table_from_bam* call for the same (path, contig, start_pos, end_pos) takes the work over
 * (gio_stats.prefetched); any other decode, another gio_prefetch or gio_prefetch_cancel discards it.  A prefetch that fails is
 * dropped silently: the decode then starts afresh and reports.  Returns 0, or -6 when no thread could be started. */
int gio_prefetch(const char *bam_path, const char *contig, int32_t start_pos, int32_t end_pos);
void gio_prefetch_cancel(void);
/* The same, the table's three arrays allocated by the CALLER: alloc(ctx, which, bytes) is called once each for which = 0 (rank),
 * 1 (off), 2 (bases) on the calling thread and returns the memory or NULL (-> -6) -- e.g. page-locked blocks (gh_host_alloc,
 * include/gretel_hip.h) kept from window to window, which gh_reads_upload then reads by DMA: no fresh pages for the decoder to
 * fault in, no staging copy for the upload.  Such a table is the caller's: gio_table_free must not be called on it. */
typedef void *(*gio_alloc_fn)(void *ctx, int which, size_t bytes);
int gio_support_table_from_bam_alloc(const char *bam_path, const char *contig, int32_t start_pos, int32_t end_pos,
                                     const uint8_t *region, int stepper_all, int32_t max_depth,
                                     gio_alloc_fn alloc, void *ctx, gio_table *out);
/* The decoder keeps its large working buffers (the inflated window, the reads' entries, keys and characters: about 130 bytes per
 * read) from one call to the next instead of unmapping them before it returns and faulting them in again -- a third of a decode's
 * time.  GIO_KEEP_MB in the environment bounds what is kept (default 512, 0: nothing); this frees it. */
void gio_release_buffers(void);

/* pysam's bam.count_coverage(contig, start0, stop, quality_threshold=0, read_callback='nofilter') as
 * gretel/snpper.py:29 uses it: counts[b][p - start0] = reads showing base b (A,C,G,T = 0..3) at 0-based
 * position p in [start0, stop); every record on the contig counts (no flag filter), deletions,
 * ref-skips, clips, insertions and N bases do not.  counts: int32[4][stop - start0], caller-allocated. */
int gio_count_coverage(const char *bam_path, const char *contig, int32_t start0, int32_t stop, int32_t *counts);

/* The same coverage as data for the GPU histogram (gh_coverage_sites, include/gretel_hip.h): the aligned (M/=/X) runs
 * of every record on the contig clipped to [start0, stop) -- per run its 0-based reference start and its bases as
 * codes A0 C1 G2 T3, 4 = anything else.  Released with gio_runs_free. */
typedef struct {
    int32_t *ref_start;  /* [n_runs] */
    int64_t *off;        /* [n_runs + 1]: codes[off[r] .. off[r+1]) are run r's bases */
    uint8_t *codes;      /* [n_bases] */
    int64_t n_runs;
    int64_t n_bases;
} gio_runs;
int gio_match_runs(const char *bam_path, const char *contig, int32_t start0, int32_t stop, gio_runs *out);
void gio_runs_free(gio_runs *r);

#ifdef __cplusplus
}
#endif
#endif

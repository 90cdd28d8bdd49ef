// bam_support.cpp -- libgretel_io.so: BGZF/BAM decoding and per-read SNP support extraction
// (C ABI: include/gretel_io.h).  Host-only C++ with zlib; replaces the pysam pileup of the
// reference (gretel/util.py:120-209) for the ingest half of load_from_bam.
#include <zlib.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "gretel_io.h"

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *gio_last_error(void) { return g_err; }

// BGZF is a series of gzip members: inflate them one after the other
static int read_bgzf(const char *path, std::vector<uint8_t> &out)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(-1, "cannot open %s", path);
    std::vector<uint8_t> in;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) in.insert(in.end(), buf, buf + n);
    fclose(fp);
    if (in.size() < 18 || in[0] != 0x1f || in[1] != 0x8b) return fail(-2, "%s is not gzip/BGZF", path);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 32) != Z_OK) return fail(-3, "inflateInit2 failed");
    zs.next_in = in.data();
    zs.avail_in = (uInt)in.size();
    out.clear();
    std::vector<uint8_t> chunk(1 << 18);
    for (;;) {
        zs.next_out = chunk.data();
        zs.avail_out = (uInt)chunk.size();
        int rc = inflate(&zs, Z_NO_FLUSH);
        out.insert(out.end(), chunk.data(), chunk.data() + (chunk.size() - zs.avail_out));
        if (rc == Z_STREAM_END) {
            if (zs.avail_in == 0) break;
            if (inflateReset(&zs) != Z_OK) { inflateEnd(&zs); return fail(-3, "inflateReset failed"); }
        } else if (rc != Z_OK) {
            inflateEnd(&zs);
            return fail(-3, "inflate failed (%d) in %s", rc, path);
        }
    }
    inflateEnd(&zs);
    return 0;
}

static inline int32_t rd32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t rdu32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint16_t rdu16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }

struct bam_file {
    std::vector<uint8_t> data;
    std::vector<std::pair<std::string, int64_t>> refs;
    size_t first_record;
};

static int open_bam(const char *path, bam_file &b)
{
    int rc = read_bgzf(path, b.data);
    if (rc) return rc;
    const std::vector<uint8_t> &d = b.data;
    if (d.size() < 12 || memcmp(d.data(), "BAM\1", 4) != 0) return fail(-4, "%s is not a BAM file", path);
    size_t o = 8 + (size_t)rd32(&d[4]);
    if (o + 4 > d.size()) return fail(-4, "truncated BAM header");
    const int n_ref = rd32(&d[o]);
    o += 4;
    for (int i = 0; i < n_ref; i++) {
        if (o + 4 > d.size()) return fail(-4, "truncated BAM header");
        const int l_name = rd32(&d[o]);
        o += 4;
        if (o + l_name + 4 > d.size()) return fail(-4, "truncated BAM header");
        std::string name((const char *)&d[o], l_name > 0 ? l_name - 1 : 0);
        o += l_name;
        b.refs.emplace_back(name, (int64_t)rd32(&d[o]));
        o += 4;
    }
    b.first_record = o;
    return 0;
}

extern "C" int gio_ref_len(const char *bam_path, const char *contig, int64_t *len)
{
    if (!bam_path || !contig || !len) return fail(-1, "null argument");
    bam_file b;
    int rc = open_bam(bam_path, b);
    if (rc) return rc;
    for (auto &r : b.refs)
        if (r.first == contig) { *len = r.second; return 0; }
    return fail(-5, "contig %s not in %s", contig, bam_path);
}

extern "C" void gio_table_free(gio_table *t)
{
    if (!t) return;
    free(t->rank); free(t->off); free(t->bases);
    memset(t, 0, sizeof *t);
}

extern "C" int gio_support_table_from_bam(const char *bam_path, const char *contig, int32_t start_pos, int32_t end_pos,
                                          const uint8_t *region, int stepper_all, gio_table *out)
{
    if (!bam_path || !contig || !region || !out || end_pos < 0) return fail(-1, "bad argument");
    memset(out, 0, sizeof *out);
    bam_file b;
    int rc = open_bam(bam_path, b);
    if (rc) return rc;
    int tid = -1;
    for (size_t i = 0; i < b.refs.size(); i++)
        if (b.refs[i].first == contig) tid = (int)i;
    if (tid < 0) return fail(-5, "contig %s not in %s", contig, bam_path);

    // csum[x] = number of SNPs in [0, x)
    std::vector<int32_t> csum((size_t)end_pos + 2, 0);
    for (int32_t p = 0; p <= end_pos; p++) csum[p + 1] = csum[p] + (region[p] ? 1 : 0);

    static const char SEQ[] = "=ACMGRSVTWYHKDBN";
    struct read_acc { int32_t rank; std::string seq; };
    std::vector<read_acc> reads;
    std::unordered_map<std::string, size_t> index;

    const std::vector<uint8_t> &d = b.data;
    size_t o = b.first_record;
    std::string chars, key;
    while (o + 4 <= d.size()) {
        const int32_t block_size = rd32(&d[o]);
        o += 4;
        if (block_size < 32 || o + (size_t)block_size > d.size()) return fail(-4, "truncated BAM record");
        const uint8_t *r = &d[o];
        o += block_size;
        const int32_t ref_id = rd32(r), pos = rd32(r + 4);
        const int l_read_name = r[8];
        const int n_cigar = rdu16(r + 12), flag = rdu16(r + 14);
        const int32_t l_seq = rd32(r + 16);
        if (ref_id != tid || (flag & (0x4 | 0x100 | 0x200 | 0x400))) continue;
        if (!stepper_all && (flag & 0x1) && !(flag & 0x2)) continue;      // orphans, stepper "samtools"
        if (l_seq == 0) continue;
        const char *name = (const char *)(r + 32);
        const uint8_t *cig = r + 32 + l_read_name;
        const uint8_t *seq = cig + 4 * (size_t)n_cigar;

        // walk the CIGAR the way htslib's pileup resolves it, column by column over the SNP positions
        chars.clear();
        int64_t ref = pos, q = 0, qalen = 0;
        const int64_t hi = end_pos;
        for (int c = 0; c < n_cigar; c++) {
            const uint32_t v = rdu32(cig + 4 * c);
            const int op = v & 15;
            const int64_t ln = v >> 4;
            if (op == 0 || op == 7 || op == 8) {                            // M = X
                int64_t lo1 = ref + 1 < 1 ? 1 : ref + 1, hi1 = ref + ln < hi ? ref + ln : hi;
                for (int64_t p1 = lo1; p1 <= hi1; p1++)
                    if (region[p1]) {
                        const int64_t qi = q + (p1 - 1 - ref);
                        const uint8_t byte = seq[qi >> 1];
                        chars.push_back(SEQ[(qi & 1) ? (byte & 15) : (byte >> 4)]);      // util.py:186-189, b[0]
                    }
                ref += ln; q += ln; qalen += ln;
            } else if (op == 2 || op == 3) {                                // D / N -> is_del column, util.py:180-182
                int64_t lo1 = ref + 1 < 1 ? 1 : ref + 1, hi1 = ref + ln < hi ? ref + ln : hi;
                for (int64_t p1 = lo1; p1 <= hi1; p1++)
                    if (region[p1]) chars.push_back('-');
                ref += ln;
            } else if (op == 1) { q += ln; qalen += ln; }                   // I
            else if (op == 4) { q += ln; }                                  // S
        }
        int64_t leftmost = (int64_t)pos + 1;                                // util.py:162
        if (leftmost < start_pos) {                                         // util.py:165-171
            if (leftmost + qalen < start_pos) continue;
            leftmost = start_pos;
        }
        if (chars.empty()) continue;
        int one_or_two = 0;
        if (flag & 0x1) one_or_two = (flag & 0x40) ? 1 : ((flag & 0x80) ? 2 : 0);
        key.assign(name);
        key += '_'; key += std::to_string(flag); key += '_'; key += std::to_string(one_or_two);   // util.py:160
        auto it = index.find(key);
        if (it == index.end()) {
            int64_t lm = leftmost > (int64_t)end_pos + 1 ? (int64_t)end_pos + 1 : leftmost;
            const int32_t rank = lm >= 1 ? csum[lm] - csum[1] : 0;          // util.py:198
            index.emplace(key, reads.size());
            reads.push_back({rank, chars});
        } else {
            reads[it->second].seq += chars;
        }
    }

    const int64_t n = (int64_t)reads.size();
    int64_t total = 0;
    for (auto &x : reads) total += (int64_t)x.seq.size();
    out->rank = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    out->off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    out->bases = (uint8_t *)malloc((size_t)(total ? total : 1));
    if (!out->rank || !out->off || !out->bases) { gio_table_free(out); return fail(-6, "out of memory"); }
    int64_t acc = 0;
    for (int64_t i = 0; i < n; i++) {
        out->rank[i] = reads[i].rank;
        out->off[i] = acc;
        memcpy(out->bases + acc, reads[i].seq.data(), reads[i].seq.size());
        acc += (int64_t)reads[i].seq.size();
    }
    out->off[n] = acc;
    out->n_reads = n;
    out->n_bases = total;
    return 0;
}

extern "C" int gio_count_coverage(const char *bam_path, const char *contig, int32_t start0, int32_t stop, int32_t *counts)
{
    if (!bam_path || !contig || !counts || start0 < 0 || stop < start0) return fail(-1, "bad argument");
    bam_file b;
    int rc = open_bam(bam_path, b);
    if (rc) return rc;
    int tid = -1;
    for (size_t i = 0; i < b.refs.size(); i++)
        if (b.refs[i].first == contig) tid = (int)i;
    if (tid < 0) return fail(-5, "contig %s not in %s", contig, bam_path);
    const int64_t len = (int64_t)stop - start0;
    memset(counts, 0, sizeof(int32_t) * 4 * (size_t)len);
    static const int8_t code2base[16] = {-1, 0, 1, -1, 2, -1, -1, -1, 3, -1, -1, -1, -1, -1, -1, -1};   // =ACMGRSVTWYHKDBN
    const std::vector<uint8_t> &d = b.data;
    size_t o = b.first_record;
    while (o + 4 <= d.size()) {
        const int32_t block_size = rd32(&d[o]);
        o += 4;
        if (block_size < 32 || o + (size_t)block_size > d.size()) return fail(-4, "truncated BAM record");
        const uint8_t *r = &d[o];
        o += block_size;
        if (rd32(r) != tid || (rdu16(r + 14) & 0x4)) continue;           // other contig / unmapped
        const int32_t pos = rd32(r + 4);
        const int n_cigar = rdu16(r + 12);
        const uint8_t *cig = r + 32 + r[8];
        const uint8_t *seq = cig + 4 * (size_t)n_cigar;
        if (rd32(r + 16) == 0) continue;
        int64_t ref = pos, q = 0;
        for (int c = 0; c < n_cigar; c++) {
            const uint32_t v = rdu32(cig + 4 * c);
            const int op = v & 15;
            const int64_t ln = v >> 4;
            if (op == 0 || op == 7 || op == 8) {
                int64_t lo = ref < start0 ? start0 : ref, hi = ref + ln < stop ? ref + ln : stop;
                for (int64_t p = lo; p < hi; p++) {
                    const int64_t qi = q + (p - ref);
                    const uint8_t byte = seq[qi >> 1];
                    const int base = code2base[(qi & 1) ? (byte & 15) : (byte >> 4)];
                    if (base >= 0) counts[(size_t)base * len + (p - start0)]++;
                }
                ref += ln; q += ln;
            } else if (op == 2 || op == 3) ref += ln;
            else if (op == 1 || op == 4) q += ln;
        }
    }
    return 0;
}

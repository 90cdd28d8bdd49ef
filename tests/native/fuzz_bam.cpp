// fuzz_bam.cpp -- malformed-BAM fuzz of the native decoder (gretel_amd/csrc/bam_support.cpp), built by
// tests/test_bam_fuzz.py with -fsanitize=address,undefined on the CPU: every mutated file must come back as a status
// code (0 or a negative error), never as an out-of-bounds access.
//   fuzz_bam <valid.bam> <contig> <end_pos> <iterations> <seed> <scratch.bam>
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "gretel_io.h"

static std::vector<uint8_t> slurp(const char *p)
{
    std::vector<uint8_t> d;
    FILE *f = fopen(p, "rb");
    if (!f) return d;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    return d;
}

static std::vector<uint8_t> gunzip_all(const std::vector<uint8_t> &in)
{
    std::vector<uint8_t> out;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    inflateInit2(&zs, 15 + 32);
    zs.next_in = const_cast<uint8_t *>(in.data());
    zs.avail_in = (uInt)in.size();
    std::vector<uint8_t> chunk(1 << 16);
    for (;;) {
        zs.next_out = chunk.data();
        zs.avail_out = (uInt)chunk.size();
        int rc = inflate(&zs, Z_NO_FLUSH);
        out.insert(out.end(), chunk.data(), chunk.data() + (chunk.size() - zs.avail_out));
        if (rc == Z_STREAM_END) {
            if (zs.avail_in == 0) break;
            inflateReset(&zs);
        } else if (rc != Z_OK) break;
    }
    inflateEnd(&zs);
    return out;
}

static void bgzf_write(const char *path, const std::vector<uint8_t> &data, size_t payload)
{
    FILE *f = fopen(path, "wb");
    for (size_t o = 0; o <= data.size(); o += payload) {
        const size_t n = o < data.size() ? std::min(payload, data.size() - o) : 0;
        std::vector<uint8_t> c(n + 1024);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = const_cast<uint8_t *>(data.data() + o);
        zs.avail_in = (uInt)n;
        zs.next_out = c.data();
        zs.avail_out = (uInt)c.size();
        deflate(&zs, Z_FINISH);
        const size_t cl = c.size() - zs.avail_out;
        deflateEnd(&zs);
        const uint16_t bsize = (uint16_t)(cl + 25);
        uint8_t head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 255), (uint8_t)(bsize >> 8)};
        fwrite(head, 1, 18, f);
        fwrite(c.data(), 1, cl, f);
        const uint32_t crc = (uint32_t)crc32(0, data.data() + o, (uInt)n), isz = (uint32_t)n;
        fwrite(&crc, 4, 1, f);
        fwrite(&isz, 4, 1, f);
        if (n == 0) break;
    }
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    const char *contig = argv[2];
    const int end_pos = atoi(argv[3]), iters = atoi(argv[4]);
    std::mt19937 rng((unsigned)atoi(argv[5]));
    const char *scratch = argv[6];
    const std::vector<uint8_t> raw = gunzip_all(slurp(argv[1]));
    if (raw.size() < 64) return 3;
    std::vector<uint8_t> region((size_t)end_pos + 1, 1);
    std::vector<int32_t> counts((size_t)4 * (end_pos + 1));
    int ok = 0, err = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> d = raw;
        const int kind = (int)(rng() % 5);
        if (kind == 0) d.resize(rng() % d.size());                                    // truncation
        else if (kind == 1) for (int q = 0; q < 1 + (int)(rng() % 8); q++) d[rng() % d.size()] = (uint8_t)rng();       // byte noise
        else if (kind == 2) {                                                          // a length field blown up
            size_t o = 12 + (rng() % (d.size() - 16));
            uint32_t v = (rng() % 2) ? 0x7fffffffu : (uint32_t)rng();
            memcpy(&d[o & ~(size_t)3], &v, 4);
        } else if (kind == 3) {                                                        // name without terminator / huge n_cigar
            for (size_t o = 0; o + 40 < d.size(); o++)
                if (d[o] == 'r' && rng() % 7 == 0) { d[o + 1 + rng() % 6] = 0xff; break; }
        } else d.insert(d.begin() + (ptrdiff_t)(rng() % d.size()), (size_t)(rng() % 64), (uint8_t)rng());   // shifted tail
        bgzf_write(scratch, d, 0xff00 >> (rng() % 6));
        gio_table t;
        const int rc = gio_support_table_from_bam(scratch, contig, 1, end_pos, region.data(), (int)(rng() % 2), &t);
        if (rc == 0) { ok++; gio_table_free(&t); } else err++;
        if (gio_count_coverage(scratch, contig, 0, end_pos, counts.data()) == 0) ok++; else err++;
        int64_t len;
        gio_ref_len(scratch, contig, &len);
    }
    // compressed-stream damage as well
    std::vector<uint8_t> comp = slurp(argv[1]);
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> d = comp;
        if (rng() % 2) d.resize(rng() % d.size());
        else for (int q = 0; q < 1 + (int)(rng() % 4); q++) d[rng() % d.size()] ^= (uint8_t)(1u << (rng() % 8));
        FILE *f = fopen(scratch, "wb");
        fwrite(d.data(), 1, d.size(), f);
        fclose(f);
        gio_table t;
        if (gio_support_table_from_bam(scratch, contig, 1, end_pos, region.data(), 0, &t) == 0) { ok++; gio_table_free(&t); } else err++;
    }
    printf("fuzz_bam done: %d ok, %d rejected\n", ok, err);
    return 0;
}

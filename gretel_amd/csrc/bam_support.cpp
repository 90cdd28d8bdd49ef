// bam_support.cpp -- libgretel_io.so: BGZF/BAM decoding and per-read SNP support extraction
// (C ABI: include/gretel_io.h).  Host-only C++; replaces the pysam pileup of the reference
// (gretel/util.py:120-209) for the ingest half of load_from_bam.
//
// The file is STREAMED: compressed bytes are read in batches and the BGZF blocks of a batch inflated in parallel
// (libdeflate when the runtime library is present, zlib otherwise) into a window of at most 32 MB.  Every thread then
// takes a byte range of the window, finds where a record plausibly starts in it, and frames and works on records from
// there (CIGAR walk from SNP to SNP, support characters, key and its hash: functions of the record alone); the chain of
// the thread in front must end exactly on that start, or the window is done again front to back.  The key table --
// first-seen order of the rows, appends to a key seen before -- is cut into partitions by hash bits, one thread each: a
// record is the first with its key or not whatever the other partitions hold, and file order lives in the records'
// numbers.  What a window cuts in two is carried over to the next.  GIO_TIMING=1 prints the time of each stage.
// With an index next to the file (<bam>.bai or <stem>.bai) the
// stream starts at the first block that can hold an alignment overlapping the window and stops at the first record
// behind it (coordinate-sorted input, as the reference's pysam fetch/pileup requires as well); without one the whole
// contig is scanned.  Every record field that is used as a length or an offset is checked against the record's size:
// malformed input is an error (-4), never an out-of-bounds read.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <string_view>
#include <thread>
#include <pthread.h>
#include <unordered_map>
#include <vector>

#include "gretel_io.h"

static thread_local char g_err[512] = "";
static thread_local gio_stats g_stats;

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *gio_last_error(void) { return g_err; }
extern "C" void gio_last_stats(gio_stats *out) { if (out) *out = g_stats; }

static inline int32_t rd32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t rdu32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint16_t rdu16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint64_t rdu64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

// ---------------------------------------------------------------------------------------------
// raw-deflate inflation of one BGZF block: libdeflate (dlopen of the runtime library; its four entry points are a
// stable ABI) or zlib
// ---------------------------------------------------------------------------------------------
struct deflate_api {
    void *(*alloc)(void) = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*free_)(void *) = nullptr;
    bool ok = false;
};

static const deflate_api &libdeflate()
{
    static deflate_api api = [] {
        deflate_api a;
        if (getenv("GIO_ZLIB") && atoi(getenv("GIO_ZLIB"))) return a;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.alloc = (void *(*)(void))dlsym(h, "libdeflate_alloc_decompressor");
        a.decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        a.free_ = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        a.ok = a.alloc && a.decompress && a.free_;
        return a;
    }();
    return api;
}

struct inflater {
    void *ld = nullptr;
    z_stream zs;
    bool z_init = false;
    inflater()
    {
        if (libdeflate().ok) ld = libdeflate().alloc();
        memset(&zs, 0, sizeof zs);
    }
    ~inflater()
    {
        if (ld) libdeflate().free_(ld);
        if (z_init) inflateEnd(&zs);
    }
    // raw deflate `in` -> exactly `out_len` bytes
    bool run(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
    {
        if (out_len == 0) return true;
        if (ld) {
            size_t got = 0;
            return libdeflate().decompress(ld, in, in_len, out, out_len, &got) == 0 && got == out_len;
        }
        if (!z_init) {
            if (inflateInit2(&zs, -15) != Z_OK) return false;
            z_init = true;
        } else if (inflateReset(&zs) != Z_OK) return false;
        zs.next_in = const_cast<uint8_t *>(in);
        zs.avail_in = (uInt)in_len;
        zs.next_out = out;
        zs.avail_out = (uInt)out_len;
        const int rc = inflate(&zs, Z_FINISH);
        return rc == Z_STREAM_END && zs.avail_out == 0;
    }
};

// ---------------------------------------------------------------------------------------------
// BGZF stream
// ---------------------------------------------------------------------------------------------
static int n_threads()
{
    static const int n = [] {
        const bool asked = getenv("GIO_THREADS") != nullptr;
        int t = asked ? atoi(getenv("GIO_THREADS")) : (int)std::thread::hardware_concurrency();
        if (t < 1) t = 1;
        // (measured on a 256-core host, C3 file: 8 threads 50 ms, 16 38 ms, 32 39 ms, 64 50 ms -- the serial parts bind; an explicit
        // GIO_THREADS may go to 64)
        // round 6, the pool and the kept buffers in place (C3 file, one MI355X host, ms inside the library: inflate / framing + records /
        // key table): 16 threads 4 / 5 / 7, 24 threads 3 / 4 / 5, 32 threads 3 / 3 / 8, 48 threads 2 / 3 / 11 -- the first two stages
        // scale, the key table's placing pass wants no more than 16 partitions
        if (t > (asked ? 64 : 24)) t = asked ? 64 : 24;
        return t;
    }();
    return n;
}

// The decoder's parallel regions -- inflate, frame + records, the key table's two passes, the table's two -- run on a pool of
// threads that is started once and kept: a C3-sized file goes through fourteen regions of up to sixteen threads, and creating and
// joining 200 threads was 3 ms of its 27.  run(n, fn) calls fn(0) .. fn(n - 1), each once, on up to n threads of which the caller is
// one, and returns when all have run; one region at a time (callers queue); the first exception of a task is rethrown in the caller.
// The pool is never destroyed (its threads are detached and sleep between regions).  After a fork the child has no workers: the
// caller then runs every task itself.
class worker_pool {
public:
    // (a forked child gets a NEW pool: none of the parent's worker threads exists there, and a mutex that another thread of the
    // parent held at the moment of the fork would stay locked for good -- the next decode in the child would never return, ADVICE r5.
    // The old object is leaked in the child, locked mutexes and all; workers are started again on demand.)
    static worker_pool *&slot() { static worker_pool *p = nullptr; return p; }
    static worker_pool &get()
    {
        static std::once_flag once;
        std::call_once(once, [] {
            slot() = new worker_pool();
            pthread_atfork(nullptr, nullptr, [] { slot() = new worker_pool(); });
        });
        return *slot();
    }
    template <typename F> void run(int n, F &&fn)
    {
        if (n <= 0) return;
        if (n == 1) { fn(0); return; }
        std::lock_guard<std::mutex> region(run_mu_);
        std::function<void(int)> f = [&fn](int t) { fn(t); };
        job j;
        j.fn = &f;
        j.n = n;
        j.left.store(n);
        {
            std::lock_guard<std::mutex> g(mu_);
            const int want = std::min(n - 1, n_threads() - 1);
            while (started_ < want) {
                try { std::thread([this] { loop(); }).detach(); } catch (...) { break; }
                started_++;
            }
            cur_ = &j;
            epoch_++;
        }
        cv_.notify_all();
        work_on(j);
        {
            std::unique_lock<std::mutex> lk(mu_);
            cur_ = nullptr;                                   // nobody picks it up from here on
            done_cv_.wait(lk, [&] { return j.left.load() == 0 && j.active == 0; });
        }
        if (j.err) std::rethrow_exception(j.err);
    }

private:
    struct job {
        std::function<void(int)> *fn = nullptr;
        int n = 0;
        std::atomic<int> next{0}, left{0};
        int active = 0;                                       // workers inside work_on (under mu_)
        std::exception_ptr err;
        std::mutex err_mu;
    };
    void work_on(job &j)
    {
        for (;;) {
            const int t = j.next.fetch_add(1);
            if (t >= j.n) break;
            try { (*j.fn)(t); }
            catch (...) {
                std::lock_guard<std::mutex> g(j.err_mu);
                if (!j.err) j.err = std::current_exception();
            }
            j.left.fetch_sub(1);
        }
    }
    void loop()
    {
        unsigned long seen = 0;
        for (;;) {
            job *j = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return epoch_ != seen; });
                seen = epoch_;
                j = cur_;
                if (j) j->active++;
            }
            if (!j) continue;
            work_on(*j);
            {
                std::lock_guard<std::mutex> g(mu_);
                j->active--;
            }
            done_cv_.notify_all();
        }
    }
    std::mutex run_mu_, mu_;
    std::condition_variable cv_, done_cv_;
    job *cur_ = nullptr;
    unsigned long epoch_ = 0;
    int started_ = 0;
};

// Large arrays on transparent huge pages: a million-read table touches ~200 MB of fresh memory, and at 4 KB a page the
// faults cost more than the decoding (madvise is a hint: where THP is off nothing changes).
// ... and they are KEPT from one decode to the next (up to GIO_KEEP_MB, default 512; 0: nothing is kept): handing 130 MB back to
// the system when a decode ends costs 8 ms of unmapping on the caller's clock (on a thread of its own it holds the address-space
// lock against the upload that follows, measured), and the next decode then faults the same pages in again -- a third of the time
// a C3-sized file takes from call to return, for a process that decodes window after window.  gio_release_buffers() frees what is
// kept.  Blocks are handed out again as they are (nobody here relies on fresh pages being zero: the key table, which does, maps its own).
struct big_cache {
    struct blk { void *p; size_t bytes; };
    std::mutex mu;
    std::vector<blk> free_blocks;
    size_t kept = 0;
    // (as the pool: a forked child starts with an empty cache of its own -- the parent's kept blocks stay mapped in the child and are
    // simply not used there; its mutex may have been held at the fork)
    static big_cache *&slot() { static big_cache *c = nullptr; return c; }
    static big_cache &get()
    {
        static std::once_flag once;
        std::call_once(once, [] {
            slot() = new big_cache();
            pthread_atfork(nullptr, nullptr, [] { slot() = new big_cache(); });
        });
        return *slot();
    }
    static size_t limit()
    {
        static const size_t l = [] {
            const char *e = getenv("GIO_KEEP_MB");
            const long mb = e ? atol(e) : 512;          // (a C3-sized file keeps ~260 MB: its inflated window and the reads' arrays)
            return (size_t)(mb < 0 ? 0 : mb) << 20;
        }();
        return l;
    }
    void *take(size_t bytes)                            // the smallest kept block that holds `bytes` without wasting three quarters of itself
    {
        std::lock_guard<std::mutex> g(mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < free_blocks.size(); i++)
            if (free_blocks[i].bytes >= bytes && free_blocks[i].bytes <= 4 * bytes && (best == (size_t)-1 || free_blocks[i].bytes < free_blocks[best].bytes)) best = i;
        if (best == (size_t)-1) return nullptr;
        void *p = free_blocks[best].p;
        kept -= free_blocks[best].bytes;
        free_blocks[best] = free_blocks.back();
        free_blocks.pop_back();
        return p;
    }
    bool give(void *p, size_t bytes)
    {
        std::lock_guard<std::mutex> g(mu);
        if (kept + bytes > limit()) return false;
        try { free_blocks.push_back(blk{p, bytes}); } catch (...) { return false; }
        kept += bytes;
        return true;
    }
    void release_all()
    {
        std::vector<blk> v;
        {
            std::lock_guard<std::mutex> g(mu);
            v.swap(free_blocks);
            kept = 0;
        }
        for (auto &b : v) free(b.p);
    }
};
static const size_t BIG_BYTES = (size_t)4 << 20;
static size_t big_round(size_t bytes) { const size_t al = (size_t)2 << 20; return (bytes + al - 1) / al * al; }
static void *big_alloc(size_t bytes)
{
    if (bytes >= BIG_BYTES) {
        const size_t al = (size_t)2 << 20, sz = big_round(bytes);
        if (void *q = big_cache::get().take(sz)) return q;
        void *p = aligned_alloc(al, sz);
        if (p) madvise(p, sz, MADV_HUGEPAGE);
        return p;
    }
    return malloc(bytes);
}
// (bytes: what big_alloc was asked for)
static void big_free(void *p, size_t bytes)
{
    if (!p) return;
    if (bytes >= BIG_BYTES && big_cache::get().give(p, big_round(bytes))) return;
    free(p);
}
template <typename T> struct big_allocator {
    typedef T value_type;
    big_allocator() = default;
    template <typename U> big_allocator(const big_allocator<U> &) {}
    T *allocate(size_t n)
    {
        void *p = big_alloc(n * sizeof(T));
        if (!p) throw std::bad_alloc();
        return (T *)p;
    }
    void deallocate(T *p, size_t n) { big_free(p, n * sizeof(T)); }
    template <typename U> bool operator==(const big_allocator<U> &) const { return true; }
    template <typename U> bool operator!=(const big_allocator<U> &) const { return false; }
};
template <typename T> using bigvec = std::vector<T, big_allocator<T>>;

// a byte window that grows without value-initialising what the inflaters are about to overwrite (zero-filling 100 MB
// on one thread, and faulting its pages in there, cost more than inflating it on eight)
// records a thread should have before another one is started (GIO_PART_RECORDS: the tests make it small)
static size_t part_records()
{
    static const size_t n = [] {
        long v = getenv("GIO_PART_RECORDS") ? atol(getenv("GIO_PART_RECORDS")) : 2048;
        return (size_t)(v < 1 ? 1 : v);
    }();
    return n;
}

class rawbuf {
public:
    rawbuf() = default;
    rawbuf(const rawbuf &) = delete;
    rawbuf &operator=(const rawbuf &) = delete;
    ~rawbuf() { big_free(p_, cap_); }
    size_t size() const { return n_; }
    uint8_t *data() { return p_; }
    const uint8_t *data() const { return p_; }
    void clear() { n_ = 0; }
    void drop_front(size_t k) { if (k) { memmove(p_, p_ + k, n_ - k); n_ -= k; } }
    // room for `c` bytes without another move (the bytes held stay)
    bool reserve(size_t c)
    {
        if (c <= cap_) return true;
        uint8_t *q = (uint8_t *)big_alloc(c);
        if (!q) return false;
        if (n_) memcpy(q, p_, n_);
        big_free(p_, cap_);
        p_ = q;
        cap_ = c;
        return true;
    }
    bool resize_uninit(size_t n)
    {
        if (n > cap_) {
            size_t c = cap_ ? cap_ : ((size_t)1 << 20);
            while (c < n) c += c / 2;
            uint8_t *q = (uint8_t *)big_alloc(c);
            if (!q) return false;
            if (n_) memcpy(q, p_, n_);
            big_free(p_, cap_);
            p_ = q;
            cap_ = c;
        }
        n_ = n;
        return true;
    }
private:
    uint8_t *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

class bgzf_stream {
public:
    ~bgzf_stream() { if (fd_ >= 0) close(fd_); }

    int open(const char *path)
    {
        fd_ = ::open(path, O_RDONLY | O_CLOEXEC);
        if (fd_ < 0) return fail(-1, "cannot open %s", path);
        path_ = path;
        struct stat sb;
        if (fstat(fd_, &sb) == 0 && S_ISREG(sb.st_mode)) { fsize_ = (uint64_t)sb.st_size; regular_ = true; }
        // (a pipe or the like is read front to back, one read after the other; what is no BGZF shows at its first block header)
        uint8_t magic[4];
        if (regular_ && (pread(fd_, magic, 4, 0) != 4 || magic[0] != 0x1f || magic[1] != 0x8b))
            return fail(-2, "%s is not gzip/BGZF", path);
        return seek(0, 0);
    }

    // continue at virtual offset (coffset, uoffset)
    int seek(uint64_t coffset, unsigned uoffset)
    {
        if (!regular_ && coffset != fpos_) return fail(-1, "seek failed in %s", path_.c_str());
        fpos_ = coffset;
        cbuf_.clear();
        out_.clear();
        rd_ = 0;
        start_ = coffset;
        taken_ = 0;
        skip_ = uoffset;
        eof_ = false;
        more_ = false;
        return 0;
    }
    // the next read takes everything up to `bytes` in one go, its slices on all threads (gio_prefetch: the file's remainder is
    // wanted anyway, and copying 7 MB out of the page cache on one thread, 4 MB at a time between the inflates, was 1.3-1.5 ms of
    // a 4.5-5.7 ms prefetch)
    void read_ahead(size_t bytes) { batch_ = bytes > batch_ ? bytes : batch_; }
    uint64_t unread_file_bytes() const { return fsize_ > fpos_ ? fsize_ - fpos_ : 0; }

    // at least `need` unread bytes in the window unless the file ends first; returns <0 on error, else bytes available
    int64_t ensure(size_t need)
    {
        while (out_.size() - rd_ < need && !eof_) {
            int rc = refill();
            if (rc) return rc;
        }
        return (int64_t)(out_.size() - rd_);
    }
    const uint8_t *ptr() const { return out_.data() + rd_; }
    // share of the file behind the last seek that has been read (for sizing what grows with the records)
    double progress() const { return fsize_ > start_ ? (double)taken_ / (double)(fsize_ - start_) : 1.0; }
    void consume(size_t n) { rd_ += n; }
    size_t window_cap() const { return WINDOW; }
    double read_seconds() const { return read_s_; }
    // a window that is filled by several refills in a row (gio_prefetch) grows in place: moving 80 MB to make room for the next
    // 35 was most of a prefetch's time.  Address space only -- pages are touched as they are written.
    void reserve_window(size_t bytes) { out_.reserve(bytes); }

private:
    struct blk { size_t coff, clen, isize, uoff; };

    int refill()
    {
        // keep the unread tail, read another batch of compressed bytes, inflate its complete blocks
        // (only when at least half of it has been read: behind an index seek the first block is entered a few KB in, and moving a
        // 115 MB window by those few KB to make room for nothing -- the refill that finds the file's end -- was 6-10 ms of a prefetch)
        if (rd_ > 0 && rd_ >= out_.size() / 2) {
            out_.drop_front(rd_);
            rd_ = 0;
        }
        // small first batches (the header, an index seek right behind it), then 4 MB at a time (the inflated window is reused from batch to batch: fresh pages are the expensive part)
        const size_t BATCH = batch_;
        if (batch_ < ((size_t)4 << 20)) batch_ *= 4;
        // (compressed bytes a capped window left behind are inflated before more are read)
        const size_t have = cbuf_.size();
        size_t got = 0;
        if (have < ((size_t)1 << 17) || !more_) {
            const auto r0 = std::chrono::steady_clock::now();
            // (no more than the file holds: what a slice reads short of that is a file that shrank under us)
            size_t want = BATCH;
            if (regular_ && (uint64_t)want > unread_file_bytes()) want = (size_t)unread_file_bytes();
            if (!cbuf_.resize_uninit(have + want)) return fail(-6, "out of memory");
            const int nr = regular_ ? (int)std::min<size_t>((size_t)n_threads(), want / ((size_t)1 << 20)) : 1;
            if (nr >= 2) {
                std::vector<size_t> part((size_t)nr, 0);
                worker_pool::get().run(nr, [&](int t) {
                    const size_t lo = want * (size_t)t / (size_t)nr, hi = want * (size_t)(t + 1) / (size_t)nr;
                    size_t done = 0;
                    while (lo + done < hi) {
                        const ssize_t k = pread(fd_, cbuf_.data() + have + lo + done, hi - lo - done, (off_t)(fpos_ + lo + done));
                        if (k <= 0) break;
                        done += (size_t)k;
                    }
                    part[(size_t)t] = done;
                });
                for (int t = 0; t < nr; t++) {
                    const size_t lo = want * (size_t)t / (size_t)nr, hi = want * (size_t)(t + 1) / (size_t)nr;
                    got += part[(size_t)t];
                    if (part[(size_t)t] != hi - lo) break;           // (a short slice ends the contiguous part)
                }
            } else {
                while (got < want) {
                    const ssize_t k = regular_ ? pread(fd_, cbuf_.data() + have + got, want - got, (off_t)(fpos_ + got))
                                               : read(fd_, cbuf_.data() + have + got, want - got);
                    if (k <= 0) break;
                    got += (size_t)k;
                }
            }
            fpos_ += got;
            cbuf_.resize_uninit(have + got);
            read_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
            g_stats.compressed_bytes += (int64_t)got;
            taken_ += got;
            if (got == 0 && !more_) {
                eof_ = true;
                if (cbuf_.size() != 0) return fail(-4, "truncated BGZF block at the end of %s", path_.c_str());
                return 0;
            }
        }
        more_ = false;
        std::vector<blk> blocks;
        size_t o = 0, total = 0;
        while (o + 18 <= cbuf_.size()) {
            const uint8_t *b = cbuf_.data() + o;
            if (b[0] != 0x1f || b[1] != 0x8b || b[2] != 8 || !(b[3] & 4)) return fail(-4, "bad BGZF block header in %s", path_.c_str());
            const size_t xlen = rdu16(b + 10);
            if (o + 12 + xlen > cbuf_.size()) break;
            size_t bsize = 0;
            for (size_t x = 12; x + 4 <= 12 + xlen;) {           // extra subfields: find BC
                const size_t slen = rdu16(b + x + 2);
                if (b[x] == 'B' && b[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) bsize = (size_t)rdu16(b + x + 4) + 1;
                x += 4 + slen;
            }
            if (bsize < 12 + xlen + 8) return fail(-4, "BGZF block without a valid BC field in %s", path_.c_str());
            if (o + bsize > cbuf_.size()) break;
            const size_t isize = rdu32(b + bsize - 4);
            if (isize > 65536) return fail(-4, "BGZF block claims %zu bytes in %s", isize, path_.c_str());
            // the inflated window stays within WINDOW bytes (what is touched once and freed again costs page faults and
            // unmapping: 116 MB for a million short reads in one piece); the blocks behind wait in cbuf_
            if (total + isize > WINDOW && !blocks.empty()) { more_ = true; break; }
            blocks.push_back({o + 12 + xlen, bsize - 12 - xlen - 8, isize, total});
            total += isize;
            o += bsize;
        }
        const size_t base = out_.size();
        if (!out_.resize_uninit(base + total)) return fail(-6, "out of memory");
        std::atomic<size_t> next(0);
        std::atomic<int> bad(0);
        auto work = [&]() {
            inflater inf;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= blocks.size()) break;
                const blk &k = blocks[i];
                if (!inf.run(cbuf_.data() + k.coff, k.clen, out_.data() + base + k.uoff, k.isize)) bad.store(1);
            }
        };
        const int nt = (int)std::min<size_t>((size_t)n_threads(), blocks.size() / 8 + 1);
        if (nt <= 1) work();
        else worker_pool::get().run(nt, [&](int) { work(); });
        if (bad.load()) return fail(-3, "inflate failed in %s", path_.c_str());
        g_stats.blocks += (int64_t)blocks.size();
        cbuf_.drop_front(o);
        if (skip_) {                                              // first refill behind a seek: start inside the first block
            if (skip_ > total) return fail(-4, "index offset beyond its block in %s", path_.c_str());
            rd_ = skip_;
            skip_ = 0;
        }
        return 0;
    }

    int fd_ = -1;
    bool regular_ = false;                                            // a regular file: its size is known, slices can be read side by side
    uint64_t fpos_ = 0;                                               // where the next read starts
    std::string path_;
    rawbuf cbuf_;                                                     // (grown without zero-filling what the read is about to write)
    rawbuf out_;
    // (GIO_WINDOW: the tests make it one block.  128 MB since the buffers are kept between decodes -- it was 32 MB to bound what a
    // decode touches once and unmaps: a C3-sized file is then one window instead of three, 13-18 ms inside the library against 18-25,
    // and 32 against 39 ms for a first decode with nothing kept)
    const size_t WINDOW = getenv("GIO_WINDOW") ? (size_t)atol(getenv("GIO_WINDOW")) : ((size_t)128 << 20);
    double read_s_ = 0.0;                                             // (GIO_TIMING: resize + fread of the compressed bytes)
    bool more_ = false;                                               // whole blocks are waiting in cbuf_
    size_t rd_ = 0, skip_ = 0, batch_ = (size_t)1 << 18;
    uint64_t fsize_ = 0, start_ = 0, taken_ = 0;
    bool eof_ = false;
};

// ---------------------------------------------------------------------------------------------
// BAM header, index
// ---------------------------------------------------------------------------------------------
struct bam_header {
    std::vector<std::pair<std::string, int64_t>> refs;
    bool sorted = false;        // @HD ... SO:coordinate
};

static int read_header(bgzf_stream &z, bam_header &h, const char *path)
{
    if (z.ensure(12) < 12 || memcmp(z.ptr(), "BAM\1", 4) != 0) return g_err[0] ? -4 : fail(-4, "%s is not a BAM file", path);
    const int32_t l_text = rd32(z.ptr() + 4);
    if (l_text < 0) return fail(-4, "bad BAM header");
    z.consume(8);
    if (z.ensure((size_t)l_text + 4) < (int64_t)l_text + 4) return fail(-4, "truncated BAM header");
    {
        // the first header line says whether the records are in coordinate order (then a scan may stop behind its window)
        const char *tx = (const char *)z.ptr();
        const size_t n = (size_t)l_text;
        size_t eol = 0;
        while (eol < n && tx[eol] != '\n') eol++;
        const std::string first(tx, eol);
        h.sorted = first.compare(0, 3, "@HD") == 0 && first.find("SO:coordinate") != std::string::npos;
    }
    z.consume((size_t)l_text);
    const int32_t n_ref = rd32(z.ptr());
    z.consume(4);
    if (n_ref < 0) return fail(-4, "bad BAM header");
    for (int i = 0; i < n_ref; i++) {
        if (z.ensure(4) < 4) return fail(-4, "truncated BAM header");
        const int32_t l_name = rd32(z.ptr());
        z.consume(4);
        if (l_name < 1 || l_name > (1 << 20) || z.ensure((size_t)l_name + 4) < (int64_t)l_name + 4) return fail(-4, "truncated BAM header");
        std::string name((const char *)z.ptr(), (size_t)l_name - 1);
        z.consume((size_t)l_name);
        h.refs.emplace_back(name, (int64_t)rd32(z.ptr()));
        z.consume(4);
    }
    return 0;
}

// smallest virtual offset an alignment overlapping 0-based position `beg` of reference `tid` can start at, from the
// linear index of a .bai (16 kb windows); false when there is no usable index
static bool bai_start(const char *bam_path, int tid, int64_t beg, uint64_t *voff)
{
    if (getenv("GIO_NO_INDEX") && atoi(getenv("GIO_NO_INDEX"))) return false;
    std::string p1 = std::string(bam_path) + ".bai", p2 = bam_path;
    if (p2.size() > 4 && p2.compare(p2.size() - 4, 4, ".bam") == 0) p2 = p2.substr(0, p2.size() - 4) + ".bai";
    FILE *fp = fopen(p1.c_str(), "rb");
    if (!fp) fp = fopen(p2.c_str(), "rb");
    if (!fp) return false;
    std::vector<uint8_t> d;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(fp);
    if (d.size() < 8 || memcmp(d.data(), "BAI\1", 4) != 0) return false;
    const int32_t n_ref = rd32(&d[4]);
    size_t o = 8;
    for (int r = 0; r < n_ref; r++) {
        if (o + 4 > d.size()) return false;
        const int32_t n_bin = rd32(&d[o]);
        o += 4;
        for (int b = 0; b < n_bin; b++) {
            if (o + 8 > d.size()) return false;
            const int32_t n_chunk = rd32(&d[o + 4]);
            o += 8;
            if (n_chunk < 0 || o + 16 * (size_t)n_chunk > d.size()) return false;
            o += 16 * (size_t)n_chunk;
        }
        if (o + 4 > d.size()) return false;
        const int32_t n_intv = rd32(&d[o]);
        o += 4;
        if (n_intv < 0 || o + 8 * (size_t)n_intv > d.size()) return false;
        if (r == tid) {
            if (n_intv == 0) return false;
            int64_t w = beg < 0 ? 0 : (beg >> 14);
            if (w >= n_intv) w = n_intv - 1;
            // windows without alignments hold 0: the nearest filled window to the left bounds the start from below
            uint64_t v = 0;
            for (int64_t q = w; q >= 0 && v == 0; q--) v = rdu64(&d[o + 8 * (size_t)q]);
            if (v == 0) return false;
            *voff = v;
            return true;
        }
        o += 8 * (size_t)n_intv;
    }
    return false;
}

static int gio_ref_len_impl(const char *bam_path, const char *contig, int64_t *len)
{
    if (!bam_path || !contig || !len) return fail(-1, "null argument");
    g_err[0] = 0;
    bgzf_stream z;
    int rc = z.open(bam_path);
    if (rc) return rc;
    bam_header h;
    if ((rc = read_header(z, h, bam_path))) return rc;
    for (auto &r : h.refs)
        if (r.first == contig) { *len = r.second; return 0; }
    return fail(-5, "contig %s not in %s", contig, bam_path);
}

extern "C" void gio_release_buffers(void) { big_cache::get().release_all(); }

extern "C" void gio_table_free(gio_table *t)
{
    if (!t) return;
    free(t->rank); free(t->off); free(t->bases);
    memset(t, 0, sizeof *t);
}

// ---------------------------------------------------------------------------------------------
// records
// ---------------------------------------------------------------------------------------------
struct bam_rec {
    int32_t ref_id, pos, l_seq;
    int flag, n_cigar, l_read_name;
    const char *name;
    const uint8_t *cigar;       // n_cigar x uint32 (the CG tag's array when the core holds the placeholder)
    const uint8_t *seq;
};

// checks every length against the record; resolves the CG:B,I tag of reads with more than 65535 CIGAR operations
static int parse_record(const uint8_t *r, int32_t block_size, bam_rec &out)
{
    if (block_size < 32) return fail(-4, "BAM record shorter than its fixed fields");
    out.ref_id = rd32(r);
    out.pos = rd32(r + 4);
    out.l_read_name = r[8];
    out.n_cigar = rdu16(r + 12);
    out.flag = rdu16(r + 14);
    out.l_seq = rd32(r + 16);
    if (out.l_seq < 0 || out.l_read_name < 1) return fail(-4, "BAM record with a negative sequence length or an empty name field");
    const int64_t var = 32 + (int64_t)out.l_read_name + 4 * (int64_t)out.n_cigar + ((int64_t)out.l_seq + 1) / 2 + (int64_t)out.l_seq;
    if (var > block_size) return fail(-4, "BAM record fields (%lld bytes) exceed its block_size %d", (long long)var, block_size);
    out.name = (const char *)(r + 32);
    if (out.name[out.l_read_name - 1] != 0) return fail(-4, "BAM read name is not NUL-terminated");
    out.cigar = r + 32 + out.l_read_name;
    out.seq = out.cigar + 4 * (size_t)out.n_cigar;
    if (out.n_cigar == 2) {
        // placeholder <l_seq>S<ref_len>N: the real CIGAR is in the CG:B,I tag (SAM spec 4.2.2)
        const uint32_t c0 = rdu32(out.cigar), c1 = rdu32(out.cigar + 4);
        if ((c0 & 15) == 4 && (int64_t)(c0 >> 4) == out.l_seq && (c1 & 15) == 3) {
            const uint8_t *a = r + var, *end = r + block_size;
            while (a + 3 <= end) {
                const char t0 = (char)a[0], t1 = (char)a[1], ty = (char)a[2];
                a += 3;
                size_t sz = 0;
                if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
                else if (ty == 's' || ty == 'S') sz = 2;
                else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
                else if (ty == 'Z' || ty == 'H') {
                    const uint8_t *z = (const uint8_t *)memchr(a, 0, (size_t)(end - a));
                    if (!z) return fail(-4, "unterminated string tag in a BAM record");
                    sz = (size_t)(z - a) + 1;
                } else if (ty == 'B') {
                    if (a + 5 > end) return fail(-4, "truncated array tag in a BAM record");
                    const char sub = (char)a[0];
                    const uint32_t cnt = rdu32(a + 1);
                    const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                    if ((uint64_t)cnt * es > (uint64_t)(end - a - 5)) return fail(-4, "array tag exceeds its BAM record");
                    if (t0 == 'C' && t1 == 'G' && sub == 'I') {
                        out.cigar = a + 5;
                        out.n_cigar = (int)cnt;
                        return 0;
                    }
                    sz = 5 + (size_t)cnt * es;
                } else return fail(-4, "unknown tag type '%c' in a BAM record", ty);
                if (sz > (size_t)(end - a)) return fail(-4, "tag exceeds its BAM record");
                a += sz;
            }
            return fail(-4, "placeholder CIGAR without a CG tag");
        }
    }
    return 0;
}

static const char SEQ[] = "=ACMGRSVTWYHKDBN";

// gio_prefetch: everything of a decode that does not need the SNP positions -- the header, the index seek, the window's BGZF
// blocks read and inflated (up to the inflated window's cap) -- on a thread of the library, so that the caller can parse its VCF
// meanwhile (gretel/cmd.py:69-78: process_vcf, then load_from_bam).  One at a time; the decode of the same (path, contig, start,
// end) takes it over, anything else discards it.  A prefetch that failed is dropped and the decode starts afresh (and reports).
struct prefetched {
    std::string path, contig;
    int32_t start = 0, end = 0;
    std::thread th;
    std::unique_ptr<bgzf_stream> z;
    bam_header hd;
    int tid = -1, rc = 0;
    bool used_index = false;
    int64_t compressed = 0, blocks = 0;
};
static std::mutex &prefetch_mu() { static std::mutex *m = new std::mutex(); return *m; }
static prefetched *&prefetch_slot()
{
    static prefetched *p = nullptr;
    static std::once_flag once;
    // (a forked child has no such thread: what the slot held is the parent's, and is left alone there)
    std::call_once(once, [] { pthread_atfork(nullptr, nullptr, [] { prefetch_slot() = nullptr; }); });
    return p;
}
static void prefetch_body(prefetched *p)
{
    memset(&g_stats, 0, sizeof g_stats);
    g_err[0] = 0;
    const auto t0 = std::chrono::steady_clock::now();
    try {
        p->z.reset(new bgzf_stream());
        int rc = p->z->open(p->path.c_str());
        if (!rc) rc = read_header(*p->z, p->hd, p->path.c_str());
        if (!rc) {
            for (size_t i = 0; i < p->hd.refs.size(); i++)
                if (p->hd.refs[i].first == p->contig) p->tid = (int)i;
            if (p->tid < 0) rc = -5;
        }
        uint64_t voff = 0;
        if (!rc && bai_start(p->path.c_str(), p->tid, (int64_t)p->start - 1, &voff)) {
            rc = p->z->seek(voff >> 16, (unsigned)(voff & 0xffff));
            p->used_index = true;
        }
        const auto t1 = std::chrono::steady_clock::now();
        if (!rc) {
            p->z->reserve_window(2 * p->z->window_cap());
            if (!getenv("GIO_NO_READ_AHEAD")) p->z->read_ahead((size_t)std::min<uint64_t>(p->z->unread_file_bytes(), (uint64_t)p->z->window_cap() / 2));
            const int64_t av = p->z->ensure(p->z->window_cap());
            if (av < 0) rc = (int)av;
        }
        if (getenv("GIO_TIMING"))
            fprintf(stderr, "gio: prefetch: open + header + index %.4f s, read + inflate %.4f s (of which reading the file %.4f)\n",
                    std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count(), p->z->read_seconds());
        p->rc = rc;
    } catch (...) { p->rc = -6; }
    p->compressed = g_stats.compressed_bytes;
    p->blocks = g_stats.blocks;
}
static void prefetch_discard(prefetched *p)
{
    if (!p) return;
    if (p->th.joinable()) p->th.join();
    delete p;
}
// the prefetch of exactly this window, finished and good -- or nullptr (whatever else was there is discarded)
static prefetched *prefetch_take(const char *bam_path, const char *contig, int32_t start_pos, int32_t end_pos)
{
    prefetched *p = nullptr;
    {
        std::lock_guard<std::mutex> g(prefetch_mu());
        p = prefetch_slot();
        prefetch_slot() = nullptr;
    }
    if (!p) return nullptr;
    if (p->th.joinable()) p->th.join();
    if (p->rc == 0 && p->path == bam_path && p->contig == contig && p->start == start_pos && p->end == end_pos) return p;
    delete p;
    return nullptr;
}

static int gio_support_table_from_bam_impl(const char *bam_path, const char *contig, int32_t start_pos, int32_t end_pos,
                                           const uint8_t *region, int stepper_all, int32_t max_depth, gio_table *out,
                                           gio_alloc_fn alloc = nullptr, void *alloc_ctx = nullptr)
{
    if (!bam_path || !contig || !region || !out || end_pos < 0) return fail(-1, "bad argument");
    memset(out, 0, sizeof *out);
    memset(&g_stats, 0, sizeof g_stats);
    g_err[0] = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    std::unique_ptr<bgzf_stream> z_own;
    bam_header hd;
    int tid = -1;
    int rc = 0;
    if (std::unique_ptr<prefetched> pf{prefetch_take(bam_path, contig, start_pos, end_pos)}) {
        // (gio_prefetch has read the header, sought and inflated the window's blocks meanwhile)
        z_own = std::move(pf->z);
        hd = std::move(pf->hd);
        tid = pf->tid;
        g_stats.used_index = pf->used_index ? 1 : 0;
        g_stats.compressed_bytes += pf->compressed;
        g_stats.blocks += pf->blocks;
        g_stats.prefetched = 1;
    } else {
        z_own.reset(new bgzf_stream());
        if ((rc = z_own->open(bam_path))) return rc;
        if ((rc = read_header(*z_own, hd, bam_path))) return rc;
        for (size_t i = 0; i < hd.refs.size(); i++)
            if (hd.refs[i].first == contig) tid = (int)i;
        if (tid < 0) return fail(-5, "contig %s not in %s", contig, bam_path);
        uint64_t voff = 0;
        if (bai_start(bam_path, tid, (int64_t)start_pos - 1, &voff)) {
            if ((rc = z_own->seek(voff >> 16, (unsigned)(voff & 0xffff)))) return rc;
            g_stats.used_index = 1;
        }
    }
    bgzf_stream &z = *z_own;

    // csum[x] = number of SNPs in [0, x)
    std::vector<int32_t> csum((size_t)end_pos + 2, 0);
    for (int32_t p = 0; p <= end_pos; p++) csum[p + 1] = csum[p] + (region[p] ? 1 : 0);
    // next_snp[x] = smallest SNP position >= x (end_pos + 1 when there is none), x in [0, end_pos + 1]
    std::vector<int32_t> next_snp((size_t)end_pos + 2, end_pos + 1);
    for (int32_t p = end_pos; p >= 0; p--) next_snp[p] = region[p] ? p : next_snp[p + 1];

    // One entry per KEPT RECORD in file order (its number = gid).  The rows of the table (util.py:191-207) are the records
    // that were the first with their key, in that order; a record whose key was seen before (the same read met through
    // another record: rare) gives the first one's row a private copy of its characters with the new ones appended.
    // The characters and keys stay where the threads wrote them (the parts of every batch live until the table is assembled).
    struct rinfo {
        const uint8_t *ch;
        const char *key;
        int32_t rank, len, key_len;
        uint32_t h_lo, h_hi;
        uint32_t dup_of1;                                   // 0: opens a row; else 1 + gid of the record that opened it
    };
    // (grown without value-initialising: a million entries are 40 MB that the fill threads are about to write anyway)
    struct rinfo_vec {
        rinfo *p = nullptr;
        size_t n = 0, cap = 0;
        ~rinfo_vec() { big_free(p, cap * sizeof(rinfo)); }
        size_t size() const { return n; }
        size_t capacity() const { return cap; }
        rinfo *data() { return p; }
        rinfo &operator[](size_t i) { return p[i]; }
        bool reserve(size_t c)
        {
            if (c <= cap) return true;
            rinfo *q = (rinfo *)big_alloc(c * sizeof(rinfo));
            if (!q) return false;
            if (n) memcpy(q, p, n * sizeof(rinfo));
            big_free(p, cap * sizeof(rinfo));
            p = q;
            cap = c;
            return true;
        }
        bool resize_uninit(size_t m)
        {
            if (m > cap && !reserve(std::max(m, cap + cap / 2))) return false;
            n = m;
            return true;
        }
    } info;
    std::vector<std::unique_ptr<uint8_t[]>> moved;          // the private copies
    // The key table is cut into PARTITIONS by bits of the key's hash, one thread each: whether a record is the first with
    // its key only depends on the records of its own partition, taken in file order.  Per partition: open addressing
    // (linear probing, at most half full); the hash's lower half says where the probe starts, a slot holds the upper half
    // and the gid -- 8 bytes, lazily-zeroed memory; a matching tag is confirmed on the lower half and the key bytes.
    struct slot { uint32_t tag, gid1; };                    // gid1 = gid + 1, 0 = empty
    // (a partition's slots come from the kept blocks like the other working buffers and go back there: mapping 16 MB of fresh pages
    // per decode and unmapping them on the way out was 2.5-3 ms of a 17 ms call.  A kept block is not zero: the partition's own
    // thread clears it before its first record -- `stale` -- so the clearing is spread over the threads as the page faults were)
    struct ptable {
        slot *p = nullptr;
        size_t n = 0, count = 0;
        bool stale = false;                                 // allocated, not cleared yet (and nothing placed)
        std::vector<uint32_t> dups;                         // this batch's records whose key was there already
        bool oom = false;
        ptable() = default;
        ptable(const ptable &) = delete;
        ptable &operator=(const ptable &) = delete;
        ~ptable() { release(p, n); }
        static size_t bytes_of(size_t slots) { return big_round(slots * sizeof(slot)); }
        static slot *alloc(size_t slots)
        {
            const size_t sz = bytes_of(slots);
            if (void *q = big_cache::get().take(sz)) return (slot *)q;
            void *q = aligned_alloc((size_t)2 << 20, sz);
            if (q) madvise(q, sz, MADV_HUGEPAGE);
            return (slot *)q;
        }
        static void release(slot *q, size_t slots)
        {
            if (q && !big_cache::get().give(q, bytes_of(slots))) free(q);
        }
        void clear_if_stale()
        {
            if (stale) { memset(p, 0, n * sizeof(slot)); stale = false; }
        }
    };
    int n_part = 1;
    while (n_part * 2 <= n_threads() && n_part * 2 <= 16) n_part *= 2;
    std::vector<ptable> tabs((size_t)n_part);
    // room for `rows` keys at half load in one partition; rehashing walks the old table front to back.  clear_now: the caller is
    // the partition's thread (or nothing runs beside it); else an empty partition's new slots are cleared by its thread later
    auto ptable_reserve = [&](ptable &T, size_t rows, bool clear_now) -> bool {
        if (T.p && rows * 2 <= T.n) return true;
        size_t want = T.n ? T.n : ((size_t)1 << 12);
        while (rows * 2 > want) want *= 2;
        slot *np = ptable::alloc(want);
        if (!np) return false;
        if (T.count == 0 && !clear_now) {
            ptable::release(T.p, T.n);
            T.p = np;
            T.n = want;
            T.stale = true;
            return true;
        }
        memset(np, 0, want * sizeof(slot));
        const size_t nm = want - 1;
        if (!T.stale)
            for (size_t i = 0; i < T.n; i++) {
                const slot e = T.p[i];
                if (!e.gid1) continue;
                size_t j = (size_t)info[e.gid1 - 1].h_lo & nm;
                while (np[j].gid1) j = (j + 1) & nm;
                np[j] = e;
            }
        ptable::release(T.p, T.n);
        T.p = np;
        T.n = want;
        T.stale = false;
        return true;
    };

    // What one record contributes, worked out by any thread: the CIGAR walk, the key and its hash depend on nothing but
    // the record.  Only the key table (first-seen order of the rows, util.py:191-207) is sequential: it takes the
    // records of a batch in file order from the threads' parts.
    struct kept { uint64_t h; int32_t rank; uint32_t key_off, key_len, ch_off, ch_len; };
    // max_depth: what the pileup's read buffer sees of a record, in file order -- every record the stepper lets through that
    // overlaps the fetched region, whether or not it shows a SNP: start, reference end, and the row it would open (-1: none)
    struct depth_cand { int32_t pos, end, kept_idx; };
    struct part {
        bigvec<depth_cand> cands;
        bigvec<kept> recs;
        bigvec<uint8_t> chars;
        bigvec<char> keys;
        int64_t n_seen = 0;          // records of the range looked at (the one that stops the scan included)
        bool stop = false;           // met the first record behind the window
        int err = 0;                 // the range ended in a malformed record
        std::string msg;
    };
    const bool used_index = g_stats.used_index != 0;
    // returns 0: next record, 1: nothing behind this record can matter, < 0: malformed (g_err of the calling thread)
    auto one_record = [&](const uint8_t *r, int32_t block_size, part &o) -> int {
        bam_rec b;
        int prc = parse_record(r, block_size, b);
        if (prc) return prc;
        // coordinate-sorted: nothing behind the window, and nothing on a later reference, can matter
        if (used_index && (b.ref_id > tid || (b.ref_id == tid && b.pos >= end_pos))) return 1;
        const int flag = b.flag;
        if (b.ref_id != tid || (flag & (0x4 | 0x100 | 0x200 | 0x400))) return 0;
        if (!stepper_all && (flag & 0x1) && !(flag & 0x2)) return 0;        // orphans, stepper "samtools"
        // what the pileup's buffer counts: the record's reference span (bam_endpos: a read that consumes none spans one base),
        // if it overlaps the fetched region [start_pos - 1, end_pos)
        auto depth_note = [&](int64_t ref_end, int32_t kept_idx) {
            if (max_depth <= 0) return;
            const int64_t e = ref_end > (int64_t)b.pos ? ref_end : (int64_t)b.pos + 1;
            if ((int64_t)b.pos < (int64_t)end_pos && e > (int64_t)start_pos - 1)
                o.cands.push_back(depth_cand{b.pos, (int32_t)(e > 0x7fffffff ? 0x7fffffff : e), kept_idx});
        };
        if (b.l_seq == 0) {
            if (max_depth > 0) {
                int64_t r = b.pos;
                for (int c = 0; c < b.n_cigar; c++) {
                    const uint32_t v = rdu32(b.cigar + 4 * (size_t)c);
                    const int op = v & 15;
                    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) r += v >> 4;
                }
                depth_note(r, -1);
            }
            return 0;
        }

        // walk the CIGAR the way htslib's pileup resolves it, column by column over the SNP positions
        const size_t ch0 = o.chars.size();
        int64_t ref = b.pos, q = 0, qalen = 0;
        const int64_t hi = end_pos;
        for (int c = 0; c < b.n_cigar; c++) {
            const uint32_t v = rdu32(b.cigar + 4 * (size_t)c);
            const int op = v & 15;
            const int64_t ln = v >> 4;
            if (op == 0 || op == 7 || op == 8) {                            // M = X
                if (q + ln > b.l_seq) { o.chars.resize(ch0); return fail(-4, "CIGAR of read %s consumes more query than its %d bases", b.name, b.l_seq); }
                int64_t lo1 = ref + 1 < 1 ? 1 : ref + 1, hi1 = ref + ln < hi ? ref + ln : hi;
                // SNPs are sparse: from one to the next through next_snp instead of testing every column
                for (int64_t p1 = lo1 <= hi1 ? next_snp[lo1] : hi1 + 1; p1 <= hi1; p1 = next_snp[p1 + 1]) {
                    const int64_t qi = q + (p1 - 1 - ref);
                    const uint8_t byte = b.seq[qi >> 1];
                    o.chars.push_back((uint8_t)SEQ[(qi & 1) ? (byte & 15) : (byte >> 4)]);          // util.py:186-189, b[0]
                }
                ref += ln; q += ln; qalen += ln;
            } else if (op == 2 || op == 3) {                                // D / N -> is_del column, util.py:180-182
                int64_t lo1 = ref + 1 < 1 ? 1 : ref + 1, hi1 = ref + ln < hi ? ref + ln : hi;
                for (int64_t p1 = lo1 <= hi1 ? next_snp[lo1] : hi1 + 1; p1 <= hi1; p1 = next_snp[p1 + 1])
                    o.chars.push_back((uint8_t)'-');
                ref += ln;
            } else if (op == 1) { q += ln; qalen += ln; }                   // I
            else if (op == 4) { q += ln; }                                  // S
            if (q > b.l_seq) { o.chars.resize(ch0); return fail(-4, "CIGAR of read %s consumes more query than its %d bases", b.name, b.l_seq); }
        }
        int64_t leftmost = (int64_t)b.pos + 1;                              // util.py:162
        if (leftmost < start_pos) {                                         // util.py:165-171
            if (leftmost + qalen < start_pos) { o.chars.resize(ch0); depth_note(ref, -1); return 0; }
            leftmost = start_pos;
        }
        if (o.chars.size() == ch0) { depth_note(ref, -1); return 0; }
        depth_note(ref, (int32_t)o.recs.size());
        int one_or_two = 0;
        if (flag & 0x1) one_or_two = (flag & 0x40) ? 1 : ((flag & 0x80) ? 2 : 0);
        // "<qname>_<flag>_<1|2|0>", util.py:160 (snprintf here was a third of the whole decode)
        const size_t k0 = o.keys.size();
        o.keys.insert(o.keys.end(), b.name, b.name + (b.l_read_name - 1));
        {
            char tail[16];
            int tn = 0;
            tail[tn++] = '_';
            char dig[8];
            int nd = 0, f = flag;
            do { dig[nd++] = (char)('0' + f % 10); f /= 10; } while (f);
            while (nd) tail[tn++] = dig[--nd];
            tail[tn++] = '_';
            tail[tn++] = (char)('0' + one_or_two);
            o.keys.insert(o.keys.end(), tail, tail + tn);
        }
        uint64_t h = 0xcbf29ce484222325ull;                                 // FNV-1a, then a finalizer to spread the low bits
        for (size_t x = k0; x < o.keys.size(); x++) { h ^= (unsigned char)o.keys[x]; h *= 0x100000001b3ull; }
        h ^= h >> 32; h *= 0x9e3779b97f4a7c15ull; h ^= h >> 29;
        int64_t lm = leftmost > (int64_t)end_pos + 1 ? (int64_t)end_pos + 1 : leftmost;
        const int32_t rank = lm >= 1 ? csum[lm] - csum[1] : 0;              // util.py:198
        o.recs.push_back(kept{h, rank, (uint32_t)k0, (uint32_t)(o.keys.size() - k0), (uint32_t)ch0, (uint32_t)(o.chars.size() - ch0)});
        return 0;
    };
    double tm[5] = {0, 0, 0, 0, 0}, kt[3] = {0, 0, 0};
    const double t_setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    std::vector<std::unique_ptr<part>> kept_parts;                         // every batch's parts: the rows point into them
    const int64_t n_ref = (int64_t)hd.refs.size();
    // Does a record plausibly start at byte x of the window?  Only a guess (fixed fields in range, lengths that fit, a
    // printable NUL-terminated name, and the same for the record behind it): a thread that starts in the middle of the
    // window starts at such a place, and the chain of the thread in front of it must END EXACTLY THERE, or the batch
    // is done again front to back.
    auto plausible1 = [&](const uint8_t *w, size_t av, size_t x, size_t *next) -> bool {
        if (x + 36 > av) return false;
        const int32_t bs = rd32(w + x);
        if (bs < 32 || bs > (1 << 28)) return false;
        const int32_t refid = rd32(w + x + 4), pos = rd32(w + x + 8), l_seq = rd32(w + x + 20), nref = rd32(w + x + 24);
        const int l_name = w[x + 12], n_cig = rdu16(w + x + 16);
        if (refid < -1 || refid >= n_ref || nref < -1 || nref >= n_ref || pos < -1 || l_seq < 0 || l_name < 1) return false;
        const int64_t var = 32 + (int64_t)l_name + 4 * (int64_t)n_cig + ((int64_t)l_seq + 1) / 2 + (int64_t)l_seq;
        if (var > bs) return false;
        if (x + 36 + (size_t)l_name <= av) {
            const uint8_t *nm = w + x + 36;
            if (nm[l_name - 1] != 0) return false;
            for (int i = 0; i + 1 < l_name; i++)
                if (nm[i] < 33 || nm[i] > 126) return false;
        }
        *next = x + 4 + (size_t)bs;
        return true;
    };
    auto find_start = [&](const uint8_t *w, size_t av, size_t from, size_t until) -> size_t {
        for (size_t x = from; x < until; x++) {
            size_t n1 = 0, n2 = 0, n3 = 0;
            if (!plausible1(w, av, x, &n1)) continue;
            if (n1 + 36 <= av && !plausible1(w, av, n1, &n2)) continue;
            if (n2 && n2 + 36 <= av && !plausible1(w, av, n2, &n3)) continue;
            return x;
        }
        return (size_t)-1;
    };
    std::vector<std::vector<uint32_t>> lists;                               // [part][partition] -> gids, per batch
    // the depth cap's state across batches: the position the iterator stands on, the reads that entered, their ends
    int64_t dc_pos = -1, dc_base = -1, dc_exp_upto = 0, dc_accepted = 0, dc_expired = 0;
    std::vector<uint32_t> dc_ends;                                          // [end - dc_base] reads that entered and end there
    // ... and of the showing that the cap cannot bind (below): reads per bucket of B positions from pos0 on, the longest reference
    // span, the last start, the parts whose candidates were set aside
    struct { std::vector<uint32_t> H; int64_t pos0 = -1, B = 16, maxspan = 0, last_pos = -1; bool off = false; std::vector<size_t> deferred; } dq;
    bool done = false;
    bool front_to_back = false;                                             // this batch again on one thread (a guessed start was wrong)
    while (!done) {
        auto T0 = now();
        int64_t av = z.ensure(4);
        if (av < 0) return (int)av;
        if (av == 0) break;
        if (av < 4) return fail(-4, "truncated BAM record");
        const uint8_t *base = z.ptr();
        auto T1 = now(); tm[0] += secs(T0, T1);
        // Threads: each takes a byte range of the window, finds the first record that starts in it, and frames AND works on
        // records from there until it reaches the start of the next thread's range (framing alone is a chain of dependent
        // cache misses, one per record: 7 ms per million on one thread).
        int nt = front_to_back ? 1 : (int)std::min<size_t>((size_t)n_threads(), (size_t)av / (part_records() * 128) + 1);
        std::vector<size_t> start((size_t)nt + 1, (size_t)av), stop_at((size_t)nt, 0);
        start[0] = 0;
        for (int t = 1; t < nt; t++) {
            start[(size_t)t] = find_start(base, (size_t)av, (size_t)av * (size_t)t / (size_t)nt, (size_t)av * (size_t)(t + 1) / (size_t)nt);
            if (start[(size_t)t] == (size_t)-1) { nt = 1; start[1] = (size_t)av; break; }       // (no start found: one thread)
        }
        const size_t p0 = kept_parts.size();
        for (int t = 0; t < nt; t++) kept_parts.emplace_back(new part());
        auto work = [&](int t) {
            part &P = *kept_parts[p0 + (size_t)t];
            const size_t lim = start[(size_t)t + 1];
            size_t o = start[(size_t)t];
            try {
                const size_t guess = (lim - o) / 96 + 16;
                P.recs.reserve(guess);
                P.keys.reserve(guess * 24);
                P.chars.reserve(guess * 16);
                while (o < lim && o + 4 <= (size_t)av) {
                    const int32_t block_size = rd32(base + o);
                    if (block_size < 32 || block_size > (1 << 28)) {
                        P.err = -4;
                        char m[64];
                        snprintf(m, sizeof m, "bad BAM record size %d", block_size);
                        P.msg = m;
                        break;
                    }
                    if (o + 4 + (size_t)block_size > (size_t)av) break;     // the window ends inside this record
                    P.n_seen++;
                    const int rc1 = one_record(base + o + 4, block_size, P);
                    if (rc1 == 1) { P.stop = true; break; }
                    if (rc1 < 0) { P.err = rc1; P.msg = g_err; break; }
                    o += 4 + (size_t)block_size;
                }
            } catch (const std::exception &) {                              // (an exception must not leave a thread)
                P.err = -6;
                P.msg = "out of memory";
            }
            stop_at[(size_t)t] = o;
        };
        if (nt <= 1) work(0);
        else {
            worker_pool::get().run(nt, work);
        }
        // every chain must end where the next one began (then, by induction from the true start 0, every start was true)
        bool chained = true;
        for (int t = 0; t + 1 < nt && chained; t++) {
            const part &P = *kept_parts[p0 + (size_t)t];
            if (P.stop || P.err) break;                                     // (nothing behind it counts)
            if (stop_at[(size_t)t] != start[(size_t)t + 1]) chained = false;
        }
        if (!chained) {
            g_stats.reframed++;
            kept_parts.resize(p0);
            front_to_back = true;
            continue;
        }
        front_to_back = false;
        size_t o = stop_at[(size_t)nt - 1];
        if (o == 0 && !kept_parts[p0]->err && !kept_parts[p0]->stop) {
            // not one whole record in the window
            kept_parts.resize(p0);
            const int32_t block_size = rd32(base);
            av = z.ensure(4 + (size_t)block_size);
            tm[4] += secs(T1, now());
            if (av < 0) return (int)av;
            if (av < 4 + (int64_t)block_size) return fail(-4, "truncated BAM record");
            continue;
        }
        auto T2 = T1;
        auto T3 = now(); tm[2] += secs(T2, T3);
        // file order: everything in front of the first record that stops the scan or is malformed counts
        int n_valid = 0;                                                    // parts that count
        for (int t = 0; t < nt && !done; t++) {
            part &P = *kept_parts[p0 + (size_t)t];
            g_stats.records += P.n_seen;
            if (P.err) return fail(P.err, "%s", P.msg.c_str());
            if (P.stop) done = true;
            n_valid = t + 1;
        }
        // pysam's pileup keeps at most max_depth reads in its buffer (bam.pileup's default 8000, gretel/util.py:137 passes none):
        // htslib's bam_plp_push drops a read that starts at the position the iterator stands on while the buffer holds more
        // than maxcnt nodes.  In file order: the FIRST read of a position is pushed while the iterator still stands on an
        // earlier one and always enters; every later read of that position finds the buffer holding the reads that entered and
        // end behind position - 1 (the columns up to there have been produced and have released the rest) plus the list's
        // sentinel node, and is dropped when that is more than max_depth.  One pass over the batch's records, the ends in a
        // histogram (the starts ascend: what has expired is a running sum).
        if (max_depth > 0) {
            // The pass below is one thread over every record (2 ms per million).  Most windows never come near the cap, and that
            // can be SHOWN batch by batch on all threads: a read in the buffer when read c is pushed starts at or before c and ends
            // at or behind it, so it starts within the longest reference span seen so far in front of c -- if no stretch of that
            // length (in buckets of dq.B positions, over every batch so far) holds max_depth reads, nothing is dropped, by
            // induction over the file order.  Such a batch's candidates are set aside; the first batch that cannot be shown
            // (or is out of order: the pass says where) takes the pass over everything set aside first -- its state then is what
            // it would have been -- and the pass runs from there on.
            bool shown = false;
            if (!dq.off) {
                struct pstat { int64_t first = -1, last = -1, span = 0; bool sorted = true; };
                std::vector<pstat> ps((size_t)n_valid);
                worker_pool::get().run(n_valid, [&](int t) {
                    const part &P = *kept_parts[p0 + (size_t)t];
                    pstat &o = ps[(size_t)t];
                    for (size_t ci = 0; ci < P.cands.size(); ci++) {
                        const depth_cand &c = P.cands[ci];
                        if (o.first < 0) o.first = c.pos;
                        if ((int64_t)c.pos < o.last) o.sorted = false;
                        o.last = c.pos;
                        if ((int64_t)c.end - (int64_t)c.pos > o.span) o.span = (int64_t)c.end - (int64_t)c.pos;
                    }
                });
                bool sorted = true;
                int64_t last = dq.last_pos, first = -1, span = dq.maxspan;
                for (int t = 0; t < n_valid; t++) {
                    const pstat &o = ps[(size_t)t];
                    if (o.first < 0) continue;                              // (pos >= 0 on a candidate: the record is mapped)
                    if (!o.sorted || o.first < last) sorted = false;
                    if (first < 0) first = o.first;
                    last = o.last;
                    if (o.span > span) span = o.span;
                }
                if (sorted && first >= 0) {
                    if (dq.pos0 < 0) {
                        dq.pos0 = first;
                        const int64_t range = (int64_t)end_pos - first + 1;
                        dq.B = 16;
                        while (range / dq.B + 2 > ((int64_t)1 << 20)) dq.B *= 2;
                        dq.H.assign((size_t)(range / dq.B + 2), 0u);
                    }
                    worker_pool::get().run(n_valid, [&](int t) {
                        const part &P = *kept_parts[p0 + (size_t)t];
                        for (size_t ci = 0; ci < P.cands.size(); ci++)
                            __atomic_fetch_add(&dq.H[(size_t)(((int64_t)P.cands[ci].pos - dq.pos0) / dq.B)], 1u, __ATOMIC_RELAXED);
                    });
                    const int64_t k = span / dq.B + 1;                      // buckets a span reaches back
                    const int64_t jlo = (first - dq.pos0) / dq.B, jhi = (last - dq.pos0) / dq.B;
                    uint64_t S = 0;
                    for (int64_t j = jlo - k < 0 ? 0 : jlo - k; j < jlo; j++) S += dq.H[(size_t)j];
                    shown = true;
                    for (int64_t j = jlo; j <= jhi && shown; j++) {
                        S += dq.H[(size_t)j];
                        if (j - k - 1 >= 0) S -= dq.H[(size_t)(j - k - 1)];
                        if (S + 1 > (uint64_t)max_depth) shown = false;
                    }
                    if (shown) { dq.last_pos = last; dq.maxspan = span; }
                } else if (sorted) {
                    shown = true;                                           // (no candidate in this batch)
                }
            }
            auto depth_pass = [&](part &P, bool deferred) -> int {
                bool any = false;
                for (size_t ci = 0; ci < P.cands.size(); ci++) {
                    const depth_cand &c = P.cands[ci];
                    bool drop = false;
                    // (the running sum below is only right over ascending starts -- as is pysam's pileup, which wants a
                    // coordinate-sorted, indexed BAM: say so instead of counting wrongly or indexing in front of the histogram)
                    if ((int64_t)c.pos < dc_pos)
                        return fail(-4, "records are not in coordinate order (a read at %lld behind one at %lld): the pileup's "
                                    "depth cap (max_depth = %d, pysam's default) needs a coordinate-sorted BAM; sort it, or pass max_depth 0",
                                    (long long)c.pos, (long long)dc_pos, (int)max_depth);
                    if ((int64_t)c.pos != dc_pos) {
                        dc_pos = c.pos;                                      // (the iterator moves here once this read is in)
                    } else {
                        while (dc_exp_upto < (int64_t)c.pos) {              // reads that end at or before pos - 1 ... end <= pos - 1
                            const int64_t x = dc_exp_upto - dc_base;
                            if (x >= 0 && (size_t)x < dc_ends.size()) dc_expired += dc_ends[(size_t)x];
                            dc_exp_upto++;
                        }
                        // what has expired is never looked at again: the histogram stays as long as the longest read span,
                        // not as long as the window (4 bytes per reference base of a chromosome otherwise)
                        if (dc_exp_upto - dc_base > (int64_t)1 << 16 && dc_base >= 0) {
                            const size_t k = std::min((size_t)(dc_exp_upto - dc_base), dc_ends.size());
                            dc_ends.erase(dc_ends.begin(), dc_ends.begin() + (std::ptrdiff_t)k);
                            dc_base += (int64_t)k;
                        }
                        drop = dc_accepted - dc_expired + 1 > (int64_t)max_depth;
                    }
                    if (drop) {
                        if (deferred) return fail(-7, "internal: the pileup's depth cap dropped a read of a batch that was shown to need none");
                        g_stats.depth_dropped++;
                        if (c.kept_idx >= 0) { P.recs[(size_t)c.kept_idx].ch_len = 0xffffffffu; any = true; }
                        continue;
                    }
                    if (dc_base < 0) { dc_base = c.pos; dc_exp_upto = c.pos; }
                    if ((int64_t)c.end < dc_base) { dc_accepted++; dc_expired++; continue; }     // (ends in front of everything still counted: entered and expired)
                    const size_t x = (size_t)((int64_t)c.end - dc_base);
                    if (x >= dc_ends.size()) dc_ends.resize(x + 4096, 0);
                    dc_ends[x]++;
                    dc_accepted++;
                }
                if (any) {
                    size_t w = 0;
                    for (size_t i = 0; i < P.recs.size(); i++)
                        if (P.recs[i].ch_len != 0xffffffffu) P.recs[w++] = P.recs[i];
                    P.recs.resize(w);
                }
                bigvec<depth_cand>().swap(P.cands);
                return 0;
            };
            if (shown) {
                for (int t = 0; t < n_valid; t++) dq.deferred.push_back(p0 + (size_t)t);
            } else {
                dq.off = true;
                for (size_t pi : dq.deferred)
                    if (int rc2 = depth_pass(*kept_parts[pi], true)) return rc2;
                dq.deferred.clear();
                for (int t = 0; t < n_valid; t++)
                    if (int rc2 = depth_pass(*kept_parts[p0 + (size_t)t], false)) return rc2;
            }
        }
        // gids of this batch: the parts' records behind one another
        std::vector<size_t> gbase((size_t)n_valid + 1, info.size());
        for (int t = 0; t < n_valid; t++) gbase[(size_t)t + 1] = gbase[(size_t)t] + kept_parts[p0 + (size_t)t]->recs.size();
        const size_t g_end = gbase[(size_t)n_valid];
        if (g_end >= 0xfffffff0ull) return fail(-6, "more than 4e9 reads in the window");
        {
            // sized for what the rest of the file will bring at this rate
            const double pr = z.progress();
            size_t est = g_end;
            if (pr > 0.0 && pr < 1.0) est = (size_t)((double)est / pr * 1.05) + 1024;
            if (est < g_end || est > (size_t)1 << 32) est = g_end;
            if (!info.reserve(est) || !info.resize_uninit(g_end)) return fail(-6, "out of memory");
            for (auto &T : tabs)
                if (!ptable_reserve(T, est / (size_t)n_part + est / (size_t)(4 * n_part) + 1024, false)) return fail(-6, "out of memory");
        }
        // (a) the parts' records into info, each part by the thread that made it
        if (lists.size() < (size_t)n_valid * (size_t)n_part) lists.resize((size_t)n_valid * (size_t)n_part);
        auto fill = [&](int t) {
            const part &P = *kept_parts[p0 + (size_t)t];
            const uint8_t *cbase = P.chars.data();
            const char *kbase = P.keys.data();
            rinfo *dst = info.data() + gbase[(size_t)t];
            const size_t nr = P.recs.size();
            // ... and their numbers sorted by partition (a partition's thread then only touches its own records)
            std::vector<uint32_t> *mine = lists.data() + (size_t)t * (size_t)n_part;
            for (int q = 0; q < n_part; q++) { mine[q].clear(); mine[q].reserve(nr / (size_t)n_part + nr / (size_t)(8 * n_part) + 16); }
            const uint32_t qmask = (uint32_t)n_part - 1u;
            const uint32_t g0 = (uint32_t)gbase[(size_t)t];
            for (size_t i = 0; i < nr; i++) {
                const kept &k = P.recs[i];
                dst[i] = rinfo{cbase + k.ch_off, kbase + k.key_off, k.rank, (int32_t)k.ch_len, (int32_t)k.key_len, (uint32_t)k.h, (uint32_t)(k.h >> 32), 0u};
                mine[(uint32_t)(k.h >> 40) & qmask].push_back(g0 + (uint32_t)i);
            }
        };
        // (b) partition q takes the records whose hash says q, in file order
        const size_t g_lo = gbase[0];
        auto place = [&](int q) {
            ptable &T = tabs[(size_t)q];
            T.dups.clear();
            T.clear_if_stale();
            const rinfo *in = info.data();
            for (int t = 0; t < n_valid; t++) {
            const std::vector<uint32_t> &lst = lists[(size_t)t * (size_t)n_part + (size_t)q];
            const size_t nl = lst.size();
            for (size_t li = 0; li < nl; li++) {
                // (two cache misses per record, both known ahead: its entry, then its slot)
                if (li + 16 < nl) __builtin_prefetch(&in[lst[li + 16]]);
                if (li + 8 < nl && T.n) __builtin_prefetch(&T.p[(size_t)in[lst[li + 8]].h_lo & (T.n - 1)]);
                const uint32_t g = lst[li];
                const rinfo &r = in[g];
                if ((T.count + 1) * 2 > T.n && !ptable_reserve(T, T.count * 2 + 1024, true)) { T.oom = true; return; }
                const size_t mask = T.n - 1;
                size_t si = (size_t)r.h_lo & mask;
                uint32_t found1 = 0;
                while (T.p[si].gid1) {
                    if (T.p[si].tag == r.h_hi) {
                        const rinfo &w = in[T.p[si].gid1 - 1];
                        if (w.h_lo == r.h_lo && w.key_len == r.key_len && memcmp(w.key, r.key, (size_t)r.key_len) == 0) {
                            found1 = T.p[si].gid1;
                            break;
                        }
                    }
                    si = (si + 1) & mask;
                }
                if (!found1) {
                    T.p[si] = slot{r.h_hi, (uint32_t)g + 1u};
                    T.count++;
                } else {
                    info[g].dup_of1 = found1;                               // (only this partition's thread writes this entry)
                    T.dups.push_back((uint32_t)g);
                }
            }
            }
        };
        auto run = [&](int n, auto &&fn) {
            worker_pool::get().run(n, fn);
        };
        auto K0 = now();
        run(n_valid, fill);
        auto K1 = now();
        if (g_end - g_lo < 4096) { for (int q = 0; q < n_part; q++) place(q); }
        else run(n_part, place);
        auto K2 = now();
        kt[0] += secs(T3, K0); kt[1] += secs(K0, K1); kt[2] += secs(K1, K2);
        for (auto &T : tabs)
            if (T.oom) return fail(-6, "out of memory");
        // (c) the records whose key was there already, in file order: their characters behind the row's
        {
            std::vector<uint32_t> dups;
            for (auto &T : tabs) dups.insert(dups.end(), T.dups.begin(), T.dups.end());
            std::sort(dups.begin(), dups.end());
            for (uint32_t g : dups) {
                rinfo &w = info[info[g].dup_of1 - 1];
                const rinfo &r = info[g];
                std::unique_ptr<uint8_t[]> cp(new (std::nothrow) uint8_t[(size_t)w.len + (size_t)r.len]);
                if (!cp) return fail(-6, "out of memory");
                memcpy(cp.get(), w.ch, (size_t)w.len);
                memcpy(cp.get() + w.len, r.ch, (size_t)r.len);
                w.ch = cp.get();
                w.len += r.len;
                moved.push_back(std::move(cp));
            }
        }
        for (int t = 0; t < nt; t++) bigvec<kept>().swap(kept_parts[p0 + (size_t)t]->recs);      // only the characters and keys are still needed
        tm[3] += secs(T3, now());
        if (done) break;
        z.consume(o);
    }
    if (getenv("GIO_TIMING"))
        fprintf(stderr, "gio: setup %.3f s, read+inflate %.3f, framing + records (%d threads) %.3f, key table (%d partitions) %.3f = sizing %.3f + entries %.3f + placing %.3f + repeats, scan done at %.3f s\n",
                t_setup, tm[0] + tm[4], n_threads(), tm[2], n_part, tm[3], kt[0], kt[1], kt[2], secs(t_begin, now()));

    // the table: the records that opened a row, in file order.  Ranges of gids on the threads: count, then place.
    const int64_t n_rec = (int64_t)info.size();
    const int nt_out = (int)std::min<int64_t>((int64_t)n_threads(), n_rec / 65536 + 1);
    std::vector<int64_t> rows_in((size_t)nt_out + 1, 0), chars_in((size_t)nt_out + 1, 0);
    std::vector<int32_t> longest((size_t)nt_out, 0);
    auto run_out = [&](auto &&fn) {
        worker_pool::get().run(nt_out, fn);
    };
    run_out([&](int t) {
        int64_t r = 0, c = 0;
        int32_t mx = 0;
        for (int64_t g = n_rec * t / nt_out, hi = n_rec * (t + 1) / nt_out; g < hi; g++)
            if (!info[(size_t)g].dup_of1) { r++; c += (int64_t)info[(size_t)g].len; if (info[(size_t)g].len > mx) mx = info[(size_t)g].len; }
        rows_in[(size_t)t + 1] = r;
        chars_in[(size_t)t + 1] = c;
        longest[(size_t)t] = mx;
    });
    for (int t = 0; t < nt_out; t++) if (longest[(size_t)t] > g_stats.max_row_len) g_stats.max_row_len = longest[(size_t)t];
    for (int t = 0; t < nt_out; t++) { rows_in[(size_t)t + 1] += rows_in[(size_t)t]; chars_in[(size_t)t + 1] += chars_in[(size_t)t]; }
    const int64_t n = rows_in[(size_t)nt_out], total = chars_in[(size_t)nt_out];
    // (the caller's allocator: e.g. page-locked memory kept from call to call, which the upload then reads by DMA and which needs
    // no fresh pages -- 17 MB per million reads otherwise)
    auto grab = [&](int which, size_t bytes) -> void * { return alloc ? alloc(alloc_ctx, which, bytes) : malloc(bytes); };
    out->rank = (int32_t *)grab(0, sizeof(int32_t) * (size_t)(n ? n : 1));
    out->off = (int64_t *)grab(1, sizeof(int64_t) * (size_t)(n + 1));
    out->bases = (uint8_t *)grab(2, (size_t)(total ? total : 1));
    if (!out->rank || !out->off || !out->bases) {
        if (alloc) memset(out, 0, sizeof *out); else gio_table_free(out);
        return fail(-6, alloc ? "the caller's allocator returned no memory" : "out of memory");
    }
    run_out([&](int t) {
        int64_t r = rows_in[(size_t)t], c = chars_in[(size_t)t];
        for (int64_t g = n_rec * t / nt_out, hi = n_rec * (t + 1) / nt_out; g < hi; g++) {
            const rinfo &x = info[(size_t)g];
            if (x.dup_of1) continue;
            out->rank[r] = x.rank;
            out->off[r] = c;
            memcpy(out->bases + c, x.ch, (size_t)x.len);
            r++;
            c += (int64_t)x.len;
        }
    });
    out->off[n] = total;
    out->n_reads = n;
    out->n_bases = total;
    g_stats.reads_kept = n;
    g_stats.libdeflate = libdeflate().ok ? 1 : 0;
    g_stats.threads = n_threads();
    g_stats.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    if (getenv("GIO_TIMING")) {
        auto R0 = now();
        kept_parts.clear();
        auto R1 = now();
        tabs.clear();
        auto R2 = now();
        moved.clear(); lists.clear();
        auto R3 = now();
        fprintf(stderr, "gio: table written at %.4f s; releasing: the threads' parts %.4f s, key table %.4f, copies + lists %.4f\n", g_stats.seconds, secs(R0, R1), secs(R1, R2), secs(R2, R3));
    }
    return 0;
}

static int gio_count_coverage_impl(const char *bam_path, const char *contig, int32_t start0, int32_t stop, int32_t *counts)
{
    if (!bam_path || !contig || !counts || start0 < 0 || stop < start0) return fail(-1, "bad argument");
    g_err[0] = 0;
    bgzf_stream z;
    int rc = z.open(bam_path);
    if (rc) return rc;
    bam_header hd;
    if ((rc = read_header(z, hd, bam_path))) return rc;
    int tid = -1;
    for (size_t i = 0; i < hd.refs.size(); i++)
        if (hd.refs[i].first == contig) tid = (int)i;
    if (tid < 0) return fail(-5, "contig %s not in %s", contig, bam_path);
    bool indexed = false;
    uint64_t voff = 0;
    if (bai_start(bam_path, tid, start0, &voff)) {
        if ((rc = z.seek(voff >> 16, (unsigned)(voff & 0xffff)))) return rc;
        indexed = true;
    }
    const int64_t len = (int64_t)stop - start0;
    memset(counts, 0, sizeof(int32_t) * 4 * (size_t)len);
    static const int8_t code2base[16] = {-1, 0, 1, -1, 2, -1, -1, -1, 3, -1, -1, -1, -1, -1, -1, -1};   // =ACMGRSVTWYHKDBN
    for (;;) {
        int64_t av = z.ensure(4);
        if (av < 0) return (int)av;
        if (av == 0) break;
        if (av < 4) return fail(-4, "truncated BAM record");
        const int32_t block_size = rd32(z.ptr());
        if (block_size < 32 || block_size > (1 << 28)) return fail(-4, "bad BAM record size %d", block_size);
        av = z.ensure(4 + (size_t)block_size);
        if (av < 0) return (int)av;
        if (av < 4 + (int64_t)block_size) return fail(-4, "truncated BAM record");
        bam_rec b;
        if ((rc = parse_record(z.ptr() + 4, block_size, b))) return rc;
        z.consume(4 + (size_t)block_size);
        // (an index, or a header that declares coordinate order: nothing behind the window can matter -- without either every
        // record of the file is visited, once per call)
        if ((indexed || hd.sorted) && (b.ref_id > tid || (b.ref_id == tid && b.pos >= stop))) break;
        if (b.ref_id != tid || (b.flag & 0x4)) continue;                  // other contig / unmapped
        if (b.l_seq == 0) continue;
        int64_t ref = b.pos, q = 0;
        for (int c = 0; c < b.n_cigar; c++) {
            const uint32_t v = rdu32(b.cigar + 4 * (size_t)c);
            const int op = v & 15;
            const int64_t ln = v >> 4;
            if (op == 0 || op == 7 || op == 8) {
                if (q + ln > b.l_seq) return fail(-4, "CIGAR of read %s consumes more query than its %d bases", b.name, b.l_seq);
                int64_t lo = ref < start0 ? start0 : ref, hi = ref + ln < stop ? ref + ln : stop;
                for (int64_t p = lo; p < hi; p++) {
                    const int64_t qi = q + (p - ref);
                    const uint8_t byte = b.seq[qi >> 1];
                    const int base = code2base[(qi & 1) ? (byte & 15) : (byte >> 4)];
                    if (base >= 0) counts[(size_t)base * len + (p - start0)]++;
                }
                ref += ln; q += ln;
            } else if (op == 2 || op == 3) ref += ln;
            else if (op == 1 || op == 4) q += ln;
            if (q > b.l_seq) return fail(-4, "CIGAR of read %s consumes more query than its %d bases", b.name, b.l_seq);
        }
    }
    return 0;
}

// The aligned (M/=/X) runs of every record on the contig, clipped to [start0, stop): what the coverage histogram of
// gretel/snpper.py:29 is made of, in the form the GPU kernel takes (gh_coverage_sites, include/gretel_hip.h): per run
// its 0-based reference start, and its bases as codes A0 C1 G2 T3, 4 = anything else (never counted).
extern "C" void gio_runs_free(gio_runs *r)
{
    if (!r) return;
    free(r->ref_start); free(r->off); free(r->codes);
    memset(r, 0, sizeof *r);
}

static int gio_match_runs_impl(const char *bam_path, const char *contig, int32_t start0, int32_t stop, gio_runs *out)
{
    if (!bam_path || !contig || !out || start0 < 0 || stop < start0) return fail(-1, "bad argument");
    memset(out, 0, sizeof *out);
    g_err[0] = 0;
    bgzf_stream z;
    int rc = z.open(bam_path);
    if (rc) return rc;
    bam_header hd;
    if ((rc = read_header(z, hd, bam_path))) return rc;
    int tid = -1;
    for (size_t i = 0; i < hd.refs.size(); i++)
        if (hd.refs[i].first == contig) tid = (int)i;
    if (tid < 0) return fail(-5, "contig %s not in %s", contig, bam_path);
    bool indexed = false;
    uint64_t voff = 0;
    if (bai_start(bam_path, tid, start0, &voff)) {
        if ((rc = z.seek(voff >> 16, (unsigned)(voff & 0xffff)))) return rc;
        indexed = true;
    }
    static const uint8_t code2base[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};   // =ACMGRSVTWYHKDBN
    std::vector<int32_t> ref_start;
    std::vector<int64_t> off(1, 0);
    std::vector<uint8_t> codes;
    for (;;) {
        int64_t av = z.ensure(4);
        if (av < 0) return (int)av;
        if (av == 0) break;
        if (av < 4) return fail(-4, "truncated BAM record");
        const int32_t block_size = rd32(z.ptr());
        if (block_size < 32 || block_size > (1 << 28)) return fail(-4, "bad BAM record size %d", block_size);
        av = z.ensure(4 + (size_t)block_size);
        if (av < 0) return (int)av;
        if (av < 4 + (int64_t)block_size) return fail(-4, "truncated BAM record");
        bam_rec b;
        if ((rc = parse_record(z.ptr() + 4, block_size, b))) return rc;
        z.consume(4 + (size_t)block_size);
        // (an index, or a header that declares coordinate order: nothing behind the window can matter -- without either every
        // record of the file is visited, once per call)
        if ((indexed || hd.sorted) && (b.ref_id > tid || (b.ref_id == tid && b.pos >= stop))) break;
        if (b.ref_id != tid || (b.flag & 0x4) || b.l_seq == 0) continue;
        int64_t ref = b.pos, q = 0;
        for (int c = 0; c < b.n_cigar; c++) {
            const uint32_t v = rdu32(b.cigar + 4 * (size_t)c);
            const int op = v & 15;
            const int64_t ln = v >> 4;
            if (op == 0 || op == 7 || op == 8) {
                if (q + ln > b.l_seq) return fail(-4, "CIGAR of read %s consumes more query than its %d bases", b.name, b.l_seq);
                const int64_t lo = ref < start0 ? start0 : ref, hi = ref + ln < stop ? ref + ln : stop;
                if (hi > lo) {
                    ref_start.push_back((int32_t)lo);
                    for (int64_t p = lo; p < hi; p++) {
                        const int64_t qi = q + (p - ref);
                        const uint8_t byte = b.seq[qi >> 1];
                        codes.push_back(code2base[(qi & 1) ? (byte & 15) : (byte >> 4)]);
                    }
                    off.push_back((int64_t)codes.size());
                }
                ref += ln; q += ln;
            } else if (op == 2 || op == 3) ref += ln;
            else if (op == 1 || op == 4) q += ln;
            if (q > b.l_seq) return fail(-4, "CIGAR of read %s consumes more query than its %d bases", b.name, b.l_seq);
        }
    }
    const size_t n = ref_start.size();
    out->ref_start = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    out->off = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    out->codes = (uint8_t *)malloc(codes.size() ? codes.size() : 1);
    if (!out->ref_start || !out->off || !out->codes) { gio_runs_free(out); return fail(-6, "out of memory"); }
    if (n) memcpy(out->ref_start, ref_start.data(), sizeof(int32_t) * n);
    memcpy(out->off, off.data(), sizeof(int64_t) * (n + 1));
    if (!codes.empty()) memcpy(out->codes, codes.data(), codes.size());
    out->n_runs = (int64_t)n;
    out->n_bases = (int64_t)codes.size();
    return 0;
}

extern "C" int gio_ref_len(const char *bam_path, const char *contig, int64_t *len)
{
    try { return gio_ref_len_impl(bam_path, contig, len); }
    catch (const std::bad_alloc &) { return fail(-6, "out of memory"); }
    catch (const std::exception &e) { return fail(-6, "%s", e.what()); }
}

extern "C" int gio_support_table_from_bam_depth(const char *bam_path, const char *contig, int32_t start_pos, int32_t end_pos,
                                                const uint8_t *region, int stepper_all, int32_t max_depth, gio_table *out)
{
    const auto t0 = std::chrono::steady_clock::now();
    int rc;
    try { rc = gio_support_table_from_bam_impl(bam_path, contig, start_pos, end_pos, region, stepper_all, max_depth, out); }
    catch (const std::bad_alloc &) { return fail(-6, "out of memory"); }
    catch (const std::exception &e) { return fail(-6, "%s", e.what()); }
    // (the call's own clock: the working buffers the function releases on its way out belong to it)
    const double all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("GIO_TIMING")) fprintf(stderr, "gio: the call %.4f s, of which %.4f s behind the table (buffers released)\n", all, all - g_stats.seconds);
    if (rc == 0) g_stats.seconds = all;
    return rc;
}

extern "C" int gio_prefetch(const char *bam_path, const char *contig, int32_t start_pos, int32_t end_pos)
{
    if (!bam_path || !contig) return fail(-1, "bad argument");
    prefetched *p = nullptr;
    try {
        p = new prefetched();
        p->path = bam_path; p->contig = contig; p->start = start_pos; p->end = end_pos;
    } catch (...) { delete p; return fail(-6, "out of memory"); }
    prefetched *old = nullptr;
    {
        std::lock_guard<std::mutex> g(prefetch_mu());
        old = prefetch_slot();
        prefetch_slot() = nullptr;
    }
    prefetch_discard(old);
    try { p->th = std::thread(prefetch_body, p); }
    catch (...) { delete p; return fail(-6, "cannot start the prefetch thread"); }
    std::lock_guard<std::mutex> g(prefetch_mu());
    prefetch_discard(prefetch_slot());                      // (another thread's, started meanwhile)
    prefetch_slot() = p;
    return 0;
}

extern "C" void gio_prefetch_cancel(void)
{
    prefetched *old = nullptr;
    {
        std::lock_guard<std::mutex> g(prefetch_mu());
        old = prefetch_slot();
        prefetch_slot() = nullptr;
    }
    prefetch_discard(old);
}

extern "C" int gio_support_table_from_bam_alloc(const char *bam_path, const char *contig, int32_t start_pos, int32_t end_pos,
                                                const uint8_t *region, int stepper_all, int32_t max_depth,
                                                gio_alloc_fn alloc, void *ctx, gio_table *out)
{
    if (!alloc) return fail(-1, "bad argument");
    const auto t0 = std::chrono::steady_clock::now();
    int rc;
    try { rc = gio_support_table_from_bam_impl(bam_path, contig, start_pos, end_pos, region, stepper_all, max_depth, out, alloc, ctx); }
    catch (const std::bad_alloc &) { if (out) memset(out, 0, sizeof *out); return fail(-6, "out of memory"); }
    catch (const std::exception &e) { if (out) memset(out, 0, sizeof *out); return fail(-6, "%s", e.what()); }
    if (rc == 0) g_stats.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

extern "C" int gio_support_table_from_bam(const char *bam_path, const char *contig, int32_t start_pos, int32_t end_pos,
                                          const uint8_t *region, int stepper_all, gio_table *out)
{
    // (the reference passes no max_depth to bam.pileup, gretel/util.py:137: pysam's default applies)
    try { return gio_support_table_from_bam_impl(bam_path, contig, start_pos, end_pos, region, stepper_all, GIO_PYSAM_MAX_DEPTH, out); }
    catch (const std::bad_alloc &) { return fail(-6, "out of memory"); }
    catch (const std::exception &e) { return fail(-6, "%s", e.what()); }
}

extern "C" int gio_count_coverage(const char *bam_path, const char *contig, int32_t start0, int32_t stop, int32_t *counts)
{
    try { return gio_count_coverage_impl(bam_path, contig, start0, stop, counts); }
    catch (const std::bad_alloc &) { return fail(-6, "out of memory"); }
    catch (const std::exception &e) { return fail(-6, "%s", e.what()); }
}

extern "C" int gio_match_runs(const char *bam_path, const char *contig, int32_t start0, int32_t stop, gio_runs *out)
{
    try { return gio_match_runs_impl(bam_path, contig, start0, stop, out); }
    catch (const std::bad_alloc &) { return fail(-6, "out of memory"); }
    catch (const std::exception &e) { return fail(-6, "%s", e.what()); }
}

// Host half of libgavisunk.so: the file formats on either side of the tagging path.
//   SUNK list + loc table loader  <- kmerpos_annot3.nim:20-69
//   FASTA/FASTQ(.gz) reader + 2-bit packer  <- `readfq(reads)`, kmerpos_annot3.nim:85, rlen.nim:13
// Native C++ because the reference's counterpart is native (Nim); no GPU involved here.
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <functional>
#include <unordered_map>
#include <unordered_set>

#include "gs_common.h"

void gs_advise_huge(void *p, size_t bytes) {
    static const bool off = getenv("GAVISUNK_NO_THP") != nullptr;  // (experiment knob: huge-page faults compact memory when it is fragmented)
    if (!off) (void)madvise(p, bytes, MADV_HUGEPAGE);
}

namespace gs {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

// ---------------------------------------------------------------------------------- line reader
// Lines of a plain or gzip-compressed file.  A second thread inflates / reads one 1 MiB block ahead while the caller
// splits the previous block into lines (for .gz inputs inflating is the larger half of the work).
// Whole-buffer gzip decompression with libdeflate when the system has it (2-3x zlib's inflate); resolved at run time,
// so the library neither needs its header nor fails without it.
struct Deflate {
    void *(*alloc)() = nullptr;
    int (*gunzip)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    Deflate() {
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        gunzip = (int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *))dlsym(h, "libdeflate_gzip_decompress_ex");
        release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        if (!alloc || !gunzip || !release) alloc = nullptr;
    }
    bool ok() const { return alloc != nullptr; }
};
static const Deflate &deflate_lib() {
    static const Deflate d;
    return d;
}

// A .gz file of moderate size, inflated in one go: members one after the other (a plain `gzip` file has one, bgzip
// files many), the output buffer sized from the trailer of the last member and grown when that guess is short.
// Returns false (and leaves `out` unspecified) whenever anything is unusual: the caller then streams with zlib.
struct RawBuf {  // uninitialised storage (a 230 MB std::vector would be zeroed first)
    std::unique_ptr<char[]> p;
    size_t cap = 0, n = 0;
    void reserve_keep(size_t want) {
        if (want <= cap) return;
        std::unique_ptr<char[]> q(new char[want]);
        if (n) memcpy(q.get(), p.get(), n);
        p = std::move(q);
        cap = want;
    }
};

static bool inflate_whole_file(const char *path, RawBuf &out) {
    const Deflate &lib = deflate_lib();
    if (!lib.ok() || getenv("GAVISUNK_NO_LIBDEFLATE")) return false;
    // (only a regular file can be opened, measured and opened again: a FIFO or /dev/stdin is read once, by gzopen below, as the
    // reference's gzopen does)
    struct stat sb;
    if (stat(path, &sb) != 0 || !S_ISREG(sb.st_mode)) return false;
    FILE *fp = fopen(path, "rb");
    if (!fp) return false;
    std::vector<unsigned char> in;
    bool good = false;
    if (fseek(fp, 0, SEEK_END) == 0) {
        const long sz = ftell(fp);
        if (sz >= 18 && sz <= (1L << 30) && fseek(fp, 0, SEEK_SET) == 0) {
            in.resize((size_t)sz);
            good = fread(in.data(), 1, in.size(), fp) == in.size() && in[0] == 0x1f && in[1] == 0x8b;
        }
    }
    fclose(fp);
    if (!good) return false;
    const size_t n = in.size();
    const size_t isize = (size_t)in[n - 4] | ((size_t)in[n - 3] << 8) | ((size_t)in[n - 2] << 16) | ((size_t)in[n - 1] << 24);
    size_t cap = std::max<size_t>(isize, n) + 4096;  // exact for a single member below 4 GiB; a first guess otherwise
    if (cap > (3ull << 30)) return false;
    void *d = lib.alloc();
    if (!d) return false;
    out.reserve_keep(cap);
    size_t ipos = 0;
    good = true;
    while (ipos < n && good) {
        size_t used_in = 0, used_out = 0;
        const int rc = lib.gunzip(d, in.data() + ipos, n - ipos, out.p.get() + out.n, out.cap - out.n, &used_in, &used_out);
        if (rc == 0 && used_in > 0) {
            ipos += used_in;
            out.n += used_out;
        } else if (rc == 3 && out.cap < (3ull << 30)) {  // LIBDEFLATE_INSUFFICIENT_SPACE: grow, redo this member
            out.reserve_keep(std::min<size_t>(out.cap * 2, 3ull << 30));
        } else {
            good = false;
        }
    }
    lib.release(d);
    return good;
}

class LineReader {
   public:
    explicit LineReader(const char *path) {
        const size_t len = strlen(path);
        if (len > 3 && !strcmp(path + len - 3, ".gz") && inflate_whole_file(path, whole_)) {
            in_memory_ = true;
            cur_ = whole_.p.get();
            end_ = whole_.n;
            return;
        }
        f_ = gzopen(path, "rb");  // transparently reads plain files too
        if (f_) gzbuffer(f_, 1 << 20);
    }
    ~LineReader() {
        if (th_.joinable()) {
            {
                std::lock_guard<std::mutex> lk(m_);
                stop_ = true;
            }
            cv_.notify_all();
            th_.join();
        }
        if (f_) gzclose(f_);
    }
    LineReader(const LineReader &) = delete;
    LineReader &operator=(const LineReader &) = delete;
    bool ok() const { return f_ != nullptr || in_memory_; }
    // Next line without its '\n' (and without a trailing '\r'); false at EOF.
    bool next(std::string &line) {
        line.clear();
        bool any = false;
        for (;;) {
            if (pos_ == end_) {
                if (!refill()) {
                    if (any && !line.empty() && line.back() == '\r') line.pop_back();
                    return any;
                }
            }
            const char *p = cur_ + pos_;
            const char *nl = (const char *)memchr(p, '\n', end_ - pos_);
            any = true;
            if (nl) {
                line.append(p, nl - p);
                pos_ = (size_t)(nl - cur_) + 1;
                if (!line.empty() && line.back() == '\r') line.pop_back();
                return true;
            }
            line.append(p, end_ - pos_);
            pos_ = end_;
        }
    }

   private:
    struct Block {
        std::vector<char> data;
        size_t n = 0;
        bool full = false;
    };
    void produce() {  // blocks are filled and consumed in the same alternating order
        for (int i = 0;; i ^= 1) {
            Block &b = blk_[i];
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return !b.full || stop_; });
                if (stop_) return;
            }
            if (b.data.empty()) b.data.resize(1 << 20);
            const int n = gzread(f_, b.data.data(), (unsigned)b.data.size());
            {
                std::lock_guard<std::mutex> lk(m_);
                if (n <= 0) {
                    eof_ = true;
                } else {
                    b.n = (size_t)n;
                    b.full = true;
                }
            }
            cv_.notify_all();
            if (n <= 0) return;
        }
    }
    bool refill() {
        if (!f_) return false;  // also the end of an in-memory file
        if (!th_.joinable() && !eof_) th_ = std::thread([this] { produce(); });
        std::unique_lock<std::mutex> lk(m_);
        if (have_) {  // hand the block just consumed back
            blk_[cons_].full = false;
            cons_ ^= 1;
            have_ = false;
            cv_.notify_all();
        }
        cv_.wait(lk, [&] { return blk_[cons_].full || eof_; });
        if (!blk_[cons_].full) return false;
        cur_ = blk_[cons_].data.data();
        pos_ = 0;
        end_ = blk_[cons_].n;
        have_ = true;
        return true;
    }

    gzFile f_ = nullptr;
    RawBuf whole_;  // the inflated file, when it was inflated in one go
    bool in_memory_ = false;
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    Block blk_[2];
    int cons_ = 0;
    bool have_ = false, eof_ = false, stop_ = false;
    const char *cur_ = nullptr;
    size_t pos_ = 0, end_ = 0;
};

// ---------------------------------------------------------------------------------- u64 -> u32 map
class CodeMap {
   public:
    void reserve(uint64_t n) {
        uint64_t cap = 64;
        while (cap < 2 * n + 2) cap <<= 1;
        rehash(cap);
    }
    // returns reference to the value slot; `fresh` tells whether the key was new
    uint32_t &at(uint64_t key, bool &fresh) {
        if ((count_ + 1) * 2 > key_.size()) rehash(key_.empty() ? 64 : key_.size() * 2);
        uint64_t h = mix(key) & mask_;
        while (key_[h] != kEmpty && key_[h] != key) h = (h + 1) & mask_;
        fresh = key_[h] == kEmpty;
        if (fresh) {
            key_[h] = key;
            count_++;
        }
        return val_[h];
    }
    bool get(uint64_t key, uint32_t &v) const {
        if (key_.empty()) return false;
        uint64_t h = mix(key) & mask_;
        while (key_[h] != kEmpty) {
            if (key_[h] == key) {
                v = val_[h];
                return true;
            }
            h = (h + 1) & mask_;
        }
        return false;
    }

   private:
    static constexpr uint64_t kEmpty = ~0ULL;
    static uint64_t mix(uint64_t x) {
        x *= 0x9E3779B97F4A7C15ULL;
        return x ^ (x >> 29);
    }
    void rehash(uint64_t cap) {
        std::vector<uint64_t> ok;
        std::vector<uint32_t> ov;
        ok.swap(key_);
        ov.swap(val_);
        key_.assign(cap, kEmpty);
        val_.assign(cap, 0);
        mask_ = cap - 1;
        for (size_t i = 0; i < ok.size(); i++)
            if (ok[i] != kEmpty) {
                uint64_t h = mix(ok[i]) & mask_;
                while (key_[h] != kEmpty) h = (h + 1) & mask_;
                key_[h] = ok[i];
                val_[h] = ov[i];
            }
    }
    std::vector<uint64_t> key_;
    std::vector<uint32_t> val_;
    uint64_t mask_ = 0, count_ = 0;
};

static bool parse_i64(const char *b, const char *e, int64_t &v) {
    if (b == e) return false;
    bool neg = false;
    if (*b == '-' || *b == '+') {
        neg = *b == '-';
        b++;
    }
    if (b == e) return false;
    uint64_t x = 0;
    for (; b < e; b++) {
        if (*b < '0' || *b > '9') return false;
        if (x > (UINT64_MAX - 9) / 10) return false;  // (Nim's parseInt raises on overflow: a malformed row)
        x = x * 10 + (uint64_t)(*b - '0');
    }
    if (x > (uint64_t)INT64_MAX + (neg ? 1u : 0u)) return false;
    v = neg ? (int64_t)(0 - x) : (int64_t)x;
    return true;
}


// ---- the location table parsed by several threads (a plain file of some size: mapped, cut at line ends) ----------------------
// Every thread parses its stretch into arrays of its own, numbering contig names and (contig, group) pairs locally in
// first-seen order; the merge walks the threads in file order, so the global numbering is the one a single pass gives.
// -> 0 done, 1 a malformed line (*err_line = its 1-based number), -1 not applicable (gzip, small, cannot map): the caller's
// line-by-line loop then reads the file.
struct LocPart {
    HugeVec<uint32_t> cid, pid;  // per row: local contig id, local (contig, group) pair id
    HugeVec<int64_t> start, group;
    HugeVec<uint64_t> code;
    std::vector<std::string> contigs;                   // local contigs, first-seen order
    std::vector<std::pair<uint32_t, int64_t>> pairs;    // local (local contig, group) pairs, first-seen order
    uint64_t lines = 0, err_line = 0;
    bool err = false;
    // the parser's state (a part is read and parsed stretch by stretch)
    struct PairHash64 {
        size_t operator()(const std::pair<uint32_t, int64_t> &p) const {
            return std::hash<uint64_t>()(((uint64_t)p.first << 40) ^ (uint64_t)p.second * 0x9E3779B97F4A7C15ULL);
        }
    };
    std::unordered_map<std::string, uint32_t> contig_id;
    std::unordered_map<std::pair<uint32_t, int64_t>, uint32_t, PairHash64> pairmap;
    std::string name;
    bool have_last = false;
    uint32_t last_cid = 0, last_pid = 0;
    int64_t last_group = 0;
};
using PairHash64 = LocPart::PairHash64;
// whole lines [b, e) of the location table, appended to the part
static void parse_loc_part(const char *b, const char *e, LocPart &o) {
    auto &contig_id = o.contig_id;
    auto &pairmap = o.pairmap;
    std::string &name = o.name;
    bool &have_last = o.have_last;
    uint32_t &last_cid = o.last_cid, &last_pid = o.last_pid;
    int64_t &last_group = o.last_group;
    while (b < e) {
        // The common row in one pass over its bytes: the contig of the row before, a tab, digits, a tab, the k-mer, a tab, digits,
        // a line feed.  Anything else (another contig, a sign, CR LF, a fifth field, a malformed row ...) takes the general code
        // below, which decides what it is.
        if (have_last) {
            const size_t nlen0 = name.size();
            const char *q = b;
            if ((size_t)(e - q) > nlen0 + 1 && q[nlen0] == '\t' && !memcmp(q, name.data(), nlen0)) {
                q += nlen0 + 1;
                int64_t st = 0;
                const char *d0 = q;
                while (q < e && (unsigned)(*q - '0') <= 9u) st = st * 10 + (*q++ - '0');
                if (q > d0 && q - d0 <= 18 && q < e && *q == '\t') {
                    q++;
                    uint64_t c = 0;
                    while (q < e && *q != '\t' && *q != '\n') c = (c << 2) | kEncode.v[(uint8_t)*q++];
                    if (q < e && *q == '\t') {
                        q++;
                        int64_t g = 0;
                        const char *g0 = q;
                        while (q < e && (unsigned)(*q - '0') <= 9u) g = g * 10 + (*q++ - '0');
                        if (q > g0 && q - g0 <= 18 && q < e && *q == '\n') {
                            o.lines++;
                            uint32_t pid;
                            if (g == last_group) {
                                pid = last_pid;
                            } else {
                                auto key = std::make_pair(last_cid, g);
                                auto lt = pairmap.find(key);
                                if (lt == pairmap.end()) {
                                    pid = (uint32_t)o.pairs.size();
                                    pairmap.emplace(key, pid);
                                    o.pairs.push_back(key);
                                } else {
                                    pid = lt->second;
                                }
                                last_group = g;
                                last_pid = pid;
                            }
                            o.cid.push_back(last_cid);
                            o.pid.push_back(pid);
                            o.start.push_back(st);
                            o.group.push_back(g);
                            o.code.push_back(c);
                            b = q + 1;
                            continue;
                        }
                    }
                }
            }
        }
        const char *nl = (const char *)memchr(b, '\n', (size_t)(e - b));
        const char *le = nl ? nl : e;
        const char *next = nl ? nl + 1 : e;
        o.lines++;
        if (le > b && le[-1] == '\r') le--;
        if (le == b) {
            b = next;
            continue;
        }
        const char *f[4], *fe[4];
        int nf = 0;
        const char *s0 = b;
        for (const char *q = b;; q++) {
            if (q == le || *q == '\t') {
                if (nf < 4) {
                    f[nf] = s0;
                    fe[nf] = q;
                }
                nf++;
                s0 = q + 1;
                if (q == le) break;
            }
        }
        int64_t start, group;
        if (nf < 4 || !parse_i64(f[1], fe[1], start) || !parse_i64(f[3], fe[3], group)) {
            o.err = true;
            o.err_line = o.lines;
            return;
        }
        uint32_t cid;
        const size_t nlen = (size_t)(fe[0] - f[0]);
        if (have_last && nlen == name.size() && !memcmp(name.data(), f[0], nlen)) {
            cid = last_cid;
        } else {
            name.assign(f[0], nlen);
            auto it = contig_id.find(name);
            if (it == contig_id.end()) {
                cid = (uint32_t)o.contigs.size();
                contig_id.emplace(name, cid);
                o.contigs.push_back(name);
            } else {
                cid = it->second;
            }
        }
        uint32_t pid;
        if (have_last && cid == last_cid && group == last_group) {
            pid = last_pid;
        } else {
            auto key = std::make_pair(cid, group);
            auto lt = pairmap.find(key);
            if (lt == pairmap.end()) {
                pid = (uint32_t)o.pairs.size();
                pairmap.emplace(key, pid);
                o.pairs.push_back(key);
            } else {
                pid = lt->second;
            }
        }
        have_last = true;
        last_cid = cid;
        last_group = group;
        last_pid = pid;
        o.cid.push_back(cid);
        o.pid.push_back(pid);
        o.start.push_back(start);
        o.group.push_back(group);
        o.code.push_back(encode_kmer(f[2], (size_t)(fe[2] - f[2])));
        b = next;
    }
}

// The bytes [lo, hi) of a file (hi = a line start or the end of the file), handed to `fn` in stretches of whole lines: through a
// mapping while that is fast, else read with pread into a buffer of the calling thread.  (A mapping of a file the page cache
// holds in small pages costs a fault per 4 KiB touched, all of them under one lock of the address space: with 16 threads on a
// 4 GiB table that was up to 1.3 s of a 0.3 s parse.)
template <typename Fn>
static bool for_line_stretches(int fd, uint64_t lo, uint64_t hi, Fn &&fn) {
    if (hi <= lo) return true;
    // A mapping first: where the page cache holds the file in huge pages it costs next to nothing (a 4 GiB table comes in at
    // 15 GB/s over 16 threads).  The first 32 MiB are timed; a thread that sees less than 400 MB/s there -- small pages: a fault
    // per 4 KiB, all threads of the process behind one lock -- reads the rest with pread.
    {
        const uint64_t page = 4096, m_lo = lo / page * page;
        void *m = mmap(nullptr, (size_t)(hi - m_lo), PROT_READ, MAP_PRIVATE, fd, (off_t)m_lo);
        if (m != MAP_FAILED) {
            const char *b = static_cast<const char *>(m) + (lo - m_lo), *e = b + (hi - lo);
            const size_t probe = (size_t)32 << 20;
            const char *cut = e;
            if ((size_t)(e - b) > 2 * probe) {
                const void *q = memrchr(b, '\n', probe);
                cut = q ? static_cast<const char *>(q) + 1 : b;
            }
            const auto t0 = std::chrono::steady_clock::now();
            if (cut > b) fn(b, cut);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const bool fast = cut == e || (double)(cut - b) / std::max(dt, 1e-9) >= 400e6;
            if (fast && cut < e) fn(cut, e);
            munmap(m, (size_t)(hi - m_lo));
            if (fast) return true;
            lo += (uint64_t)(cut - b);
        }
    }
    std::vector<char> buf((size_t)4 << 20);
    size_t have = 0;  // bytes of an unfinished line at the front of buf
    uint64_t at = lo;
    while (at < hi || have) {
        if (at < hi) {
            if (buf.size() - have < ((size_t)1 << 20)) buf.resize(buf.size() * 2);  // (a line longer than the buffer)
            const size_t want = (size_t)std::min<uint64_t>(buf.size() - have, hi - at);
            const ssize_t r = pread(fd, buf.data() + have, want, (off_t)at);
            if (r <= 0) return false;
            at += (uint64_t)r;
            have += (size_t)r;
        }
        size_t usable = have;
        if (at < hi) {  // more to come: up to the last line end in the buffer
            const void *q = memrchr(buf.data(), '\n', have);
            if (!q) continue;
            usable = (size_t)(static_cast<const char *>(q) - buf.data()) + 1;
        }
        fn(buf.data(), buf.data() + usable);
        memmove(buf.data(), buf.data() + usable, have - usable);
        have -= usable;
    }
    return true;
}
// the first line start at or after `q` (or the end of the file), looked for with small reads
static uint64_t next_line_start(int fd, uint64_t q, uint64_t n) {
    char tmp[4096];
    while (q < n) {
        const ssize_t r = pread(fd, tmp, (size_t)std::min<uint64_t>(sizeof tmp, n - q), (off_t)q);
        if (r <= 0) return n;
        const void *nl = memchr(tmp, '\n', (size_t)r);
        if (nl) return q + (uint64_t)(static_cast<const char *>(nl) - tmp) + 1;
        q += (uint64_t)r;
    }
    return n;
}
// the k-mer list (one k-mer per line) parsed by several threads: -> false when not applicable (gzip, small, cannot map)
static bool parse_db_parallel(const char *path, HugeVec<uint64_t> &codes, int &k) {
    if (getenv("GAVISUNK_LOC_THREADS") && atoi(getenv("GAVISUNK_LOC_THREADS")) <= 1) return false;
    {  // (a FIFO is read once, by the serial reader: it must not be opened and closed here first)
        struct stat st0;
        if (stat(path, &st0) != 0 || !S_ISREG(st0.st_mode)) return false;
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < (4 << 20)) {
        close(fd);
        return false;
    }
    const size_t n = (size_t)st.st_size;
    struct FdCloser {
        int fd;
        ~FdCloser() { close(fd); }
    } fdc{fd};
    unsigned char magic[2] = {0, 0};
    if (pread(fd, magic, 2, 0) != 2 || (magic[0] == 0x1f && magic[1] == 0x8b)) return false;
    unsigned T = std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 2));  // (the location table is parsed beside this)
    if (const char *e = getenv("GAVISUNK_LOC_THREADS")) T = (unsigned)std::max(1, atoi(e));
    T = (unsigned)std::min<size_t>(T, n >> 20);
    std::vector<uint64_t> cut(T + 1, n);
    cut[0] = 0;
    for (unsigned t = 1; t < T; t++) cut[t] = next_line_start(fd, std::max<uint64_t>((uint64_t)n * t / T, cut[t - 1]), n);
    std::vector<HugeVec<uint64_t>> parts(T);
    std::vector<char> okv(T, 1);
    auto work = [&](unsigned t) {
        parts[t].reserve((size_t)(cut[t + 1] - cut[t]) / 16 + 16);
        okv[t] = for_line_stretches(fd, cut[t], cut[t + 1], [&](const char *b, const char *e) {
            while (b < e) {  // every line is a k-mer, the empty ones too (the reference encodes whatever `lines` yields)
                uint64_t c = 0;
                const char *q = b;
                while (q < e && *q != '\n') c = (c << 2) | kEncode.v[(uint8_t)*q++];  // (encoded while the line end is looked for)
                if (q > b && q[-1] == '\r') c = encode_kmer(b, (size_t)(q - 1 - b));  // CR LF: without the CR
                parts[t].push_back(c);
                b = q < e ? q + 1 : e;
            }
        }) ? 1 : 0;
    };
    {
        gs_parallel_for(T, work);
    }
    for (char o : okv)
        if (!o) return false;
    {  // k = len(first line), kmerpos_annot3.nim:28-38
        char head[256];
        const ssize_t r = pread(fd, head, sizeof head, 0);
        const char *nl = r > 0 ? (const char *)memchr(head, '\n', (size_t)r) : nullptr;
        if (!nl && (size_t)r == sizeof head) return false;  // (no k-mer is that long: let the line reader say what it is)
        const char *le = nl ? nl : head + std::max<ssize_t>(r, 0);
        if (le > head && le[-1] == '\r') le--;
        k = (int)(le - head);
    }
    std::vector<size_t> at(T + 1, 0);
    for (unsigned t = 0; t < T; t++) at[t + 1] = at[t] + parts[t].size();
    codes.resize(at[T]);  // (uninitialised: every thread copies its part, and frees it)
    {
        auto place = [&](unsigned t) {
            if (!parts[t].empty()) memcpy(codes.data() + at[t], parts[t].data(), parts[t].size() * sizeof(uint64_t));
            HugeVec<uint64_t>().swap(parts[t]);
        };
        gs_parallel_for(T, place);
    }
    return true;
}

static int parse_loc_parallel(const char *path, gs_sunkdb *d, HugeVec<uint64_t> &row_code, uint64_t *err_line) {
    if (getenv("GAVISUNK_LOC_THREADS") && atoi(getenv("GAVISUNK_LOC_THREADS")) <= 1) return -1;
    {  // (a FIFO is read once, by the serial reader: it must not be opened and closed here first)
        struct stat st0;
        if (stat(path, &st0) != 0 || !S_ISREG(st0.st_mode)) return -1;
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < (4 << 20)) {
        close(fd);
        return -1;
    }
    const size_t n = (size_t)st.st_size;
    struct FdCloser {
        int fd;
        ~FdCloser() { close(fd); }
    } fdc{fd};
    unsigned char magic[2] = {0, 0};
    if (pread(fd, magic, 2, 0) != 2 || (magic[0] == 0x1f && magic[1] == 0x8b)) return -1;  // gzip: the line reader inflates it
    unsigned T = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("GAVISUNK_LOC_THREADS")) T = (unsigned)std::max(1, atoi(e));
    T = (unsigned)std::min<size_t>(T, n >> 20);
    std::vector<uint64_t> cut(T + 1, n);
    cut[0] = 0;
    for (unsigned t = 1; t < T; t++) cut[t] = next_line_start(fd, std::max<uint64_t>((uint64_t)n * t / T, cut[t - 1]), n);  // the first line start at or after n t / T
    std::vector<LocPart> parts(T);
    std::vector<char> okv(T, 1);
    auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tp0 = tnow();
    {
        auto work = [&](unsigned t) {
            LocPart &o = parts[t];
            const size_t guess = (size_t)(cut[t + 1] - cut[t]) / 40 + 16;
            o.cid.reserve(guess), o.pid.reserve(guess), o.start.reserve(guess), o.group.reserve(guess), o.code.reserve(guess);
            okv[t] = for_line_stretches(fd, cut[t], cut[t + 1], [&](const char *b, const char *e) {
                if (!o.err) parse_loc_part(b, e, o);
            }) ? 1 : 0;
        };
        gs_parallel_for(T, work);
    }
    for (char o : okv)
        if (!o) return -1;
    const double tp1 = tnow();
    const double tp2 = tp1;
    uint64_t lines_before = 0;
    for (unsigned t = 0; t < T; t++) {
        if (parts[t].err) {
            *err_line = lines_before + parts[t].err_line;
            return 1;
        }
        lines_before += parts[t].lines;
    }
    // merge, in file order: global contig ids, then global (contig, group) ids
    std::unordered_map<std::string, uint32_t> contig_id;
    std::unordered_map<std::pair<uint32_t, int64_t>, uint32_t, PairHash64> locmap;
    std::vector<std::vector<uint32_t>> cmap(T), pmap(T);
    std::vector<size_t> row0(T + 1, 0);
    for (unsigned t = 0; t < T; t++) {
        LocPart &o = parts[t];
        cmap[t].resize(o.contigs.size());
        for (size_t c = 0; c < o.contigs.size(); c++) {
            auto it = contig_id.find(o.contigs[c]);
            if (it == contig_id.end()) {
                it = contig_id.emplace(o.contigs[c], (uint32_t)d->contigs.size()).first;
                d->contigs.push_back(o.contigs[c]);
            }
            cmap[t][c] = it->second;
        }
        pmap[t].resize(o.pairs.size());
        for (size_t q = 0; q < o.pairs.size(); q++) {
            const auto key = std::make_pair(cmap[t][o.pairs[q].first], o.pairs[q].second);
            auto it = locmap.find(key);
            if (it == locmap.end()) it = locmap.emplace(key, (uint32_t)locmap.size()).first;
            pmap[t][q] = it->second;
        }
        row0[t + 1] = row0[t] + o.cid.size();
    }
    const size_t nrows = row0[T];
    d->row_contig.resize(nrows);
    d->row_start.resize(nrows);
    d->row_group.resize(nrows);
    d->row_loc.resize(nrows);
    row_code.resize(nrows);
    {
        auto place = [&](unsigned t) {
            LocPart &o = parts[t];
            const size_t b = row0[t];
            for (size_t i = 0; i < o.cid.size(); i++) {
                d->row_contig[b + i] = cmap[t][o.cid[i]];
                d->row_loc[b + i] = pmap[t][o.pid[i]];
            }
            if (!o.cid.empty()) {
                memcpy(d->row_start.data() + b, o.start.data(), o.start.size() * sizeof(int64_t));
                memcpy(d->row_group.data() + b, o.group.data(), o.group.size() * sizeof(int64_t));
                memcpy(row_code.data() + b, o.code.data(), o.code.size() * sizeof(uint64_t));
            }
            parts[t] = LocPart();  // (640 MB per 2e7 rows go back to the system here, on this thread)
        };
        gs_parallel_for(T, place);
    }
    if (getenv("GAVISUNK_TIMING"))
        fprintf(stderr, "[sunkdb] loc: %u threads, parse %.3f s, munmap %.3f s, merge + place %.3f s\n", T, tp1 - tp0, tp2 - tp1, tnow() - tp2);
    return 0;
}
}  // namespace gs

using namespace gs;

extern "C" {

const char *gs_last_error(void) { return g_err.c_str(); }
const char *gs_version(void) { return "gavisunk_amd 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------- sunkdb
// The join of the parsed k-mer list with the location table, on the host, for callers that ask for the key arrays (the table
// itself is joined on the device: gs_table_create_from_db).  Distinct db codes in first-seen order (a k-mer listed twice is one
// key: the reference's HashSet, kmerpos_annot3.nim:20-38), each with the LAST loc row that carries it (:68), GS_MISSING without
// one.  The hashing is split over threads by hash class; rows stay in file order inside a class.
static void join_host(gs_sunkdb *d) {
    std::lock_guard<std::mutex> lock(d->join_mu);
    if (d->joined) return;
    constexpr unsigned kMapParts = 8;
    auto map_part = [](uint64_t code) { return (unsigned)((code * 0xD6E8FEB86659FD93ULL) >> 61); };
    auto run_parts = [&](size_t n, const std::function<void(unsigned)> &fn) {
        if (n < (1u << 16)) {
            for (unsigned p2 = 0; p2 < kMapParts; p2++) fn(p2);
        } else {
            gs_parallel_for(kMapParts, fn);
        }
    };
    // distinct db codes, first occurrences, file order
    HugeVec<uint64_t> codes = std::move(d->db_codes);
    const size_t n = codes.size();
    std::vector<uint8_t> is_first(n, 0);
    run_parts(n, [&](unsigned part) {
        CodeMap seen;
        seen.reserve(n / 8 + n / 32 + 16);
        for (size_t i = 0; i < n; i++)
            if (map_part(codes[i]) == part) {
                bool fresh;
                seen.at(codes[i], fresh);
                is_first[i] = fresh;
            }
    });
    size_t nk = 0;
    for (size_t i = 0; i < n; i++)
        if (is_first[i]) codes[nk++] = codes[i];
    d->keys.assign(codes.begin(), codes.begin() + nk);
    HugeVec<uint64_t>().swap(codes);
    // loc code -> last row with that code
    CodeMap rowmap[kMapParts];
    const HugeVec<uint64_t> &row_code = d->row_code;
    const size_t nrows = row_code.size();
    run_parts(nrows, [&](unsigned part) {
        CodeMap &m = rowmap[part];
        m.reserve(nrows / kMapParts + nrows / (4 * kMapParts) + 16);
        for (size_t i = 0; i < nrows; i++)
            if (map_part(row_code[i]) == part) {
                bool fresh;
                m.at(row_code[i], fresh) = (uint32_t)i;  // later row overwrites (kmerpos_annot3.nim:68)
            }
    });
    d->key_row.resize(nk);
    d->key_loc.resize(nk);
    auto join_range = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            uint32_t row;
            if (rowmap[map_part(d->keys[i])].get(d->keys[i], row)) {
                d->key_row[i] = row;
                d->key_loc[i] = d->row_loc[row];
            } else {
                d->key_row[i] = GS_MISSING;
                d->key_loc[i] = GS_MISSING;
            }
        }
    };
    const unsigned nthr = nk < (1u << 16) ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    gs_parallel_for(nthr, [&](unsigned t) { join_range(nk * t / nthr, nk * (t + 1) / nthr); });
    HugeVec<uint64_t>().swap(d->row_code);
    d->joined = true;
}
static const gs_sunkdb *joined(const gs_sunkdb *db) {
    if (!db->joined) join_host(const_cast<gs_sunkdb *>(db));
    return db;
}

int gs_sunkdb_load(const char *db_path, const char *loc_path, gs_sunkdb **out) {
    if (!db_path || !loc_path || !out) {
        set_error("gs_sunkdb_load: null argument");
        return GS_E_ARG;
    }
    *out = nullptr;
    auto *d = new gs_sunkdb();
    std::string line;
    auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = tnow();
    // The two files are independent until the join at the end: the k-mer list is read by a second thread while this one
    // reads the location table (both are dominated by their hash-map inserts).
    int db_rc = GS_OK;
    bool db_open = true;
    std::thread db_thread([&]() {
        std::string l;
        HugeVec<uint64_t> codes;  // local while growing: the two threads must not share cache lines of *d
        int k = 0;
        if (!parse_db_parallel(db_path, codes, k)) {
            LineReader r(db_path);
            if (!r.ok()) {
                db_open = false;
                db_rc = GS_E_IO;
                return;
            }
            bool first = true;
            while (r.next(l)) {
                if (first) {  // k = len(first line), kmerpos_annot3.nim:28-38
                    k = (int)l.size();
                    first = false;
                }
                codes.push_back(encode_kmer(l.data(), l.size()));
            }
        }
        d->k = k;
        d->db_codes = std::move(codes);  // (duplicates and all: the join makes the key set)
        if (k < 1 || k > 32) db_rc = GS_E_ARG;
    });
    struct Joiner {  // the thread is joined on every way out
        std::thread &t;
        ~Joiner() {
            if (t.joinable()) t.join();
        }
    } joiner{db_thread};
    // the reference reads the k-mer list first: its errors win over errors in the location table
    auto db_error = [&]() {
        if (!db_open) set_error("cannot open the file %s", db_path);
        else set_error("%s: k = %d is outside 1..32 (2 bits per base in a uint64)", db_path, d->k);
        return db_rc;
    };
    HugeVec<uint64_t> row_code;
    uint64_t par_err_line = 0;
    const int par = parse_loc_parallel(loc_path, d, row_code, &par_err_line);
    const double t_parsed = tnow();
    if (par == 1) {
        db_thread.join();
        const int rc = db_rc != GS_OK ? db_error()
                                      : (set_error("%s:%llu: expected `chrom<TAB>start<TAB>kmer<TAB>group`", loc_path, (unsigned long long)par_err_line),
                                         GS_E_IO);
        delete d;
        return rc;
    }
    if (par < 0) {
        LineReader r(loc_path);
        if (!r.ok()) {
            db_thread.join();
            const int rc = db_rc != GS_OK ? db_error() : (set_error("cannot open the file %s", loc_path), GS_E_IO);
            delete d;
            return rc;
        }
        std::unordered_map<std::string, uint32_t> contig_id;
        struct PairHash {
            size_t operator()(const std::pair<uint32_t, int64_t> &p) const {
                return std::hash<uint64_t>()(((uint64_t)p.first << 40) ^ (uint64_t)p.second * 0x9E3779B97F4A7C15ULL);
            }
        };
        std::unordered_map<std::pair<uint32_t, int64_t>, uint32_t, PairHash> locmap;
        uint64_t lineno = 0;
        std::string name;
        bool have_last = false;
        uint32_t last_cid = 0, last_lid = 0;
        int64_t last_group = 0;
        while (r.next(line)) {
            lineno++;
            if (line.empty()) continue;
            const char *p = line.data(), *e = p + line.size();
            const char *f[4];
            const char *fe[4];
            int nf = 0;
            const char *s = p;
            for (const char *q = p;; q++) {
                if (q == e || *q == '\t') {
                    if (nf < 4) {
                        f[nf] = s;
                        fe[nf] = q;
                    }
                    nf++;
                    s = q + 1;
                    if (q == e) break;
                }
            }
            int64_t start, group;
            if (nf < 4 || !parse_i64(f[1], fe[1], start) || !parse_i64(f[3], fe[3], group)) {
                db_thread.join();
                const int rc = db_rc != GS_OK ? db_error()
                                              : (set_error("%s:%llu: expected `chrom<TAB>start<TAB>kmer<TAB>group`", loc_path,
                                                           (unsigned long long)lineno),
                                                 GS_E_IO);
                delete d;
                return rc;
            }
            // consecutive rows nearly always share the contig and very often the group: skip the hash lookups then
            uint32_t cid;
            const size_t nlen = (size_t)(fe[0] - f[0]);
            if (have_last && nlen == name.size() && !memcmp(name.data(), f[0], nlen)) {
                cid = last_cid;
            } else {
                name.assign(f[0], nlen);
                auto it = contig_id.find(name);
                if (it == contig_id.end()) {
                    cid = (uint32_t)d->contigs.size();
                    contig_id.emplace(name, cid);
                    d->contigs.push_back(name);
                } else {
                    cid = it->second;
                }
            }
            const uint32_t row = (uint32_t)d->row_contig.size();
            d->row_contig.push_back(cid);
            d->row_start.push_back(start);
            d->row_group.push_back(group);
            uint32_t lid;
            if (have_last && cid == last_cid && group == last_group) {
                lid = last_lid;
            } else {
                auto key = std::make_pair(cid, group);
                auto lt = locmap.find(key);
                if (lt == locmap.end()) {
                    lid = (uint32_t)locmap.size();
                    locmap.emplace(key, lid);
                } else {
                    lid = lt->second;
                }
            }
            have_last = true;
            last_cid = cid;
            last_group = group;
            last_lid = lid;
            d->row_loc.push_back(lid);
            (void)row;
            row_code.push_back(encode_kmer(f[2], fe[2] - f[2]));
        }
    }
    const double t_loc = tnow();
    db_thread.join();
    if (db_rc != GS_OK) {
        const int rc = db_error();
        delete d;
        return rc;
    }
    d->row_code = std::move(row_code);
    if (getenv("GAVISUNK_TIMING"))
        fprintf(stderr, "[sunkdb] loc text %.3f s, +db wait %.3f s (the join is made by whoever needs it)\n", t_parsed - t_start, tnow() - t_loc);
    *out = d;
    return GS_OK;
}

void gs_sunkdb_free(gs_sunkdb *db) { delete db; }
int gs_sunkdb_k(const gs_sunkdb *db) { return db->k; }
// (the four below make the host join on first use; a handle from gs_sidecar_open has no keys: they live in its table)
uint64_t gs_sunkdb_n_keys(const gs_sunkdb *db) { return joined(db)->keys.size(); }
const uint64_t *gs_sunkdb_keys(const gs_sunkdb *db) { return joined(db)->keys.data(); }
const uint32_t *gs_sunkdb_key_row(const gs_sunkdb *db) { return joined(db)->key_row.data(); }
const uint32_t *gs_sunkdb_key_loc(const gs_sunkdb *db) { return joined(db)->key_loc.data(); }
uint64_t gs_sunkdb_n_rows(const gs_sunkdb *db) { return db->row_contig.size(); }
const uint32_t *gs_sunkdb_row_contig(const gs_sunkdb *db) { return db->row_contig.data(); }
const int64_t *gs_sunkdb_row_start(const gs_sunkdb *db) { return db->row_start.data(); }
const int64_t *gs_sunkdb_row_group(const gs_sunkdb *db) { return db->row_group.data(); }
const uint32_t *gs_sunkdb_row_loc(const gs_sunkdb *db) { return db->row_loc.data(); }
uint32_t gs_sunkdb_n_contigs(const gs_sunkdb *db) { return (uint32_t)db->contigs.size(); }
const char *gs_sunkdb_contig_name(const gs_sunkdb *db, uint32_t c) {
    return c < db->contigs.size() ? db->contigs[c].c_str() : nullptr;
}

}  // extern "C"

// ------------------------------------------------------------------------------------- fastx
struct gs_fastx {
    LineReader reader;
    std::string pending;  // a header line already consumed from the stream
    bool have_pending = false;
    bool eof = false;
    explicit gs_fastx(const char *p) : reader(p) {}
};

namespace gs {

// Appends one read (ASCII, possibly in several pieces) to a batch, packing as it goes.
class Packer {
   public:
    explicit Packer(gs_batch *b) : b_(b) {}
    void begin(const char *name, size_t name_len) {
        b_->name_off.push_back(b_->names.size());
        b_->names.insert(b_->names.end(), name, name + name_len);
        b_->names.push_back('\0');
        cur_ = 0;
        nb_ = 0;
        len_ = 0;
    }
    void add(const char *s, size_t n) {
        for (size_t i = 0; i < n; i++) {
            cur_ = (cur_ << 2) | enc((uint8_t)s[i]);
            if (++nb_ == 32) {
                b_->words.push_back(cur_);
                cur_ = 0;
                nb_ = 0;
            }
        }
        len_ += n;
    }
    void end() {
        if (nb_) b_->words.push_back(cur_ << (2 * (32 - nb_)));
        b_->words.push_back(0);  // >= 1 zero word after every read ('A' padding, Q6)
        b_->word_off.push_back(b_->words.size());
        b_->len.push_back((uint32_t)len_);
        b_->bases += len_;
    }

   private:
    gs_batch *b_;
    uint64_t cur_ = 0;
    int nb_ = 0;
    size_t len_ = 0;
};

}  // namespace gs

extern "C" {

int gs_fastx_open(const char *path, gs_fastx **out) {
    if (!path || !out) {
        set_error("gs_fastx_open: null argument");
        return GS_E_ARG;
    }
    auto *fx = new gs_fastx(path);
    if (!fx->reader.ok()) {
        set_error("cannot open the file %s", path);
        delete fx;
        *out = nullptr;
        return GS_E_IO;
    }
    *out = fx;
    return GS_OK;
}

void gs_fastx_close(gs_fastx *fx) { delete fx; }

int gs_fastx_next_batch(gs_fastx *fx, uint64_t max_bases, uint32_t max_reads, gs_batch **out) {
    if (!fx || !out) {
        set_error("gs_fastx_next_batch: null argument");
        return GS_E_ARG;
    }
    *out = nullptr;
    if (fx->eof && !fx->have_pending) return GS_OK;
    auto *b = new gs_batch();
    b->word_off.push_back(0);
    Packer pk(b);
    std::string line;
    uint32_t n = 0;
    while (n < max_reads && (n == 0 || b->bases < max_bases)) {
        // locate a header: kseq scans for the next '>' or '@'
        std::string hdr;
        bool got = false;
        if (fx->have_pending) {
            hdr.swap(fx->pending);
            fx->have_pending = false;
            got = true;
        } else {
            while (fx->reader.next(line)) {
                size_t p = line.find_first_of(">@");
                if (p != std::string::npos) {
                    hdr = line.substr(p);
                    got = true;
                    break;
                }
            }
        }
        if (!got) {
            fx->eof = true;
            break;
        }
        // name = up to the first whitespace
        size_t ne = 1;
        while (ne < hdr.size() && !isspace((unsigned char)hdr[ne])) ne++;
        pk.begin(hdr.data() + 1, ne - 1);
        size_t seqlen = 0;
        char stop = 0;
        while (fx->reader.next(line)) {
            if (!line.empty() && (line[0] == '>' || line[0] == '@' || line[0] == '+')) {
                stop = line[0];
                break;
            }
            pk.add(line.data(), line.size());
            seqlen += line.size();
        }
        pk.end();
        n++;
        if (stop == '>' || stop == '@') {
            fx->pending = line;
            fx->have_pending = true;
        } else if (stop == '+') {  // FASTQ: qualities span as many bytes as the sequence
            size_t q = 0;
            while (q < seqlen && fx->reader.next(line)) q += line.size();
            if (seqlen > 0 && q < seqlen) {
                fx->eof = true;  // truncated record: kseq stops here
                break;
            }
        } else {
            fx->eof = true;
            break;
        }
    }
    if (n == 0) {
        delete b;
        return GS_OK;
    }
    *out = b;
    return GS_OK;
}

int gs_batch_from_ascii(const uint8_t *seq, const uint64_t *off, uint32_t n_reads, gs_batch **out) {
    if (!off || !out || (!seq && n_reads && off[n_reads] > 0)) {
        set_error("gs_batch_from_ascii: null argument");
        return GS_E_ARG;
    }
    auto *b = new gs_batch();
    b->word_off.push_back(0);
    b->words.reserve(off[n_reads] / 32 + 2ull * n_reads + 2);
    Packer pk(b);
    char nm[32];
    for (uint32_t r = 0; r < n_reads; r++) {
        int l = snprintf(nm, sizeof nm, "%u", r);
        pk.begin(nm, (size_t)l);
        pk.add((const char *)seq + off[r], off[r + 1] - off[r]);
        pk.end();
    }
    *out = b;
    return GS_OK;
}

void gs_batch_free(gs_batch *b) { delete b; }
uint32_t gs_batch_n_reads(const gs_batch *b) { return (uint32_t)b->len.size(); }
uint64_t gs_batch_n_bases(const gs_batch *b) { return b->bases; }
const uint64_t *gs_batch_words(const gs_batch *b) { return b->words.data(); }
uint64_t gs_batch_n_words(const gs_batch *b) { return b->words.size(); }
const uint64_t *gs_batch_word_off(const gs_batch *b) { return b->word_off.data(); }
const uint32_t *gs_batch_len(const gs_batch *b) { return b->len.data(); }
const char *gs_batch_name(const gs_batch *b, uint32_t r) {
    return r < b->name_off.size() ? b->names.data() + b->name_off[r] : nullptr;
}
int gs_batch_names(const gs_batch *b, const char **base, const uint64_t **off, uint64_t *bytes) {
    if (!b || !base || !off || !bytes) {
        set_error("gs_batch_names: null argument");
        return GS_E_ARG;
    }
    *base = b->names.data();
    *off = b->name_off.data();
    *bytes = b->names.size();
    return GS_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------- diag_filter_step2
extern "C" {

// diag_filter_step2.nim:16-22 (best contig per read, a later line for the same read overwrites) and :41-64 (rname_prev starts
// as "string"; a row passes when its contig is its read's best contig)
int gs_diag_filter_step2_files(const char *sunkpos_path, const char *diag_path, const char *out_path, uint64_t *rows_out) {
    if (!sunkpos_path || !diag_path) {
        set_error("gs_diag_filter_step2_files: null argument");
        return GS_E_ARG;
    }
    std::unordered_map<std::string, std::string> best;
    std::string line;
    {
        LineReader rd(diag_path);
        if (!rd.ok()) {
            set_error("cannot open the file %s", diag_path);
            return GS_E_IO;
        }
        while (rd.next(line)) {
            const size_t t1 = line.find('\t');
            if (t1 == std::string::npos) continue;  // fewer than two fields
            size_t t2 = line.find('\t', t1 + 1);
            if (t2 == std::string::npos) t2 = line.size();
            best[line.substr(0, t1)] = line.substr(t1 + 1, t2 - t1 - 1);
        }
    }
    LineReader rd(sunkpos_path);
    if (!rd.ok()) {
        set_error("cannot open the file %s", sunkpos_path);
        return GS_E_IO;
    }
    FILE *out = out_path ? fopen(out_path, "wb") : stdout;
    if (!out) {
        set_error("cannot open the file %s for writing", out_path);
        return GS_E_IO;
    }
    std::vector<char> obuf(1 << 20);
    if (out_path) setvbuf(out, obuf.data(), _IOFBF, obuf.size());
    std::string prev = "string", cur, name;
    uint64_t rows = 0;
    while (rd.next(line)) {
        if (line.empty()) continue;
        const size_t t1 = line.find('\t');
        name.assign(line, 0, t1 == std::string::npos ? line.size() : t1);
        if (name != prev) {
            auto it = best.find(name);
            cur = it == best.end() ? std::string() : it->second;
            prev = name;
        }
        if (t1 == std::string::npos) continue;
        const size_t t2 = line.find('\t', t1 + 1);
        if (t2 == std::string::npos) continue;  // fewer than three fields
        size_t t3 = line.find('\t', t2 + 1);
        if (t3 == std::string::npos) t3 = line.size();
        if (line.compare(t2 + 1, t3 - t2 - 1, cur) == 0) {
            fwrite(line.data(), 1, line.size(), out);
            fputc('\n', out);
            rows++;
        }
    }
    if (out_path)
        fclose(out);
    else
        fflush(out);
    if (rows_out) *rows_out = rows;
    return GS_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------- output rows
// `name \t i \t chrom \t start \t group \n` (kmerpos_annot3.nim:93), formatted and written natively: a genome-scale pass emits
// 1e7 .. 1e8 rows, far beyond what a per-row Python format survives.
// (the pinned staging of the device formatter is gs_format.hip's: this default stands in where the host parsers are built alone --
// tools/gz_asan.sh)
__attribute__((weak)) void rowwriter_release_pinned(gs_rowwriter *) {}

static inline char *put_u64(char *p, uint64_t v) {
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
static inline char *put_i64(char *p, int64_t v) {
    if (v < 0) {
        *p++ = '-';
        return put_u64(p, (uint64_t)(-(v + 1)) + 1u);
    }
    return put_u64(p, (uint64_t)v);
}

extern "C" {

int gs_rowwriter_open(const char *path, gs_rowwriter **out) {
    if (!path || !out) {
        set_error("gs_rowwriter_open: null argument");
        return GS_E_ARG;
    }
    FILE *f = fopen(path, "wb");
    if (!f) {
        set_error("cannot create the file %s", path);
        return GS_E_IO;
    }
    gs_rowwriter *w = new gs_rowwriter;
    w->f = f;
    w->buf.reserve(8u << 20);
    *out = w;
    return GS_OK;
}
// n rows of one batch: read names from (names_base + name_off[read], name_len[read]) -- the text block a device-parsed batch came
// from (gs_textindex_*) -- or, when name_off is NULL, from `batch` (gs_batch_name)
int gs_rowwriter_rows(gs_rowwriter *w, const gs_hit *hits, uint64_t n, const gs_sunkdb *db, const uint8_t *names_base,
                      const uint64_t *name_off, const uint32_t *name_len, const gs_batch *batch) {
    if (!w || !db || (n && !hits) || (!name_off && !batch) || (name_off && (!names_base || !name_len))) {
        set_error("gs_rowwriter_rows: null argument");
        return GS_E_ARG;
    }
    const uint64_t n_rows = db->row_contig.size();
    for (uint64_t i = 0; i < n; i++) {
        const gs_hit h = hits[i];
        if (h.payload >= n_rows) {
            set_error("gs_rowwriter_rows: payload %u is no loc row", h.payload);
            return GS_E_ARG;
        }
        const char *nm;
        size_t nl;
        if (name_off) {
            nm = reinterpret_cast<const char *>(names_base) + name_off[h.read];
            nl = name_len[h.read];
        } else {
            if (h.read >= batch->len.size()) {
                set_error("gs_rowwriter_rows: read %u is not in the batch", h.read);
                return GS_E_ARG;
            }
            nm = batch->names.data() + batch->name_off[h.read];
            nl = strlen(nm);
        }
        const std::string &cn = db->contigs[db->row_contig[h.payload]];
        const size_t o = w->buf.size();
        w->buf.resize(o + nl + cn.size() + 80);
        char *p = w->buf.data() + o;
        memcpy(p, nm, nl);
        p += nl;
        *p++ = '\t';
        p = put_u64(p, h.pos);
        *p++ = '\t';
        memcpy(p, cn.data(), cn.size());
        p += cn.size();
        *p++ = '\t';
        p = put_i64(p, db->row_start[h.payload]);
        *p++ = '\t';
        p = put_i64(p, db->row_group[h.payload]);
        *p++ = '\n';
        w->buf.resize((size_t)(p - w->buf.data()));
        if (w->buf.size() > (7u << 20) && !w->flush()) {
            set_error("gs_rowwriter_rows: write failed");
            return GS_E_IO;
        }
    }
    w->rows += n;
    return GS_OK;
}
// `name \t len \n` (workflow/src/rlen.nim:13-14) for every record of a block of text that holds WHOLE records in one of the regular
// forms -- FASTA (a '>' line, then sequence lines up to the next '>' line) or four-line FASTQ ('@' line, sequence, '+' line,
// qualities as long as the sequence) -- appended to the writer's file.  The block is only scanned (memchr): no packing, no copy of
// the bases.  Whatever the block-free reader (gs_fastx_*, kseq's rules) might read differently -- a sequence line that starts with
// '>', '@' or '+', an empty sequence, a quality line of another length, text in front of the first header -- is not guessed at:
// GS_E_IRREGULAR, and the caller reads the file from its start with gs_fastx_*.  *n_records = records of this block.
int gs_rowwriter_rlen(gs_rowwriter *w, const uint8_t *text, uint64_t n, uint64_t *n_records) {
    if (!w || (n && !text)) {
        set_error("gs_rowwriter_rlen: null argument");
        return GS_E_ARG;
    }
    if (n_records) *n_records = 0;
    const char *p = reinterpret_cast<const char *>(text), *const e = p + n;
    // a line [p, le) without its '\n' and a trailing '\r'; q = start of the next line
    auto line = [&](const char *from, const char *&le, const char *&q) {
        const char *nl = static_cast<const char *>(memchr(from, '\n', (size_t)(e - from)));
        le = nl ? nl : e;
        q = nl ? nl + 1 : e;
        if (le > from && le[-1] == '\r') le--;
    };
    auto irregular = [&](const char *what) {
        set_error("gs_rowwriter_rlen: %s", what);
        return GS_E_IRREGULAR;
    };
    uint64_t recs = 0;
    while (p < e) {
        const char *le, *q;
        line(p, le, q);
        if (le == p) {  // (kseq skips empty lines between records)
            p = q;
            continue;
        }
        const char kind = *p;
        if (kind != '>' && kind != '@') return irregular("text in front of a record header");
        const char *nm = p + 1, *ne = nm;
        while (ne < le && !isspace((unsigned char)*ne)) ne++;
        p = q;
        uint64_t len = 0;
        if (kind == '@') {
            if (p >= e) return irregular("a FASTQ record without its sequence line");
            const char *se, *sq;
            line(p, se, sq);
            len = (uint64_t)(se - p);
            if (!len || *p == '>' || *p == '@' || *p == '+') return irregular("a FASTQ sequence line that is empty or starts with > @ +");
            p = sq;
            if (p >= e) return irregular("a FASTQ record without its + line");
            const char *pe, *pq;
            line(p, pe, pq);
            if (pe == p || *p != '+') return irregular("a FASTQ record whose third line is no + line (multi-line FASTQ?)");
            p = pq;
            const char *qe, *qq;
            if (p >= e) return irregular("a FASTQ record without its quality line");
            line(p, qe, qq);
            if ((uint64_t)(qe - p) != len) return irregular("a FASTQ quality line as long as something else than the sequence");
            p = qq;
        } else {
            while (p < e && *p != '>') {
                const char *se, *sq;
                line(p, se, sq);
                if (se > p && (*p == '@' || *p == '+')) return irregular("a FASTA sequence line that starts with @ or +");
                len += (uint64_t)(se - p);
                p = sq;
            }
        }
        const size_t nl = (size_t)(ne - nm), o = w->buf.size();
        w->buf.resize(o + nl + 32);
        char *d = w->buf.data() + o;
        memcpy(d, nm, nl);
        d += nl;
        *d++ = '\t';
        d = put_u64(d, len);
        *d++ = '\n';
        w->buf.resize((size_t)(d - w->buf.data()));
        recs++;
        if (w->buf.size() > (7u << 20) && !w->flush()) {
            set_error("gs_rowwriter_rlen: write failed");
            return GS_E_IO;
        }
    }
    w->rows += recs;
    if (n_records) *n_records = recs;
    return GS_OK;
}
// flushes and closes; *rows_out = rows written
int gs_rowwriter_close(gs_rowwriter *w, uint64_t *rows_out) {
    if (!w) return GS_OK;
    bool ok = w->flush();
    ok = fclose(w->f) == 0 && ok;
    if (rows_out) *rows_out = w->rows;
    rowwriter_release_pinned(w);
    delete w;
    if (!ok) {
        set_error("gs_rowwriter_close: write failed");
        return GS_E_IO;
    }
    return GS_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------- sunkpos
extern "C" {

int gs_sunkpos_load(const char *path, gs_sunkpos **out) {
    if (!path || !out) {
        set_error("gs_sunkpos_load: null argument");
        return GS_E_ARG;
    }
    *out = nullptr;
    LineReader r(path);
    if (!r.ok()) {
        set_error("cannot open the file %s", path);
        return GS_E_IO;
    }
    auto *d = new gs_sunkpos();
    std::unordered_map<std::string, uint32_t> contig_id;
    std::string line, name, prev;
    bool first = true;
    uint64_t lineno = 0;
    d->row_text_off.push_back(0);
    while (r.next(line)) {
        lineno++;
        if (line.empty()) continue;
        const char *p = line.data(), *e = p + line.size();
        const char *f[5], *fe[5];
        int nf = 0;
        const char *s0 = p;
        for (const char *q = p;; q++) {
            if (q == e || *q == '\t') {
                if (nf < 5) {
                    f[nf] = s0;
                    fe[nf] = q;
                }
                nf++;
                s0 = q + 1;
                if (q == e) break;
            }
        }
        int64_t pos, start, group;
        if (nf < 5 || !parse_i64(f[1], fe[1], pos) || !parse_i64(f[3], fe[3], start) || !parse_i64(f[4], fe[4], group)) {
            set_error("%s:%llu: expected `rname<TAB>pos<TAB>contig<TAB>start<TAB>group`", path, (unsigned long long)lineno);
            delete d;
            return GS_E_IO;
        }
        name.assign(f[0], fe[0] - f[0]);
        if (first || name != prev) {  // the reference flushes on every change of name (diag_filter_v3.nim:64)
            d->seg_names.push_back(name);
            prev = name;
            first = false;
        }
        name.assign(f[2], fe[2] - f[2]);
        auto it = contig_id.find(name);
        uint32_t cid;
        if (it == contig_id.end()) {
            cid = (uint32_t)d->contigs.size();
            contig_id.emplace(name, cid);
            d->contigs.push_back(name);
        } else {
            cid = it->second;
        }
        d->row_seg.push_back((uint32_t)d->seg_names.size() - 1);
        d->row_pos.push_back(pos);
        d->row_contig.push_back(cid);
        d->row_start.push_back(start);
        d->row_group.push_back(group);
        d->text.insert(d->text.end(), line.begin(), line.end());
        d->text.push_back('\0');
        d->row_text_off.push_back(d->text.size());
    }
    *out = d;
    return GS_OK;
}

void gs_sunkpos_free(gs_sunkpos *s) { delete s; }
uint64_t gs_sunkpos_n_rows(const gs_sunkpos *s) { return s->row_seg.size(); }
uint32_t gs_sunkpos_n_segs(const gs_sunkpos *s) { return (uint32_t)s->seg_names.size(); }
const uint32_t *gs_sunkpos_row_seg(const gs_sunkpos *s) { return s->row_seg.data(); }
const int64_t *gs_sunkpos_row_pos(const gs_sunkpos *s) { return s->row_pos.data(); }
const uint32_t *gs_sunkpos_row_contig(const gs_sunkpos *s) { return s->row_contig.data(); }
const int64_t *gs_sunkpos_row_start(const gs_sunkpos *s) { return s->row_start.data(); }
const int64_t *gs_sunkpos_row_group(const gs_sunkpos *s) { return s->row_group.data(); }
const char *gs_sunkpos_seg_name(const gs_sunkpos *s, uint32_t seg) {
    return seg < s->seg_names.size() ? s->seg_names[seg].c_str() : nullptr;
}
uint32_t gs_sunkpos_n_contigs(const gs_sunkpos *s) { return (uint32_t)s->contigs.size(); }
const char *gs_sunkpos_contig_name(const gs_sunkpos *s, uint32_t c) {
    return c < s->contigs.size() ? s->contigs[c].c_str() : nullptr;
}
const char *gs_sunkpos_row_text(const gs_sunkpos *s, uint64_t row, uint32_t *len) {
    if (row >= s->row_seg.size()) return nullptr;
    if (len) *len = (uint32_t)(s->row_text_off[row + 1] - s->row_text_off[row] - 1);
    return s->text.data() + s->row_text_off[row];
}

// ---- what the row-set functions (gs_chain.hip) need from a `.sunkpos` file beside its columns -------------------------------
// The distinct read names in first-seen order: name id of every ROW, rank of every name in sorted (byte-wise = code-point) order
// -- `groupby("rname")` iterates in that order (process-by-contig_lowmem_AR.py:135).  n_names and the arrays are built on the
// first call and live as long as the handle.
static void sunkpos_name_table(gs_sunkpos *s) {
    if (!s->name_of_seg.empty() || s->seg_names.empty()) return;
    std::unordered_map<std::string, uint32_t> ids;
    s->name_of_seg.resize(s->seg_names.size());
    for (size_t i = 0; i < s->seg_names.size(); i++) {
        auto it = ids.find(s->seg_names[i]);
        if (it == ids.end()) {
            it = ids.emplace(s->seg_names[i], (uint32_t)s->names.size()).first;
            s->names.push_back((uint32_t)i);  // (a name = the first segment that carries it)
        }
        s->name_of_seg[i] = it->second;
    }
    s->row_name.resize(s->row_seg.size());
    for (size_t r = 0; r < s->row_seg.size(); r++) s->row_name[r] = s->name_of_seg[s->row_seg[r]];
    std::vector<uint32_t> order(s->names.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return s->seg_names[s->names[a]] < s->seg_names[s->names[b]]; });
    s->name_rank.resize(order.size());
    for (size_t i = 0; i < order.size(); i++) s->name_rank[order[i]] = (uint32_t)i;
}
uint32_t gs_sunkpos_n_names(gs_sunkpos *s) {
    sunkpos_name_table(s);
    return (uint32_t)s->names.size();
}
const uint32_t *gs_sunkpos_row_name(gs_sunkpos *s) {
    sunkpos_name_table(s);
    return s->row_name.data();
}
const uint32_t *gs_sunkpos_name_rank(gs_sunkpos *s) {
    sunkpos_name_table(s);
    return s->name_rank.data();
}
const char *gs_sunkpos_name(gs_sunkpos *s, uint32_t name) {
    sunkpos_name_table(s);
    return name < s->names.size() ? s->seg_names[s->names[name]].c_str() : nullptr;
}
// name_long[n_names] <- 1 where the `.rlen` file (rows `rname \t len`) holds a row of that name with len >= minlen
// (`lendf.query("len >= @minlen").index`, process-by-contig_lowmem_AR.py:106-108)
int gs_sunkpos_names_long(gs_sunkpos *s, const char *rlen_path, int64_t minlen, uint8_t *name_long) {
    if (!s || !rlen_path || !name_long) {
        set_error("gs_sunkpos_names_long: null argument");
        return GS_E_ARG;
    }
    sunkpos_name_table(s);
    std::unordered_map<std::string, uint32_t> ids;
    for (size_t i = 0; i < s->names.size(); i++) ids.emplace(s->seg_names[s->names[i]], (uint32_t)i);
    std::fill(name_long, name_long + s->names.size(), 0);
    LineReader r(rlen_path);
    if (!r.ok()) {
        set_error("cannot open the file %s", rlen_path);
        return GS_E_IO;
    }
    std::string line, nm;
    while (r.next(line)) {
        const size_t t = line.find('\t');
        if (t == std::string::npos) continue;
        size_t t2 = line.find('\t', t + 1);
        if (t2 == std::string::npos) t2 = line.size();
        int64_t len;
        if (!parse_i64(line.data() + t + 1, line.data() + t2, len)) continue;
        if (len < minlen) continue;
        nm.assign(line, 0, t);
        auto it = ids.find(nm);
        if (it != ids.end()) name_long[it->second] = 1;
    }
    return GS_OK;
}
// row_bad[n_rows] <- 1 where the row's `chrom:ID` is a line of the bad-SUNK list (process-by-contig_lowmem_AR.py:66-72: the
// script compares the strings chrom + ":" + str(ID))
int gs_sunkpos_rows_bad(gs_sunkpos *s, const char *bad_path, uint8_t *row_bad) {
    if (!s || !bad_path || !row_bad) {
        set_error("gs_sunkpos_rows_bad: null argument");
        return GS_E_ARG;
    }
    std::unordered_map<std::string, uint32_t> cid;
    for (size_t c = 0; c < s->contigs.size(); c++) cid.emplace(s->contigs[c], (uint32_t)c);
    std::unordered_set<uint64_t> bad;  // contig id << 40 ^ ID (IDs are contig coordinates: below 2^40)
    LineReader r(bad_path);
    if (!r.ok()) {
        set_error("cannot open the file %s", bad_path);
        return GS_E_IO;
    }
    std::string line, nm;
    while (r.next(line)) {  // (`splitlines()`: the text of every line, CR and all, is an element; only exact strings can match)
        const size_t t = line.rfind(':');
        if (t == std::string::npos) continue;
        int64_t id;
        if (!parse_i64(line.data() + t + 1, line.data() + line.size(), id) || id < 0) continue;
        if (std::to_string(id) != line.substr(t + 1)) continue;  // ("007" is not str(7))
        nm.assign(line, 0, t);
        auto it = cid.find(nm);
        if (it != cid.end()) bad.insert(((uint64_t)it->second << 40) ^ (uint64_t)id);
    }
    const size_t n = s->row_seg.size();
    for (size_t i = 0; i < n; i++)
        row_bad[i] = (s->row_group[i] >= 0 && bad.count(((uint64_t)s->row_contig[i] << 40) ^ (uint64_t)s->row_group[i])) ? 1 : 0;
    return GS_OK;
}
// dense ids of the rows' distinct (contig, ID) pairs, in first-seen order: row_key[n_rows]; the pair of key j =
// key_contig[j], key_id[j] (valid until the next call / the handle's end).  -> number of keys
uint32_t gs_sunkpos_row_keys(gs_sunkpos *s, uint32_t *row_key, const uint32_t **key_contig, const int64_t **key_id) {
    std::unordered_map<uint64_t, uint32_t> ids;
    s->key_contig.clear();
    s->key_id.clear();
    const size_t n = s->row_seg.size();
    for (size_t i = 0; i < n; i++) {
        const uint64_t k = ((uint64_t)s->row_contig[i] << 40) ^ (uint64_t)s->row_group[i];
        auto it = ids.find(k);
        if (it == ids.end()) {
            it = ids.emplace(k, (uint32_t)s->key_contig.size()).first;
            s->key_contig.push_back(s->row_contig[i]);
            s->key_id.push_back(s->row_group[i]);
        }
        row_key[i] = it->second;
    }
    if (key_contig) *key_contig = s->key_contig.data();
    if (key_id) *key_id = s->key_id.data();
    return (uint32_t)s->key_contig.size();
}

}  // extern "C"

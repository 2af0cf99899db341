// `.sunkpos` text made where the rows are: the tagger's only product is one line per emitted hit,
//     name \t pos \t chrom \t start \t group \n                                   (workflow/src/kmerpos_annot3.nim:93)
// and a genome-scale pass emits 1e7 .. 1e8 of them.  Round 4 formatted them on ONE host thread with three random lookups per row
// into host arrays of the whole location table (2.2 Mrow/s).  Here a pass's hit list stays in device memory: one kernel measures
// every line (the loc record of a row is one 16-byte gather, gs_locrows), an exclusive scan places them, a second kernel writes
// them -- a wave stages its 64 lines in LDS and stores them as one contiguous stretch -- and the finished text crosses PCIe in
// 32 MiB pieces through pinned buffers while host threads pwrite() the pieces before.  Same bytes as gs_rowwriter_rows.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <future>
#include <string>

#include "gs_chain.h"

using namespace gs;
using namespace gs::chain;

namespace {

struct FmtArgs {
    const gs_hit *hits;
    uint64_t n;
    const uint4 *rec;           // x = start, y = contig, z = group
    const uint8_t *names;       // the batch's read names back to back
    const uint32_t *name_off;   // n_reads + 1
    const uint8_t *cnames;      // contig names back to back
    const uint32_t *cname_off;  // n_contigs + 1
};

__device__ inline uint32_t ndigits(uint32_t v) {
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u : v < 10000000u ? 7u : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
__global__ void k_row_len(FmtArgs a, uint64_t *len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) {
        if (i == a.n) len[i] = 0;
        return;
    }
    const gs_hit h = a.hits[i];
    const uint4 r = a.rec[h.payload];
    len[i] = (a.name_off[h.read + 1] - a.name_off[h.read]) + (a.cname_off[r.y + 1] - a.cname_off[r.y]) + ndigits(h.pos) + ndigits(r.x) + ndigits(r.z) + 5u;
}
template <typename P>
__device__ inline P put_dec(P p, uint32_t v) {
    const uint32_t nd = ndigits(v);
    for (uint32_t k = nd; k > 0; k--) {
        p[k - 1] = (uint8_t)('0' + v % 10u);
        v /= 10u;
    }
    return p + nd;
}
template <typename P>
__device__ inline void put_row(P p, const FmtArgs &a, const gs_hit h, const uint4 r) {
    for (uint32_t q = a.name_off[h.read], e = a.name_off[h.read + 1]; q < e; q++) *p++ = a.names[q];
    *p++ = '\t';
    p = put_dec(p, h.pos);
    *p++ = '\t';
    for (uint32_t q = a.cname_off[r.y], e = a.cname_off[r.y + 1]; q < e; q++) *p++ = a.cnames[q];
    *p++ = '\t';
    p = put_dec(p, r.x);
    *p++ = '\t';
    p = put_dec(p, r.z);
    *p++ = '\n';
}
constexpr int kFmtWaves = 2, kFmtStage = 8192;  // 16 KiB of LDS per workgroup (see gs_diag.hip on why not more)
__global__ __launch_bounds__(64 * kFmtWaves) void k_row_write(FmtArgs a, const uint64_t *off, uint8_t *text) {
    __shared__ uint8_t stage[kFmtWaves][kFmtStage];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t i0 = ((uint64_t)blockIdx.x * kFmtWaves + w) * 64u;
    if (i0 >= a.n) return;
    const uint64_t i = i0 + lane;
    const uint64_t i1 = min(i0 + 64u, a.n);
    const uint64_t base = off[i0], total = off[i1] - base;
    gs_hit h{};
    uint4 r{};
    if (i < a.n) {
        h = a.hits[i];
        r = a.rec[h.payload];
    }
    if (total <= (uint64_t)kFmtStage) {  // the wave's 64 lines in LDS, then out as one contiguous stretch of bytes
        if (i < a.n) put_row(&stage[w][off[i] - base], a, h, r);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint8_t *dst = text + base;
        const uint32_t n = (uint32_t)total;
        // bytes up to the first 4-byte boundary of the destination one by one, then dwords, then the tail
        const uint32_t head = min(n, (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u));
        if (lane < head) dst[lane] = stage[w][lane];
        const uint32_t nd = (n - head) >> 2;
        for (uint32_t q = lane; q < nd; q += 64) {
            const uint8_t *s = &stage[w][head + 4u * q];
            reinterpret_cast<uint32_t *>(dst + head)[q] = (uint32_t)s[0] | (uint32_t)s[1] << 8 | (uint32_t)s[2] << 16 | (uint32_t)s[3] << 24;
        }
        const uint32_t done = head + 4u * nd;
        if (done + lane < n) dst[done + lane] = stage[w][done + lane];
    } else if (i < a.n) {  // (very long read names: every lane writes its own line)
        put_row(text + off[i], a, h, r);
    }
}
__global__ void k_check_hits(const gs_hit *hits, uint64_t n, uint64_t n_rows, uint32_t n_reads, const uint4 *rec, uint32_t n_contigs, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (i < n) {
        const gs_hit h = hits[i];
        bad = h.payload >= n_rows || h.read >= n_reads || rec[h.payload].y >= n_contigs;
    }
    if (__ballot(bad) && (threadIdx.x & 63u) == 0) atomicOr(flag, 1u);
}

// the formatted rows of the last pass over `r`, in device memory (a pool block: pool_free)
int format_rows(Ctx &c, gs_table *t, gs_reads *r, const gs_locrows *lr, uint32_t n_contigs, const char *const *contig_names, const uint8_t *names_base,
                const uint64_t *name_off, const uint32_t *name_len, uint32_t n_names, const gs_batch *batch, uint8_t **d_text_out, uint64_t *n_bytes, uint64_t *n_rows_out,
                const uint64_t *row_cuts, uint32_t n_cuts, uint64_t *byte_cuts) {
    *d_text_out = nullptr;
    *n_bytes = 0;
    *n_rows_out = 0;
    for (uint32_t q = 0; q < n_cuts; q++) byte_cuts[q] = 0;
    const void *d_hits = nullptr;
    uint64_t n = 0;
    const int rc_hits = gs_tag_hits_device(t, r, &d_hits, &n);
    if (rc_hits != GS_OK && rc_hits != GS_E_KEYERROR) return rc_hits;  // (KeyError: the rows before the offending window are the output, kmerpos_annot3.nim:90)
    if (!n) return GS_OK;
    for (uint32_t q = 0; q < n_cuts; q++)
        if (row_cuts[q] > n || (q && row_cuts[q] < row_cuts[q - 1])) {
            set_error("gs_rowwriter_rows_device_cuts: row cuts must ascend and end at or before the pass's %llu rows", (unsigned long long)n);
            return GS_E_ARG;
        }
    // the batch's read names and the contig names, back to back, with offsets
    const uint32_t n_reads = name_off ? n_names : (uint32_t)batch->len.size();
    std::vector<uint32_t> noff((size_t)n_reads + 1, 0), coff((size_t)n_contigs + 1, 0);
    std::string nblob, cblob;
    for (uint32_t q = 0; q < n_reads; q++) {
        const char *nm;
        size_t nl;
        if (name_off) {
            nm = reinterpret_cast<const char *>(names_base) + name_off[q];
            nl = name_len[q];
        } else {
            nm = batch->names.data() + batch->name_off[q];
            nl = strlen(nm);
        }
        if (nblob.size() + nl > 0xFFFFFFF0ull) {
            set_error("gs_rowwriter_rows_device: more than 4 GiB of read names in one batch");
            return GS_E_CAPACITY;
        }
        nblob.append(nm, nl);
        noff[q + 1] = (uint32_t)nblob.size();
    }
    for (uint32_t q = 0; q < n_contigs; q++) {
        cblob.append(contig_names[q]);
        coff[q + 1] = (uint32_t)cblob.size();
    }
    FmtArgs a{};
    a.hits = static_cast<const gs_hit *>(d_hits);
    a.n = n;
    a.rec = lr->rec;
    uint8_t *d_nb, *d_cb;
    uint32_t *d_no, *d_co, *d_flag;
    GSC_TRY(c.mem.put(&d_nb, reinterpret_cast<const uint8_t *>(nblob.data()), nblob.size()));
    GSC_TRY(c.mem.put(&d_no, noff.data(), noff.size()));
    GSC_TRY(c.mem.put(&d_cb, reinterpret_cast<const uint8_t *>(cblob.data()), cblob.size()));
    GSC_TRY(c.mem.put(&d_co, coff.data(), coff.size()));
    a.names = d_nb;
    a.name_off = d_no;
    a.cnames = d_cb;
    a.cname_off = d_co;
    GSC_TRY(c.mem.get(&d_flag, 1));
    GSC_TRY(c.zero(d_flag, 1));
    hipLaunchKernelGGL(k_check_hits, dim3(blocks_of(n)), dim3(256), 0, c.st, a.hits, n, lr->n_rows, n_reads, lr->rec, n_contigs, d_flag);
    uint32_t bad = 0;
    GSC_TRY(c.fetch(&bad, d_flag));
    if (bad) {
        set_error("gs_rowwriter_rows_device: a hit names a loc row, a read or a contig that does not exist (%llu loc rows, %u reads, %u contigs)",
                  (unsigned long long)lr->n_rows, n_reads, n_contigs);
        return GS_E_ARG;
    }
    uint64_t *d_len, *d_off;
    GSC_TRY(c.mem.get(&d_len, n + 1));
    GSC_TRY(c.mem.get(&d_off, n + 1));
    hipLaunchKernelGGL(k_row_len, dim3(blocks_of(n + 1)), dim3(256), 0, c.st, a, d_len);
    GSC_TRY(exclusive_sum(c, d_len, d_off, n + 1));
    uint64_t total = 0;
    GSC_TRY(c.fetch(&total, d_off + n));
    for (uint32_t q = 0; q < n_cuts; q++)  // (a handful: the scatter chunks of --emulate-chunks)
        GSC_HIP(hipMemcpyAsync(byte_cuts + q, d_off + row_cuts[q], sizeof(uint64_t), hipMemcpyDeviceToHost, c.st));
    if (n_cuts) GSC_TRY(c.sync());
    uint8_t *d_text = static_cast<uint8_t *>(pool_alloc(c.device, std::max<uint64_t>(total, 1)));
    if (!d_text) return GS_E_NOMEM;
    hipLaunchKernelGGL(k_row_write, dim3(blocks_of(n, 64 * kFmtWaves)), dim3(64 * kFmtWaves), 0, c.st, a, d_off, d_text);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c.st);
    if (e != hipSuccess) {
        pool_free(c.device, d_text);
        set_error("gs_rowwriter_rows_device: %s", hipGetErrorString(e));
        return GS_E_NODEVICE;
    }
    *d_text_out = d_text;
    *n_bytes = total;
    *n_rows_out = n;
    return GS_OK;
}

constexpr size_t kPiece = (size_t)32 << 20;

// at < 0: the output has no positions (a pipe, /dev/stdout into one): the bytes go behind what was written before
bool pwrite_all(int fd, const uint8_t *p, size_t n, off_t at) {
    while (n) {
        const ssize_t w = at < 0 ? write(fd, p, n) : pwrite(fd, p, n, at);
        if (w <= 0) {
            if (w < 0 && errno == EINTR) continue;
            return false;
        }
        p += w;
        n -= (size_t)w;
        if (at >= 0) at += w;
    }
    return true;
}

}  // namespace

void rowwriter_release_pinned(gs_rowwriter *w) {
    for (void *&p : w->pinned)
        if (p) {
            (void)hipHostFree(p);
            p = nullptr;
        }
}

extern "C" {

// The rows of the last pass over `r` (its hit list is still in device memory), formatted on the device and appended to the file.
// Names as in gs_rowwriter_rows: (names_base + name_off[read], name_len[read]) for the n_names reads of the batch or, when name_off
// is NULL, `batch`.
int gs_rowwriter_rows_device(gs_rowwriter *w, gs_table *t, gs_reads *r, const gs_locrows *lr, uint32_t n_contigs, const char *const *contig_names,
                             const uint8_t *names_base, const uint64_t *name_off, const uint32_t *name_len, uint32_t n_names, const gs_batch *batch) {
    return gs_rowwriter_rows_device_cuts(w, t, r, lr, n_contigs, contig_names, names_base, name_off, name_len, n_names, batch, nullptr, 0, nullptr);
}

// ... and where in this call's text given rows start: byte_cuts[i] = bytes of the rows in front of row row_cuts[i] (ascending, at most the
// pass's row count: that one gives the call's total).  What --emulate-chunks needs: a batch tagged in chunk-major order is written as one
// text, and every scatter chunk's stretch of it is known (kmerpos_annot3.py run_chunked).
int gs_rowwriter_rows_device_cuts(gs_rowwriter *w, gs_table *t, gs_reads *r, const gs_locrows *lr, uint32_t n_contigs, const char *const *contig_names,
                                  const uint8_t *names_base, const uint64_t *name_off, const uint32_t *name_len, uint32_t n_names, const gs_batch *batch,
                                  const uint64_t *row_cuts, uint32_t n_cuts, uint64_t *byte_cuts) {
    if (!w || !t || !r || !lr || (n_contigs && !contig_names) || (!name_off && !batch) || (name_off && (!names_base || !name_len)) ||
        (n_cuts && (!row_cuts || !byte_cuts))) {
        set_error("gs_rowwriter_rows_device: null argument");
        return GS_E_ARG;
    }
    GSC_TRY(check_device(lr->device, "gs_rowwriter_rows_device"));
    static const bool timing = getenv("GAVISUNK_TIMING") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    Ctx c(lr->device);
    uint8_t *d_text = nullptr;
    uint64_t total = 0, n = 0;
    GSC_TRY(format_rows(c, t, r, lr, n_contigs, contig_names, names_base, name_off, name_len, n_names, batch, &d_text, &total, &n, row_cuts, n_cuts, byte_cuts));
    if (!n) return GS_OK;
    const double ms_format = ms_since(t_start);
    struct Free {
        int dev;
        void *p;
        ~Free() { pool_free(dev, p); }
    } guard{c.device, d_text};
    if (!w->flush() || fflush(w->f) != 0) {
        set_error("gs_rowwriter_rows_device: write failed");
        return GS_E_IO;
    }
    // (the reference opens <out> with open(ofile, fmWrite), kmerpos_annot3.nim:80: a pipe or /dev/stdout is as good as a file.  Such an
    // output has no positions: its pieces are written one after the other, each behind the one before)
    const off_t at0 = ftello(w->f);
    const int fd = fileno(w->f);
    const bool seekable = at0 >= 0;
    if (fd < 0 || (!seekable && errno != ESPIPE)) {
        set_error("gs_rowwriter_rows_device: the output file cannot be written");
        return GS_E_IO;
    }
    // device -> pinned piece -> pwrite on another thread, four pieces in flight
    std::future<bool> busy[4];
    bool ok = true;
    int rc = GS_OK;
    uint64_t done = 0;
    for (int k = 0; done < total && rc == GS_OK; k = (k + 1) & 3) {
        if (busy[k].valid()) ok = busy[k].get() && ok;
        if (!ok) break;  // (a reader that went away, a full disk: no point in formatting on)
        if (!w->pinned[k] && hipHostMalloc(&w->pinned[k], kPiece, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            w->pinned[k] = nullptr;
            set_error("gs_rowwriter_rows_device: no pinned memory for the output staging");
            rc = GS_E_NOMEM;
            break;
        }
        const size_t m = (size_t)std::min<uint64_t>(kPiece, total - done);
        hipError_t e = hipMemcpyAsync(w->pinned[k], d_text + done, m, hipMemcpyDeviceToHost, c.st);
        if (e == hipSuccess) e = hipStreamSynchronize(c.st);
        if (e != hipSuccess) {
            set_error("gs_rowwriter_rows_device: %s", hipGetErrorString(e));
            rc = GS_E_NODEVICE;
            break;
        }
        const uint8_t *src = static_cast<const uint8_t *>(w->pinned[k]);
        const off_t at = seekable ? at0 + (off_t)done : (off_t)-1;
        if (!seekable && busy[(k + 3) & 3].valid()) ok = busy[(k + 3) & 3].get() && ok;  // (in order: the piece before is out)
        try {
            busy[k] = std::async(std::launch::async, pwrite_all, fd, src, m, at);
        } catch (const std::exception &) {  // (no thread to be had: on this one)
            ok = pwrite_all(fd, src, m, at) && ok;
        }
        done += m;
    }
    for (auto &b : busy)
        if (b.valid()) ok = b.get() && ok;
    if (rc != GS_OK) return rc;
    if (!ok || (seekable && fseeko(w->f, at0 + (off_t)total, SEEK_SET) != 0)) {
        set_error("gs_rowwriter_rows_device: write failed");
        return GS_E_IO;
    }
    w->rows += n;
    if (timing)
        fprintf(stderr, "[gs_rowwriter_rows_device] %llu rows, %llu bytes: formatted on the device in %.2f ms, written in %.2f ms\n", (unsigned long long)n,
                (unsigned long long)total, ms_format, ms_since(t_start) - ms_format);
    return GS_OK;
}

}  // extern "C"

// Diagonal filter of GAVISUNK (workflow/src/diag_filter_v3.nim, diag_filter_step2.nim) on the GPU.
//
// Per read (run of rows with the same name) and per contig of this haplotype's .fai:
//   n_- = pos + start, n_+ = start - pos;  median = int(percentile(n, 50))  [sort, linear interpolation
//   at (len-1)/2 in float64, truncation toward zero: arraymancer 0.7.10, diag_filter_v3.nim:86,113];
//   good = #distinct groups with |n - median| < 2500 (:91-94);
//   the best contig maximises good, then minimises the hit count, "-" is tried before "+", and contigs
//   are visited in the slot order of Nim's Table[string, ...] (:76) -- a serial hash-table replay the
//   host does while it lays the rows out by (read, contig); everything numeric runs in two kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "gs_common.h"

using namespace gs;

namespace {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GS_E_NODEVICE;                                                             \
        }                                                                                     \
    } while (0)

// ---- Nim >= 1.4 `hash(string)` = MurmurHash3_x86_32, seed 0 (symbol murmurHash_pureZhashes in the
// reference's diag_filter_v3 binary); Table = open addressing, linear probing, 64 initial slots for
// initTable(), doubles when dataLen*2 < counter*3 or dataLen-counter < 4 before an insert; clear()
// keeps the capacity; pairs() walks the slots in index order.
uint32_t murmur3_32(const char *data, size_t len) {
    const uint32_t c1 = 0xCC9E2D51u, c2 = 0x1B873593u;
    uint32_t h = 0;
    const size_t nb = len / 4;
    for (size_t i = 0; i < nb; i++) {
        uint32_t k;
        memcpy(&k, data + 4 * i, 4);
        k *= c1;
        k = (k << 15) | (k >> 17);
        k *= c2;
        h ^= k;
        h = (h << 13) | (h >> 19);
        h = h * 5 + 0xE6546B64u;
    }
    uint32_t k = 0;
    for (size_t i = len; i > 4 * nb; i--) k = (k << 8) | (uint8_t)data[i - 1];
    k *= c1;
    k = (k << 15) | (k >> 17);
    k *= c2;
    h ^= k;
    h ^= (uint32_t)len;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

class NimTableOrder {
   public:
    NimTableOrder() : key_(64, kFree), hc_(64, 0) {}
    void clear() {
        std::fill(key_.begin(), key_.end(), kFree);
        counter_ = 0;
    }
    void add(uint32_t contig, uint32_t hc) {  // contigs of one read are distinct ids <=> distinct names
        size_t mask = key_.size() - 1, h = hc & mask;
        while (key_[h] != kFree) {
            if (key_[h] == contig) return;
            h = (h + 1) & mask;
        }
        const size_t n = key_.size();
        if (n * 2 < counter_ * 3 || n - counter_ < 4) {
            std::vector<uint32_t> nk(2 * n, kFree), nh(2 * n, 0);
            for (size_t i = 0; i < n; i++)
                if (key_[i] != kFree) {
                    size_t j = hc_[i] & (2 * n - 1);
                    while (nk[j] != kFree) j = (j + 1) & (2 * n - 1);
                    nk[j] = key_[i];
                    nh[j] = hc_[i];
                }
            key_.swap(nk);
            hc_.swap(nh);
            mask = key_.size() - 1;
            h = hc & mask;
            while (key_[h] != kFree) h = (h + 1) & mask;
        }
        key_[h] = contig;
        hc_[h] = hc;
        counter_++;
    }
    template <typename F>
    void for_each(F f) const {
        for (size_t i = 0; i < key_.size(); i++)
            if (key_[i] != kFree) f(key_[i]);
    }

   private:
    static constexpr uint32_t kFree = 0xFFFFFFFFu;
    std::vector<uint32_t> key_, hc_;
    size_t counter_ = 0;
};

// ---- device
__device__ void heapsort64(int64_t *a, int n) {
    for (int start = n / 2 - 1; start >= 0; start--) {
        int root = start;
        for (;;) {
            int child = 2 * root + 1;
            if (child >= n) break;
            if (child + 1 < n && a[child] < a[child + 1]) child++;
            if (a[root] >= a[child]) break;
            const int64_t t = a[root]; a[root] = a[child]; a[child] = t;
            root = child;
        }
    }
    for (int end = n - 1; end > 0; end--) {
        const int64_t t0 = a[0]; a[0] = a[end]; a[end] = t0;
        int root = 0;
        for (;;) {
            int child = 2 * root + 1;
            if (child >= end) break;
            if (child + 1 < end && a[child] < a[child + 1]) child++;
            if (a[root] >= a[child]) break;
            const int64_t t = a[root]; a[root] = a[child]; a[child] = t;
            root = child;
        }
    }
}

// one thread per (read, contig) pair
__global__ void k_diag_stats(uint32_t n_pairs, const uint64_t *pair_off, const uint32_t *pair_rows, const int64_t *pos,
                             const int64_t *start, const int64_t *group, int64_t *scratch_a, int64_t *scratch_b,
                             uint32_t *good_minus, uint32_t *good_plus, int64_t bandwidth) {
    const uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= n_pairs) return;
    const uint64_t lo = pair_off[pi], hi = pair_off[pi + 1];
    const int n = (int)(hi - lo);
    if (n <= 1) {  // `if p.shape[0] == 1: continue` (diag_filter_v3.nim:82)
        good_minus[pi] = 0;
        good_plus[pi] = 0;
        return;
    }
    int64_t *sa = scratch_a + lo, *sb = scratch_b + lo;
    for (int dir = 0; dir < 2; dir++) {
        for (int i = 0; i < n; i++) {
            const uint32_t r = pair_rows[lo + i];
            sa[i] = dir == 0 ? pos[r] + start[r] : start[r] - pos[r];
        }
        heapsort64(sa, n);
        const double f = (double)(n - 1) * 50.0 / 100.0;
        const int i0 = (int)floor(f);
        const double frac = f - (double)i0;
        double res;
        if (f == (double)i0) res = (double)sa[i0];
        else res = (double)sa[i0] + frac * ((double)sa[i0 + 1] - (double)sa[i0]);
        const int64_t median = (int64_t)res;  // truncation toward zero
        int m = 0;
        for (int i = 0; i < n; i++) {
            const uint32_t r = pair_rows[lo + i];
            const int64_t v = dir == 0 ? pos[r] + start[r] : start[r] - pos[r];
            int64_t dlt = v - median;
            if (dlt < 0) dlt = -dlt;
            if (dlt < bandwidth) sb[m++] = group[r];
        }
        heapsort64(sb, m);
        uint32_t distinct = 0;
        for (int i = 0; i < m; i++) distinct += (i == 0 || sb[i] != sb[i - 1]) ? 1u : 0u;
        (dir == 0 ? good_minus : good_plus)[pi] = distinct;
    }
}

// one thread per read: the reference's selection sequence over its contigs in Table slot order
__global__ void k_diag_select(uint32_t n_segs, const uint32_t *seg_pair_off, const uint64_t *pair_off,
                              const uint32_t *pair_contig, const uint32_t *good_minus, const uint32_t *good_plus,
                              uint32_t *best_contig, uint32_t *best_good, uint8_t *best_dir) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    uint32_t maxgood = 0, bestc = GS_NONE;
    uint64_t maxhitlen = 0;
    uint8_t dirc = 0;
    for (uint32_t pi = seg_pair_off[s]; pi < seg_pair_off[s + 1]; pi++) {
        const uint64_t hitlen = pair_off[pi + 1] - pair_off[pi];
        if (hitlen == 1) continue;
        for (int dir = 0; dir < 2; dir++) {
            const uint32_t good = dir == 0 ? good_minus[pi] : good_plus[pi];
            if (good > maxgood) {  // :98-102
                maxgood = good;
                maxhitlen = hitlen;
                bestc = pair_contig[pi];
                dirc = dir == 0 ? '-' : '+';
            }
            if (good == maxgood && hitlen < maxhitlen) {  // :103-107
                maxgood = good;
                maxhitlen = hitlen;
                bestc = pair_contig[pi];
                dirc = dir == 0 ? '-' : '+';
            }
        }
    }
    const bool print = maxgood > 1;  // :141
    best_contig[s] = print ? bestc : GS_NONE;
    best_good[s] = maxgood;
    best_dir[s] = dirc;
}

__global__ void k_diag_keep(uint64_t n_rows, const uint32_t *row_seg, const uint32_t *row_contig,
                            const uint32_t *best_contig, uint8_t *keep) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const uint32_t b = best_contig[row_seg[i]];
    keep[i] = (b != GS_NONE && b == row_contig[i]) ? 1 : 0;
}

struct DevBuf {
    std::vector<void *> ptrs;
    ~DevBuf() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    int put(T **d, const T *h, uint64_t n) {
        *d = nullptr;
        HIP_TRY(hipMalloc((void **)d, std::max<uint64_t>(n, 1) * sizeof(T)));
        ptrs.push_back(*d);
        if (h && n) HIP_TRY(hipMemcpy(*d, h, n * sizeof(T), hipMemcpyHostToDevice));
        return GS_OK;
    }
};

}  // namespace

extern "C" int gs_diag_filter(int device, const gs_sunkpos *s, const uint8_t *in_fai, uint32_t *best_contig,
                              uint32_t *best_good, uint8_t *best_dir, uint8_t *keep) {
    if (!s || !in_fai || !best_contig || !best_good || !best_dir || !keep) {
        set_error("gs_diag_filter: null argument");
        return GS_E_ARG;
    }
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_diag_filter: HIP device %d not available -- there is no CPU fallback", device);
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    const uint64_t n_rows = s->row_seg.size();
    const uint32_t n_segs = (uint32_t)s->seg_names.size();
    // Host walk: per read, the Table's slot order of its contigs; rows laid out by (read, contig).
    std::vector<uint32_t> chash(s->contigs.size());
    for (size_t c = 0; c < s->contigs.size(); c++) {
        uint32_t h = murmur3_32(s->contigs[c].data(), s->contigs[c].size());
        chash[c] = h ? h : 314159265u;  // Nim's `genHash`: a zero hash code marks an empty slot
    }
    std::vector<uint32_t> seg_pair_off(n_segs + 1, 0), pair_contig, pair_rows;
    std::vector<uint64_t> pair_off(1, 0);
    pair_rows.reserve(n_rows);
    NimTableOrder table;
    std::vector<uint32_t> slot_of_contig(s->contigs.size(), 0), counts;
    uint64_t r = 0;
    for (uint32_t seg = 0; seg < n_segs; seg++) {
        const uint64_t r0 = r;
        table.clear();
        while (r < n_rows && s->row_seg[r] == seg) {
            table.add(s->row_contig[r], chash[s->row_contig[r]]);  // entry is created even outside the fai (:145-146)
            r++;
        }
        seg_pair_off[seg] = (uint32_t)pair_contig.size();
        const size_t p0 = pair_contig.size();
        table.for_each([&](uint32_t c) {
            if (in_fai[c]) {  // rows of other contigs are never stored (:147-150); their empty entries cannot win
                slot_of_contig[c] = (uint32_t)(pair_contig.size() - p0);
                pair_contig.push_back(c);
            }
        });
        const size_t np = pair_contig.size() - p0;
        counts.assign(np + 1, 0);
        for (uint64_t i = r0; i < r; i++)
            if (in_fai[s->row_contig[i]]) counts[slot_of_contig[s->row_contig[i]] + 1]++;
        for (size_t i = 0; i < np; i++) counts[i + 1] += counts[i];
        const uint64_t base = pair_rows.size();
        pair_rows.resize(base + counts[np]);
        for (size_t i = 0; i < np; i++) pair_off.push_back(base + counts[i + 1]);
        std::vector<uint32_t> fill(counts.begin(), counts.end() - 1);
        for (uint64_t i = r0; i < r; i++)
            if (in_fai[s->row_contig[i]]) pair_rows[base + fill[slot_of_contig[s->row_contig[i]]]++] = (uint32_t)i;
    }
    seg_pair_off[n_segs] = (uint32_t)pair_contig.size();
    const uint32_t n_pairs = (uint32_t)pair_contig.size();

    DevBuf db;
    int rc;
    int64_t *d_pos, *d_start, *d_group, *d_sa, *d_sb;
    uint32_t *d_row_seg, *d_row_contig, *d_pair_rows, *d_pair_contig, *d_seg_pair_off, *d_gm, *d_gp, *d_bc, *d_bg;
    uint64_t *d_pair_off;
    uint8_t *d_bd, *d_keep;
    if ((rc = db.put(&d_pos, s->row_pos.data(), n_rows)) || (rc = db.put(&d_start, s->row_start.data(), n_rows)) ||
        (rc = db.put(&d_group, s->row_group.data(), n_rows)) || (rc = db.put(&d_row_seg, s->row_seg.data(), n_rows)) ||
        (rc = db.put(&d_row_contig, s->row_contig.data(), n_rows)) ||
        (rc = db.put(&d_pair_rows, pair_rows.data(), pair_rows.size())) ||
        (rc = db.put(&d_pair_contig, pair_contig.data(), n_pairs)) ||
        (rc = db.put(&d_pair_off, pair_off.data(), pair_off.size())) ||
        (rc = db.put(&d_seg_pair_off, seg_pair_off.data(), seg_pair_off.size())) ||
        (rc = db.put(&d_sa, (const int64_t *)nullptr, pair_rows.size())) ||
        (rc = db.put(&d_sb, (const int64_t *)nullptr, pair_rows.size())) ||
        (rc = db.put(&d_gm, (const uint32_t *)nullptr, n_pairs)) || (rc = db.put(&d_gp, (const uint32_t *)nullptr, n_pairs)) ||
        (rc = db.put(&d_bc, (const uint32_t *)nullptr, n_segs)) || (rc = db.put(&d_bg, (const uint32_t *)nullptr, n_segs)) ||
        (rc = db.put(&d_bd, (const uint8_t *)nullptr, n_segs)) || (rc = db.put(&d_keep, (const uint8_t *)nullptr, n_rows)))
        return rc;
    if (n_pairs)
        hipLaunchKernelGGL(k_diag_stats, dim3((n_pairs + 127) / 128), dim3(128), 0, 0, n_pairs, d_pair_off, d_pair_rows,
                           d_pos, d_start, d_group, d_sa, d_sb, d_gm, d_gp, (int64_t)2500);
    if (n_segs)
        hipLaunchKernelGGL(k_diag_select, dim3((n_segs + 127) / 128), dim3(128), 0, 0, n_segs, d_seg_pair_off, d_pair_off,
                           d_pair_contig, d_gm, d_gp, d_bc, d_bg, d_bd);
    if (n_rows)
        hipLaunchKernelGGL(k_diag_keep, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, 0, n_rows, d_row_seg,
                           d_row_contig, d_bc, d_keep);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (n_segs) {
        HIP_TRY(hipMemcpy(best_contig, d_bc, n_segs * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(best_good, d_bg, n_segs * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(best_dir, d_bd, n_segs, hipMemcpyDeviceToHost));
    }
    if (n_rows) HIP_TRY(hipMemcpy(keep, d_keep, n_rows, hipMemcpyDeviceToHost));
    return GS_OK;
}

// The device-resident second half of the engine: the tagger's hit list (HBM) -> rows -> diagonal filter -> bad-SUNK histogram ->
// file-level filtering of process-by-contig -> pair test / component -> contig-wide union, all on device arrays.  The file
// drop-ins (diag_filter_v3, badsunks, process-by-contig) are thin users of the same functions over rows parsed from text.
//
// Reference semantics (cited where they are implemented):
//   workflow/src/diag_filter_v3.nim:63-229, diag_filter_step2.nim:45-64
//   workflow/scripts/badsunks_AR.py:20-103
//   workflow/scripts/process-by-contig_lowmem_AR.py:50-260
#include <hip/hip_runtime.h>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/functional.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "gs_common.h"

using namespace gs;


// kmer.loc rows on the device: what turns a hit's payload (loc row) into contig / start / ID / dense (contig, ID) key
struct gs_locrows {
    int device = 0;
    uint64_t n_rows = 0;
    uint32_t n_locs = 0;
    bool owned = false;
    const uint64_t *pos = nullptr;    // contig << 32 | start
    const uint32_t *group = nullptr;  // ID
    const uint32_t *loc = nullptr;    // dense id of (contig, ID)
};

// `.sunkpos` rows as arrays in device memory, file order
struct gs_rowset {
    int device = 0;
    uint64_t n = 0, cap = 0;
    uint32_t *name = nullptr, *pos = nullptr, *contig = nullptr, *start = nullptr, *id = nullptr, *key = nullptr;
    bool has_key = true;
    // segments (runs of rows with one name) and the diagonal filter's verdict, valid while n == seg_n
    uint64_t seg_n = ~0ULL;
    uint32_t n_segs = 0;
    uint32_t *seg = nullptr;  // per row
    uint8_t *keep = nullptr;  // per row
    bool has_keep = false;
};

struct gs_pairstage {
    std::vector<uint32_t> row_contig, row_name;
    std::vector<int64_t> row_id;
    std::vector<uint32_t> region_contig;
    std::vector<int64_t> region_start, region_end;
    std::vector<uint32_t> contigs_in;  // distinct contigs of the input rows, ascending
    uint64_t counters[8] = {0};        // rows in | not bad | reads with >= 2 IDs | de-duplicated | long enough | read groups | pairs' reads printed | -
};

namespace {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GS_E_NODEVICE;                                                             \
        }                                                                                     \
    } while (0)
#define GS_TRY(expr)              \
    do {                          \
        int _rc = (expr);         \
        if (_rc != GS_OK) return _rc; \
    } while (0)

constexpr uint32_t kNone = 0xFFFFFFFFu;

struct Dev {  // device allocations of one call
    std::vector<void *> ptrs;
    ~Dev() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    int get(T **d, uint64_t n) {
        *d = nullptr;
        hipError_t e = hipMalloc((void **)d, std::max<uint64_t>(n, 1) * sizeof(T));
        if (e != hipSuccess) {
            set_error("device allocation of %llu bytes failed: %s", (unsigned long long)(n * sizeof(T)), hipGetErrorString(e));
            return GS_E_NOMEM;
        }
        ptrs.push_back(*d);
        return GS_OK;
    }
    template <typename T>
    int put(T **d, const T *h, uint64_t n) {
        GS_TRY(get(d, n));
        if (h && n) HIP_TRY(hipMemcpy(*d, h, n * sizeof(T), hipMemcpyHostToDevice));
        return GS_OK;
    }
    void release(void *p) {  // free early
        for (auto &q : ptrs)
            if (q == p) {
                (void)hipFree(p);
                q = nullptr;
            }
    }
};

inline unsigned blocks_of(uint64_t n, unsigned per = 256) { return (unsigned)std::max<uint64_t>((n + per - 1) / per, 1); }

template <typename T, typename Op>
int inclusive_scan(const T *in, T *out, uint64_t n, Op op) {
    if (!n) return GS_OK;
    size_t tb = 0;
    HIP_TRY(rocprim::inclusive_scan(nullptr, tb, in, out, (size_t)n, op));
    void *tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
    hipError_t e = rocprim::inclusive_scan(tmp, tb, in, out, (size_t)n, op);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(tmp);
    HIP_TRY(e);
    return GS_OK;
}
template <typename T>
int exclusive_sum(const T *in, T *out, uint64_t n) {
    if (!n) return GS_OK;
    size_t tb = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tb, in, out, (T)0, (size_t)n, rocprim::plus<T>()));
    void *tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
    hipError_t e = rocprim::exclusive_scan(tmp, tb, in, out, (T)0, (size_t)n, rocprim::plus<T>());
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(tmp);
    HIP_TRY(e);
    return GS_OK;
}
// indices i in [0, n) with flag[i] != 0, ascending -> out; *n_out = how many
int select_indices(const uint8_t *flag, uint32_t *out, uint64_t n, uint64_t *n_out) {
    *n_out = 0;
    if (!n) return GS_OK;
    unsigned long long *d_count = nullptr;
    HIP_TRY(hipMalloc((void **)&d_count, sizeof(unsigned long long)));
    rocprim::counting_iterator<uint32_t> iota(0u);
    size_t tb = 0;
    hipError_t e = rocprim::select(nullptr, tb, iota, flag, out, d_count, (size_t)n);
    void *tmp = nullptr;
    if (e == hipSuccess) e = hipMalloc(&tmp, std::max<size_t>(tb, 16));
    if (e == hipSuccess) e = rocprim::select(tmp, tb, iota, flag, out, d_count, (size_t)n);
    unsigned long long c = 0;
    if (e == hipSuccess) e = hipMemcpy(&c, d_count, sizeof(c), hipMemcpyDeviceToHost);
    if (tmp) (void)hipFree(tmp);
    (void)hipFree(d_count);
    HIP_TRY(e);
    *n_out = c;
    return GS_OK;
}

int check_device(int device, const char *who) {
    if (gs_device_count() <= device || device < 0) {
        set_error("%s: HIP device %d not available -- there is no CPU fallback", who, device);
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    // rocprim returns whatever hipGetLastError holds after its launches: an error another library left behind on this thread (RCCL
    // probes host pointers with hipPointerGetAttributes, which records "invalid argument") must not fail the next scan
    (void)hipGetLastError();
    return GS_OK;
}

// ---- Nim >= 1.4 `hash(string)` = MurmurHash3_x86_32, seed 0; a zero hash code is replaced (Nim's `genHash`) ----
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
    return h ? h : 314159265u;
}

// ------------------------------------------------------------------------------------------
// rows
// ------------------------------------------------------------------------------------------
__global__ void k_rows_from_hits(const gs_hit *hits, uint64_t n, uint32_t name_base, const uint64_t *lpos, const uint32_t *lgroup,
                                 const uint32_t *lloc, uint32_t *name, uint32_t *pos, uint32_t *contig, uint32_t *start, uint32_t *id,
                                 uint32_t *key) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const gs_hit h = hits[i];
    const uint64_t p = lpos[h.payload];
    name[i] = name_base + h.read;
    pos[i] = h.pos;
    contig[i] = (uint32_t)(p >> 32);
    start[i] = (uint32_t)p;
    id[i] = lgroup[h.payload];
    key[i] = lloc[h.payload];
}
__global__ void k_pack_pos(const uint32_t *contig, const int64_t *start, const int64_t *group, uint64_t n, uint64_t *pos, uint32_t *g32) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pos[i] = ((uint64_t)contig[i] << 32) | (uint32_t)start[i];
    g32[i] = (uint32_t)group[i];
}
template <typename T>
__global__ void k_gather(const T *in, const uint32_t *idx, uint64_t n, T *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}
__global__ void k_iota(uint32_t *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}

int rowset_reserve(gs_rowset *rs, uint64_t want) {
    if (want <= rs->cap) return GS_OK;
    const uint64_t cap = std::max<uint64_t>(want, rs->cap + rs->cap / 2 + 1024);
    uint32_t **cols[6] = {&rs->name, &rs->pos, &rs->contig, &rs->start, &rs->id, &rs->key};
    for (auto c : cols) {
        uint32_t *nw = nullptr;
        hipError_t e = hipMalloc((void **)&nw, cap * sizeof(uint32_t));
        if (e != hipSuccess) {
            set_error("gs_rowset: device allocation of %llu rows failed: %s", (unsigned long long)cap, hipGetErrorString(e));
            return GS_E_NOMEM;
        }
        if (rs->n) HIP_TRY(hipMemcpy(nw, *c, rs->n * sizeof(uint32_t), hipMemcpyDeviceToDevice));
        if (*c) (void)hipFree(*c);
        *c = nw;
    }
    rs->cap = cap;
    return GS_OK;
}
void rowset_drop_derived(gs_rowset *rs) {
    if (rs->seg) (void)hipFree(rs->seg);
    if (rs->keep) (void)hipFree(rs->keep);
    rs->seg = nullptr;
    rs->keep = nullptr;
    rs->seg_n = ~0ULL;
    rs->has_keep = false;
}

// ------------------------------------------------------------------------------------------
// diagonal filter
// ------------------------------------------------------------------------------------------
__global__ void k_head_u32(const uint32_t *v, uint64_t n, uint32_t *flag) {  // 1 where v[i] != v[i-1]
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || v[i] != v[i - 1]) ? 1u : 0u;
}
__global__ void k_head_u64(const uint64_t *v, uint64_t n, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || v[i] != v[i - 1]) ? 1u : 0u;
}
__global__ void k_head_u64_b(const uint64_t *v, uint64_t n, uint8_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || v[i] != v[i - 1]) ? 1 : 0;
}
__global__ void k_dec(uint32_t *v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] -= 1u;
}
__global__ void k_seg_keys(const uint32_t *seg, const uint32_t *contig, uint64_t n, uint64_t *key, uint32_t *val) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        key[i] = ((uint64_t)seg[i] << 32) | contig[i];
        val[i] = (uint32_t)i;
    }
}
// heads of the sorted (seg, contig) keys -> the pairs: pair_off, pair_key; and the first pair of every segment
__global__ void k_pair_fill(const uint64_t *key, const uint32_t *head, const uint32_t *pidx, uint64_t n, uint64_t *pair_off,
                            uint64_t *pair_key) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && head[j]) {
        pair_off[pidx[j]] = j;
        pair_key[pidx[j]] = key[j];
    }
}
__global__ void k_seg_pair_off(const uint64_t *pair_key, uint32_t n_pairs, uint32_t *seg_pair_off) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const uint32_t s = (uint32_t)(pair_key[p] >> 32);
    if (p == 0 || (uint32_t)(pair_key[p - 1] >> 32) != s) seg_pair_off[s] = p;
}
// Nim's Table: 64 slots to begin with; before a NEW key goes in, the table doubles when dataLen * 2 < counter * 3 or
// dataLen - counter < 4 (counter = entries so far); clear() keeps the capacity.  A table of n slots therefore takes
// floor(2 n / 3) + 1 entries without growing: the capacity a read with d distinct contigs needs.
__host__ __device__ inline uint32_t nim_cap_for(uint32_t d) {
    uint32_t n = 64;
    while (d > 0 && (2u * n) / 3u + 1u < d) n *= 2u;
    // (the second trigger, n - c < 4, is implied for n >= 64)
    return n;
}
__global__ void k_seg_need(const uint32_t *seg_pair_off, uint32_t n_segs, uint32_t *need) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_segs) need[s] = nim_cap_for(seg_pair_off[s + 1] - seg_pair_off[s]);
}
// the slot of every (read, contig) entry in the read's Table: insertion in order of first appearance, linear probing.
// Reads during which the table does not grow (all but a handful per run): the occupied slots are the pairs placed so far.
// A read that makes it grow is listed in grow_list and replayed exactly by k_slots_grow.
__global__ void k_slots(uint32_t n_segs, const uint32_t *seg_pair_off, const uint32_t *need, const uint32_t *need_prefmax,
                        const uint64_t *pair_off, const uint64_t *pair_key, const uint32_t *rows_sorted, const uint32_t *chash,
                        uint32_t *pair_slot, uint32_t *grow_list, uint32_t *grow_count, uint32_t grow_cap, uint32_t cap0) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    const uint32_t cap_before = s ? max(cap0, need_prefmax[s - 1]) : cap0;
    const uint32_t p0 = seg_pair_off[s], p1 = seg_pair_off[s + 1];
    if (need[s] > cap_before) {
        const uint32_t g = atomicAdd(grow_count, 1u);
        if (g < grow_cap) grow_list[g] = s;
        return;
    }
    const uint32_t mask = cap_before - 1u;
    for (uint32_t p = p0; p < p1; p++) pair_slot[p] = kNone;
    for (uint32_t step = p0; step < p1; step++) {
        // the next contig to appear in the read: the smallest first row among the pairs not placed yet
        uint32_t best = kNone, bestrow = kNone;
        for (uint32_t p = p0; p < p1; p++) {
            if (pair_slot[p] != kNone) continue;
            const uint32_t fr = rows_sorted[pair_off[p]];
            if (fr < bestrow) {
                bestrow = fr;
                best = p;
            }
        }
        uint32_t h = chash[(uint32_t)pair_key[best]] & mask;
        for (;;) {
            bool taken = false;
            for (uint32_t p = p0; p < p1; p++)
                if (pair_slot[p] == h) {
                    taken = true;
                    break;
                }
            if (!taken) break;
            h = (h + 1u) & mask;
        }
        pair_slot[best] = h;
    }
}
// exact replay of a growing Table (one thread per such read; scratch: 4 arrays of the final capacity)
__global__ void k_slots_grow(uint32_t n_grow, const uint32_t *grow_list, const uint64_t *scratch_off, uint32_t *scratch,
                             const uint32_t *seg_pair_off, const uint32_t *need_prefmax, const uint64_t *pair_off,
                             const uint64_t *pair_key, const uint32_t *rows_sorted, const uint32_t *chash, uint32_t *pair_slot,
                             uint32_t cap0) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_grow) return;
    const uint32_t s = grow_list[g];
    uint32_t n = s ? max(cap0, need_prefmax[s - 1]) : cap0;
    const uint32_t p0 = seg_pair_off[s], p1 = seg_pair_off[s + 1];
    const uint32_t capf = max(n, nim_cap_for(p1 - p0));
    uint32_t *key = scratch + scratch_off[g], *hc = key + capf, *nk = hc + capf, *nh = nk + capf;
    for (uint32_t i = 0; i < capf; i++) key[i] = kNone;
    for (uint32_t p = p0; p < p1; p++) pair_slot[p] = kNone;
    uint32_t counter = 0;
    for (uint32_t step = p0; step < p1; step++) {
        uint32_t best = kNone, bestrow = kNone;
        for (uint32_t p = p0; p < p1; p++) {
            if (pair_slot[p] != kNone) continue;
            const uint32_t fr = rows_sorted[pair_off[p]];
            if (fr < bestrow) {
                bestrow = fr;
                best = p;
            }
        }
        const uint32_t h0 = chash[(uint32_t)pair_key[best]];
        if (n * 2u < counter * 3u || n - counter < 4u) {  // enlarge: the old slots re-inserted in index order
            for (uint32_t i = 0; i < 2u * n; i++) nk[i] = kNone;
            for (uint32_t i = 0; i < n; i++)
                if (key[i] != kNone) {
                    uint32_t j = hc[i] & (2u * n - 1u);
                    while (nk[j] != kNone) j = (j + 1u) & (2u * n - 1u);
                    nk[j] = key[i];
                    nh[j] = hc[i];
                }
            n *= 2u;
            for (uint32_t i = 0; i < n; i++) {
                key[i] = nk[i];
                hc[i] = nh[i];
            }
        }
        uint32_t h = h0 & (n - 1u);
        while (key[h] != kNone) h = (h + 1u) & (n - 1u);
        key[h] = best;  // (the pair index stands for the contig)
        hc[h] = h0;
        counter++;
        pair_slot[best] = 0u;  // placed (the real slot comes at the end)
    }
    for (uint32_t i = 0; i < n; i++)
        if (key[i] != kNone) pair_slot[key[i]] = i;
}

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
// one thread per (read, contig): median diagonal (sort, fp64 interpolation at (len - 1) / 2, truncation toward zero:
// diag_filter_v3.nim:86,113), distinct groups within 2500 of it (:91-94), "-" (pos + start) and "+" (start - pos)
__global__ void k_diag_stats_rows(uint32_t n_pairs, const uint64_t *pair_off, const uint64_t *pair_key, const uint32_t *rows_sorted,
                                  const uint32_t *pos, const uint32_t *start, const uint32_t *id, const uint8_t *in_fai,
                                  int64_t *scratch_a, int64_t *scratch_b, uint32_t *good_minus, uint32_t *good_plus, int64_t bandwidth) {
    const uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= n_pairs) return;
    const uint64_t lo = pair_off[pi], hi = pair_off[pi + 1];
    const int n = (int)(hi - lo);
    if (n <= 1 || !in_fai[(uint32_t)pair_key[pi]]) {  // `if p.shape[0] == 1: continue` (:82); rows of other contigs are never stored (:147-150)
        good_minus[pi] = 0;
        good_plus[pi] = 0;
        return;
    }
    int64_t *sa = scratch_a + lo, *sb = scratch_b + lo;
    for (int dir = 0; dir < 2; dir++) {
        for (int i = 0; i < n; i++) {
            const uint32_t r = rows_sorted[lo + i];
            sa[i] = dir == 0 ? (int64_t)pos[r] + (int64_t)start[r] : (int64_t)start[r] - (int64_t)pos[r];
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
            const uint32_t r = rows_sorted[lo + i];
            const int64_t v = dir == 0 ? (int64_t)pos[r] + (int64_t)start[r] : (int64_t)start[r] - (int64_t)pos[r];
            int64_t dlt = v - median;
            if (dlt < 0) dlt = -dlt;
            if (dlt < bandwidth) sb[m++] = (int64_t)id[r];
        }
        heapsort64(sb, m);
        uint32_t distinct = 0;
        for (int i = 0; i < m; i++) distinct += (i == 0 || sb[i] != sb[i - 1]) ? 1u : 0u;
        (dir == 0 ? good_minus : good_plus)[pi] = distinct;
    }
}
// one thread per read: the reference's selection sequence (:98-107) over its contigs in Table slot order
__global__ void k_diag_select_slots(uint32_t n_segs, const uint32_t *seg_pair_off, const uint64_t *pair_off, const uint64_t *pair_key,
                                    const uint32_t *pair_slot, const uint8_t *in_fai, const uint32_t *good_minus, const uint32_t *good_plus,
                                    uint32_t *best_contig, uint32_t *best_good, uint8_t *best_dir) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    uint32_t maxgood = 0, bestc = GS_NONE;
    uint64_t maxhitlen = 0;
    uint8_t dirc = 0;
    const uint32_t p0 = seg_pair_off[s], p1 = seg_pair_off[s + 1];
    int64_t last = -1;
    for (uint32_t step = p0; step < p1; step++) {
        uint32_t pi = kNone, sl = kNone;
        for (uint32_t p = p0; p < p1; p++)
            if ((int64_t)pair_slot[p] > last && pair_slot[p] < sl) {
                sl = pair_slot[p];
                pi = p;
            }
        last = (int64_t)sl;
        const uint32_t c = (uint32_t)pair_key[pi];
        if (!in_fai[c]) continue;
        const uint64_t hitlen = pair_off[pi + 1] - pair_off[pi];
        if (hitlen == 1) continue;
        for (int dir = 0; dir < 2; dir++) {
            const uint32_t good = dir == 0 ? good_minus[pi] : good_plus[pi];
            if (good > maxgood) {  // :98-102
                maxgood = good;
                maxhitlen = hitlen;
                bestc = c;
                dirc = dir == 0 ? '-' : '+';
            }
            if (good == maxgood && hitlen < maxhitlen) {  // :103-107
                maxgood = good;
                maxhitlen = hitlen;
                bestc = c;
                dirc = dir == 0 ? '-' : '+';
            }
        }
    }
    const bool print = maxgood > 1;  // :141
    best_contig[s] = print ? bestc : GS_NONE;
    best_good[s] = maxgood;
    best_dir[s] = dirc;
}
__global__ void k_keep_rows(uint64_t n, const uint32_t *seg, const uint32_t *contig, const uint32_t *best_contig, uint8_t *keep) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = best_contig[seg[i]];
    keep[i] = (b != GS_NONE && b == contig[i]) ? 1 : 0;  // diag_filter_step2.nim:45-64
}
__global__ void k_seg_first(const uint32_t *flag, const uint32_t *seg, const uint32_t *name, uint64_t n, uint32_t *seg_name) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) seg_name[seg[i]] = name[i];
}

// ------------------------------------------------------------------------------------------
// bad SUNKs
// ------------------------------------------------------------------------------------------
__global__ void k_count_keys(const uint32_t *key, uint64_t n, uint32_t *count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&count[key[i]], 1u);
}
__global__ void k_hist_counts(const uint32_t *count, const uint8_t *key_hap, uint8_t hapbit, uint32_t n_keys, uint32_t *hist,
                              uint32_t hist_len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_keys && count[i] && (key_hap[i] & hapbit)) atomicAdd(&hist[min(count[i], hist_len - 1)], 1u);
}
__global__ void k_mark_bad(const uint32_t *count, const uint8_t *key_hap, uint8_t hapbit, uint32_t n_keys, double limit, uint8_t *bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_keys || !count[i]) return;  // (value_counts holds the keys that occur)
    const bool own = (key_hap[i] & hapbit) != 0;
    const double c = (double)count[i];
    if (own ? (c > limit || count[i] < 2u) : (c > limit)) bad[i] = 1;  // badsunks_AR.py:48-51
}

// ------------------------------------------------------------------------------------------
// process-by-contig: file-level filtering
// ------------------------------------------------------------------------------------------
__global__ void k_not_bad(const uint32_t *key, const uint8_t *key_bad, const uint8_t *row_bad, uint64_t n, uint8_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool bad = false;
    if (key_bad) bad = key_bad[key[i]] != 0;
    if (row_bad) bad = bad || row_bad[i] != 0;
    flag[i] = bad ? 0 : 1;
}
// sort keys of the rows idx[j]: group = contig << 32 | rank of the read's name; second = the u32 column
__global__ void k_group_key(const uint32_t *idx, uint64_t n, const uint32_t *contig, const uint32_t *name, const uint32_t *name_rank,
                            uint64_t *gkey) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) gkey[j] = ((uint64_t)contig[idx[j]] << 32) | name_rank[name[idx[j]]];
}
__global__ void k_col_key(const uint32_t *idx, uint64_t n, const uint32_t *col, uint64_t *k) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) k[j] = col[idx[j]];
}
// after the sort by (group, ID): distinct IDs per group
__global__ void k_count_ids(const uint64_t *gkey, const uint32_t *idx, const uint32_t *id, const uint32_t *gidx, uint64_t n, uint32_t *cnt) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (j == 0 || gkey[j] != gkey[j - 1] || id[idx[j]] != id[idx[j - 1]]) atomicAdd(&cnt[gidx[j]], 1u);
}
__global__ void k_flag_multi(const uint32_t *gidx, const uint32_t *cnt, uint64_t n, const uint32_t *idx, uint8_t *row_flag) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) row_flag[idx[j]] = cnt[gidx[j]] > 1u ? 1 : 0;  // reads with at least two distinct IDs (:91-97)
}
// rows sorted by (group, start), original order inside: an exact duplicate of an earlier row goes (drop_duplicates, :104);
// then the reads of at least 10 kb (:106-108)
__global__ void k_dedup_long(const uint64_t *gkey, const uint32_t *idx, uint64_t n, const uint32_t *start, const uint32_t *pos,
                             const uint32_t *id, const uint32_t *name, const uint8_t *name_long, uint8_t *flag_dedup, uint8_t *flag_out) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t r = idx[j];
    bool dup = false;
    for (uint64_t q = j; q > 0; q--) {
        const uint32_t o = idx[q - 1];
        if (gkey[q - 1] != gkey[j] || start[o] != start[r]) break;
        if (pos[o] == pos[r] && id[o] == id[r]) {
            dup = true;
            break;
        }
    }
    flag_dedup[j] = dup ? 0 : 1;
    flag_out[j] = (!dup && name_long[name[r]]) ? 1 : 0;
}
__global__ void k_final_rows(const uint32_t *sel, uint64_t n, const uint64_t *gkey, const uint32_t *idx, const uint32_t *pos,
                             const uint32_t *start, const uint32_t *id, uint64_t *fkey, int64_t *fpos, int64_t *fstart, int64_t *fid,
                             uint32_t *frow) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t q = sel[j], r = idx[q];
    fkey[j] = gkey[q];
    fpos[j] = (int64_t)pos[r];
    fstart[j] = (int64_t)start[r];
    fid[j] = (int64_t)id[r];
    frow[j] = r;
}
__global__ void k_group_off(const uint32_t *head, const uint32_t *gidx, uint64_t n, uint64_t *off) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && head[j]) off[gidx[j]] = j;
}
// the `ID \t rname` rows of every read group that prints something, in group order
__global__ void k_out_rows(uint32_t n_groups, const uint64_t *off, const uint32_t *out_n, const uint64_t *out_off, const int64_t *out_ids,
                           const uint32_t *frow, const uint32_t *contig, const uint32_t *name, uint32_t *o_contig, uint32_t *o_name,
                           int64_t *o_id, uint32_t *o_group) {
    const uint32_t g = blockIdx.x;
    const uint32_t r0 = frow[off[g]];
    for (uint32_t i = threadIdx.x; i < out_n[g]; i += blockDim.x) {
        const uint64_t w = out_off[g] + i;
        o_contig[w] = contig[r0];
        o_name[w] = name[r0];
        o_id[w] = out_ids[off[g] + i];
        o_group[w] = g;
    }
}
__global__ void k_vertex_key(const uint32_t *o_contig, const int64_t *o_id, uint64_t n, uint64_t *vk) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vk[i] = ((uint64_t)o_contig[i] << 32) | (uint32_t)o_id[i];
}
__global__ void k_unique_fill(const uint64_t *sorted, const uint32_t *head, const uint32_t *vidx, uint64_t n, uint64_t *uniq, int64_t *vid) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && head[j]) {
        uniq[vidx[j]] = sorted[j];
        vid[vidx[j]] = (int64_t)(uint32_t)sorted[j];
    }
}
__global__ void k_lookup_vertex(const uint64_t *vk, uint64_t n, const uint64_t *uniq, uint32_t nv, uint32_t *row_vertex) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t lo = 0, hi = nv;
    const uint64_t want = vk[i];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (uniq[mid] < want) lo = mid + 1;
        else hi = mid;
    }
    row_vertex[i] = lo;
}
__global__ void k_pack_u32x2(const uint32_t *hi, const uint32_t *lo, uint64_t n, uint64_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint64_t)hi[i] << 32 | lo[i];
}
__global__ void k_widen(const uint32_t *in, uint64_t *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ void k_region_flag(const uint32_t *size, uint32_t nv, uint8_t *flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nv) flag[i] = size[i] > 2u ? 1 : 0;  // components with more than two SUNKs (:247)
}

// stable sort of `idx` (rows) by a 64-bit key (all `bits` low bits): keys and idx are permuted together, result back in the same arrays
int sort_by(uint64_t *key, uint32_t *idx, uint64_t n, int bits, Dev &db) {
    if (n < 2) return GS_OK;
    uint64_t *k2;
    uint32_t *i2;
    GS_TRY(db.get(&k2, n));
    GS_TRY(db.get(&i2, n));
    GS_TRY(sort_pairs_u64_u32(key, k2, idx, i2, n, bits, nullptr));
    HIP_TRY(hipMemcpy(key, k2, n * sizeof(uint64_t), hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy(idx, i2, n * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    db.release(k2);
    db.release(i2);
    return GS_OK;
}

// every v[i] < bound?  (ids that index caller-sized arrays: a wrong count must fail loudly, not read out of bounds)
__global__ void k_any_ge(const uint32_t *v, uint64_t n, uint32_t bound, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && v[i] >= bound) atomicOr(flag, 1u);
}
int check_ids_below(const uint32_t *d_ids, uint64_t n, uint32_t bound, const char *what) {
    if (!n) return GS_OK;
    uint32_t *d_flag = nullptr, h = 0;
    HIP_TRY(hipMalloc((void **)&d_flag, sizeof(uint32_t)));
    hipError_t e = hipMemset(d_flag, 0, sizeof(uint32_t));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_any_ge, dim3(blocks_of(n)), dim3(256), 0, 0, d_ids, n, bound, d_flag);
        e = hipMemcpy(&h, d_flag, sizeof h, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_flag);
    HIP_TRY(e);
    if (h) {
        set_error("%s: a row's id is not below %u", what, bound);
        return GS_E_ARG;
    }
    return GS_OK;
}

int compute_segments(gs_rowset *rs) {
    if (rs->seg_n == rs->n && rs->seg) return GS_OK;
    rowset_drop_derived(rs);
    const uint64_t n = rs->n;
    HIP_TRY(hipMalloc((void **)&rs->seg, std::max<uint64_t>(n, 1) * sizeof(uint32_t)));
    rs->n_segs = 0;
    if (n) {
        hipLaunchKernelGGL(k_head_u32, dim3(blocks_of(n)), dim3(256), 0, 0, rs->name, n, rs->seg);
        GS_TRY(inclusive_scan(rs->seg, rs->seg, n, rocprim::plus<uint32_t>()));
        HIP_TRY(hipMemcpy(&rs->n_segs, rs->seg + n - 1, sizeof(uint32_t), hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_dec, dim3(blocks_of(n)), dim3(256), 0, 0, rs->seg, n);
        HIP_TRY(hipGetLastError());
    }
    rs->seg_n = n;
    return GS_OK;
}


// process-by-contig_lowmem_AR.py:219-260 over T output rows in device memory (contig, ID, group = the read the row belongs to; a
// read's rows are adjacent): vertices = distinct (contig, ID), an edge between consecutive rows of a read, components with more
// than two SUNKs -> [min, max] per contig.  Commutative over reads: the rows of several ranks, concatenated, give the union of
// the whole run (gs_pairstage_from_rows).
int union_of_rows(Dev &db, uint64_t T, const uint32_t *d_oc, const int64_t *d_oi, const uint32_t *d_og, gs_pairstage *ps) {
    uint64_t *d_vk, *d_vs, *d_uniq;
    uint32_t *d_vh, *d_vi, *d_rv, *d_size;
    int64_t *d_vid, *d_min, *d_max;
    GS_TRY(db.get(&d_vk, T));
    GS_TRY(db.get(&d_vs, T));
    hipLaunchKernelGGL(k_vertex_key, dim3(blocks_of(T)), dim3(256), 0, 0, d_oc, d_oi, T, d_vk);
    GS_TRY(sort_keys_u64(d_vk, d_vs, T, nullptr));
    GS_TRY(db.get(&d_vh, T));
    GS_TRY(db.get(&d_vi, T));
    hipLaunchKernelGGL(k_head_u64, dim3(blocks_of(T)), dim3(256), 0, 0, d_vs, T, d_vh);
    GS_TRY(inclusive_scan(d_vh, d_vi, T, rocprim::plus<uint32_t>()));
    uint32_t nv = 0;
    HIP_TRY(hipMemcpy(&nv, d_vi + T - 1, 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_dec, dim3(blocks_of(T)), dim3(256), 0, 0, d_vi, T);
    GS_TRY(db.get(&d_uniq, nv));
    GS_TRY(db.get(&d_vid, nv));
    hipLaunchKernelGGL(k_unique_fill, dim3(blocks_of(T)), dim3(256), 0, 0, d_vs, d_vh, d_vi, T, d_uniq, d_vid);
    GS_TRY(db.get(&d_rv, T));
    hipLaunchKernelGGL(k_lookup_vertex, dim3(blocks_of(T)), dim3(256), 0, 0, d_vk, T, d_uniq, nv, d_rv);
    GS_TRY(db.get(&d_size, nv));
    GS_TRY(db.get(&d_min, nv));
    GS_TRY(db.get(&d_max, nv));
    HIP_TRY(hipDeviceSynchronize());
    GS_TRY(union_regions_device(T, d_og, d_rv, nv, d_vid, d_size, d_min, d_max));
    uint8_t *d_rf;
    uint32_t *d_rsel;
    GS_TRY(db.get(&d_rf, nv));
    GS_TRY(db.get(&d_rsel, nv));
    hipLaunchKernelGGL(k_region_flag, dim3(blocks_of(nv)), dim3(256), 0, 0, d_size, nv, d_rf);
    uint64_t nr = 0;
    GS_TRY(select_indices(d_rf, d_rsel, nv, &nr));
    if (nr) {
        std::vector<uint32_t> sel(nr);
        std::vector<uint64_t> uq(nv);
        std::vector<int64_t> mn(nv), mx(nv);
        HIP_TRY(hipMemcpy(sel.data(), d_rsel, nr * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(uq.data(), d_uniq, (size_t)nv * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(mn.data(), d_min, (size_t)nv * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(mx.data(), d_max, (size_t)nv * 8, hipMemcpyDeviceToHost));
        struct Reg {
            uint32_t c;
            int64_t s, e;
        };
        std::vector<Reg> regs;
        for (uint32_t v : sel) regs.push_back(Reg{(uint32_t)(uq[v] >> 32), mn[v], mx[v]});
        std::sort(regs.begin(), regs.end(), [](const Reg &a, const Reg &b) {  // `sort_values(by='start')` per contig (:254); (start, end) as the stub always wrote
            return a.c != b.c ? a.c < b.c : (a.s != b.s ? a.s < b.s : a.e < b.e);
        });
        for (const Reg &r : regs) {
            ps->region_contig.push_back(r.c);
            ps->region_start.push_back(r.s);
            ps->region_end.push_back(r.e);
        }
    }
    return GS_OK;
}
}  // namespace

extern "C" {

// ---- loc rows ----------------------------------------------------------------------------------
int gs_locrows_create(int device, uint64_t n_rows, const uint32_t *row_contig, const int64_t *row_start, const int64_t *row_group,
                      const uint32_t *row_loc, gs_locrows **out) {
    if (!out || (n_rows && (!row_contig || !row_start || !row_group || !row_loc))) {
        set_error("gs_locrows_create: null argument");
        return GS_E_ARG;
    }
    GS_TRY(check_device(device, "gs_locrows_create"));
    Dev tmp;
    uint32_t *d_c;
    int64_t *d_s, *d_g;
    GS_TRY(tmp.put(&d_c, row_contig, n_rows));
    GS_TRY(tmp.put(&d_s, row_start, n_rows));
    GS_TRY(tmp.put(&d_g, row_group, n_rows));
    uint64_t *pos = nullptr;
    uint32_t *grp = nullptr, *loc = nullptr;
    HIP_TRY(hipMalloc((void **)&pos, std::max<uint64_t>(n_rows, 1) * sizeof(uint64_t)));
    HIP_TRY(hipMalloc((void **)&grp, std::max<uint64_t>(n_rows, 1) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&loc, std::max<uint64_t>(n_rows, 1) * sizeof(uint32_t)));
    if (n_rows) {
        hipLaunchKernelGGL(k_pack_pos, dim3(blocks_of(n_rows)), dim3(256), 0, 0, d_c, d_s, d_g, n_rows, pos, grp);
        HIP_TRY(hipMemcpy(loc, row_loc, n_rows * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipDeviceSynchronize());
    }
    uint32_t n_locs = 0;
    for (uint64_t i = 0; i < n_rows; i++)
        if (row_loc[i] != GS_MISSING) n_locs = std::max(n_locs, row_loc[i] + 1u);
    gs_locrows *lr = new gs_locrows;
    lr->device = device;
    lr->n_rows = n_rows;
    lr->n_locs = n_locs;
    lr->owned = true;
    lr->pos = pos;
    lr->group = grp;
    lr->loc = loc;
    *out = lr;
    return GS_OK;
}

int gs_locrows_from_sunkset_dev(int device, const gs_sunkset_dev *s, gs_locrows **out) {
    if (!s || !out) {
        set_error("gs_locrows_from_sunkset_dev: null argument");
        return GS_E_ARG;
    }
    gs_locrows *lr = new gs_locrows;
    lr->device = device;
    lr->n_rows = gs_sunkset_dev_n(s);
    lr->n_locs = (uint32_t)gs_sunkset_dev_n_groups(s);
    lr->owned = false;  // the arrays stay the SUNK set's: it must outlive this handle
    lr->pos = static_cast<const uint64_t *>(gs_sunkset_dev_pos(s));
    lr->group = static_cast<const uint32_t *>(gs_sunkset_dev_group(s));
    lr->loc = static_cast<const uint32_t *>(gs_sunkset_dev_loc(s));
    *out = lr;
    return GS_OK;
}

void gs_locrows_free(gs_locrows *lr) {
    if (!lr) return;
    if (lr->owned) {
        (void)hipSetDevice(lr->device);
        (void)hipFree((void *)lr->pos);
        (void)hipFree((void *)lr->group);
        (void)hipFree((void *)lr->loc);
    }
    delete lr;
}
uint32_t gs_locrows_n_locs(const gs_locrows *lr) { return lr ? lr->n_locs : 0; }
uint64_t gs_locrows_n_rows(const gs_locrows *lr) { return lr ? lr->n_rows : 0; }

// contig of every dense (contig, ID) key (host array of n_locs entries; GS_NONE for keys no row carries)
int gs_locrows_key_contig(const gs_locrows *lr, uint32_t *key_contig) {
    if (!lr || !key_contig) {
        set_error("gs_locrows_key_contig: null argument");
        return GS_E_ARG;
    }
    GS_TRY(check_device(lr->device, "gs_locrows_key_contig"));
    std::fill(key_contig, key_contig + lr->n_locs, GS_NONE);
    const uint64_t chunk = 1ULL << 26;
    std::vector<uint64_t> p(std::min(chunk, std::max<uint64_t>(lr->n_rows, 1)));
    std::vector<uint32_t> l(p.size());
    for (uint64_t a = 0; a < lr->n_rows; a += chunk) {
        const uint64_t m = std::min(chunk, lr->n_rows - a);
        HIP_TRY(hipMemcpy(p.data(), lr->pos + a, m * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(l.data(), lr->loc + a, m * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < m; i++)
            if (l[i] < lr->n_locs) key_contig[l[i]] = (uint32_t)(p[i] >> 32);
    }
    return GS_OK;
}

// ---- row sets ----------------------------------------------------------------------------------
int gs_rowset_create(int device, uint64_t cap_hint, gs_rowset **out) {
    if (!out) {
        set_error("gs_rowset_create: null argument");
        return GS_E_ARG;
    }
    GS_TRY(check_device(device, "gs_rowset_create"));
    gs_rowset *rs = new gs_rowset;
    rs->device = device;
    int rc = rowset_reserve(rs, std::max<uint64_t>(cap_hint, 1024));
    if (rc) {
        delete rs;
        return rc;
    }
    *out = rs;
    return GS_OK;
}
void gs_rowset_free(gs_rowset *rs) {
    if (!rs) return;
    (void)hipSetDevice(rs->device);
    rowset_drop_derived(rs);
    for (uint32_t *p : {rs->name, rs->pos, rs->contig, rs->start, rs->id, rs->key})
        if (p) (void)hipFree(p);
    delete rs;
}
uint64_t gs_rowset_n(const gs_rowset *rs) { return rs ? rs->n : 0; }

int gs_rowset_append_pass(gs_rowset *rs, gs_table *t, gs_reads *r, uint32_t name_base, const gs_locrows *lr) {
    if (!rs || !t || !r || !lr) {
        set_error("gs_rowset_append_pass: null argument");
        return GS_E_ARG;
    }
    GS_TRY(check_device(rs->device, "gs_rowset_append_pass"));
    const void *d_hits = nullptr;
    uint64_t n = 0;
    GS_TRY(gs_tag_hits_device(t, r, &d_hits, &n));
    if (rs->n + n >= 0xFFFFFFFFULL) {
        set_error("gs_rowset_append_pass: more than 2^32 rows in one row set");
        return GS_E_CAPACITY;
    }
    GS_TRY(rowset_reserve(rs, rs->n + n));
    if (n) {
        const uint64_t o = rs->n;
        hipLaunchKernelGGL(k_rows_from_hits, dim3(blocks_of(n)), dim3(256), 0, 0, static_cast<const gs_hit *>(d_hits), n, name_base, lr->pos,
                           lr->group, lr->loc, rs->name + o, rs->pos + o, rs->contig + o, rs->start + o, rs->id + o, rs->key + o);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    rs->n += n;
    rowset_drop_derived(rs);
    return GS_OK;
}

int gs_rowset_append_host(gs_rowset *rs, uint64_t n, const uint32_t *name, const uint32_t *pos, const uint32_t *contig,
                          const uint32_t *start, const uint32_t *id, const uint32_t *key) {
    if (!rs || (n && (!name || !pos || !contig || !start || !id))) {
        set_error("gs_rowset_append_host: null argument");
        return GS_E_ARG;
    }
    GS_TRY(check_device(rs->device, "gs_rowset_append_host"));
    if (rs->n + n >= 0xFFFFFFFFULL) {
        set_error("gs_rowset_append_host: more than 2^32 rows in one row set");
        return GS_E_CAPACITY;
    }
    GS_TRY(rowset_reserve(rs, rs->n + n));
    if (n) {
        const uint64_t o = rs->n;
        HIP_TRY(hipMemcpy(rs->name + o, name, n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(rs->pos + o, pos, n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(rs->contig + o, contig, n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(rs->start + o, start, n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(rs->id + o, id, n * 4, hipMemcpyHostToDevice));
        if (key) HIP_TRY(hipMemcpy(rs->key + o, key, n * 4, hipMemcpyHostToDevice));
        else rs->has_key = false;
    }
    rs->n += n;
    rowset_drop_derived(rs);
    return GS_OK;
}

int gs_rowset_download(const gs_rowset *rs, uint32_t *name, uint32_t *pos, uint32_t *contig, uint32_t *start, uint32_t *id, uint32_t *key) {
    if (!rs) {
        set_error("gs_rowset_download: null argument");
        return GS_E_ARG;
    }
    GS_TRY(check_device(rs->device, "gs_rowset_download"));
    const uint32_t *src[6] = {rs->name, rs->pos, rs->contig, rs->start, rs->id, rs->key};
    uint32_t *dst[6] = {name, pos, contig, start, id, key};
    for (int c = 0; c < 6; c++)
        if (dst[c] && rs->n) HIP_TRY(hipMemcpy(dst[c], src[c], rs->n * 4, hipMemcpyDeviceToHost));
    return GS_OK;
}

int gs_rowset_segments(gs_rowset *rs, uint32_t *n_segs) {
    if (!rs || !n_segs) {
        set_error("gs_rowset_segments: null argument");
        return GS_E_ARG;
    }
    GS_TRY(check_device(rs->device, "gs_rowset_segments"));
    GS_TRY(compute_segments(rs));
    *n_segs = rs->n_segs;
    return GS_OK;
}

// diag_filter_v3 over a row set.  Host outputs (each may be NULL), n_segs entries (gs_rowset_segments):
//   seg_name = name id of the segment's rows; best_contig = contig id or GS_NONE (prints nothing); best_good; best_dir
int gs_rowset_diag_filter(gs_rowset *rs, uint32_t n_contigs, const char *const *contig_names, const uint8_t *in_fai, uint32_t *seg_name,
                          uint32_t *best_contig, uint32_t *best_good, uint8_t *best_dir, uint64_t *n_kept) {
    return gs_rowset_diag_filter_from(rs, n_contigs, contig_names, in_fai, 64, seg_name, best_contig, best_good, best_dir, n_kept, nullptr);
}

int gs_rowset_diag_filter_from(gs_rowset *rs, uint32_t n_contigs, const char *const *contig_names, const uint8_t *in_fai, uint32_t cap_in,
                               uint32_t *seg_name, uint32_t *best_contig, uint32_t *best_good, uint8_t *best_dir, uint64_t *n_kept,
                               uint32_t *cap_out) {
    if (!rs || !contig_names || !in_fai || cap_in < 64 || (cap_in & (cap_in - 1))) {
        set_error("gs_rowset_diag_filter: bad argument (cap_in = the Table capacity carried in: a power of two >= 64)");
        return GS_E_ARG;
    }
    const uint32_t cap0 = cap_in;
    if (cap_out) *cap_out = cap0;
    GS_TRY(check_device(rs->device, "gs_rowset_diag_filter"));
    GS_TRY(check_ids_below(rs->contig, rs->n, n_contigs, "gs_rowset_diag_filter: contig"));
    GS_TRY(compute_segments(rs));
    const uint64_t n = rs->n;
    const uint32_t n_segs = rs->n_segs;
    if (rs->keep) (void)hipFree(rs->keep);
    rs->keep = nullptr;
    HIP_TRY(hipMalloc((void **)&rs->keep, std::max<uint64_t>(n, 1)));
    rs->has_keep = true;
    if (n_kept) *n_kept = 0;
    if (!n) return GS_OK;
    std::vector<uint32_t> chash(n_contigs);
    for (uint32_t c = 0; c < n_contigs; c++) chash[c] = murmur3_32(contig_names[c], strlen(contig_names[c]));
    Dev db;
    uint32_t *d_chash;
    uint8_t *d_infai;
    GS_TRY(db.put(&d_chash, chash.data(), n_contigs));
    GS_TRY(db.put(&d_infai, in_fai, n_contigs));
    // rows by (read, contig), original order inside
    uint64_t *d_key;
    uint32_t *d_rows, *d_head, *d_pidx;
    GS_TRY(db.get(&d_key, n));
    GS_TRY(db.get(&d_rows, n));
    hipLaunchKernelGGL(k_seg_keys, dim3(blocks_of(n)), dim3(256), 0, 0, rs->seg, rs->contig, n, d_key, d_rows);
    GS_TRY(sort_by(d_key, d_rows, n, 64, db));
    GS_TRY(db.get(&d_head, n));
    GS_TRY(db.get(&d_pidx, n));
    hipLaunchKernelGGL(k_head_u64, dim3(blocks_of(n)), dim3(256), 0, 0, d_key, n, d_head);
    GS_TRY(inclusive_scan(d_head, d_pidx, n, rocprim::plus<uint32_t>()));
    uint32_t n_pairs = 0;
    HIP_TRY(hipMemcpy(&n_pairs, d_pidx + n - 1, sizeof(uint32_t), hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_dec, dim3(blocks_of(n)), dim3(256), 0, 0, d_pidx, n);
    uint64_t *d_pair_off, *d_pair_key;
    uint32_t *d_seg_pair_off, *d_need, *d_prefmax, *d_slot, *d_grow, *d_grow_n;
    GS_TRY(db.get(&d_pair_off, (uint64_t)n_pairs + 1));
    GS_TRY(db.get(&d_pair_key, (uint64_t)n_pairs + 1));
    hipLaunchKernelGGL(k_pair_fill, dim3(blocks_of(n)), dim3(256), 0, 0, d_key, d_head, d_pidx, n, d_pair_off, d_pair_key);
    HIP_TRY(hipMemcpy(d_pair_off + n_pairs, &n, sizeof(uint64_t), hipMemcpyHostToDevice));
    db.release(d_head);
    db.release(d_pidx);
    db.release(d_key);
    GS_TRY(db.get(&d_seg_pair_off, (uint64_t)n_segs + 1));
    hipLaunchKernelGGL(k_seg_pair_off, dim3(blocks_of(n_pairs)), dim3(256), 0, 0, d_pair_key, n_pairs, d_seg_pair_off);
    HIP_TRY(hipMemcpy(d_seg_pair_off + n_segs, &n_pairs, sizeof(uint32_t), hipMemcpyHostToDevice));
    // the Table's capacity before every read: a running maximum of what the reads so far needed
    GS_TRY(db.get(&d_need, n_segs));
    GS_TRY(db.get(&d_prefmax, n_segs));
    hipLaunchKernelGGL(k_seg_need, dim3(blocks_of(n_segs)), dim3(256), 0, 0, d_seg_pair_off, n_segs, d_need);
    GS_TRY(inclusive_scan(d_need, d_prefmax, n_segs, rocprim::maximum<uint32_t>()));
    GS_TRY(db.get(&d_slot, n_pairs));
    const uint32_t grow_cap = 64;  // (every growth at least doubles the capacity: there cannot be more than 26)
    GS_TRY(db.get(&d_grow, grow_cap));
    GS_TRY(db.get(&d_grow_n, 1));
    HIP_TRY(hipMemset(d_grow_n, 0, sizeof(uint32_t)));
    hipLaunchKernelGGL(k_slots, dim3(blocks_of(n_segs, 128)), dim3(128), 0, 0, n_segs, d_seg_pair_off, d_need, d_prefmax, d_pair_off,
                       d_pair_key, d_rows, d_chash, d_slot, d_grow, d_grow_n, grow_cap, cap0);
    if (cap_out) {  // the capacity the Table leaves this row set with
        uint32_t last = 0;
        HIP_TRY(hipMemcpy(&last, d_prefmax + n_segs - 1, sizeof(uint32_t), hipMemcpyDeviceToHost));
        *cap_out = std::max(cap0, last);
    }
    uint32_t n_grow = 0;
    HIP_TRY(hipMemcpy(&n_grow, d_grow_n, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (n_grow > grow_cap) {
        set_error("gs_rowset_diag_filter: internal: %u growing tables", n_grow);
        return GS_E_ARG;
    }
    if (n_grow) {
        std::vector<uint32_t> gl(n_grow), spo((size_t)n_segs + 1), need(n_segs), pm(n_segs);
        HIP_TRY(hipMemcpy(gl.data(), d_grow, n_grow * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(need.data(), d_need, (size_t)n_segs * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(pm.data(), d_prefmax, (size_t)n_segs * 4, hipMemcpyDeviceToHost));
        std::vector<uint64_t> so(n_grow + 1, 0);
        for (uint32_t g = 0; g < n_grow; g++) {
            const uint32_t s = gl[g];
            const uint32_t capf = std::max(need[s], s ? std::max(cap0, pm[s - 1]) : cap0);
            so[g + 1] = so[g] + 4ULL * capf;
        }
        uint64_t *d_so;
        uint32_t *d_scr;
        GS_TRY(db.put(&d_so, so.data(), so.size()));
        GS_TRY(db.get(&d_scr, so[n_grow]));
        hipLaunchKernelGGL(k_slots_grow, dim3(blocks_of(n_grow, 64)), dim3(64), 0, 0, n_grow, d_grow, d_so, d_scr, d_seg_pair_off, d_prefmax,
                           d_pair_off, d_pair_key, d_rows, d_chash, d_slot, cap0);
    }
    // statistics per (read, contig), selection per read, verdict per row
    int64_t *d_sa, *d_sb;
    uint32_t *d_gm, *d_gp, *d_bc, *d_bg;
    uint8_t *d_bd;
    GS_TRY(db.get(&d_sa, n));
    GS_TRY(db.get(&d_sb, n));
    GS_TRY(db.get(&d_gm, n_pairs));
    GS_TRY(db.get(&d_gp, n_pairs));
    GS_TRY(db.get(&d_bc, n_segs));
    GS_TRY(db.get(&d_bg, n_segs));
    GS_TRY(db.get(&d_bd, n_segs));
    hipLaunchKernelGGL(k_diag_stats_rows, dim3(blocks_of(n_pairs, 128)), dim3(128), 0, 0, n_pairs, d_pair_off, d_pair_key, d_rows, rs->pos,
                       rs->start, rs->id, d_infai, d_sa, d_sb, d_gm, d_gp, (int64_t)2500);
    hipLaunchKernelGGL(k_diag_select_slots, dim3(blocks_of(n_segs, 128)), dim3(128), 0, 0, n_segs, d_seg_pair_off, d_pair_off, d_pair_key,
                       d_slot, d_infai, d_gm, d_gp, d_bc, d_bg, d_bd);
    hipLaunchKernelGGL(k_keep_rows, dim3(blocks_of(n)), dim3(256), 0, 0, n, rs->seg, rs->contig, d_bc, rs->keep);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (best_contig) HIP_TRY(hipMemcpy(best_contig, d_bc, (size_t)n_segs * 4, hipMemcpyDeviceToHost));
    if (best_good) HIP_TRY(hipMemcpy(best_good, d_bg, (size_t)n_segs * 4, hipMemcpyDeviceToHost));
    if (best_dir) HIP_TRY(hipMemcpy(best_dir, d_bd, n_segs, hipMemcpyDeviceToHost));
    if (seg_name) {
        uint32_t *d_flag, *d_sn;
        GS_TRY(db.get(&d_flag, n));
        GS_TRY(db.get(&d_sn, n_segs));
        hipLaunchKernelGGL(k_head_u32, dim3(blocks_of(n)), dim3(256), 0, 0, rs->name, n, d_flag);
        hipLaunchKernelGGL(k_seg_first, dim3(blocks_of(n)), dim3(256), 0, 0, d_flag, rs->seg, rs->name, n, d_sn);
        HIP_TRY(hipMemcpy(seg_name, d_sn, (size_t)n_segs * 4, hipMemcpyDeviceToHost));
    }
    if (n_kept) {
        uint32_t *d_sel;
        GS_TRY(db.get(&d_sel, n));
        GS_TRY(select_indices(rs->keep, d_sel, n, n_kept));
    }
    return GS_OK;
}

int gs_rowset_keep_flags(const gs_rowset *rs, uint8_t *keep) {
    if (!rs || !keep || !rs->has_keep) {
        set_error("gs_rowset_keep_flags: no diagonal filter has run on this row set");
        return GS_E_ARG;
    }
    GS_TRY(check_device(rs->device, "gs_rowset_keep_flags"));
    if (rs->n) HIP_TRY(hipMemcpy(keep, rs->keep, rs->n, hipMemcpyDeviceToHost));
    return GS_OK;
}

// diag_filter_step2: the rows whose contig is their read's best contig stay, in order
int gs_rowset_keep_filtered(gs_rowset *rs) {
    if (!rs || !rs->has_keep) {
        set_error("gs_rowset_keep_filtered: no diagonal filter has run on this row set");
        return GS_E_ARG;
    }
    GS_TRY(check_device(rs->device, "gs_rowset_keep_filtered"));
    const uint64_t n = rs->n;
    Dev db;
    uint32_t *d_sel, *d_tmp;
    GS_TRY(db.get(&d_sel, n));
    uint64_t m = 0;
    GS_TRY(select_indices(rs->keep, d_sel, n, &m));
    GS_TRY(db.get(&d_tmp, m));
    for (uint32_t *col : {rs->name, rs->pos, rs->contig, rs->start, rs->id, rs->key}) {
        if (!m) break;
        hipLaunchKernelGGL(k_gather<uint32_t>, dim3(blocks_of(m)), dim3(256), 0, 0, col, d_sel, m, d_tmp);
        HIP_TRY(hipMemcpy(col, d_tmp, m * 4, hipMemcpyDeviceToDevice));
    }
    rs->n = m;
    rowset_drop_derived(rs);
    return GS_OK;
}

// badsunks_AR.py:20-103 over the two haplotypes' (diagonal-filtered) row sets.  Keys = the rows' dense (contig, ID) ids;
// key_hap[key]: bit 0 = the key's contig is in hap1's fai, bit 1 = in hap2's.  bad_host[n_keys] (may be NULL) and *d_bad (may be
// NULL; device array of n_keys flags, gs_device_free) <- the union of the four bad sets; meancov[h] = the script's
// "mean coverage" of haplotype h (mode of the reads per SUNK group on its own contigs, smallest on ties).
int gs_rowset_bad_keys(const gs_rowset *hap1, const gs_rowset *hap2, uint32_t n_keys, const uint8_t *key_hap, uint8_t *bad_host,
                       uint32_t meancov[2], void **d_bad_out) {
    return gs_rowset_bad_keys_reduced(hap1, hap2, n_keys, key_hap, nullptr, nullptr, bad_host, meancov, d_bad_out);
}

int gs_rowset_bad_keys_reduced(const gs_rowset *hap1, const gs_rowset *hap2, uint32_t n_keys, const uint8_t *key_hap, gs_reduce_u32_fn reduce,
                               void *reduce_ctx, uint8_t *bad_host, uint32_t meancov[2], void **d_bad_out) {
    if (!hap1 || !hap2 || !key_hap || !n_keys || !hap1->has_key || !hap2->has_key || hap1->device != hap2->device) {
        set_error("gs_rowset_bad_keys: bad argument (both row sets need keys and one device)");
        return GS_E_ARG;
    }
    GS_TRY(check_device(hap1->device, "gs_rowset_bad_keys"));
    GS_TRY(check_ids_below(hap1->key, hap1->n, n_keys, "gs_rowset_bad_keys: key"));
    GS_TRY(check_ids_below(hap2->key, hap2->n, n_keys, "gs_rowset_bad_keys: key"));
    Dev db;
    uint8_t *d_hap, *d_bad = nullptr;
    uint32_t *d_count, *d_hist;
    const uint32_t hist_len = 1u << 16;
    GS_TRY(db.put(&d_hap, key_hap, n_keys));
    GS_TRY(db.get(&d_count, n_keys));
    GS_TRY(db.get(&d_hist, hist_len));
    HIP_TRY(hipMalloc((void **)&d_bad, n_keys));
    HIP_TRY(hipMemset(d_bad, 0, n_keys));
    std::vector<uint32_t> hist(hist_len);
    int rc = GS_OK;
    for (int h = 0; h < 2 && rc == GS_OK; h++) {
        const gs_rowset *rs = h == 0 ? hap1 : hap2;
        const uint8_t bit = (uint8_t)(1u << h);
        hipError_t e = hipMemset(d_count, 0, (size_t)n_keys * 4);
        if (e == hipSuccess) e = hipMemset(d_hist, 0, hist_len * 4);
        if (e == hipSuccess && rs->n) hipLaunchKernelGGL(k_count_keys, dim3(blocks_of(rs->n)), dim3(256), 0, 0, rs->key, rs->n, d_count);
        if (e == hipSuccess && reduce) {  // reads sharded over ranks: the counts of all of them (one sum per haplotype)
            e = hipDeviceSynchronize();
            if (e == hipSuccess && (rc = reduce(reduce_ctx, d_count, n_keys)) != GS_OK) break;
        }
        if (e == hipSuccess) hipLaunchKernelGGL(k_hist_counts, dim3(blocks_of(n_keys)), dim3(256), 0, 0, d_count, d_hap, bit, n_keys, d_hist, hist_len);
        if (e == hipSuccess) e = hipMemcpy(hist.data(), d_hist, hist_len * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            set_error("gs_rowset_bad_keys: %s", hipGetErrorString(e));
            rc = GS_E_NODEVICE;
            break;
        }
        uint64_t best = 0, mode = 0;
        for (uint32_t c = 1; c < hist_len - 1; c++)
            if (hist[c] > best) {
                best = hist[c];
                mode = c;
            }
        if (hist[hist_len - 1] && hist[hist_len - 1] >= best) {  // absurdly deep data: exact mode on the host
            std::vector<uint32_t> cnt(n_keys);
            std::vector<uint32_t> big;
            if (hipMemcpy(cnt.data(), d_count, (size_t)n_keys * 4, hipMemcpyDeviceToHost) != hipSuccess) {
                set_error("gs_rowset_bad_keys: copy failed");
                rc = GS_E_NODEVICE;
                break;
            }
            for (uint32_t i = 0; i < n_keys; i++)
                if (cnt[i] >= hist_len - 1 && (key_hap[i] & bit)) big.push_back(cnt[i]);
            std::sort(big.begin(), big.end());
            for (size_t i = 0; i < big.size();) {
                size_t j = i;
                while (j < big.size() && big[j] == big[i]) j++;
                if (j - i > best) {
                    best = j - i;
                    mode = big[i];
                }
                i = j;
            }
        }
        if (!best) {  // (the script dies on mode().values[0])
            set_error("gs_rowset_bad_keys: no SUNK of haplotype %d's own contigs among its rows", h + 1);
            rc = GS_E_ARG;
            break;
        }
        if (meancov) meancov[h] = (uint32_t)mode;
        const double limit = (double)mode + std::sqrt((double)mode) * 4.0;  // badsunks_AR.py:46
        hipLaunchKernelGGL(k_mark_bad, dim3(blocks_of(n_keys)), dim3(256), 0, 0, d_count, d_hap, bit, n_keys, limit, d_bad);
    }
    if (rc == GS_OK && hipDeviceSynchronize() != hipSuccess) {
        set_error("gs_rowset_bad_keys: kernel failed");
        rc = GS_E_NODEVICE;
    }
    if (rc == GS_OK && bad_host && hipMemcpy(bad_host, d_bad, n_keys, hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("gs_rowset_bad_keys: copy failed");
        rc = GS_E_NODEVICE;
    }
    if (rc != GS_OK || !d_bad_out) (void)hipFree(d_bad);
    else *d_bad_out = d_bad;
    return rc;
}
int gs_device_free(int device, void *p) {
    GS_TRY(check_device(device, "gs_device_free"));
    if (p) HIP_TRY(hipFree(p));
    return GS_OK;
}

// process-by-contig_lowmem_AR.py:50-260 over a row set (rows of several contigs are handled contig by contig, as the workflow's
// per-contig files are):
//   d_key_bad  device flags per key (gs_rowset_bad_keys), or NULL;  row_bad  host flags per row (`chrom:ID` in bad_sunks), or NULL
//   name_rank[n_names]  position of every name id in sorted-name order (`groupby("rname")`, :135)
//   name_long[n_names]  1 = the read is at least 10 kb long per the .rlen file (:106-108)
int gs_rowset_pair_stage(const gs_rowset *rs, const void *d_key_bad, const uint8_t *row_bad, uint32_t n_names, const uint32_t *name_rank,
                         const uint8_t *name_long, gs_pairstage **out) {
    if (!rs || !out || (n_names && (!name_rank || !name_long)) || (d_key_bad && !rs->has_key)) {
        set_error("gs_rowset_pair_stage: bad argument");
        return GS_E_ARG;
    }
    GS_TRY(check_device(rs->device, "gs_rowset_pair_stage"));
    GS_TRY(check_ids_below(rs->name, rs->n, n_names, "gs_rowset_pair_stage: name"));
    gs_pairstage *ps = new gs_pairstage;
    struct Guard {
        gs_pairstage *p;
        ~Guard() { delete p; }
    } guard{ps};
    const uint64_t n = rs->n;
    ps->counters[0] = n;
    Dev db;
    if (!n) {
        *out = ps;
        guard.p = nullptr;
        return GS_OK;
    }
    uint32_t *d_rank;
    uint8_t *d_long, *d_rowbad = nullptr;
    GS_TRY(db.put(&d_rank, name_rank, n_names));
    GS_TRY(db.put(&d_long, name_long, n_names));
    if (row_bad) GS_TRY(db.put(&d_rowbad, row_bad, n));
    // distinct contigs of the input (the one-line output of a contig without result names it, :92-94,202-204)
    {
        uint64_t *d_ck, *d_cs, *d_cu;
        uint8_t *d_f8;
        uint32_t *d_csel;
        GS_TRY(db.get(&d_ck, n));
        GS_TRY(db.get(&d_cs, n));
        GS_TRY(db.get(&d_f8, n));
        GS_TRY(db.get(&d_csel, n));
        hipLaunchKernelGGL(k_widen, dim3(blocks_of(n)), dim3(256), 0, 0, rs->contig, d_ck, n);
        GS_TRY(sort_keys_u64(d_ck, d_cs, n, nullptr));
        hipLaunchKernelGGL(k_head_u64_b, dim3(blocks_of(n)), dim3(256), 0, 0, d_cs, n, d_f8);
        uint64_t nc = 0;
        GS_TRY(select_indices(d_f8, d_csel, n, &nc));
        GS_TRY(db.get(&d_cu, nc));
        hipLaunchKernelGGL(k_gather<uint64_t>, dim3(blocks_of(nc)), dim3(256), 0, 0, d_cs, d_csel, nc, d_cu);
        std::vector<uint64_t> cu(nc);
        HIP_TRY(hipMemcpy(cu.data(), d_cu, nc * 8, hipMemcpyDeviceToHost));
        for (uint64_t c : cu) ps->contigs_in.push_back((uint32_t)c);
        db.release(d_ck);
        db.release(d_cs);
        db.release(d_f8);
        db.release(d_csel);
    }
    // 1. rows whose chrom:ID is not a bad SUNK (:66-72)
    uint8_t *d_flag;
    uint32_t *d_idx;
    GS_TRY(db.get(&d_flag, n));
    GS_TRY(db.get(&d_idx, n));
    hipLaunchKernelGGL(k_not_bad, dim3(blocks_of(n)), dim3(256), 0, 0, rs->key, static_cast<const uint8_t *>(d_key_bad), d_rowbad, n, d_flag);
    uint64_t n0 = 0;
    GS_TRY(select_indices(d_flag, d_idx, n, &n0));
    ps->counters[1] = n0;
    if (!n0) {
        *out = ps;
        guard.p = nullptr;
        return GS_OK;
    }
    // 2. reads (per contig) with at least two distinct IDs (:91-97): sort by (contig, name, ID), count the ID heads per group
    uint64_t *d_gkey, *d_k2;
    uint32_t *d_head, *d_gidx, *d_cnt;
    GS_TRY(db.get(&d_gkey, n0));
    GS_TRY(db.get(&d_k2, n0));
    hipLaunchKernelGGL(k_col_key, dim3(blocks_of(n0)), dim3(256), 0, 0, d_idx, n0, rs->id, d_k2);
    GS_TRY(sort_by(d_k2, d_idx, n0, 32, db));
    hipLaunchKernelGGL(k_group_key, dim3(blocks_of(n0)), dim3(256), 0, 0, d_idx, n0, rs->contig, rs->name, d_rank, d_gkey);
    GS_TRY(sort_by(d_gkey, d_idx, n0, 64, db));
    GS_TRY(db.get(&d_head, n0));
    GS_TRY(db.get(&d_gidx, n0));
    GS_TRY(db.get(&d_cnt, n0));
    hipLaunchKernelGGL(k_head_u64, dim3(blocks_of(n0)), dim3(256), 0, 0, d_gkey, n0, d_head);
    GS_TRY(inclusive_scan(d_head, d_gidx, n0, rocprim::plus<uint32_t>()));
    hipLaunchKernelGGL(k_dec, dim3(blocks_of(n0)), dim3(256), 0, 0, d_gidx, n0);
    HIP_TRY(hipMemset(d_cnt, 0, n0 * 4));
    hipLaunchKernelGGL(k_count_ids, dim3(blocks_of(n0)), dim3(256), 0, 0, d_gkey, d_idx, rs->id, d_gidx, n0, d_cnt);
    HIP_TRY(hipMemset(d_flag, 0, n));
    hipLaunchKernelGGL(k_flag_multi, dim3(blocks_of(n0)), dim3(256), 0, 0, d_gidx, d_cnt, n0, d_idx, d_flag);
    uint64_t n1 = 0;
    GS_TRY(select_indices(d_flag, d_idx, n, &n1));  // (file order again)
    ps->counters[2] = n1;
    if (!n1) {
        *out = ps;
        guard.p = nullptr;
        return GS_OK;
    }
    // 3. sort by (contig, name, start), stable (:100); exact duplicates (:104); reads of at least 10 kb (:106-108)
    hipLaunchKernelGGL(k_col_key, dim3(blocks_of(n1)), dim3(256), 0, 0, d_idx, n1, rs->start, d_k2);
    GS_TRY(sort_by(d_k2, d_idx, n1, 32, db));
    hipLaunchKernelGGL(k_group_key, dim3(blocks_of(n1)), dim3(256), 0, 0, d_idx, n1, rs->contig, rs->name, d_rank, d_gkey);
    GS_TRY(sort_by(d_gkey, d_idx, n1, 64, db));
    uint8_t *d_fd, *d_fo;
    uint32_t *d_sel;
    GS_TRY(db.get(&d_fd, n1));
    GS_TRY(db.get(&d_fo, n1));
    GS_TRY(db.get(&d_sel, n1));
    hipLaunchKernelGGL(k_dedup_long, dim3(blocks_of(n1)), dim3(256), 0, 0, d_gkey, d_idx, n1, rs->start, rs->pos, rs->id, rs->name, d_long,
                       d_fd, d_fo);
    uint64_t n2 = 0, n3 = 0;
    GS_TRY(select_indices(d_fd, d_sel, n1, &n2));
    GS_TRY(select_indices(d_fo, d_sel, n1, &n3));
    ps->counters[3] = n2;
    ps->counters[4] = n3;
    if (!n3) {
        *out = ps;
        guard.p = nullptr;
        return GS_OK;
    }
    // 4. the read groups and the pair stage proper (:135-198)
    uint64_t *d_fkey, *d_off;
    int64_t *d_fpos, *d_fstart, *d_fid, *d_outids;
    uint32_t *d_frow, *d_outn;
    GS_TRY(db.get(&d_fkey, n3));
    GS_TRY(db.get(&d_fpos, n3));
    GS_TRY(db.get(&d_fstart, n3));
    GS_TRY(db.get(&d_fid, n3));
    GS_TRY(db.get(&d_frow, n3));
    hipLaunchKernelGGL(k_final_rows, dim3(blocks_of(n3)), dim3(256), 0, 0, d_sel, n3, d_gkey, d_idx, rs->pos, rs->start, rs->id, d_fkey,
                       d_fpos, d_fstart, d_fid, d_frow);
    hipLaunchKernelGGL(k_head_u64, dim3(blocks_of(n3)), dim3(256), 0, 0, d_fkey, n3, d_head);
    GS_TRY(inclusive_scan(d_head, d_gidx, n3, rocprim::plus<uint32_t>()));
    uint32_t n_groups = 0;
    HIP_TRY(hipMemcpy(&n_groups, d_gidx + n3 - 1, 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_dec, dim3(blocks_of(n3)), dim3(256), 0, 0, d_gidx, n3);
    ps->counters[5] = n_groups;
    GS_TRY(db.get(&d_off, (uint64_t)n_groups + 1));
    hipLaunchKernelGGL(k_group_off, dim3(blocks_of(n3)), dim3(256), 0, 0, d_head, d_gidx, n3, d_off);
    HIP_TRY(hipMemcpy(d_off + n_groups, &n3, 8, hipMemcpyHostToDevice));
    GS_TRY(db.get(&d_outids, n3));
    GS_TRY(db.get(&d_outn, (uint64_t)n_groups + 1));
    HIP_TRY(hipMemset(d_outn, 0, ((size_t)n_groups + 1) * 4));
    HIP_TRY(hipDeviceSynchronize());
    GS_TRY(pair_component_device(n_groups, n3, d_off, d_fpos, d_fstart, d_fid, d_outn, d_outids));
    // 5. output rows
    uint64_t *d_outn64, *d_outoff;
    GS_TRY(db.get(&d_outn64, (uint64_t)n_groups + 1));
    GS_TRY(db.get(&d_outoff, (uint64_t)n_groups + 1));
    hipLaunchKernelGGL(k_widen, dim3(blocks_of((uint64_t)n_groups + 1)), dim3(256), 0, 0, d_outn, d_outn64, (uint64_t)n_groups + 1);
    GS_TRY(exclusive_sum(d_outn64, d_outoff, (uint64_t)n_groups + 1));
    uint64_t T = 0;
    HIP_TRY(hipMemcpy(&T, d_outoff + n_groups, 8, hipMemcpyDeviceToHost));
    if (!T) {
        *out = ps;
        guard.p = nullptr;
        return GS_OK;
    }
    uint32_t *d_oc, *d_on, *d_og;
    int64_t *d_oi;
    GS_TRY(db.get(&d_oc, T));
    GS_TRY(db.get(&d_on, T));
    GS_TRY(db.get(&d_og, T));
    GS_TRY(db.get(&d_oi, T));
    hipLaunchKernelGGL(k_out_rows, dim3(n_groups), dim3(64), 0, 0, n_groups, d_off, d_outn, d_outoff, d_outids, d_frow, rs->contig, rs->name,
                       d_oc, d_on, d_oi, d_og);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    ps->row_contig.resize(T);
    ps->row_name.resize(T);
    ps->row_id.resize(T);
    HIP_TRY(hipMemcpy(ps->row_contig.data(), d_oc, T * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ps->row_name.data(), d_on, T * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ps->row_id.data(), d_oi, T * 8, hipMemcpyDeviceToHost));
    // 6. contig-wide union of the reads' ID sets (:219-260)
    GS_TRY(union_of_rows(db, T, d_oc, d_oi, d_og, ps));
    ps->counters[6] = T;
    *out = ps;
    guard.p = nullptr;
    return GS_OK;
}

// The last step of a read-sharded run on its root: the TSV rows of all ranks (sorted by contig, then by the reads' name order;
// a read's rows adjacent and in the order its rank wrote them) -> the same result object a one-GPU run gives: rows as handed in,
// the validated intervals from the contig-wide union over ALL of them.
int gs_pairstage_from_rows(int device, uint64_t n_rows, const uint32_t *row_contig, const uint32_t *row_name, const int64_t *row_id,
                           uint32_t n_contigs_in, const uint32_t *contigs_in, gs_pairstage **out) {
    if (!out || (n_rows && (!row_contig || !row_name || !row_id)) || (n_contigs_in && !contigs_in)) {
        set_error("gs_pairstage_from_rows: null argument");
        return GS_E_ARG;
    }
    GS_TRY(check_device(device, "gs_pairstage_from_rows"));
    gs_pairstage *ps = new gs_pairstage;
    struct Guard {
        gs_pairstage *p;
        ~Guard() { delete p; }
    } guard{ps};
    ps->contigs_in.assign(contigs_in, contigs_in + n_contigs_in);
    ps->row_contig.assign(row_contig, row_contig + n_rows);
    ps->row_name.assign(row_name, row_name + n_rows);
    ps->row_id.assign(row_id, row_id + n_rows);
    ps->counters[6] = n_rows;
    if (n_rows) {
        Dev db;
        uint32_t *d_oc, *d_on, *d_head, *d_og;
        int64_t *d_oi;
        uint64_t *d_gk;
        GS_TRY(db.put(&d_oc, row_contig, n_rows));
        GS_TRY(db.put(&d_on, row_name, n_rows));
        GS_TRY(db.put(&d_oi, row_id, n_rows));
        GS_TRY(db.get(&d_gk, n_rows));
        GS_TRY(db.get(&d_head, n_rows));
        GS_TRY(db.get(&d_og, n_rows));
        hipLaunchKernelGGL(k_pack_u32x2, dim3(blocks_of(n_rows)), dim3(256), 0, 0, d_oc, d_on, n_rows, d_gk);
        hipLaunchKernelGGL(k_head_u64, dim3(blocks_of(n_rows)), dim3(256), 0, 0, d_gk, n_rows, d_head);
        GS_TRY(inclusive_scan(d_head, d_og, n_rows, rocprim::plus<uint32_t>()));  // (group numbers from 1: only equality matters)
        GS_TRY(union_of_rows(db, n_rows, d_oc, d_oi, d_og, ps));
    }
    *out = ps;
    guard.p = nullptr;
    return GS_OK;
}

void gs_pairstage_free(gs_pairstage *ps) { delete ps; }
uint64_t gs_pairstage_n_rows(const gs_pairstage *ps) { return ps ? ps->row_id.size() : 0; }
const uint32_t *gs_pairstage_row_contig(const gs_pairstage *ps) { return ps->row_contig.data(); }
const uint32_t *gs_pairstage_row_name(const gs_pairstage *ps) { return ps->row_name.data(); }
const int64_t *gs_pairstage_row_id(const gs_pairstage *ps) { return ps->row_id.data(); }
uint32_t gs_pairstage_n_regions(const gs_pairstage *ps) { return ps ? (uint32_t)ps->region_contig.size() : 0; }
const uint32_t *gs_pairstage_region_contig(const gs_pairstage *ps) { return ps->region_contig.data(); }
const int64_t *gs_pairstage_region_start(const gs_pairstage *ps) { return ps->region_start.data(); }
const int64_t *gs_pairstage_region_end(const gs_pairstage *ps) { return ps->region_end.data(); }
uint32_t gs_pairstage_n_contigs_in(const gs_pairstage *ps) { return ps ? (uint32_t)ps->contigs_in.size() : 0; }
const uint32_t *gs_pairstage_contigs_in(const gs_pairstage *ps) { return ps->contigs_in.data(); }
const uint64_t *gs_pairstage_counters(const gs_pairstage *ps) { return ps->counters; }

}  // extern "C"

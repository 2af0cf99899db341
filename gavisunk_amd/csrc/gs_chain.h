// Shared by the translation units of the device-resident second half (gs_chain.hip: rows, step 2, bad SUNKs, pair stage, union;
// gs_diag.hip: the diagonal filter): the handles behind include/gavisunk.h's gs_locrows / gs_rowset / gs_pairstage, and the
// scan / select / sort helpers, which take their temporary storage from the device pool and run on the chain stream.
#pragma once
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/functional.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>
#include <cstring>
#include <vector>

#include "gs_common.h"
#include "gs_devpool.h"

// kmer.loc rows on the device: what turns a hit's payload (loc row) into contig / start / ID / dense (contig, ID) key.  One 16-byte
// record per row {start, contig, ID, key}: a row of a hit list is ONE random 16-byte load (three arrays were three requests per row,
// at the same request wall the tagger's probe kernel sits on)
struct gs_locrows {
    int device = 0;
    uint64_t n_rows = 0;
    uint32_t n_locs = 0;
    uint4 *rec = nullptr;  // x = start, y = contig, z = ID (group), w = dense id of (contig, ID)
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
    uint32_t *seg = nullptr;      // per row: its segment
    uint32_t *seg_off = nullptr;  // per segment: its first row (n_segs + 1 entries)
    uint8_t *keep = nullptr;      // per row
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

namespace gs {
namespace chain {

#define GSC_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            gs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GS_E_NODEVICE;                                                             \
        }                                                                                     \
    } while (0)
#define GSC_TRY(expr)             \
    do {                          \
        int _rc = (expr);         \
        if (_rc != GS_OK) return _rc; \
    } while (0)

constexpr uint32_t kNone = 0xFFFFFFFFu;

inline unsigned blocks_of(uint64_t n, unsigned per = 256) { return (unsigned)std::max<uint64_t>((n + per - 1) / per, 1); }

// one call of the second half: the device, its stream, its temporaries
struct Ctx {
    int device;
    hipStream_t st;
    PoolScope mem;
    explicit Ctx(int dev) : device(dev), st(chain_stream(dev)), mem(dev) {}
    int sync() {
        GSC_HIP(hipStreamSynchronize(st));
        return GS_OK;
    }
    // a value kernels of this stream wrote, on the host
    template <typename T>
    int fetch(T *host, const T *dev, uint64_t n = 1) {
        GSC_HIP(hipMemcpyAsync(host, dev, n * sizeof(T), hipMemcpyDeviceToHost, st));
        GSC_HIP(hipStreamSynchronize(st));
        return GS_OK;
    }
    template <typename T>
    int zero(T *dev, uint64_t n) {
        if (n) GSC_HIP(hipMemsetAsync(dev, 0, n * sizeof(T), st));
        return GS_OK;
    }
};

int check_device(int device, const char *who);  // gs_chain.hip

template <typename T, typename Op>
int inclusive_scan(Ctx &c, const T *in, T *out, uint64_t n, Op op) {
    if (!n) return GS_OK;
    size_t tb = 0;
    GSC_HIP(rocprim::inclusive_scan(nullptr, tb, in, out, (size_t)n, op, c.st));
    uint8_t *tmp;
    GSC_TRY(c.mem.get(&tmp, std::max<size_t>(tb, 16)));
    GSC_HIP(rocprim::inclusive_scan(tmp, tb, in, out, (size_t)n, op, c.st));
    c.mem.release(tmp);
    return GS_OK;
}
template <typename T>
int exclusive_sum(Ctx &c, const T *in, T *out, uint64_t n) {
    if (!n) return GS_OK;
    size_t tb = 0;
    GSC_HIP(rocprim::exclusive_scan(nullptr, tb, in, out, (T)0, (size_t)n, rocprim::plus<T>(), c.st));
    uint8_t *tmp;
    GSC_TRY(c.mem.get(&tmp, std::max<size_t>(tb, 16)));
    GSC_HIP(rocprim::exclusive_scan(tmp, tb, in, out, (T)0, (size_t)n, rocprim::plus<T>(), c.st));
    c.mem.release(tmp);
    return GS_OK;
}
// indices i in [0, n) with flag[i] != 0, ascending -> out; *n_out = how many (waits for the stream)
inline int select_indices(Ctx &c, const uint8_t *flag, uint32_t *out, uint64_t n, uint64_t *n_out) {
    *n_out = 0;
    if (!n) return GS_OK;
    unsigned long long *d_count;
    GSC_TRY(c.mem.get(&d_count, 1));
    rocprim::counting_iterator<uint32_t> iota(0u);
    size_t tb = 0;
    GSC_HIP(rocprim::select(nullptr, tb, iota, flag, out, d_count, (size_t)n, c.st));
    uint8_t *tmp;
    GSC_TRY(c.mem.get(&tmp, std::max<size_t>(tb, 16)));
    GSC_HIP(rocprim::select(tmp, tb, iota, flag, out, d_count, (size_t)n, c.st));
    unsigned long long cnt = 0;
    GSC_TRY(c.fetch(&cnt, d_count));
    c.mem.release(tmp);
    c.mem.release(d_count);
    *n_out = cnt;
    return GS_OK;
}
// stable radix sort of (key, value) pairs by the key bits [0, bits): both arrays are replaced by their sorted versions
// (the second buffers come from, and go back to, the pool; `*key` / `*val` may point to them afterwards -- they stay in c.mem)
template <typename K, typename V>
int sort_pairs(Ctx &c, K **key, V **val, uint64_t n, int bits) {
    if (n < 2 || bits <= 0) return GS_OK;
    K *k2;
    V *v2;
    GSC_TRY(c.mem.get(&k2, n));
    GSC_TRY(c.mem.get(&v2, n));
    rocprim::double_buffer<K> kb(*key, k2);
    rocprim::double_buffer<V> vb(*val, v2);
    size_t tb = 0;
    GSC_HIP(rocprim::radix_sort_pairs(nullptr, tb, kb, vb, (size_t)n, 0u, (unsigned)bits, c.st));
    uint8_t *tmp;
    GSC_TRY(c.mem.get(&tmp, std::max<size_t>(tb, 16)));
    GSC_HIP(rocprim::radix_sort_pairs(tmp, tb, kb, vb, (size_t)n, 0u, (unsigned)bits, c.st));
    c.mem.release(tmp);
    K *kc = kb.current(), *ka = kb.alternate();
    V *vc = vb.current(), *va = vb.alternate();
    *key = kc;
    *val = vc;
    c.mem.release(ka);
    c.mem.release(va);
    return GS_OK;
}
template <typename K>
int sort_keys(Ctx &c, K **key, uint64_t n, int bits) {
    if (n < 2 || bits <= 0) return GS_OK;
    K *k2;
    GSC_TRY(c.mem.get(&k2, n));
    rocprim::double_buffer<K> kb(*key, k2);
    size_t tb = 0;
    GSC_HIP(rocprim::radix_sort_keys(nullptr, tb, kb, (size_t)n, 0u, (unsigned)bits, c.st));
    uint8_t *tmp;
    GSC_TRY(c.mem.get(&tmp, std::max<size_t>(tb, 16)));
    GSC_HIP(rocprim::radix_sort_keys(tmp, tb, kb, (size_t)n, 0u, (unsigned)bits, c.st));
    c.mem.release(tmp);
    K *kc = kb.current(), *ka = kb.alternate();
    *key = kc;
    c.mem.release(ka);
    return GS_OK;
}
inline int bits_for(uint64_t max_value) {  // key bits a radix sort has to look at
    int b = 1;
    while (b < 64 && (max_value >> b)) b++;
    return b;
}

int compute_segments(Ctx &c, gs_rowset *rs);  // gs_chain.hip: rs->seg, rs->seg_off, rs->n_segs
void rowset_drop_derived(gs_rowset *rs);

}  // namespace chain
}  // namespace gs

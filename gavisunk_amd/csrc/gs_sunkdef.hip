// SUNK definition on the GPU: what GAVISUNK obtains from
// `jellyfish count -C -U 1` + `mrsfast -e 0` + `bedtools merge/intersect` (workflow/rules/defineSUNKs.smk:22-127)
// -- the canonical k-mers that occur exactly once over hap1+hap2, their positions, and the merged
// overlapping/book-ended [start, start+k) intervals (group id = start of the merged interval).
//
//   k_window_codes   canonical code of every window (windows with a non-ACGT byte are not k-mers)
//   radix sort       (code, contig<<32|pos) by code           (rocPRIM: plain device-wide primitive)
//   k_mark_unique    code differs from both neighbours
//   select           -> the singly-unique ones
//   radix sort       by (contig, pos)  = kmer.loc order
//   k_group_heads / inclusive max-scan / k_group_ids   group = start of the run with gaps <= k
#include <hip/hip_runtime.h>

#include <cstring>
#include <algorithm>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/functional.hpp>

#include "gs_common.h"

using namespace gs;

struct gs_sunkset {
    int k = 0;
    std::vector<uint64_t> code;
    std::vector<uint32_t> contig;
    std::vector<int64_t> start, group;
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

__device__ __forceinline__ int base_code(uint8_t c) {  // -1: not a nucleotide jellyfish counts
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

// one thread per 64 consecutive windows of a contig
constexpr int kChunk = 64;
__global__ void k_window_codes(const uint8_t *seq, const uint64_t *coff, const uint64_t *chunk_off, uint32_t n_contigs,
                               int k, uint64_t invalid_key, uint64_t *keys, uint64_t *vals) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= chunk_off[n_contigs]) return;
    // contig of this chunk: binary search over the chunk prefix
    uint32_t lo = 0, hi = n_contigs;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (chunk_off[mid] <= t) lo = mid;
        else hi = mid;
    }
    const uint32_t c = lo;
    const uint64_t len = coff[c + 1] - coff[c];
    const uint64_t nwin = len >= (uint64_t)k ? len - k + 1 : 0;
    const uint64_t w0 = (t - chunk_off[c]) * kChunk;
    const uint8_t *s = seq + coff[c];
    const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
    const int rshift = 2 * (k - 1);
    uint64_t fwd = 0, rev = 0;
    int good = 0;  // consecutive valid bases ending here
    const uint64_t wend = min(w0 + (uint64_t)kChunk, nwin);
    for (uint64_t e = w0; e < wend + k - 1; e++) {
        const int b = base_code(s[e]);
        if (b < 0) {
            good = 0;
            fwd = rev = 0;
        } else {
            good++;
            fwd = ((fwd << 2) | (uint64_t)b) & kmask;
            rev = (rev >> 2) | ((uint64_t)(3 - b) << rshift);
        }
        if (e + 1 >= w0 + (uint64_t)k) {
            const uint64_t w = e + 1 - k;
            const uint64_t g = coff[c] + w;  // global slot of the window
            keys[g] = good >= k ? (fwd < rev ? fwd : rev) : invalid_key;
            vals[g] = ((uint64_t)c << 32) | w;
        }
    }
}

__global__ void k_fill_invalid(uint64_t *keys, uint64_t *vals, const uint64_t *coff, uint32_t n_contigs, int k,
                               uint64_t invalid_key) {
    // the last k-1 slots of every contig hold no window
    const uint32_t c = blockIdx.x;
    const uint64_t len = coff[c + 1] - coff[c];
    const uint64_t nwin = len >= (uint64_t)k ? len - k + 1 : 0;
    for (uint64_t i = nwin + threadIdx.x; i < len; i += blockDim.x) {
        keys[coff[c] + i] = invalid_key;
        vals[coff[c] + i] = 0;
    }
}

__global__ void k_mark_unique(const uint64_t *keys, uint64_t n, uint64_t invalid_key, uint8_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t x = keys[i];
    const bool u = x != invalid_key && (i == 0 || keys[i - 1] != x) && (i + 1 == n || keys[i + 1] != x);
    flag[i] = u ? 1 : 0;
}

__global__ void k_group_heads(const uint64_t *pos, uint64_t n, int k, uint64_t *head) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool is_new = i == 0 || (pos[i] >> 32) != (pos[i - 1] >> 32) || (pos[i] & 0xFFFFFFFFu) - (pos[i - 1] & 0xFFFFFFFFu) > (uint64_t)k;
    head[i] = is_new ? i + 1 : 0;  // index + 1 of the run's head; max-scan carries it forward
}

__global__ void k_group_ids(const uint64_t *pos, const uint64_t *head, uint64_t n, int64_t *group) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) group[i] = (int64_t)(pos[head[i] - 1] & 0xFFFFFFFFu);
}

struct Bufs {
    std::vector<void *> p;
    ~Bufs() {
        for (void *x : p) (void)hipFree(x);
    }
    template <typename T>
    int get(T **d, uint64_t n) {
        *d = nullptr;
        HIP_TRY(hipMalloc((void **)d, std::max<uint64_t>(n, 1) * sizeof(T)));
        p.push_back(*d);
        return GS_OK;
    }
};

}  // namespace

extern "C" {

int gs_sunks_define(int device, int k, uint32_t n_contigs, const uint8_t *seq, const uint64_t *off, gs_sunkset **out) {
    if (!out || k < 1 || k > 32 || (n_contigs && (!seq || !off))) {
        set_error("gs_sunks_define: bad argument (k must be 1..32)");
        return GS_E_ARG;
    }
    *out = nullptr;
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_sunks_define: HIP device %d not available -- there is no CPU fallback", device);
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    auto *res = new gs_sunkset();
    res->k = k;
    const uint64_t n = n_contigs ? off[n_contigs] : 0;
    if (n == 0) {
        *out = res;
        return GS_OK;
    }
    for (uint32_t c = 0; c < n_contigs; c++)
        if (off[c + 1] - off[c] > 0xFFFFFFFFull) {
            set_error("gs_sunks_define: contig %u is longer than 2^32 bases", c);
            delete res;
            return GS_E_ARG;
        }
    std::vector<uint64_t> chunk_off(n_contigs + 1, 0);
    for (uint32_t c = 0; c < n_contigs; c++) {
        const uint64_t len = off[c + 1] - off[c];
        const uint64_t nwin = len >= (uint64_t)k ? len - k + 1 : 0;
        chunk_off[c + 1] = chunk_off[c] + (nwin + kChunk - 1) / kChunk;
    }
    Bufs b;
    int rc;
    uint8_t *d_seq, *d_flag;
    uint64_t *d_coff, *d_chunk, *d_k0, *d_v0, *d_k1, *d_v1, *d_cnt;
    if ((rc = b.get(&d_seq, n + 64)) || (rc = b.get(&d_coff, (uint64_t)n_contigs + 1)) ||
        (rc = b.get(&d_chunk, (uint64_t)n_contigs + 1)) || (rc = b.get(&d_k0, n)) || (rc = b.get(&d_v0, n)) ||
        (rc = b.get(&d_k1, n)) || (rc = b.get(&d_v1, n)) || (rc = b.get(&d_flag, n)) || (rc = b.get(&d_cnt, 1))) {
        delete res;
        return rc;
    }
    HIP_TRY(hipMemcpy(d_seq, seq, n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_coff, off, ((uint64_t)n_contigs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_chunk, chunk_off.data(), chunk_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    const uint64_t invalid_key = k == 32 ? ~0ULL : (1ULL << (2 * k));  // sorts after every real code (k = 32: poly-T is never canonical)
    const int end_bit = k == 32 ? 64 : 2 * k + 1;
    hipLaunchKernelGGL(k_fill_invalid, dim3(n_contigs), dim3(64), 0, 0, d_k0, d_v0, d_coff, n_contigs, k, invalid_key);
    if (chunk_off[n_contigs])
        hipLaunchKernelGGL(k_window_codes, dim3((unsigned)((chunk_off[n_contigs] + 127) / 128)), dim3(128), 0, 0, d_seq, d_coff,
                           d_chunk, n_contigs, k, invalid_key, d_k0, d_v0);
    HIP_TRY(hipGetLastError());
    // sort by code
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_k0, d_k1, d_v0, d_v1, n, 0u, (unsigned)end_bit));
    size_t tb2 = 0, tb3 = 0, tb4 = 0;
    HIP_TRY(rocprim::select(nullptr, tb2, d_v1, d_flag, d_v0, d_cnt, n));
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb3, d_v0, d_v1, d_k0, d_k1, n, 0u, 64u));
    HIP_TRY(rocprim::inclusive_scan(nullptr, tb4, d_k0, d_k1, n, rocprim::maximum<uint64_t>()));
    tmp_bytes = std::max(std::max(tmp_bytes, tb2), std::max(tb3, tb4));
    void *d_tmp;
    if ((rc = b.get((uint8_t **)&d_tmp, tmp_bytes))) {
        delete res;
        return rc;
    }
    size_t tbb = tmp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(d_tmp, tbb, d_k0, d_k1, d_v0, d_v1, n, 0u, (unsigned)end_bit));
    hipLaunchKernelGGL(k_mark_unique, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_k1, n, invalid_key, d_flag);
    // singly-unique: positions -> v0, codes -> k0
    tbb = tmp_bytes;
    HIP_TRY(rocprim::select(d_tmp, tbb, d_v1, d_flag, d_v0, d_cnt, n));
    tbb = tmp_bytes;
    HIP_TRY(rocprim::select(d_tmp, tbb, d_k1, d_flag, d_k0, d_cnt, n));
    uint64_t m = 0;
    HIP_TRY(hipMemcpy(&m, d_cnt, sizeof m, hipMemcpyDeviceToHost));
    res->code.resize(m);
    res->contig.resize(m);
    res->start.resize(m);
    res->group.resize(m);
    if (m) {
        // kmer.loc order: by (contig, pos)
        tbb = tmp_bytes;
        HIP_TRY(rocprim::radix_sort_pairs(d_tmp, tbb, d_v0, d_v1, d_k0, d_k1, m, 0u, 64u));
        // groups: head index of each run (gap <= k merges: overlapping or book-ended intervals), carried by a max-scan
        hipLaunchKernelGGL(k_group_heads, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, d_v1, m, k, d_k0);
        tbb = tmp_bytes;
        HIP_TRY(rocprim::inclusive_scan(d_tmp, tbb, d_k0, d_v0, m, rocprim::maximum<uint64_t>()));
        int64_t *d_group = reinterpret_cast<int64_t *>(d_k0);
        hipLaunchKernelGGL(k_group_ids, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, d_v1, d_v0, m, d_group);
        HIP_TRY(hipGetLastError());
        std::vector<uint64_t> posv(m);
        HIP_TRY(hipMemcpy(res->code.data(), d_k1, m * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(posv.data(), d_v1, m * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(res->group.data(), d_group, m * sizeof(int64_t), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < m; i++) {
            res->contig[i] = (uint32_t)(posv[i] >> 32);
            res->start[i] = (int64_t)(posv[i] & 0xFFFFFFFFu);
        }
    }
    *out = res;
    return GS_OK;
}

void gs_sunkset_free(gs_sunkset *s) { delete s; }
uint64_t gs_sunkset_n(const gs_sunkset *s) { return s->code.size(); }
const uint64_t *gs_sunkset_code(const gs_sunkset *s) { return s->code.data(); }
const uint32_t *gs_sunkset_contig(const gs_sunkset *s) { return s->contig.data(); }
const int64_t *gs_sunkset_start(const gs_sunkset *s) { return s->start.data(); }
const int64_t *gs_sunkset_group(const gs_sunkset *s) { return s->group.data(); }

}  // extern "C"

// The device-resident second half of the engine: the tagger's hit list (HBM) -> rows -> diagonal filter (gs_diag.hip) -> step 2 ->
// bad-SUNK histogram -> file-level filtering of process-by-contig -> pair test / component -> contig-wide union, all on device
// arrays.  The file drop-ins (diag_filter_v3, badsunks, process-by-contig) are thin users of the same functions over rows parsed
// from text.  Everything runs on one non-blocking stream per device with its temporaries from a caching pool (gs_devpool.h), so
// that the tagger's next pass can run beside it.
//
// Reference semantics (cited where they are implemented):
//   workflow/src/diag_filter_v3.nim:63-229 (gs_diag.hip), diag_filter_step2.nim:45-64
//   workflow/scripts/badsunks_AR.py:20-103
//   workflow/scripts/process-by-contig_lowmem_AR.py:50-260
#include <hip/hip_runtime.h>

#include <rocprim/device/device_reduce.hpp>

#include <cmath>
#include <map>
#include <unordered_map>

#include "gs_chain.h"

using namespace gs;
using namespace gs::chain;

// ------------------------------------------------------------------------------------------
// the pool and the stream
// ------------------------------------------------------------------------------------------
namespace {
constexpr int kMaxDevices = 16;
struct Pool {
    std::mutex mu;
    std::multimap<size_t, void *> idle;            // size -> block
    std::unordered_map<void *, size_t> owned;      // every block the pool ever handed out and still owns
    size_t idle_bytes = 0;
    hipStream_t stream = nullptr;
};
Pool g_pool[kMaxDevices];

size_t pool_limit() {  // idle bytes the pool may hold before it gives blocks back to the driver
    static const size_t v = [] {
        const char *e = getenv("GAVISUNK_POOL_MB");
        return (size_t)(e ? std::max(0L, atol(e)) : 32768L) << 20;
    }();
    return v;
}
size_t pool_round(size_t bytes) {  // to 1/8 of the size's power of two: a slightly different row count finds the same block
    bytes = std::max<size_t>(bytes, 256);
    int lg = 0;
    while ((bytes >> (lg + 1)) != 0) lg++;
    const size_t step = std::max<size_t>((size_t)1 << std::max(lg - 3, 0), 256);
    return (bytes + step - 1) / step * step;
}
void pool_release_idle(Pool &p) {  // caller holds p.mu; the stream is drained first: an idle block may still be in use by queued work
    if (p.idle.empty()) return;
    if (p.stream) (void)hipStreamSynchronize(p.stream);
    for (auto &kv : p.idle) {
        (void)hipFree(kv.second);
        p.owned.erase(kv.second);
    }
    p.idle.clear();
    p.idle_bytes = 0;
}
}  // namespace

hipStream_t gs::chain_stream(int device) {
    Pool &p = g_pool[device & (kMaxDevices - 1)];
    std::lock_guard<std::mutex> g(p.mu);
    if (!p.stream) {
        (void)hipSetDevice(device);
        // the highest priority the device offers: the chain is a string of short kernels that wait for each other (and for the host
        // between them); beside a tagging pass whose probe kernels fill every CU (all 160 KiB of LDS included) they would otherwise
        // queue behind workgroups of a kernel that is bound by the memory system's request rate, not by the CUs it holds
        // (experiment knob: the chain confined to N of the CUs, every (CUs / N)-th one -- see profiles/r05)
        if (const char *e = getenv("GAVISUNK_CHAIN_CUS")) {
            hipDeviceProp_t prop;
            const int want = atoi(e);
            if (want > 0 && hipGetDeviceProperties(&prop, device) == hipSuccess && want < prop.multiProcessorCount) {
                const int ncu = prop.multiProcessorCount, step = std::max(ncu / want, 1);
                std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
                for (int i = 0; i < ncu; i += step) mask[(size_t)i / 32] |= 1u << (i % 32);
                if (hipExtStreamCreateWithCUMask(&p.stream, (uint32_t)mask.size(), mask.data()) == hipSuccess) return p.stream;
                (void)hipGetLastError();
                p.stream = nullptr;
            }
        }
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) greatest = 0;
        if (getenv("GAVISUNK_CHAIN_PRIORITY") && !atoi(getenv("GAVISUNK_CHAIN_PRIORITY"))) greatest = least;  // (A/B knob of the timing experiments)
        if (hipStreamCreateWithPriority(&p.stream, hipStreamNonBlocking, greatest) != hipSuccess) {
            (void)hipGetLastError();
            if (hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking) != hipSuccess) p.stream = nullptr;  // (the NULL stream: correct, not concurrent)
        }
    }
    return p.stream;
}

void *gs::pool_alloc(int device, size_t bytes) {
    Pool &p = g_pool[device & (kMaxDevices - 1)];
    const size_t want = pool_round(bytes);
    std::lock_guard<std::mutex> g(p.mu);
    auto it = p.idle.lower_bound(want);
    if (it != p.idle.end() && it->first <= want + want / 2) {
        void *b = it->second;
        p.idle_bytes -= it->first;
        p.idle.erase(it);
        return b;
    }
    void *b = nullptr;
    hipError_t e = hipMalloc(&b, want);
    if (e != hipSuccess) {  // make room: everything idle goes back to the driver, then once more
        (void)hipGetLastError();
        pool_release_idle(p);
        e = hipMalloc(&b, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("device allocation of %llu bytes failed: %s", (unsigned long long)want, hipGetErrorString(e));
        return nullptr;
    }
    p.owned[b] = want;
    return b;
}

void gs::pool_free(int device, void *b) {
    if (!b) return;
    Pool &p = g_pool[device & (kMaxDevices - 1)];
    std::lock_guard<std::mutex> g(p.mu);
    auto it = p.owned.find(b);
    if (it == p.owned.end()) {  // not ours (hipMalloc'ed elsewhere)
        (void)hipFree(b);
        return;
    }
    p.idle.emplace(it->second, b);
    p.idle_bytes += it->second;
    if (p.idle_bytes > pool_limit()) pool_release_idle(p);
}

void gs::pool_trim(int device) {
    Pool &p = g_pool[device & (kMaxDevices - 1)];
    std::lock_guard<std::mutex> g(p.mu);
    pool_release_idle(p);
}

int gs::chain::check_device(int device, const char *who) {
    if (gs_device_count() <= device || device < 0 || device >= kMaxDevices) {
        set_error("%s: HIP device %d not available -- there is no CPU fallback", who, device);
        return GS_E_NODEVICE;
    }
    GSC_HIP(hipSetDevice(device));
    // rocprim returns whatever hipGetLastError holds after its launches: an error another library left behind on this thread (RCCL
    // probes host pointers with hipPointerGetAttributes, which records "invalid argument") must not fail the next scan.  Our own
    // launches are checked where they are made.
    (void)hipGetLastError();
    return GS_OK;
}

namespace {

// ------------------------------------------------------------------------------------------
// rows
// ------------------------------------------------------------------------------------------
__global__ void k_rows_from_hits(const gs_hit *hits, uint64_t n, uint32_t name_base, const uint4 *rec, uint32_t *name, uint32_t *pos,
                                 uint32_t *contig, uint32_t *start, uint32_t *id, uint32_t *key) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const gs_hit h = hits[i];
    const uint4 r = rec[h.payload];
    name[i] = name_base + h.read;
    pos[i] = h.pos;
    contig[i] = r.y;
    start[i] = r.x;
    id[i] = r.z;
    key[i] = r.w;
}
__global__ void k_pack_rec(const uint32_t *contig, const int64_t *start, const int64_t *group, const uint32_t *loc, uint64_t n, uint4 *rec) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rec[i] = make_uint4((uint32_t)start[i], contig[i], (uint32_t)group[i], loc[i]);
}
__global__ void k_pack_rec_dev(const uint64_t *pos, const uint32_t *group, const uint32_t *loc, uint64_t n, uint4 *rec) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rec[i] = make_uint4((uint32_t)pos[i], (uint32_t)(pos[i] >> 32), group[i], loc[i]);
}
template <typename T>
__global__ void k_gather(const T *in, const uint32_t *idx, uint64_t n, T *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}
struct Cols6 {
    uint32_t *c[6];
};
__global__ void k_gather6(Cols6 in, const uint32_t *idx, uint64_t n, Cols6 out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = idx[i];
#pragma unroll
    for (int k = 0; k < 6; k++) out.c[k][i] = in.c[k][r];
}

int rowset_reserve(gs_rowset *rs, uint64_t want) {
    if (want <= rs->cap) return GS_OK;
    const uint64_t cap = std::max<uint64_t>(want, 2 * rs->cap + 1024);
    hipStream_t st = chain_stream(rs->device);
    uint32_t **cols[6] = {&rs->name, &rs->pos, &rs->contig, &rs->start, &rs->id, &rs->key};
    for (auto c : cols) {
        uint32_t *nw = static_cast<uint32_t *>(pool_alloc(rs->device, cap * sizeof(uint32_t)));
        if (!nw) return GS_E_NOMEM;
        if (rs->n) GSC_HIP(hipMemcpyAsync(nw, *c, rs->n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        pool_free(rs->device, *c);
        *c = nw;
    }
    rs->cap = cap;
    return GS_OK;
}

// ------------------------------------------------------------------------------------------
// segments
// ------------------------------------------------------------------------------------------
__global__ void k_head_u32(const uint32_t *v, uint64_t n, uint32_t *flag) {  // 1 where v[i] != v[i-1]
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || v[i] != v[i - 1]) ? 1u : 0u;
}
__global__ void k_head_u64(const uint64_t *v, uint64_t n, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || v[i] != v[i - 1]) ? 1u : 0u;
}
// seg[] holds the inclusive scan of the head flags (1-based): -> 0-based, and the first row of every segment
__global__ void k_seg_finish(const uint32_t *name, uint64_t n, uint32_t *seg, uint32_t *seg_off) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = seg[i] - 1u;
    seg[i] = s;
    if (i == 0 || name[i] != name[i - 1]) seg_off[s] = (uint32_t)i;
}
__global__ void k_dec(uint32_t *v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] -= 1u;
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
// the largest contig id and the largest start of the rows (how many key bits the sorts need), contigs that occur
__global__ void k_col_max(const uint32_t *contig, const uint32_t *start, uint64_t n, uint32_t *out) {
    uint32_t c = 0, s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        c = max(c, contig[i]);
        s = max(s, start[i]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        c = max(c, (uint32_t)__shfl_xor((int)c, o, 64));
        s = max(s, (uint32_t)__shfl_xor((int)s, o, 64));
    }
    if ((threadIdx.x & 63u) == 0) {  // (a fixed grid: a few thousand atomics on one address, not one per 64 rows)
        atomicMax(&out[0], c);
        atomicMax(&out[1], s);
    }
}
__global__ void k_mark_present(const uint32_t *contig, uint64_t n, uint8_t *present) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) present[contig[i]] = 1;
}
// sort keys of the rows idx[j]: group = contig << rank_bits | rank of the read's name
__global__ void k_group_key(const uint32_t *idx, uint64_t n, const uint32_t *contig, const uint32_t *name, const uint32_t *name_rank,
                            int rank_bits, uint64_t *gkey) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) gkey[j] = ((uint64_t)contig[idx[j]] << rank_bits) | name_rank[name[idx[j]]];
}
__global__ void k_group_start_key(const uint32_t *idx, uint64_t n, const uint32_t *contig, const uint32_t *name, const uint32_t *name_rank,
                                  const uint32_t *start, int rank_bits, int start_bits, uint64_t *key) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t r = idx[j];
    key[j] = ((((uint64_t)contig[r] << rank_bits) | name_rank[name[r]]) << start_bits) | start[r];
}
__global__ void k_col_key(const uint32_t *idx, uint64_t n, const uint32_t *col, uint64_t *k) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) k[j] = col[idx[j]];
}
__global__ void k_shift_right(uint64_t *k, uint64_t n, int bits) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) k[j] >>= bits;
}
// rows sorted by (group, start): the ID of every group's first row ...
__global__ void k_group_first_id(const uint32_t *head, const uint32_t *gidx, const uint32_t *idx, const uint32_t *id, uint64_t n, uint32_t *first_id) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && head[j]) first_id[gidx[j]] = id[idx[j]];
}
// ... and the groups in which some row carries another one: the reads with at least two distinct IDs (:91-97)
__global__ void k_group_multi(const uint32_t *gidx, const uint32_t *idx, const uint32_t *id, const uint32_t *first_id, uint64_t n, uint8_t *multi) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && id[idx[j]] != first_id[gidx[j]]) multi[gidx[j]] = 1;
}
// rows sorted by (group, start), original order inside: the rows of reads with >= 2 IDs (:91-97); of those, an exact duplicate of an
// earlier row goes (drop_duplicates, :104); then the reads of at least 10 kb (:106-108).  counts: [0] in reads with >= 2 IDs,
// [1] after drop_duplicates, [2] in long reads
__global__ void k_dedup_long(const uint64_t *gkey, const uint32_t *gidx, const uint8_t *multi, const uint32_t *idx, uint64_t n, const uint32_t *start,
                             const uint32_t *pos, const uint32_t *id, const uint32_t *name, const uint8_t *name_long, uint8_t *flag_out,
                             unsigned long long *counts) {
    uint32_t c0 = 0, c1 = 0, c2 = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        bool out = false;
        if (multi[gidx[j]]) {
            c0++;
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
            c1 += dup ? 0u : 1u;
            out = !dup && name_long[name[r]];
            c2 += out ? 1u : 0u;
        }
        flag_out[j] = out ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) {
        c0 += __shfl_xor((int)c0, o, 64);
        c1 += __shfl_xor((int)c1, o, 64);
        c2 += __shfl_xor((int)c2, o, 64);
    }
    if ((threadIdx.x & 63u) == 0) {  // (a fixed grid: one atomic per wave at the end)
        if (c0) atomicAdd(&counts[0], (unsigned long long)c0);
        if (c1) atomicAdd(&counts[1], (unsigned long long)c1);
        if (c2) atomicAdd(&counts[2], (unsigned long long)c2);
    }
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

// every v[i] < bound?  (ids that index caller-sized arrays: a wrong count must fail loudly, not read out of bounds)
__global__ void k_any_ge(const uint32_t *v, uint64_t n, uint32_t bound, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = i < n && v[i] >= bound;
    if (__ballot(bad) && (threadIdx.x & 63u) == 0) atomicOr(flag, 1u);
}
int check_ids_below(Ctx &c, const uint32_t *d_ids, uint64_t n, uint32_t bound, const char *what) {
    if (!n) return GS_OK;
    uint32_t *d_flag, h = 0;
    GSC_TRY(c.mem.get(&d_flag, 1));
    GSC_TRY(c.zero(d_flag, 1));
    hipLaunchKernelGGL(k_any_ge, dim3(blocks_of(n)), dim3(256), 0, c.st, d_ids, n, bound, d_flag);
    GSC_TRY(c.fetch(&h, d_flag));
    c.mem.release(d_flag);
    if (h) {
        set_error("%s: a row's id is not below %u", what, bound);
        return GS_E_ARG;
    }
    return GS_OK;
}

// process-by-contig_lowmem_AR.py:219-260 over T output rows in device memory (contig, ID, group = the read the row belongs to; a
// read's rows are adjacent): vertices = distinct (contig, ID), an edge between consecutive rows of a read, components with more
// than two SUNKs -> [min, max] per contig.  Commutative over reads: the rows of several ranks, concatenated, give the union of
// the whole run (gs_pairstage_from_rows).
int union_of_rows(Ctx &c, uint64_t T, const uint32_t *d_oc, const int64_t *d_oi, const uint32_t *d_og, gs_pairstage *ps) {
    uint64_t *d_vk, *d_vs, *d_uniq;
    uint32_t *d_vh, *d_vi, *d_rv, *d_size;
    int64_t *d_vid, *d_min, *d_max;
    GSC_TRY(c.mem.get(&d_vk, T));
    GSC_TRY(c.mem.get(&d_vs, T));
    hipLaunchKernelGGL(k_vertex_key, dim3(blocks_of(T)), dim3(256), 0, c.st, d_oc, d_oi, T, d_vk);
    GSC_HIP(hipMemcpyAsync(d_vs, d_vk, T * sizeof(uint64_t), hipMemcpyDeviceToDevice, c.st));
    GSC_TRY(sort_keys(c, &d_vs, T, 64));
    GSC_TRY(c.mem.get(&d_vh, T));
    GSC_TRY(c.mem.get(&d_vi, T));
    hipLaunchKernelGGL(k_head_u64, dim3(blocks_of(T)), dim3(256), 0, c.st, d_vs, T, d_vh);
    GSC_TRY(inclusive_scan(c, d_vh, d_vi, T, rocprim::plus<uint32_t>()));
    uint32_t nv = 0;
    GSC_TRY(c.fetch(&nv, d_vi + T - 1));
    hipLaunchKernelGGL(k_dec, dim3(blocks_of(T)), dim3(256), 0, c.st, d_vi, T);
    GSC_TRY(c.mem.get(&d_uniq, nv));
    GSC_TRY(c.mem.get(&d_vid, nv));
    hipLaunchKernelGGL(k_unique_fill, dim3(blocks_of(T)), dim3(256), 0, c.st, d_vs, d_vh, d_vi, T, d_uniq, d_vid);
    GSC_TRY(c.mem.get(&d_rv, T));
    hipLaunchKernelGGL(k_lookup_vertex, dim3(blocks_of(T)), dim3(256), 0, c.st, d_vk, T, d_uniq, nv, d_rv);
    GSC_TRY(c.mem.get(&d_size, nv));
    GSC_TRY(c.mem.get(&d_min, nv));
    GSC_TRY(c.mem.get(&d_max, nv));
    GSC_HIP(hipGetLastError());
    GSC_TRY(union_regions_device(c.device, T, d_og, d_rv, nv, d_vid, d_size, d_min, d_max));
    uint8_t *d_rf;
    uint32_t *d_rsel;
    GSC_TRY(c.mem.get(&d_rf, nv));
    GSC_TRY(c.mem.get(&d_rsel, nv));
    hipLaunchKernelGGL(k_region_flag, dim3(blocks_of(nv)), dim3(256), 0, c.st, d_size, nv, d_rf);
    uint64_t nr = 0;
    GSC_TRY(select_indices(c, d_rf, d_rsel, nv, &nr));
    if (nr) {
        uint64_t *d_ru;
        int64_t *d_rmin, *d_rmax;
        GSC_TRY(c.mem.get(&d_ru, nr));
        GSC_TRY(c.mem.get(&d_rmin, nr));
        GSC_TRY(c.mem.get(&d_rmax, nr));
        hipLaunchKernelGGL(k_gather<uint64_t>, dim3(blocks_of(nr)), dim3(256), 0, c.st, d_uniq, d_rsel, nr, d_ru);
        hipLaunchKernelGGL(k_gather<int64_t>, dim3(blocks_of(nr)), dim3(256), 0, c.st, d_min, d_rsel, nr, d_rmin);
        hipLaunchKernelGGL(k_gather<int64_t>, dim3(blocks_of(nr)), dim3(256), 0, c.st, d_max, d_rsel, nr, d_rmax);
        std::vector<uint64_t> uq(nr);
        std::vector<int64_t> mn(nr), mx(nr);
        GSC_HIP(hipMemcpyAsync(uq.data(), d_ru, nr * 8, hipMemcpyDeviceToHost, c.st));
        GSC_HIP(hipMemcpyAsync(mn.data(), d_rmin, nr * 8, hipMemcpyDeviceToHost, c.st));
        GSC_HIP(hipMemcpyAsync(mx.data(), d_rmax, nr * 8, hipMemcpyDeviceToHost, c.st));
        GSC_TRY(c.sync());
        struct Reg {
            uint32_t c;
            int64_t s, e;
        };
        std::vector<Reg> regs(nr);
        for (uint64_t v = 0; v < nr; v++) regs[v] = Reg{(uint32_t)(uq[v] >> 32), mn[v], mx[v]};
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

void gs::chain::rowset_drop_derived(gs_rowset *rs) {
    pool_free(rs->device, rs->seg);
    pool_free(rs->device, rs->seg_off);
    pool_free(rs->device, rs->keep);
    rs->seg = rs->seg_off = nullptr;
    rs->keep = nullptr;
    rs->seg_n = ~0ULL;
    rs->has_keep = false;
}

int gs::chain::compute_segments(Ctx &c, gs_rowset *rs) {
    if (rs->seg_n == rs->n && rs->seg) return GS_OK;
    rowset_drop_derived(rs);
    const uint64_t n = rs->n;
    rs->seg = static_cast<uint32_t *>(pool_alloc(rs->device, std::max<uint64_t>(n, 1) * sizeof(uint32_t)));
    if (!rs->seg) return GS_E_NOMEM;
    rs->n_segs = 0;
    if (n) {
        hipLaunchKernelGGL(k_head_u32, dim3(blocks_of(n)), dim3(256), 0, c.st, rs->name, n, rs->seg);
        GSC_TRY(inclusive_scan(c, rs->seg, rs->seg, n, rocprim::plus<uint32_t>()));
        GSC_TRY(c.fetch(&rs->n_segs, rs->seg + n - 1));
    }
    rs->seg_off = static_cast<uint32_t *>(pool_alloc(rs->device, ((uint64_t)rs->n_segs + 1) * sizeof(uint32_t)));
    if (!rs->seg_off) return GS_E_NOMEM;
    if (n) {
        hipLaunchKernelGGL(k_seg_finish, dim3(blocks_of(n)), dim3(256), 0, c.st, rs->name, n, rs->seg, rs->seg_off);
        GSC_HIP(hipGetLastError());
    }
    const uint32_t n32 = (uint32_t)n;
    GSC_HIP(hipMemcpyAsync(rs->seg_off + rs->n_segs, &n32, sizeof(uint32_t), hipMemcpyHostToDevice, c.st));
    GSC_TRY(c.sync());  // (n32 is a local)
    rs->seg_n = n;
    return GS_OK;
}

extern "C" {

// ---- loc rows ----------------------------------------------------------------------------------
int gs_locrows_create(int device, uint64_t n_rows, const uint32_t *row_contig, const int64_t *row_start, const int64_t *row_group,
                      const uint32_t *row_loc, gs_locrows **out) {
    if (!out || (n_rows && (!row_contig || !row_start || !row_group || !row_loc))) {
        set_error("gs_locrows_create: null argument");
        return GS_E_ARG;
    }
    GSC_TRY(check_device(device, "gs_locrows_create"));
    hipStream_t st = chain_stream(device);
    // (four transient uploads of the whole location table: straight from the driver and back to it, not through the pool)
    struct Tmp {
        void *p[4] = {nullptr, nullptr, nullptr, nullptr};
        ~Tmp() {
            for (void *q : p)
                if (q) (void)hipFree(q);
        }
    } tmp;
    const uint64_t m = std::max<uint64_t>(n_rows, 1);
    GSC_HIP(hipMalloc(&tmp.p[0], m * sizeof(uint32_t)));
    GSC_HIP(hipMalloc(&tmp.p[1], m * sizeof(int64_t)));
    GSC_HIP(hipMalloc(&tmp.p[2], m * sizeof(int64_t)));
    GSC_HIP(hipMalloc(&tmp.p[3], m * sizeof(uint32_t)));
    uint4 *rec = nullptr;
    GSC_HIP(hipMalloc((void **)&rec, m * sizeof(uint4)));
    if (n_rows) {
        hipError_t e = hipMemcpyAsync(tmp.p[0], row_contig, n_rows * sizeof(uint32_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(tmp.p[1], row_start, n_rows * sizeof(int64_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(tmp.p[2], row_group, n_rows * sizeof(int64_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(tmp.p[3], row_loc, n_rows * sizeof(uint32_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_pack_rec, dim3(blocks_of(n_rows)), dim3(256), 0, st, static_cast<const uint32_t *>(tmp.p[0]),
                               static_cast<const int64_t *>(tmp.p[1]), static_cast<const int64_t *>(tmp.p[2]), static_cast<const uint32_t *>(tmp.p[3]),
                               n_rows, rec);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            (void)hipFree(rec);
            set_error("gs_locrows_create: %s", hipGetErrorString(e));
            return GS_E_NODEVICE;
        }
    }
    uint32_t n_locs = 0;
    for (uint64_t i = 0; i < n_rows; i++)
        if (row_loc[i] != GS_MISSING) n_locs = std::max(n_locs, row_loc[i] + 1u);
    gs_locrows *lr = new gs_locrows;
    lr->device = device;
    lr->n_rows = n_rows;
    lr->n_locs = n_locs;
    lr->rec = rec;
    *out = lr;
    return GS_OK;
}

// the loc rows of a SUNK set defined on the device: packed from its pos / group / loc arrays (the handle does not refer to the set
// afterwards)
int gs_locrows_from_sunkset_dev(int device, const gs_sunkset_dev *s, gs_locrows **out) {
    if (!s || !out) {
        set_error("gs_locrows_from_sunkset_dev: null argument");
        return GS_E_ARG;
    }
    GSC_TRY(check_device(device, "gs_locrows_from_sunkset_dev"));
    const uint64_t n_rows = gs_sunkset_dev_n(s);
    uint4 *rec = nullptr;
    GSC_HIP(hipMalloc((void **)&rec, std::max<uint64_t>(n_rows, 1) * sizeof(uint4)));
    if (n_rows) {
        GSC_HIP(hipDeviceSynchronize());  // (the set's arrays may have been written on another stream)
        hipStream_t st = chain_stream(device);
        hipLaunchKernelGGL(k_pack_rec_dev, dim3(blocks_of(n_rows)), dim3(256), 0, st, static_cast<const uint64_t *>(gs_sunkset_dev_pos(s)),
                           static_cast<const uint32_t *>(gs_sunkset_dev_group(s)), static_cast<const uint32_t *>(gs_sunkset_dev_loc(s)), n_rows, rec);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            (void)hipFree(rec);
            set_error("gs_locrows_from_sunkset_dev: %s", hipGetErrorString(e));
            return GS_E_NODEVICE;
        }
    }
    gs_locrows *lr = new gs_locrows;
    lr->device = device;
    lr->n_rows = n_rows;
    lr->n_locs = (uint32_t)gs_sunkset_dev_n_groups(s);
    lr->rec = rec;
    *out = lr;
    return GS_OK;
}

void gs_locrows_free(gs_locrows *lr) {
    if (!lr) return;
    (void)hipSetDevice(lr->device);
    (void)hipStreamSynchronize(chain_stream(lr->device));
    (void)hipFree(lr->rec);
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
    GSC_TRY(check_device(lr->device, "gs_locrows_key_contig"));
    std::fill(key_contig, key_contig + lr->n_locs, GS_NONE);
    const uint64_t chunk = 1ULL << 24;
    std::vector<uint4> p(std::min(chunk, std::max<uint64_t>(lr->n_rows, 1)));
    for (uint64_t a = 0; a < lr->n_rows; a += chunk) {
        const uint64_t m = std::min(chunk, lr->n_rows - a);
        GSC_HIP(hipMemcpy(p.data(), lr->rec + a, m * sizeof(uint4), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < m; i++)
            if (p[i].w < lr->n_locs) key_contig[p[i].w] = p[i].y;
    }
    return GS_OK;
}

// ---- row sets ----------------------------------------------------------------------------------
int gs_rowset_create(int device, uint64_t cap_hint, gs_rowset **out) {
    if (!out) {
        set_error("gs_rowset_create: null argument");
        return GS_E_ARG;
    }
    GSC_TRY(check_device(device, "gs_rowset_create"));
    gs_rowset *rs = new gs_rowset;
    rs->device = device;
    int rc = rowset_reserve(rs, std::max<uint64_t>(cap_hint, 1024));
    if (rc) {
        gs_rowset_free(rs);
        return rc;
    }
    *out = rs;
    return GS_OK;
}
void gs_rowset_free(gs_rowset *rs) {
    if (!rs) return;
    (void)hipSetDevice(rs->device);
    rowset_drop_derived(rs);
    for (uint32_t *p : {rs->name, rs->pos, rs->contig, rs->start, rs->id, rs->key}) pool_free(rs->device, p);
    delete rs;
}
uint64_t gs_rowset_n(const gs_rowset *rs) { return rs ? rs->n : 0; }

int gs_rowset_append_pass(gs_rowset *rs, gs_table *t, gs_reads *r, uint32_t name_base, const gs_locrows *lr) {
    if (!rs || !t || !r || !lr) {
        set_error("gs_rowset_append_pass: null argument");
        return GS_E_ARG;
    }
    GSC_TRY(check_device(rs->device, "gs_rowset_append_pass"));
    const void *d_hits = nullptr;
    uint64_t n = 0;
    GSC_TRY(gs_tag_hits_device(t, r, &d_hits, &n));
    if (rs->n + n >= 0xFFFFFFFFULL) {
        set_error("gs_rowset_append_pass: more than 2^32 rows in one row set");
        return GS_E_CAPACITY;
    }
    GSC_TRY(rowset_reserve(rs, rs->n + n));
    if (n) {
        hipStream_t st = chain_stream(rs->device);
        const uint64_t o = rs->n;
        hipLaunchKernelGGL(k_rows_from_hits, dim3(blocks_of(n)), dim3(256), 0, st, static_cast<const gs_hit *>(d_hits), n, name_base, lr->rec,
                           rs->name + o, rs->pos + o, rs->contig + o, rs->start + o, rs->id + o, rs->key + o);
        GSC_HIP(hipGetLastError());
        GSC_HIP(hipStreamSynchronize(st));  // the batch's hit list may be overwritten by its next pass once this returns
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
    GSC_TRY(check_device(rs->device, "gs_rowset_append_host"));
    if (rs->n + n >= 0xFFFFFFFFULL) {
        set_error("gs_rowset_append_host: more than 2^32 rows in one row set");
        return GS_E_CAPACITY;
    }
    GSC_TRY(rowset_reserve(rs, rs->n + n));
    if (n) {
        hipStream_t st = chain_stream(rs->device);
        const uint64_t o = rs->n;
        GSC_HIP(hipMemcpyAsync(rs->name + o, name, n * 4, hipMemcpyHostToDevice, st));
        GSC_HIP(hipMemcpyAsync(rs->pos + o, pos, n * 4, hipMemcpyHostToDevice, st));
        GSC_HIP(hipMemcpyAsync(rs->contig + o, contig, n * 4, hipMemcpyHostToDevice, st));
        GSC_HIP(hipMemcpyAsync(rs->start + o, start, n * 4, hipMemcpyHostToDevice, st));
        GSC_HIP(hipMemcpyAsync(rs->id + o, id, n * 4, hipMemcpyHostToDevice, st));
        if (key) GSC_HIP(hipMemcpyAsync(rs->key + o, key, n * 4, hipMemcpyHostToDevice, st));
        else rs->has_key = false;
        GSC_HIP(hipStreamSynchronize(st));
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
    GSC_TRY(check_device(rs->device, "gs_rowset_download"));
    hipStream_t st = chain_stream(rs->device);
    const uint32_t *src[6] = {rs->name, rs->pos, rs->contig, rs->start, rs->id, rs->key};
    uint32_t *dst[6] = {name, pos, contig, start, id, key};
    for (int c = 0; c < 6; c++)
        if (dst[c] && rs->n) GSC_HIP(hipMemcpyAsync(dst[c], src[c], rs->n * 4, hipMemcpyDeviceToHost, st));
    GSC_HIP(hipStreamSynchronize(st));
    return GS_OK;
}

int gs_rowset_segments(gs_rowset *rs, uint32_t *n_segs) {
    if (!rs || !n_segs) {
        set_error("gs_rowset_segments: null argument");
        return GS_E_ARG;
    }
    GSC_TRY(check_device(rs->device, "gs_rowset_segments"));
    Ctx c(rs->device);
    GSC_TRY(compute_segments(c, rs));
    *n_segs = rs->n_segs;
    return GS_OK;
}

int gs_rowset_keep_flags(const gs_rowset *rs, uint8_t *keep) {
    if (!rs || !keep || !rs->has_keep) {
        set_error("gs_rowset_keep_flags: no diagonal filter has run on this row set");
        return GS_E_ARG;
    }
    GSC_TRY(check_device(rs->device, "gs_rowset_keep_flags"));
    if (rs->n) {
        hipStream_t st = chain_stream(rs->device);
        GSC_HIP(hipMemcpyAsync(keep, rs->keep, rs->n, hipMemcpyDeviceToHost, st));
        GSC_HIP(hipStreamSynchronize(st));
    }
    return GS_OK;
}

// diag_filter_step2: the rows whose contig is their read's best contig stay, in order
int gs_rowset_keep_filtered(gs_rowset *rs) {
    if (!rs || !rs->has_keep) {
        set_error("gs_rowset_keep_filtered: no diagonal filter has run on this row set");
        return GS_E_ARG;
    }
    GSC_TRY(check_device(rs->device, "gs_rowset_keep_filtered"));
    Ctx c(rs->device);
    const uint64_t n = rs->n;
    uint32_t *d_sel;
    GSC_TRY(c.mem.get(&d_sel, n));
    uint64_t m = 0;
    GSC_TRY(select_indices(c, rs->keep, d_sel, n, &m));
    // the six columns move to arrays of the new size (one pass over the selection; the old ones go back to the pool)
    const uint64_t cap = std::max<uint64_t>(m, 1024);
    Cols6 in{{rs->name, rs->pos, rs->contig, rs->start, rs->id, rs->key}}, out{};
    for (int k = 0; k < 6; k++) {
        out.c[k] = static_cast<uint32_t *>(pool_alloc(rs->device, cap * sizeof(uint32_t)));
        if (!out.c[k]) {
            for (int q = 0; q < k; q++) pool_free(rs->device, out.c[q]);
            return GS_E_NOMEM;
        }
    }
    if (m) {
        hipLaunchKernelGGL(k_gather6, dim3(blocks_of(m)), dim3(256), 0, c.st, in, d_sel, m, out);
        GSC_HIP(hipGetLastError());
    }
    for (int k = 0; k < 6; k++) pool_free(rs->device, in.c[k]);
    rs->name = out.c[0];
    rs->pos = out.c[1];
    rs->contig = out.c[2];
    rs->start = out.c[3];
    rs->id = out.c[4];
    rs->key = out.c[5];
    rs->cap = cap;
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
    // With a reduce callback every rank has to enter it once per haplotype whatever happens locally (the other ranks are in the
    // same collective): a local failure is remembered, the callback still runs (on whatever the counters hold), the error
    // is returned afterwards.
    int rc = check_device(hap1->device, "gs_rowset_bad_keys");
    if (rc != GS_OK && !reduce) return rc;
    Ctx c(hap1->device);
    uint8_t *d_hap = nullptr, *d_bad = nullptr;
    uint32_t *d_count = nullptr, *d_hist = nullptr;
    const uint32_t hist_len = 1u << 16;
    auto step = [&](int r) {
        if (rc == GS_OK) rc = r;
    };
    if (rc == GS_OK) step(check_ids_below(c, hap1->key, hap1->n, n_keys, "gs_rowset_bad_keys: key"));
    if (rc == GS_OK) step(check_ids_below(c, hap2->key, hap2->n, n_keys, "gs_rowset_bad_keys: key"));
    if (rc == GS_OK) step(c.mem.put(&d_hap, key_hap, n_keys));
    if (rc == GS_OK) step(c.mem.get(&d_hist, hist_len));
    if (rc == GS_OK) {
        d_bad = static_cast<uint8_t *>(pool_alloc(hap1->device, n_keys));
        if (!d_bad) rc = GS_E_NOMEM;
    }
    if (rc == GS_OK && hipMemsetAsync(d_bad, 0, n_keys, c.st) != hipSuccess) rc = GS_E_NODEVICE;
    {
        int r = c.mem.get(&d_count, n_keys);  // (needed by the callback even when something above failed)
        if (r != GS_OK) {
            pool_free(hap1->device, d_bad);
            return r;  // no counters to hand to the collective: nothing this rank can do but fail (the others time out in it)
        }
    }
    std::vector<uint32_t> hist(hist_len);
    for (int h = 0; h < 2; h++) {
        const gs_rowset *rs = h == 0 ? hap1 : hap2;
        const uint8_t bit = (uint8_t)(1u << h);
        hipError_t e = hipMemsetAsync(d_count, 0, (size_t)n_keys * 4, c.st);
        if (e == hipSuccess && rc == GS_OK) e = hipMemsetAsync(d_hist, 0, hist_len * 4, c.st);
        if (e == hipSuccess && rc == GS_OK && rs->n) hipLaunchKernelGGL(k_count_keys, dim3(blocks_of(rs->n)), dim3(256), 0, c.st, rs->key, rs->n, d_count);
        if (e != hipSuccess && rc == GS_OK) {
            set_error("gs_rowset_bad_keys: %s", hipGetErrorString(e));
            rc = GS_E_NODEVICE;
        }
        if (reduce) {  // reads sharded over ranks: the counts of all of them (one sum per haplotype)
            (void)hipStreamSynchronize(c.st);
            const int r = reduce(reduce_ctx, d_count, n_keys);
            if (rc == GS_OK) rc = r;
        }
        if (rc != GS_OK) continue;
        hipLaunchKernelGGL(k_hist_counts, dim3(blocks_of(n_keys)), dim3(256), 0, c.st, d_count, d_hap, bit, n_keys, d_hist, hist_len);
        if ((rc = c.fetch(hist.data(), d_hist, hist_len)) != GS_OK) continue;
        uint64_t best = 0, mode = 0;
        for (uint32_t q = 1; q < hist_len - 1; q++)
            if (hist[q] > best) {
                best = hist[q];
                mode = q;
            }
        if (hist[hist_len - 1] && hist[hist_len - 1] >= best) {  // absurdly deep data: exact mode on the host
            std::vector<uint32_t> cnt(n_keys);
            std::vector<uint32_t> big;
            if ((rc = c.fetch(cnt.data(), d_count, n_keys)) != GS_OK) continue;
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
            continue;
        }
        if (meancov) meancov[h] = (uint32_t)mode;
        const double limit = (double)mode + std::sqrt((double)mode) * 4.0;  // badsunks_AR.py:46
        hipLaunchKernelGGL(k_mark_bad, dim3(blocks_of(n_keys)), dim3(256), 0, c.st, d_count, d_hap, bit, n_keys, limit, d_bad);
    }
    if (rc == GS_OK && hipGetLastError() != hipSuccess) {
        set_error("gs_rowset_bad_keys: kernel launch failed");
        rc = GS_E_NODEVICE;
    }
    if (rc == GS_OK && bad_host) rc = c.fetch(bad_host, d_bad, n_keys);
    else if (rc == GS_OK) rc = c.sync();
    if (rc != GS_OK || !d_bad_out) pool_free(hap1->device, d_bad);
    else *d_bad_out = d_bad;
    return rc;
}
int gs_device_free(int device, void *p) {
    GSC_TRY(check_device(device, "gs_device_free"));
    pool_free(device, p);
    return GS_OK;
}
int gs_device_pool_trim(int device) {
    GSC_TRY(check_device(device, "gs_device_pool_trim"));
    pool_trim(device);
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
    GSC_TRY(check_device(rs->device, "gs_rowset_pair_stage"));
    Ctx c(rs->device);
    GSC_TRY(check_ids_below(c, rs->name, rs->n, n_names, "gs_rowset_pair_stage: name"));
    gs_pairstage *ps = new gs_pairstage;
    struct Guard {
        gs_pairstage *p;
        ~Guard() { delete p; }
    } guard{ps};
    auto done = [&]() {
        *out = ps;
        guard.p = nullptr;
        return GS_OK;
    };
    const uint64_t n = rs->n;
    ps->counters[0] = n;
    if (!n) return done();
    uint32_t *d_rank;
    uint8_t *d_long, *d_rowbad = nullptr;
    GSC_TRY(c.mem.put(&d_rank, name_rank, n_names));
    GSC_TRY(c.mem.put(&d_long, name_long, n_names));
    if (row_bad) GSC_TRY(c.mem.put(&d_rowbad, row_bad, n));
    uint32_t max_rank = 0;
    for (uint32_t i = 0; i < n_names; i++) max_rank = std::max(max_rank, name_rank[i]);
    // the largest contig id / start (key bits of the sort), and the distinct contigs of the input (the one-line output of a contig
    // without result names it, :92-94,202-204)
    uint32_t mx[2] = {0, 0};
    {
        uint32_t *d_mx;
        GSC_TRY(c.mem.get(&d_mx, 2));
        GSC_TRY(c.zero(d_mx, 2));
        hipLaunchKernelGGL(k_col_max, dim3(std::min(blocks_of(n), 2048u)), dim3(256), 0, c.st, rs->contig, rs->start, n, d_mx);
        GSC_TRY(c.fetch(mx, d_mx, 2));
        uint8_t *d_present;
        uint32_t *d_csel;
        const uint64_t nc_all = (uint64_t)mx[0] + 1;
        if (nc_all > (1ULL << 28)) {
            set_error("gs_rowset_pair_stage: contig id %u: ids are expected to be dense", mx[0]);
            return GS_E_ARG;
        }
        GSC_TRY(c.mem.get(&d_present, nc_all));
        GSC_TRY(c.mem.get(&d_csel, nc_all));
        GSC_TRY(c.zero(d_present, nc_all));
        hipLaunchKernelGGL(k_mark_present, dim3(blocks_of(n)), dim3(256), 0, c.st, rs->contig, n, d_present);
        uint64_t nc = 0;
        GSC_TRY(select_indices(c, d_present, d_csel, nc_all, &nc));
        ps->contigs_in.resize(nc);
        GSC_TRY(c.fetch(ps->contigs_in.data(), d_csel, nc));
        c.mem.release(d_present);
        c.mem.release(d_csel);
    }
    // 1. rows whose chrom:ID is not a bad SUNK (:66-72)
    uint8_t *d_flag;
    uint32_t *d_idx;
    GSC_TRY(c.mem.get(&d_flag, n));
    GSC_TRY(c.mem.get(&d_idx, n));
    hipLaunchKernelGGL(k_not_bad, dim3(blocks_of(n)), dim3(256), 0, c.st, rs->key, static_cast<const uint8_t *>(d_key_bad), d_rowbad, n, d_flag);
    uint64_t n0 = 0;
    GSC_TRY(select_indices(c, d_flag, d_idx, n, &n0));
    ps->counters[1] = n0;
    if (!n0) return done();
    // 2 + 3. ONE stable sort by (contig, name rank, start) (:100; file order inside equal keys, as sort_values gives) lays out the read
    // groups; a read (per contig) has at least two distinct IDs (:91-97) iff some row of its group carries another ID than the
    // group's first row, and dropping the other groups before or after the sort leaves the same rows in the same order.
    // (Round 4 sorted twice by 64 + 32 key bits, once by (group, ID) to count IDs and once by (group, start).)
    const int cb = bits_for(mx[0]), rb = bits_for(max_rank), sb = bits_for(mx[1]);
    uint64_t *d_gkey;
    GSC_TRY(c.mem.get(&d_gkey, n0));
    if (cb + rb + sb <= 64) {
        hipLaunchKernelGGL(k_group_start_key, dim3(blocks_of(n0)), dim3(256), 0, c.st, d_idx, n0, rs->contig, rs->name, d_rank, rs->start, rb, sb, d_gkey);
        GSC_TRY(sort_pairs(c, &d_gkey, &d_idx, n0, cb + rb + sb));
        hipLaunchKernelGGL(k_shift_right, dim3(blocks_of(n0)), dim3(256), 0, c.st, d_gkey, n0, sb);
    } else {  // (the key does not fit 64 bits: by start, then by group; both stable)
        uint64_t *d_k2;
        GSC_TRY(c.mem.get(&d_k2, n0));
        hipLaunchKernelGGL(k_col_key, dim3(blocks_of(n0)), dim3(256), 0, c.st, d_idx, n0, rs->start, d_k2);
        GSC_TRY(sort_pairs(c, &d_k2, &d_idx, n0, sb));
        c.mem.release(d_k2);
        hipLaunchKernelGGL(k_group_key, dim3(blocks_of(n0)), dim3(256), 0, c.st, d_idx, n0, rs->contig, rs->name, d_rank, rb, d_gkey);
        GSC_TRY(sort_pairs(c, &d_gkey, &d_idx, n0, cb + rb));
    }
    uint32_t *d_head, *d_gidx, *d_first;
    uint8_t *d_multi, *d_fo;
    uint32_t *d_sel;
    unsigned long long *d_counts;
    GSC_TRY(c.mem.get(&d_head, n0));
    GSC_TRY(c.mem.get(&d_gidx, n0));
    hipLaunchKernelGGL(k_head_u64, dim3(blocks_of(n0)), dim3(256), 0, c.st, d_gkey, n0, d_head);
    GSC_TRY(inclusive_scan(c, d_head, d_gidx, n0, rocprim::plus<uint32_t>()));
    hipLaunchKernelGGL(k_dec, dim3(blocks_of(n0)), dim3(256), 0, c.st, d_gidx, n0);
    GSC_TRY(c.mem.get(&d_first, n0));  // (at most n0 groups)
    GSC_TRY(c.mem.get(&d_multi, n0));
    GSC_TRY(c.zero(d_multi, n0));
    hipLaunchKernelGGL(k_group_first_id, dim3(blocks_of(n0)), dim3(256), 0, c.st, d_head, d_gidx, d_idx, rs->id, n0, d_first);
    hipLaunchKernelGGL(k_group_multi, dim3(blocks_of(n0)), dim3(256), 0, c.st, d_gidx, d_idx, rs->id, d_first, n0, d_multi);
    // exact duplicates (:104); reads of at least 10 kb (:106-108)
    GSC_TRY(c.mem.get(&d_fo, n0));
    GSC_TRY(c.mem.get(&d_sel, n0));
    GSC_TRY(c.mem.get(&d_counts, 3));
    GSC_TRY(c.zero(d_counts, 3));
    hipLaunchKernelGGL(k_dedup_long, dim3(std::min(blocks_of(n0), 2048u)), dim3(256), 0, c.st, d_gkey, d_gidx, d_multi, d_idx, n0, rs->start, rs->pos, rs->id, rs->name,
                       d_long, d_fo, d_counts);
    uint64_t n3 = 0;
    GSC_TRY(select_indices(c, d_fo, d_sel, n0, &n3));
    unsigned long long counts[3] = {0, 0, 0};
    GSC_TRY(c.fetch(counts, d_counts, 3));
    ps->counters[2] = counts[0];
    ps->counters[3] = counts[1];
    ps->counters[4] = n3;
    c.mem.release(d_first);
    c.mem.release(d_multi);
    c.mem.release(d_fo);
    if (!n3) return done();
    // 4. the read groups and the pair stage proper (:135-198)
    uint64_t *d_fkey, *d_off;
    int64_t *d_fpos, *d_fstart, *d_fid, *d_outids;
    uint32_t *d_frow, *d_outn;
    GSC_TRY(c.mem.get(&d_fkey, n3));
    GSC_TRY(c.mem.get(&d_fpos, n3));
    GSC_TRY(c.mem.get(&d_fstart, n3));
    GSC_TRY(c.mem.get(&d_fid, n3));
    GSC_TRY(c.mem.get(&d_frow, n3));
    hipLaunchKernelGGL(k_final_rows, dim3(blocks_of(n3)), dim3(256), 0, c.st, d_sel, n3, d_gkey, d_idx, rs->pos, rs->start, rs->id, d_fkey,
                       d_fpos, d_fstart, d_fid, d_frow);
    hipLaunchKernelGGL(k_head_u64, dim3(blocks_of(n3)), dim3(256), 0, c.st, d_fkey, n3, d_head);
    GSC_TRY(inclusive_scan(c, d_head, d_gidx, n3, rocprim::plus<uint32_t>()));
    uint32_t n_groups = 0;
    GSC_TRY(c.fetch(&n_groups, d_gidx + n3 - 1));
    hipLaunchKernelGGL(k_dec, dim3(blocks_of(n3)), dim3(256), 0, c.st, d_gidx, n3);
    ps->counters[5] = n_groups;
    GSC_TRY(c.mem.get(&d_off, (uint64_t)n_groups + 1));
    hipLaunchKernelGGL(k_group_off, dim3(blocks_of(n3)), dim3(256), 0, c.st, d_head, d_gidx, n3, d_off);
    GSC_HIP(hipMemcpyAsync(d_off + n_groups, &n3, 8, hipMemcpyHostToDevice, c.st));
    GSC_TRY(c.mem.get(&d_outids, n3));
    GSC_TRY(c.mem.get(&d_outn, (uint64_t)n_groups + 1));
    GSC_TRY(c.zero(d_outn, (uint64_t)n_groups + 1));
    GSC_HIP(hipGetLastError());
    GSC_TRY(c.sync());
    c.mem.release(d_gkey);
    c.mem.release(d_idx);
    c.mem.release(d_sel);
    c.mem.release(d_fkey);
    GSC_TRY(pair_component_device(c.device, n_groups, n3, d_off, d_fpos, d_fstart, d_fid, d_outn, d_outids));
    // 5. output rows
    uint64_t *d_outn64, *d_outoff;
    GSC_TRY(c.mem.get(&d_outn64, (uint64_t)n_groups + 1));
    GSC_TRY(c.mem.get(&d_outoff, (uint64_t)n_groups + 1));
    hipLaunchKernelGGL(k_widen, dim3(blocks_of((uint64_t)n_groups + 1)), dim3(256), 0, c.st, d_outn, d_outn64, (uint64_t)n_groups + 1);
    GSC_TRY(exclusive_sum(c, d_outn64, d_outoff, (uint64_t)n_groups + 1));
    uint64_t T = 0;
    GSC_TRY(c.fetch(&T, d_outoff + n_groups));
    if (!T) return done();
    uint32_t *d_oc, *d_on, *d_og;
    int64_t *d_oi;
    GSC_TRY(c.mem.get(&d_oc, T));
    GSC_TRY(c.mem.get(&d_on, T));
    GSC_TRY(c.mem.get(&d_og, T));
    GSC_TRY(c.mem.get(&d_oi, T));
    hipLaunchKernelGGL(k_out_rows, dim3(n_groups), dim3(64), 0, c.st, n_groups, d_off, d_outn, d_outoff, d_outids, d_frow, rs->contig, rs->name,
                       d_oc, d_on, d_oi, d_og);
    GSC_HIP(hipGetLastError());
    ps->row_contig.resize(T);
    ps->row_name.resize(T);
    ps->row_id.resize(T);
    GSC_HIP(hipMemcpyAsync(ps->row_contig.data(), d_oc, T * 4, hipMemcpyDeviceToHost, c.st));
    GSC_HIP(hipMemcpyAsync(ps->row_name.data(), d_on, T * 4, hipMemcpyDeviceToHost, c.st));
    GSC_HIP(hipMemcpyAsync(ps->row_id.data(), d_oi, T * 8, hipMemcpyDeviceToHost, c.st));
    // 6. contig-wide union of the reads' ID sets (:219-260)
    GSC_TRY(union_of_rows(c, T, d_oc, d_oi, d_og, ps));
    GSC_TRY(c.sync());
    ps->counters[6] = T;
    return done();
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
    GSC_TRY(check_device(device, "gs_pairstage_from_rows"));
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
        Ctx c(device);
        uint32_t *d_oc, *d_on, *d_head, *d_og;
        int64_t *d_oi;
        uint64_t *d_gk;
        GSC_TRY(c.mem.put(&d_oc, row_contig, n_rows));
        GSC_TRY(c.mem.put(&d_on, row_name, n_rows));
        GSC_TRY(c.mem.put(&d_oi, row_id, n_rows));
        GSC_TRY(c.mem.get(&d_gk, n_rows));
        GSC_TRY(c.mem.get(&d_head, n_rows));
        GSC_TRY(c.mem.get(&d_og, n_rows));
        hipLaunchKernelGGL(k_pack_u32x2, dim3(blocks_of(n_rows)), dim3(256), 0, c.st, d_oc, d_on, n_rows, d_gk);
        hipLaunchKernelGGL(k_head_u64, dim3(blocks_of(n_rows)), dim3(256), 0, c.st, d_gk, n_rows, d_head);
        GSC_TRY(inclusive_scan(c, d_head, d_og, n_rows, rocprim::plus<uint32_t>()));  // (group numbers from 1: only equality matters)
        GSC_TRY(union_of_rows(c, n_rows, d_oc, d_oi, d_og, ps));
        GSC_TRY(c.sync());
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

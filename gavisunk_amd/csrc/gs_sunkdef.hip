// SUNK definition on the GPU: what GAVISUNK obtains from
// `jellyfish count -C -U 1` + `mrsfast -e 0` + `bedtools merge/intersect` (workflow/rules/defineSUNKs.smk:22-127)
// -- the canonical k-mers that occur exactly once over hap1+hap2, their positions, and the merged
// overlapping/book-ended [start, start+k) intervals (group id = start of the merged interval).
//
//   k_window_codes   canonical code of every window (windows with a non-ACGT byte are not k-mers), in P passes over the
//                    code space so that hap1 + hap2 of a human genome (6e9 windows) fit the sorts: count, scan, emit
//   radix sort       (code, contig<<32|pos) by code           (rocPRIM: plain device-wide primitive)
//   k_mark_unique    code differs from both neighbours
//   select           -> the singly-unique ones (of this pass; the passes' survivors are concatenated)
//   radix sort       by (contig, pos)  = kmer.loc order
//   k_group_flags / inclusive sum-scan / k_group_loc   group = start of the run with gaps <= k, loc = dense id of the run
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <memory>
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

#define GS_TRY(expr)                  \
    do {                              \
        int _rc = (expr);             \
        if (_rc != GS_OK) return _rc; \
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

// one thread per 64 consecutive windows of a contig.  The definition runs in P passes over the code space (P a power of two;
// a pass takes the windows whose canonical code falls into its partition: uniform by a multiplicative hash), so that an
// assembly of any size -- hap1 + hap2 of a human genome is 6e9 windows -- is counted with sorts that fit device memory:
//   EMIT = false   per chunk, the number of k-mers (windows without a non-ACGT byte) of partition `part`
//   EMIT = true    those k-mers' (canonical code, contig << 32 | window), compacted, at chunk_base[chunk] ..
constexpr int kChunk = 64;
__device__ __forceinline__ uint32_t code_partition(uint64_t c, uint32_t pmask) { return (uint32_t)((c * 0x9E3779B97F4A7C15ULL) >> 56) & pmask; }
template <bool EMIT>
__global__ void k_window_codes(const uint8_t *seq, const uint64_t *coff, const uint64_t *chunk_off, uint32_t n_contigs, int k,
                               uint32_t pmask, uint32_t part, uint64_t *chunk_count, const uint64_t *chunk_base, uint64_t *keys,
                               uint64_t *vals) {
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
    uint64_t o = EMIT ? chunk_base[t] : 0;
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
        if (e + 1 >= w0 + (uint64_t)k && good >= k) {
            const uint64_t code = fwd < rev ? fwd : rev;
            if (code_partition(code, pmask) == part) {
                if (EMIT) {
                    keys[o] = code;
                    vals[o] = ((uint64_t)c << 32) | (e + 1 - k);
                }
                o++;
            }
        }
    }
    if (!EMIT) chunk_count[t] = o;
}

__global__ void k_mark_unique(const uint64_t *keys, uint64_t n, uint8_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t x = keys[i];
    flag[i] = ((i == 0 || keys[i - 1] != x) && (i + 1 == n || keys[i + 1] != x)) ? 1 : 0;
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

int gs::sort_pairs_u64_u32(const uint64_t *k_in, uint64_t *k_out, const uint32_t *v_in, uint32_t *v_out, uint64_t n, int end_bit,
                           void *stream) {
    if (!n) return GS_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    size_t tb = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, k_in, k_out, v_in, v_out, n, 0u, (unsigned)end_bit, st));
    void *tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
    hipError_t e = rocprim::radix_sort_pairs(tmp, tb, k_in, k_out, v_in, v_out, n, 0u, (unsigned)end_bit, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (e != hipSuccess) {
        set_error("radix sort failed: %s", hipGetErrorString(e));
        return GS_E_NODEVICE;
    }
    return GS_OK;
}

int gs::sort_keys_u64(const uint64_t *k_in, uint64_t *k_out, uint64_t n, void *stream) {
    if (!n) return GS_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    size_t tb = 0;
    HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, k_in, k_out, n, 0u, 64u, st));
    void *tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
    hipError_t e = rocprim::radix_sort_keys(tmp, tb, k_in, k_out, n, 0u, 64u, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (e != hipSuccess) {
        set_error("radix sort failed: %s", hipGetErrorString(e));
        return GS_E_NODEVICE;
    }
    return GS_OK;
}

// ---- the join of <kmers.db> with <kmer.loc> on the device (what the reference builds as HashSet + Table on the host,
// kmerpos_annot3.nim:20-72): sorts and a merge instead of 1e8 .. 1e9 hash-map inserts ----
namespace {
__global__ void k_dbj_iota(uint32_t *o, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = (uint32_t)i;
}
__global__ void k_dbj_tail(const uint64_t *k, uint64_t n, uint8_t *f) {  // 1 at the last entry of a run of equal codes
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f[i] = (i + 1 == n || k[i + 1] != k[i]) ? 1 : 0;
}
__global__ void k_dbj_head(const uint64_t *k, uint64_t n, uint8_t *f) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f[i] = (i == 0 || k[i - 1] != k[i]) ? 1 : 0;
}
__global__ void k_dbj_gather(const uint64_t *in, const uint32_t *idx, uint64_t n, uint64_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}
// every distinct db code looked up among the distinct loc codes (both ascending): the row that wins and its location id
__global__ void k_dbj_join(const uint64_t *keys, uint64_t nk, const uint64_t *lcode, const uint32_t *lrow, uint64_t nl, const uint32_t *row_loc,
                           uint32_t *key_loc, uint32_t *key_row) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nk) return;
    const uint64_t x = keys[i];
    uint64_t lo = 0, hi = nl;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (lcode[mid] < x) lo = mid + 1;
        else hi = mid;
    }
    if (lo < nl && lcode[lo] == x) {
        key_row[i] = lrow[lo];
        key_loc[i] = row_loc[lrow[lo]];
    } else {
        key_row[i] = GS_MISSING;  // a db k-mer without loc row: the reference's KeyError when a read meets it (kmerpos_annot3.nim:90)
        key_loc[i] = GS_MISSING;
    }
}
inline unsigned dbj_blocks(uint64_t n) { return (unsigned)std::max<uint64_t>((n + 255) / 256, 1); }
}  // namespace

int gs::db_join_device(int device, const gs_sunkdb *db, uint64_t **d_keys_out, uint32_t **d_loc_out, uint32_t **d_payload_out, uint64_t *n_keys) {
    *d_keys_out = nullptr;
    *d_loc_out = *d_payload_out = nullptr;
    *n_keys = 0;
    const uint64_t nd = db->db_codes.size(), nr = db->row_code.size();
    if (nd > 0xFFFFFFF0ull || nr > 0xFFFFFFF0ull) {
        set_error("gs_table_create_from_db: more than 2^32 rows");
        return GS_E_ARG;
    }
    HIP_TRY(hipSetDevice(device));
    if (!nd) return GS_OK;
    const int bits = std::min(64, 2 * std::max(db->k, 1));
    Bufs keep;  // what outlives the steps
    unsigned long long *d_cnt = nullptr;
    GS_TRY(keep.get(&d_cnt, 1));
    // ---- loc side: distinct codes (ascending), each with its LAST row ----
    uint64_t *d_lcode = nullptr;
    uint32_t *d_lrow = nullptr, *d_rowloc = nullptr;
    uint64_t nl = 0;
    GS_TRY(keep.get(&d_rowloc, nr));
    if (nr) {
        HIP_TRY(hipMemcpy(d_rowloc, db->row_loc.data(), nr * sizeof(uint32_t), hipMemcpyHostToDevice));
        Bufs tmp;
        uint64_t *d_rc, *d_rc2;
        uint32_t *d_ri, *d_ri2, *d_sel;
        uint8_t *d_f;
        GS_TRY(tmp.get(&d_rc, nr));
        GS_TRY(tmp.get(&d_rc2, nr));
        GS_TRY(tmp.get(&d_ri, nr));
        GS_TRY(tmp.get(&d_ri2, nr));
        GS_TRY(tmp.get(&d_f, nr));
        GS_TRY(tmp.get(&d_sel, nr));
        HIP_TRY(hipMemcpy(d_rc, db->row_code.data(), nr * sizeof(uint64_t), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_dbj_iota, dim3(dbj_blocks(nr)), dim3(256), 0, 0, d_ri, nr);
        GS_TRY(sort_pairs_u64_u32(d_rc, d_rc2, d_ri, d_ri2, nr, bits, nullptr));  // stable: rows ascend inside a run of equal codes
        hipLaunchKernelGGL(k_dbj_tail, dim3(dbj_blocks(nr)), dim3(256), 0, 0, d_rc2, nr, d_f);
        GS_TRY(select_flagged_u32(d_ri2, d_f, d_sel, d_cnt, nr, &nl, nullptr));  // the last row of every code (kmerpos_annot3.nim:68)
        GS_TRY(keep.get(&d_lcode, nl));
        GS_TRY(keep.get(&d_lrow, nl));
        HIP_TRY(hipMemcpy(d_lrow, d_sel, nl * sizeof(uint32_t), hipMemcpyDeviceToDevice));
        hipLaunchKernelGGL(k_dbj_gather, dim3(dbj_blocks(nl)), dim3(256), 0, 0, d_rc, d_lrow, nl, d_lcode);  // (ascending: tails in sorted order)
        HIP_TRY(hipDeviceSynchronize());
    }
    // ---- db side: distinct codes (a k-mer listed twice is one key: the reference's HashSet) ----
    uint64_t *d_keys = nullptr;
    uint64_t nk = 0;
    {
        Bufs tmp;
        uint64_t *d_dc, *d_dc2;
        uint32_t *d_pos, *d_sel;
        uint8_t *d_f;
        GS_TRY(tmp.get(&d_dc, nd));
        GS_TRY(tmp.get(&d_dc2, nd));
        GS_TRY(tmp.get(&d_pos, nd));
        GS_TRY(tmp.get(&d_sel, nd));
        GS_TRY(tmp.get(&d_f, nd));
        HIP_TRY(hipMemcpy(d_dc, db->db_codes.data(), nd * sizeof(uint64_t), hipMemcpyHostToDevice));
        GS_TRY(sort_keys_u64(d_dc, d_dc2, nd, nullptr));
        hipLaunchKernelGGL(k_dbj_head, dim3(dbj_blocks(nd)), dim3(256), 0, 0, d_dc2, nd, d_f);
        hipLaunchKernelGGL(k_dbj_iota, dim3(dbj_blocks(nd)), dim3(256), 0, 0, d_pos, nd);
        GS_TRY(select_flagged_u32(d_pos, d_f, d_sel, d_cnt, nd, &nk, nullptr));
        HIP_TRY(hipMalloc((void **)&d_keys, std::max<uint64_t>(nk, 1) * sizeof(uint64_t)));
        hipLaunchKernelGGL(k_dbj_gather, dim3(dbj_blocks(nk)), dim3(256), 0, 0, d_dc2, d_sel, nk, d_keys);
        if (hipDeviceSynchronize() != hipSuccess) {
            (void)hipFree(d_keys);
            set_error("gs_table_create_from_db: the key sort failed");
            return GS_E_NODEVICE;
        }
    }
    // ---- the join ----
    uint32_t *d_loc = nullptr, *d_pay = nullptr;
    hipError_t e = hipMalloc((void **)&d_loc, std::max<uint64_t>(nk, 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&d_pay, std::max<uint64_t>(nk, 1) * sizeof(uint32_t));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_dbj_join, dim3(dbj_blocks(nk)), dim3(256), 0, 0, d_keys, nk, d_lcode, d_lrow, nl, d_rowloc, d_loc, d_pay);
        e = hipDeviceSynchronize();
    }
    if (e != hipSuccess) {
        (void)hipFree(d_keys);
        (void)hipFree(d_loc);
        (void)hipFree(d_pay);
        set_error("gs_table_create_from_db: %s", hipGetErrorString(e));
        return GS_E_NODEVICE;
    }
    *d_keys_out = d_keys;
    *d_loc_out = d_loc;
    *d_payload_out = d_pay;
    *n_keys = nk;
    return GS_OK;
}

extern "C" {
// The table of a parsed <kmers.db> + <kmer.loc> pair: the join (distinct db codes, each with the last loc row that carries it) is
// made on the device (gs::db_join_device: two radix sorts and a merge) and feeds the build directly -- what the reference does
// with 1e8 .. 1e9 HashSet / Table inserts per run (kmerpos_annot3.nim:20-72).  A handle whose host join exists already (somebody
// asked for gs_sunkdb_keys) is built from that.
int gs_table_create_from_db(int device, const gs_sunkdb *db, gs_table **out) {
    if (!db || !out) {
        set_error("gs_table_create_from_db: null argument");
        return GS_E_ARG;
    }
    *out = nullptr;
    // the lazy host join (gs_sunkdb_keys / _key_row / _key_loc from another thread) MOVES db_codes out and frees row_code under this
    // mutex: the table is built from one consistent state of the handle, whichever it is
    std::lock_guard<std::mutex> lock(const_cast<gs_sunkdb *>(db)->join_mu);
    if (db->joined)
        return gs_table_create(device, db->k, db->keys.size(), db->keys.data(), db->key_loc.data(), db->key_row.data(), out);
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_table_create_from_db: HIP device %d not available (%d devices) -- there is no CPU fallback", device, gs_device_count());
        return GS_E_NODEVICE;
    }
    uint64_t *dk = nullptr;
    uint32_t *dl = nullptr, *dp = nullptr;
    uint64_t n = 0;
    int rc = db_join_device(device, db, &dk, &dl, &dp, &n);
    if (rc == GS_OK) rc = gs_table_create_device(device, db->k, n, dk, dl, dp, out);
    (void)hipFree(dk);
    (void)hipFree(dl);
    (void)hipFree(dp);
    return rc;
}
}  // extern "C"

struct gs_sunkset_dev {
    int device = 0, k = 0;
    uint64_t n = 0, n_groups = 0;
    uint64_t *code = nullptr;  // canonical code, kmer.loc order
    uint64_t *pos = nullptr;   // contig << 32 | start
    uint32_t *group = nullptr; // start of the merged interval
    uint32_t *loc = nullptr;   // dense id of (contig, group)
};

namespace {

__global__ void k_group_flags(const uint64_t *pos, uint64_t n, int k, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || (pos[i] >> 32) != (pos[i - 1] >> 32) || (pos[i] & 0xFFFFFFFFu) - (pos[i - 1] & 0xFFFFFFFFu) > (uint64_t)k) ? 1u : 0u;
}
// rank[i] = inclusive count of run heads up to i;  loc = rank - 1;  group = start of the run's head (looked up through
// the head's row: head_row[rank - 1], filled by the heads themselves)
__global__ void k_head_rows(const uint32_t *flag, const uint32_t *rank, const uint64_t *pos, uint64_t n, uint32_t *head_start) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) head_start[rank[i] - 1] = (uint32_t)(pos[i] & 0xFFFFFFFFu);
}
__global__ void k_group_loc(const uint32_t *rank, const uint32_t *head_start, uint64_t n, uint32_t *group, uint32_t *loc) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        loc[i] = rank[i] - 1;
        group[i] = head_start[rank[i] - 1];
    }
}

// The whole definition on device memory.  d_seq: ASCII bases of all contigs back to back (device), off: host offsets.
// Results are freshly allocated device arrays owned by `res`.
int sunks_define_core(int device, int k, uint32_t n_contigs, const uint8_t *d_seq, const uint64_t *off, gs_sunkset_dev *res) {
    res->device = device;
    res->k = k;
    const uint64_t n = n_contigs ? off[n_contigs] : 0;
    if (n == 0) return GS_OK;
    for (uint32_t c = 0; c < n_contigs; c++)
        if (off[c + 1] - off[c] > 0xFFFFFFFFull) {
            set_error("gs_sunks_define: contig %u is longer than 2^32 bases", c);
            return GS_E_ARG;
        }
    std::vector<uint64_t> chunk_off(n_contigs + 1, 0);
    uint64_t n_windows = 0;
    for (uint32_t c = 0; c < n_contigs; c++) {
        const uint64_t len = off[c + 1] - off[c];
        const uint64_t nwin = len >= (uint64_t)k ? len - k + 1 : 0;
        chunk_off[c + 1] = chunk_off[c] + (nwin + kChunk - 1) / kChunk;
        n_windows += nwin;
    }
    const uint64_t n_chunks = chunk_off[n_contigs];
    uint64_t m = 0;
    uint64_t *d_code = nullptr, *d_pos = nullptr;  // survivors of the first stage
    if (n_chunks) {
        // stage 1: window codes, sorted by code, the singly-unique ones kept -- in P passes over the code space, P = the
        // smallest power of two whose per-pass buffers (4 x 8 B + 1 B per window, + the sort's own) fit the free device memory
        // and whose windows stay below 2^32 (GAVISUNK_DEFINE_PASSES forces more: test knob)
        Bufs b;
        int rc;
        uint64_t *d_coff, *d_chunk, *d_count, *d_base, *d_cnt;
        if ((rc = b.get(&d_coff, (uint64_t)n_contigs + 1)) || (rc = b.get(&d_chunk, (uint64_t)n_contigs + 1)) ||
            (rc = b.get(&d_count, n_chunks + 1)) || (rc = b.get(&d_base, n_chunks + 1)) || (rc = b.get(&d_cnt, 1)))
            return rc;
        HIP_TRY(hipMemcpy(d_coff, off, ((uint64_t)n_contigs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_chunk, chunk_off.data(), chunk_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        uint32_t P = 1;
        while (P < 256 && ((double)n_windows / P * 1.05 > 4.0e9 || (double)n_windows / P * 36.0 * 1.05 > 0.80 * (double)free_b)) P *= 2;
        if (const char *e = getenv("GAVISUNK_DEFINE_PASSES")) {
            const int v = atoi(e);
            if (v >= 1 && v <= 256 && (v & (v - 1)) == 0) P = std::max<uint32_t>(P, (uint32_t)v);
        }
        const int end_bit = k == 32 ? 64 : 2 * k;
        const unsigned cblocks = (unsigned)((n_chunks + 127) / 128);
        size_t scan_tmp_bytes = 0;
        HIP_TRY(rocprim::exclusive_scan(nullptr, scan_tmp_bytes, d_count, d_base, (uint64_t)0, n_chunks + 1, rocprim::plus<uint64_t>()));
        void *d_scan_tmp;
        if ((rc = b.get((uint8_t **)&d_scan_tmp, scan_tmp_bytes))) return rc;
        HIP_TRY(hipMemset(d_count + n_chunks, 0, sizeof(uint64_t)));  // (the scan's last output = the pass's window count)
        std::vector<std::pair<uint64_t *, uint64_t *>> parts;  // per pass: codes, positions of its singly-unique k-mers
        std::vector<uint64_t> part_n;
        auto drop_parts = [&]() {
            for (auto &pp : parts) {
                (void)hipFree(pp.first);
                (void)hipFree(pp.second);
            }
        };
        for (uint32_t part = 0; part < P; part++) {
            hipLaunchKernelGGL((k_window_codes<false>), dim3(cblocks), dim3(128), 0, 0, d_seq, d_coff, d_chunk, n_contigs, k, P - 1u, part,
                               d_count, (const uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr);
            size_t stb = scan_tmp_bytes;
            hipError_t e = rocprim::exclusive_scan(d_scan_tmp, stb, d_count, d_base, (uint64_t)0, n_chunks + 1, rocprim::plus<uint64_t>());
            uint64_t np = 0;
            if (e == hipSuccess) e = hipMemcpy(&np, d_base + n_chunks, sizeof np, hipMemcpyDeviceToHost);
            if (e != hipSuccess) {
                drop_parts();
                set_error("gs_sunks_define: counting pass failed: %s", hipGetErrorString(e));
                return GS_E_NODEVICE;
            }
            if (!np) continue;
            Bufs pb;  // this pass's buffers
            uint8_t *d_flag;
            uint64_t *d_k0, *d_v0, *d_k1, *d_v1;
            if ((rc = pb.get(&d_k0, np)) || (rc = pb.get(&d_v0, np)) || (rc = pb.get(&d_k1, np)) || (rc = pb.get(&d_v1, np)) ||
                (rc = pb.get(&d_flag, np))) {
                drop_parts();
                return rc;
            }
            hipLaunchKernelGGL((k_window_codes<true>), dim3(cblocks), dim3(128), 0, 0, d_seq, d_coff, d_chunk, n_contigs, k, P - 1u, part,
                               (uint64_t *)nullptr, (const uint64_t *)d_base, d_k0, d_v0);
            size_t tmp_bytes = 0, tb2 = 0;
            e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_k0, d_k1, d_v0, d_v1, np, 0u, (unsigned)end_bit);
            if (e == hipSuccess) e = rocprim::select(nullptr, tb2, d_v1, d_flag, d_v0, d_cnt, np);
            tmp_bytes = std::max(tmp_bytes, tb2);
            void *d_tmp = nullptr;
            if (e == hipSuccess && (rc = pb.get((uint8_t **)&d_tmp, tmp_bytes))) {
                drop_parts();
                return rc;
            }
            size_t tbb = tmp_bytes;
            if (e == hipSuccess) e = rocprim::radix_sort_pairs(d_tmp, tbb, d_k0, d_k1, d_v0, d_v1, np, 0u, (unsigned)end_bit);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_mark_unique, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, 0, d_k1, np, d_flag);
                tbb = tmp_bytes;  // singly-unique: positions -> v0, codes -> k0
                e = rocprim::select(d_tmp, tbb, d_v1, d_flag, d_v0, d_cnt, np);
            }
            tbb = tmp_bytes;
            if (e == hipSuccess) e = rocprim::select(d_tmp, tbb, d_k1, d_flag, d_k0, d_cnt, np);
            uint64_t mp = 0;
            if (e == hipSuccess) e = hipMemcpy(&mp, d_cnt, sizeof mp, hipMemcpyDeviceToHost);
            uint64_t *pc = nullptr, *pp = nullptr;
            if (e == hipSuccess && mp) {
                e = hipMalloc((void **)&pc, mp * sizeof(uint64_t));
                if (e == hipSuccess) e = hipMalloc((void **)&pp, mp * sizeof(uint64_t));
                if (e == hipSuccess) e = hipMemcpy(pc, d_k0, mp * sizeof(uint64_t), hipMemcpyDeviceToDevice);
                if (e == hipSuccess) e = hipMemcpy(pp, d_v0, mp * sizeof(uint64_t), hipMemcpyDeviceToDevice);
                if (e != hipSuccess) {
                    (void)hipFree(pc);
                    (void)hipFree(pp);
                }
            }
            if (e != hipSuccess) {
                drop_parts();
                set_error("gs_sunks_define: pass %u of %u failed: %s", part + 1, P, hipGetErrorString(e));
                return e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_NODEVICE;
            }
            if (mp) {
                parts.push_back({pc, pp});
                part_n.push_back(mp);
                m += mp;
            }
        }
        if (m > 0xFFFFFFF0ull) {
            drop_parts();
            set_error("gs_sunks_define: %llu singly-unique k-mers (the table and kmer.loc rows are indexed with 32 bits)", (unsigned long long)m);
            return GS_E_ARG;
        }
        if (parts.size() == 1) {
            d_code = parts[0].first;
            d_pos = parts[0].second;
        } else if (m) {
            hipError_t e = hipMalloc((void **)&d_code, m * sizeof(uint64_t));
            if (e == hipSuccess) e = hipMalloc((void **)&d_pos, m * sizeof(uint64_t));
            uint64_t at = 0;
            for (size_t i = 0; i < parts.size() && e == hipSuccess; i++) {
                e = hipMemcpy(d_code + at, parts[i].first, part_n[i] * sizeof(uint64_t), hipMemcpyDeviceToDevice);
                if (e == hipSuccess) e = hipMemcpy(d_pos + at, parts[i].second, part_n[i] * sizeof(uint64_t), hipMemcpyDeviceToDevice);
                at += part_n[i];
            }
            drop_parts();
            if (e != hipSuccess) {
                (void)hipFree(d_code);
                (void)hipFree(d_pos);
                set_error("gs_sunks_define: out of device memory");
                return GS_E_NOMEM;
            }
        }
    }
    res->n = m;
    if (!m) return GS_OK;
    Bufs b;  // stage 2: kmer.loc order = by (contig, pos); groups
    b.p.push_back(d_code);
    b.p.push_back(d_pos);
    int rc;
    uint32_t *d_flag32, *d_rank, *d_head;
    void *d_tmp;
    HIP_TRY(hipMalloc((void **)&res->code, m * sizeof(uint64_t)));
    HIP_TRY(hipMalloc((void **)&res->pos, m * sizeof(uint64_t)));
    size_t t1 = 0, t2 = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, t1, d_pos, res->pos, d_code, res->code, m, 0u, 64u));
    HIP_TRY(rocprim::inclusive_scan(nullptr, t2, (uint32_t *)nullptr, (uint32_t *)nullptr, m, rocprim::plus<uint32_t>()));
    size_t tmp_bytes = std::max(t1, t2);
    if ((rc = b.get((uint8_t **)&d_tmp, tmp_bytes))) return rc;
    size_t tbb = tmp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(d_tmp, tbb, d_pos, res->pos, d_code, res->code, m, 0u, 64u));
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(d_code);
    (void)hipFree(d_pos);
    b.p.erase(b.p.begin(), b.p.begin() + 2);
    if ((rc = b.get(&d_flag32, m)) || (rc = b.get(&d_rank, m))) return rc;
    HIP_TRY(hipMalloc((void **)&res->group, m * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&res->loc, m * sizeof(uint32_t)));
    const unsigned gb = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL(k_group_flags, dim3(gb), dim3(256), 0, 0, res->pos, m, k, d_flag32);
    tbb = tmp_bytes;
    HIP_TRY(rocprim::inclusive_scan(d_tmp, tbb, d_flag32, d_rank, m, rocprim::plus<uint32_t>()));
    uint32_t ng = 0;
    HIP_TRY(hipMemcpy(&ng, d_rank + (m - 1), sizeof ng, hipMemcpyDeviceToHost));
    res->n_groups = ng;
    if ((rc = b.get(&d_head, ng))) return rc;
    hipLaunchKernelGGL(k_head_rows, dim3(gb), dim3(256), 0, 0, d_flag32, d_rank, res->pos, m, d_head);
    hipLaunchKernelGGL(k_group_loc, dim3(gb), dim3(256), 0, 0, d_rank, d_head, m, res->group, res->loc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return GS_OK;
}

int check_define_args(const char *who, int device, int k, uint32_t n_contigs, const void *seq, const uint64_t *off, void *out) {
    if (!out || k < 1 || k > 32 || (n_contigs && (!seq || !off))) {
        set_error("%s: bad argument (k must be 1..32)", who);
        return GS_E_ARG;
    }
    if (gs_device_count() <= device || device < 0) {
        set_error("%s: HIP device %d not available -- there is no CPU fallback", who, device);
        return GS_E_NODEVICE;
    }
    return GS_OK;
}

}  // namespace

int gs::select_flagged_u32(const uint32_t *in, const uint8_t *flag, uint32_t *out, unsigned long long *d_count, uint64_t n,
                           uint64_t *n_out, void *stream) {
    *n_out = 0;
    if (!n) return GS_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    size_t tb = 0;
    HIP_TRY(rocprim::select(nullptr, tb, in, flag, out, d_count, n, st));
    void *tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
    hipError_t e = rocprim::select(tmp, tb, in, flag, out, d_count, n, st);
    unsigned long long cnt = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&cnt, d_count, sizeof cnt, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (e != hipSuccess) {
        set_error("select failed: %s", hipGetErrorString(e));
        return GS_E_NODEVICE;
    }
    *n_out = cnt;
    return GS_OK;
}

extern "C" {

void gs_sunkset_dev_free(gs_sunkset_dev *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipFree(s->code);
    (void)hipFree(s->pos);
    (void)hipFree(s->group);
    (void)hipFree(s->loc);
    delete s;
}
uint64_t gs_sunkset_dev_n(const gs_sunkset_dev *s) { return s ? s->n : 0; }
uint64_t gs_sunkset_dev_n_groups(const gs_sunkset_dev *s) { return s ? s->n_groups : 0; }
const void *gs_sunkset_dev_code(const gs_sunkset_dev *s) { return s ? s->code : nullptr; }
const void *gs_sunkset_dev_pos(const gs_sunkset_dev *s) { return s ? s->pos : nullptr; }
const void *gs_sunkset_dev_group(const gs_sunkset_dev *s) { return s ? s->group : nullptr; }
const void *gs_sunkset_dev_loc(const gs_sunkset_dev *s) { return s ? s->loc : nullptr; }

int gs_sunks_define_device(int device, int k, uint32_t n_contigs, const void *d_seq, const uint64_t *off, gs_sunkset_dev **out) {
    int rc = check_define_args("gs_sunks_define_device", device, k, n_contigs, d_seq, off, out);
    if (rc) return rc;
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());  // the sequence may have been written on another stream
    auto *res = new gs_sunkset_dev();
    if ((rc = sunks_define_core(device, k, n_contigs, static_cast<const uint8_t *>(d_seq), off, res))) {
        gs_sunkset_dev_free(res);
        return rc;
    }
    *out = res;
    return GS_OK;
}

// The two text files the tagger reads (workflow/rules/defineSUNKs.smk:59,126), written natively from a device-resident SUNK set:
// <db> = one canonical k-mer per line in ascending code order, <loc> = `contig \t start \t k-mer \t group` rows in (contig,
// start) order.  (The Python writer of the small-assembly path formats a row at a time: minutes at 1e8 rows.)
int gs_sunkset_dev_write_files(const gs_sunkset_dev *s, const char *const *contig_names, uint32_t n_names, const char *db_path,
                               const char *loc_path) {
    if (!s || !db_path || !loc_path || (n_names && !contig_names)) {
        set_error("gs_sunkset_dev_write_files: bad argument");
        return GS_E_ARG;
    }
    HIP_TRY(hipSetDevice(s->device));
    const uint64_t n = s->n;
    const int k = s->k;
    std::vector<uint64_t> code(n), pos(n);
    std::vector<uint32_t> grp(n);
    if (n) {
        HIP_TRY(hipMemcpy(code.data(), s->code, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(pos.data(), s->pos, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(grp.data(), s->group, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    auto put_kmer = [k](char *o, uint64_t c) {
        for (int j = 0; j < k; j++) o[j] = "ACGT"[(c >> (2 * (k - 1 - j))) & 3];
    };
    auto put_uint = [](char *o, uint64_t v) {
        char tmp[24];
        int m = 0;
        do {
            tmp[m++] = (char)('0' + v % 10);
            v /= 10;
        } while (v);
        for (int j = 0; j < m; j++) o[j] = tmp[m - 1 - j];
        return m;
    };
    {
        FILE *f = fopen(loc_path, "wb");
        if (!f) {
            set_error("cannot create %s", loc_path);
            return GS_E_IO;
        }
        std::vector<char> buf;
        buf.reserve(1 << 22);
        for (uint64_t i = 0; i < n; i++) {
            const uint32_t c = (uint32_t)(pos[i] >> 32);
            if (c >= n_names) {
                fclose(f);
                set_error("gs_sunkset_dev_write_files: contig %u has no name", c);
                return GS_E_ARG;
            }
            const char *nm = contig_names[c];
            const size_t ln = strlen(nm);
            const size_t at = buf.size();
            buf.resize(at + ln + k + 48);
            char *o = buf.data() + at;
            memcpy(o, nm, ln);
            o += ln;
            *o++ = '\t';
            o += put_uint(o, pos[i] & 0xFFFFFFFFull);
            *o++ = '\t';
            put_kmer(o, code[i]);
            o += k;
            *o++ = '\t';
            o += put_uint(o, grp[i]);
            *o++ = '\n';
            buf.resize((size_t)(o - buf.data()));
            if (buf.size() > (1u << 22) - 256) {
                fwrite(buf.data(), 1, buf.size(), f);
                buf.clear();
            }
        }
        fwrite(buf.data(), 1, buf.size(), f);
        if (fclose(f) != 0) {
            set_error("writing %s failed", loc_path);
            return GS_E_IO;
        }
    }
    std::sort(code.begin(), code.end());
    FILE *f = fopen(db_path, "wb");
    if (!f) {
        set_error("cannot create %s", db_path);
        return GS_E_IO;
    }
    std::vector<char> buf((size_t)(k + 1) * 65536);
    for (uint64_t i0 = 0; i0 < n; i0 += 65536) {
        const uint64_t m = std::min<uint64_t>(65536, n - i0);
        for (uint64_t j = 0; j < m; j++) {
            put_kmer(buf.data() + j * (k + 1), code[i0 + j]);
            buf[j * (k + 1) + k] = '\n';
        }
        fwrite(buf.data(), 1, (size_t)m * (k + 1), f);
    }
    if (fclose(f) != 0) {
        set_error("writing %s failed", db_path);
        return GS_E_IO;
    }
    return GS_OK;
}

int gs_copy_to_host(int device, void *dst, const void *d_src, uint64_t bytes) {
    if (bytes && (!dst || !d_src)) {
        set_error("gs_copy_to_host: bad argument");
        return GS_E_ARG;
    }
    HIP_TRY(hipSetDevice(device));
    if (bytes) HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return GS_OK;
}

int gs_sunks_define(int device, int k, uint32_t n_contigs, const uint8_t *seq, const uint64_t *off, gs_sunkset **out) {
    int rc = check_define_args("gs_sunks_define", device, k, n_contigs, seq, off, out);
    if (rc) return rc;
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    auto res = std::make_unique<gs_sunkset>();
    res->k = k;
    const uint64_t n = n_contigs ? off[n_contigs] : 0;
    if (n == 0) {
        *out = res.release();
        return GS_OK;
    }
    Bufs b;
    uint8_t *d_seq;
    if ((rc = b.get(&d_seq, n + 64))) return rc;
    HIP_TRY(hipMemcpy(d_seq, seq, n, hipMemcpyHostToDevice));
    gs_sunkset_dev *dv = new gs_sunkset_dev();
    std::unique_ptr<gs_sunkset_dev, void (*)(gs_sunkset_dev *)> guard(dv, gs_sunkset_dev_free);
    if ((rc = sunks_define_core(device, k, n_contigs, d_seq, off, dv))) return rc;
    const uint64_t m = dv->n;
    res->code.resize(m);
    res->contig.resize(m);
    res->start.resize(m);
    res->group.resize(m);
    if (m) {
        std::vector<uint64_t> posv(m);
        std::vector<uint32_t> grp(m);
        HIP_TRY(hipMemcpy(res->code.data(), dv->code, m * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(posv.data(), dv->pos, m * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(grp.data(), dv->group, m * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < m; i++) {
            res->contig[i] = (uint32_t)(posv[i] >> 32);
            res->start[i] = (int64_t)(posv[i] & 0xFFFFFFFFu);
            res->group[i] = (int64_t)grp[i];
        }
    }
    *out = res.release();
    return GS_OK;
}

void gs_sunkset_free(gs_sunkset *s) { delete s; }
uint64_t gs_sunkset_n(const gs_sunkset *s) { return s->code.size(); }
const uint64_t *gs_sunkset_code(const gs_sunkset *s) { return s->code.data(); }
const uint32_t *gs_sunkset_contig(const gs_sunkset *s) { return s->contig.data(); }
const int64_t *gs_sunkset_start(const gs_sunkset *s) { return s->start.data(); }
const int64_t *gs_sunkset_group(const gs_sunkset *s) { return s->group.data(); }

}  // extern "C"

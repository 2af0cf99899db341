// Inter-SUNK distance test of GAVISUNK (workflow/scripts/process-by-contig_lowmem_AR.py:135-200),
// per read, on the GPU.
//
// Input per read: its rows (pos on the read, start on the contig, SUNK group ID), already filtered,
// sorted by start and de-duplicated by the caller (lines 50-108 of the script: file-level pandas work).
//   k_pair_count   one workgroup per read: all pairs i<j in fp64 exactly as scipy's pdist + numpy do it
//                  (ratio = |dpos| / |dstart|, keep unless ratio >= 1.1 or ratio <= 0.9, so inf is out and
//                  NaN stays, :140-147), majority orientation among kept pairs (tie -> 0, :150-152),
//                  and the number of surviving pairs per row i (:153).
//   k_pair_edges   writes the surviving pairs in (i, j) order = the order of `sub2`'s rows (:157-158).
//   k_pair_graph   one thread per read: multi-position clean-up (:160-181), union-find over the IDs in
//                  first-appearance order, largest component with the lowest label on ties, its IDs in
//                  vertex order (:189-198).  Serial per read, parallel over reads: the edge lists are small.
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

__device__ __forceinline__ bool pair_kept(int64_t pi, int64_t pj, int64_t si, int64_t sj) {
    const double pd = fabs((double)pi - (double)pj), sd = fabs((double)si - (double)sj);
    const double ratio = pd / sd;  // IEEE: x/0 = inf, 0/0 = NaN
    return !((ratio >= 1.1) || (ratio <= 0.9));
}

constexpr int kPairThreads = 256;

__global__ __launch_bounds__(kPairThreads) void k_pair_count(const uint64_t *read_off, const int64_t *pos, const int64_t *start,
                                                             uint32_t *row_edges, uint8_t *orient, uint8_t *any_kept) {
    __shared__ unsigned long long s_ones, s_zeros;
    const uint32_t r = blockIdx.x;
    const uint64_t lo = read_off[r];
    const int n = (int)(read_off[r + 1] - lo);
    if (threadIdx.x == 0) s_ones = s_zeros = 0;
    __syncthreads();
    unsigned long long ones = 0, zeros = 0;
    for (int i = threadIdx.x; i < n; i += kPairThreads)
        for (int j = i + 1; j < n; j++)
            if (pair_kept(pos[lo + i], pos[lo + j], start[lo + i], start[lo + j])) {
                if (pos[lo + i] > pos[lo + j]) ones++;
                else zeros++;
            }
    atomicAdd(&s_ones, ones);
    atomicAdd(&s_zeros, zeros);
    __syncthreads();
    const bool o = s_ones > s_zeros;  // np.unique sorts [0, 1]; argmax takes the first maximum -> tie gives 0
    if (threadIdx.x == 0) {
        orient[r] = o ? 1 : 0;
        any_kept[r] = (s_ones + s_zeros) > 0 ? 1 : 0;
    }
    for (int i = threadIdx.x; i < n; i += kPairThreads) {
        uint32_t c = 0;
        for (int j = i + 1; j < n; j++)
            if (pair_kept(pos[lo + i], pos[lo + j], start[lo + i], start[lo + j]) && ((pos[lo + i] > pos[lo + j]) == o)) c++;
        row_edges[lo + i] = c;
    }
}

__global__ __launch_bounds__(kPairThreads) void k_pair_edges(const uint64_t *read_off, const int64_t *pos, const int64_t *start,
                                                             const uint8_t *orient, const uint64_t *row_edge_off,
                                                             uint32_t *edge_i, uint32_t *edge_j) {
    const uint32_t r = blockIdx.x;
    const uint64_t lo = read_off[r];
    const int n = (int)(read_off[r + 1] - lo);
    const bool o = orient[r] != 0;
    for (int i = threadIdx.x; i < n; i += kPairThreads) {
        uint64_t w = row_edge_off[lo + i];
        for (int j = i + 1; j < n; j++)
            if (pair_kept(pos[lo + i], pos[lo + j], start[lo + i], start[lo + j]) && ((pos[lo + i] > pos[lo + j]) == o)) {
                edge_i[w] = (uint32_t)i;
                edge_j[w] = (uint32_t)j;
                w++;
            }
    }
}

__device__ uint32_t uf_find(uint32_t *parent, uint32_t x) {
    while (parent[x] != x) {
        parent[x] = parent[parent[x]];
        x = parent[x];
    }
    return x;
}

// scratch per row (all uint32): deg, rank, vert, parent, size
__global__ void k_pair_graph(uint32_t n_reads, const uint64_t *read_off, const int64_t *pos, const int64_t *id,
                             const uint8_t *any_kept, const uint64_t *row_edge_off, const uint32_t *edge_i,
                             const uint32_t *edge_j, uint8_t *edge_bad, uint32_t *deg, uint32_t *rank, uint32_t *vert,
                             uint32_t *parent, uint32_t *csize, int64_t *out_ids, uint32_t *out_n) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t lo = read_off[r];
    const int n = (int)(read_off[r + 1] - lo);
    out_n[r] = 0;
    if (n < 2 || !any_kept[r]) return;  // `if sum(mask) < 1: continue` (:148)
    const uint64_t e0 = row_edge_off[lo], e1 = row_edge_off[lo + n];
    const uint64_t E = e1 - e0;
    const int64_t *P = pos + lo, *ID = id + lo;
    uint32_t *dg = deg + lo, *rk = rank + lo, *vt = vert + lo, *pa = parent + lo, *cs = csize + lo;
    // mergedf = [first endpoints in edge order] ++ [second endpoints in edge order] (:160-163):
    // per row: how often it occurs (deg) and where it first occurs (rank)
    for (int i = 0; i < n; i++) {
        dg[i] = 0;
        rk[i] = 0xFFFFFFFFu;
    }
    for (uint64_t e = 0; e < E; e++) {
        const uint32_t a = edge_i[e0 + e], b = edge_j[e0 + e];
        dg[a]++;
        dg[b]++;
        rk[a] = min(rk[a], (uint32_t)e);
        rk[b] = min(rk[b], (uint32_t)(E + e));
        edge_bad[e0 + e] = 0;
    }
    // IDs seen with more than one pos keep their most frequent pos (value_counts().index[0]; ties: the pos that
    // appears first in mergedf), and lose the pairs that start at that ID and touch the good pos on neither side
    for (int a = 0; a < n; a++) {
        if (!dg[a]) continue;
        bool first_of_id = true, multi = false;
        for (int b = 0; b < n; b++) {
            if (b == a || !dg[b] || ID[b] != ID[a]) continue;
            if (b < a) first_of_id = false;
            if (P[b] != P[a]) multi = true;
        }
        if (!first_of_id || !multi) continue;
        int64_t goodpos = 0;
        uint64_t best_cnt = 0;
        uint32_t best_rank = 0xFFFFFFFFu;
        for (int b = 0; b < n; b++) {
            if (!dg[b] || ID[b] != ID[a]) continue;
            bool first_of_pos = true;
            uint64_t cnt = 0;
            uint32_t rnk = 0xFFFFFFFFu;
            for (int c = 0; c < n; c++) {
                if (!dg[c] || ID[c] != ID[a] || P[c] != P[b]) continue;
                if (c < b) first_of_pos = false;
                cnt += dg[c];
                rnk = min(rnk, rk[c]);
            }
            if (!first_of_pos) continue;
            if (cnt > best_cnt || (cnt == best_cnt && rnk < best_rank)) {
                best_cnt = cnt;
                best_rank = rnk;
                goodpos = P[b];
            }
        }
        for (uint64_t e = 0; e < E; e++) {
            const uint32_t x = edge_i[e0 + e], y = edge_j[e0 + e];
            if (ID[x] == ID[a] && !(P[x] == goodpos || P[y] == goodpos)) edge_bad[e0 + e] = 1;  // :175-176
        }
    }
    // graph over ID values, vertices numbered by first appearance (source, then target, edge by edge)
    uint32_t nv = 0;
    auto vertex_of = [&](uint32_t row) -> uint32_t {
        for (uint32_t v = 0; v < nv; v++)
            if (ID[vt[v]] == ID[row]) return v;
        vt[nv] = row;
        pa[nv] = nv;
        return nv++;
    };
    for (uint64_t e = 0; e < E; e++) {
        if (edge_bad[e0 + e]) continue;
        const uint32_t va = vertex_of(edge_i[e0 + e]);
        const uint32_t vb = vertex_of(edge_j[e0 + e]);
        const uint32_t ra = uf_find(pa, va), rb = uf_find(pa, vb);
        if (ra != rb) pa[max(ra, rb)] = min(ra, rb);  // the root is the component's lowest vertex = its label order
    }
    if (!nv) return;
    for (uint32_t v = 0; v < nv; v++) cs[v] = 0;
    for (uint32_t v = 0; v < nv; v++) cs[uf_find(pa, v)]++;
    uint32_t best_root = 0, best = 0;
    for (uint32_t v = 0; v < nv; v++)
        if (pa[v] == v && cs[v] > best) {  // strict: the first (lowest-label) maximum wins
            best = cs[v];
            best_root = v;
        }
    uint32_t m = 0;
    for (uint32_t v = 0; v < nv; v++)
        if (uf_find(pa, v) == best_root) out_ids[lo + m++] = ID[vt[v]];
    out_n[r] = m;
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

extern "C" int gs_pair_component(int device, uint32_t n_reads, const uint64_t *read_off, const int64_t *pos,
                                 const int64_t *start, const int64_t *id, uint32_t *out_n, int64_t *out_ids) {
    if (!read_off || !out_n || (n_reads && read_off[n_reads] && (!pos || !start || !id || !out_ids))) {
        set_error("gs_pair_component: null argument");
        return GS_E_ARG;
    }
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_pair_component: HIP device %d not available -- there is no CPU fallback", device);
        return GS_E_NODEVICE;
    }
    if (!n_reads) return GS_OK;
    HIP_TRY(hipSetDevice(device));
    const uint64_t n_rows = read_off[n_reads];
    DevBuf db;
    int rc;
    uint64_t *d_off, *d_reoff;
    int64_t *d_pos, *d_start, *d_id, *d_out;
    uint32_t *d_rowe, *d_outn, *d_deg, *d_rank, *d_vert, *d_parent, *d_size;
    uint8_t *d_orient, *d_any;
    if ((rc = db.put(&d_off, read_off, (uint64_t)n_reads + 1)) || (rc = db.put(&d_pos, pos, n_rows)) ||
        (rc = db.put(&d_start, start, n_rows)) || (rc = db.put(&d_id, id, n_rows)) ||
        (rc = db.put(&d_rowe, (const uint32_t *)nullptr, n_rows + 1)) || (rc = db.put(&d_orient, (const uint8_t *)nullptr, n_reads)) ||
        (rc = db.put(&d_any, (const uint8_t *)nullptr, n_reads)))
        return rc;
    hipLaunchKernelGGL(k_pair_count, dim3(n_reads), dim3(kPairThreads), 0, 0, d_off, d_pos, d_start, d_rowe, d_orient, d_any);
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> rowe(n_rows + 1, 0);
    if (n_rows) HIP_TRY(hipMemcpy(rowe.data(), d_rowe, n_rows * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint64_t> reoff(n_rows + 1, 0);
    for (uint64_t i = 0; i < n_rows; i++) reoff[i + 1] = reoff[i] + rowe[i];
    const uint64_t n_edges = reoff[n_rows];
    uint32_t *d_ei, *d_ej;
    uint8_t *d_bad;
    if ((rc = db.put(&d_reoff, reoff.data(), n_rows + 1)) || (rc = db.put(&d_ei, (const uint32_t *)nullptr, n_edges)) ||
        (rc = db.put(&d_ej, (const uint32_t *)nullptr, n_edges)) || (rc = db.put(&d_bad, (const uint8_t *)nullptr, n_edges)) ||
        (rc = db.put(&d_deg, (const uint32_t *)nullptr, n_rows)) || (rc = db.put(&d_rank, (const uint32_t *)nullptr, n_rows)) ||
        (rc = db.put(&d_vert, (const uint32_t *)nullptr, n_rows)) || (rc = db.put(&d_parent, (const uint32_t *)nullptr, n_rows)) ||
        (rc = db.put(&d_size, (const uint32_t *)nullptr, n_rows)) || (rc = db.put(&d_out, (const int64_t *)nullptr, n_rows)) ||
        (rc = db.put(&d_outn, (const uint32_t *)nullptr, n_reads)))
        return rc;
    hipLaunchKernelGGL(k_pair_edges, dim3(n_reads), dim3(kPairThreads), 0, 0, d_off, d_pos, d_start, d_orient, d_reoff, d_ei, d_ej);
    hipLaunchKernelGGL(k_pair_graph, dim3((n_reads + 63) / 64), dim3(64), 0, 0, n_reads, d_off, d_pos, d_id, d_any, d_reoff,
                       d_ei, d_ej, d_bad, d_deg, d_rank, d_vert, d_parent, d_size, d_out, d_outn);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_n, d_outn, n_reads * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (n_rows) HIP_TRY(hipMemcpy(out_ids, d_out, n_rows * sizeof(int64_t), hipMemcpyDeviceToHost));
    return GS_OK;
}

// ------------------------------------------------------------------------------------------
// Bad-SUNK histogram (workflow/scripts/badsunks_AR.py:20-51):
// reads per `chrom:ID` key, then the histogram of those counts (its arg-max is the script's "mean coverage").
// ------------------------------------------------------------------------------------------
namespace {
__global__ void k_key_count(const uint32_t *row_key, uint64_t n_rows, uint32_t *key_count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows) atomicAdd(&key_count[row_key[i]], 1u);
}
__global__ void k_count_hist(const uint32_t *key_count, const uint8_t *key_sel, uint32_t n_keys, uint32_t *hist,
                             uint32_t hist_len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_keys && key_sel[i]) atomicAdd(&hist[min(key_count[i], hist_len - 1)], 1u);
}
}  // namespace

extern "C" int gs_key_histogram(int device, uint64_t n_rows, const uint32_t *row_key, uint32_t n_keys,
                                const uint8_t *key_sel, uint32_t *key_count, uint32_t *hist, uint32_t hist_len) {
    if ((n_rows && !row_key) || !key_sel || !key_count || !hist || hist_len < 2) {
        set_error("gs_key_histogram: bad argument");
        return GS_E_ARG;
    }
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_key_histogram: HIP device %d not available -- there is no CPU fallback", device);
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    DevBuf db;
    int rc;
    uint32_t *d_rk, *d_kc, *d_h;
    uint8_t *d_sel;
    if ((rc = db.put(&d_rk, row_key, n_rows)) || (rc = db.put(&d_kc, (const uint32_t *)nullptr, n_keys)) ||
        (rc = db.put(&d_h, (const uint32_t *)nullptr, hist_len)) || (rc = db.put(&d_sel, key_sel, n_keys)))
        return rc;
    HIP_TRY(hipMemset(d_kc, 0, std::max<uint64_t>(n_keys, 1) * sizeof(uint32_t)));
    HIP_TRY(hipMemset(d_h, 0, hist_len * sizeof(uint32_t)));
    if (n_rows) hipLaunchKernelGGL(k_key_count, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, 0, d_rk, n_rows, d_kc);
    if (n_keys) hipLaunchKernelGGL(k_count_hist, dim3((n_keys + 255) / 256), dim3(256), 0, 0, d_kc, d_sel, n_keys, d_h, hist_len);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (n_keys) HIP_TRY(hipMemcpy(key_count, d_kc, n_keys * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hist, d_h, hist_len * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return GS_OK;
}

// ------------------------------------------------------------------------------------------
// Contig-wide union of the reads' validated ID sets (process-by-contig_lowmem_AR.py:219-260, SURVEY.md section 8f
// row N3): connected components of "appears in the same read", components with more than two SUNKs become
// [min ID, max ID] intervals.  Lock-free union-find (link the larger root under the smaller with atomicCAS),
// then per-component size / min / max with atomics.
// ------------------------------------------------------------------------------------------
namespace {
__device__ uint32_t uf_root(uint32_t *parent, uint32_t x) {
    uint32_t p = parent[x];
    while (p != x) {
        const uint32_t gp = parent[p];
        if (gp != p) atomicCAS(&parent[x], p, gp);  // path halving; a stale write only shortens a valid path
        x = p;
        p = parent[x];
    }
    return x;
}
__global__ void k_uf_init(uint32_t *parent, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = i;
}
// rows of one read are consecutive; an edge joins row i to row i+1 of the same read
__global__ void k_uf_union(uint32_t *parent, const uint32_t *row_vertex, const uint32_t *row_read, uint64_t n_rows) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n_rows || row_read[i] != row_read[i + 1]) return;
    uint32_t a = uf_root(parent, row_vertex[i]), b = uf_root(parent, row_vertex[i + 1]);
    while (a != b) {
        if (a < b) {
            const uint32_t t = a;
            a = b;
            b = t;
        }
        const uint32_t old = atomicCAS(&parent[a], a, b);  // hang the larger root under the smaller
        if (old == a) break;
        a = uf_root(parent, a);
        b = uf_root(parent, b);
    }
}
__global__ void k_uf_stats(uint32_t *parent, const long long *vertex_id, uint32_t n, uint32_t *csize, long long *cmin,
                           long long *cmax) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = uf_root(parent, i);
    atomicAdd(&csize[r], 1u);
    atomicMin(&cmin[r], vertex_id[i]);
    atomicMax(&cmax[r], vertex_id[i]);
}
}  // namespace

extern "C" int gs_union_regions(int device, uint64_t n_rows, const uint32_t *row_read, const uint32_t *row_vertex,
                                uint32_t n_vertices, const int64_t *vertex_id, uint32_t *comp_size, int64_t *comp_min,
                                int64_t *comp_max) {
    if ((n_rows && (!row_read || !row_vertex)) || (n_vertices && (!vertex_id || !comp_size || !comp_min || !comp_max))) {
        set_error("gs_union_regions: null argument");
        return GS_E_ARG;
    }
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_union_regions: HIP device %d not available -- there is no CPU fallback", device);
        return GS_E_NODEVICE;
    }
    if (!n_vertices) return GS_OK;
    HIP_TRY(hipSetDevice(device));
    DevBuf db;
    int rc;
    uint32_t *d_rr, *d_rv, *d_parent, *d_size;
    long long *d_vid, *d_min, *d_max;
    if ((rc = db.put(&d_rr, row_read, n_rows)) || (rc = db.put(&d_rv, row_vertex, n_rows)) ||
        (rc = db.put(&d_parent, (const uint32_t *)nullptr, n_vertices)) || (rc = db.put(&d_size, (const uint32_t *)nullptr, n_vertices)) ||
        (rc = db.put(&d_vid, (const long long *)vertex_id, n_vertices)) || (rc = db.put(&d_min, (const long long *)nullptr, n_vertices)) ||
        (rc = db.put(&d_max, (const long long *)nullptr, n_vertices)))
        return rc;
    HIP_TRY(hipMemset(d_size, 0, n_vertices * sizeof(uint32_t)));
    HIP_TRY(hipMemset(d_min, 0x7F, n_vertices * sizeof(long long)));  // 0x7F7F...: above every ID
    HIP_TRY(hipMemset(d_max, 0x80, n_vertices * sizeof(long long)));  // 0x8080...: below every ID
    hipLaunchKernelGGL(k_uf_init, dim3((n_vertices + 255) / 256), dim3(256), 0, 0, d_parent, n_vertices);
    if (n_rows > 1)
        hipLaunchKernelGGL(k_uf_union, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, 0, d_parent, d_rv, d_rr, n_rows);
    hipLaunchKernelGGL(k_uf_stats, dim3((n_vertices + 255) / 256), dim3(256), 0, 0, d_parent, d_vid, n_vertices, d_size, d_min, d_max);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(comp_size, d_size, n_vertices * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(comp_min, d_min, n_vertices * sizeof(long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(comp_max, d_max, n_vertices * sizeof(long long), hipMemcpyDeviceToHost));
    return GS_OK;
}

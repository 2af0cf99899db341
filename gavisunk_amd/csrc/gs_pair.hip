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
//   k_pair_graph   one workgroup per read: multi-position clean-up (:160-181), lock-free union-find over the IDs in
//                  first-appearance order, largest component with the lowest label on ties, its IDs in
//                  vertex order (:189-198).  O(n^2 + E) work per read, spread over 256 threads.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gs_common.h"
#include "gs_devpool.h"

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

// The three kernels take one workgroup per read.  A read has tens of rows on a haploid genome and hundreds to thousands on a
// diploid one: the reads of a chunk are split by size (`list`), 64 threads for those of up to kSmallRead rows, 256 for the others.
constexpr int kSmallRead = 64;

// reads of at most 64 rows, one wave each: lane i holds row i; the pair test (an fp64 division per pair: the cost of this stage) is
// made ONCE -- lane i walks all j by cross-lane reads and keeps two 64-bit masks, "pair (i, j) kept" and "pos[i] > pos[j]" -- and
// orientation, edge counts and (k_pair_edges_small) the ordered edge list all come from the masks.  (The general form below tests
// every pair three times.)
__global__ __launch_bounds__(64) void k_pair_count_small(const uint32_t *list, const uint64_t *read_off, const int64_t *pos, const int64_t *start,
                                                         uint32_t *row_edges, uint8_t *orient, uint8_t *any_kept, uint64_t *edge_mask) {
    const uint32_t r = list[blockIdx.x];
    const uint64_t lo = read_off[r];
    const int n = (int)(read_off[r + 1] - lo);
    const int i = threadIdx.x;
    int64_t pi = 0, si = 0;
    if (i < n) {
        pi = pos[lo + i];
        si = start[lo + i];
    }
    uint64_t kept = 0, dir = 0;
    for (int j = 1; j < n; j++) {  // wave-uniform trip count
        const int64_t pj = __shfl(pi, j, 64), sj = __shfl(si, j, 64);
        if (i < j) {
            if (pair_kept(pi, pj, si, sj)) kept |= 1ULL << j;
            if (pi > pj) dir |= 1ULL << j;
        }
    }
    uint32_t ones = (uint32_t)__popcll(kept & dir), zeros = (uint32_t)__popcll(kept & ~dir);
    for (int o = 32; o > 0; o >>= 1) {
        ones += __shfl_xor((int)ones, o, 64);
        zeros += __shfl_xor((int)zeros, o, 64);
    }
    const bool o = ones > zeros;  // np.unique sorts [0, 1]; argmax takes the first maximum -> tie gives 0
    if (i == 0) {
        orient[r] = o ? 1 : 0;
        any_kept[r] = (ones + zeros) > 0 ? 1 : 0;
    }
    if (i < n) {
        const uint64_t m = kept & (o ? dir : ~dir);
        row_edges[lo + i] = (uint32_t)__popcll(m);
        edge_mask[lo + i] = m;
    }
}
__global__ __launch_bounds__(64) void k_pair_edges_small(const uint32_t *list, const uint64_t *read_off, const uint64_t *edge_mask,
                                                         const uint64_t *row_edge_off, uint32_t *edge_i, uint32_t *edge_j) {
    const uint32_t r = list[blockIdx.x];
    const uint64_t lo = read_off[r];
    const int n = (int)(read_off[r + 1] - lo);
    const int i = threadIdx.x;
    if (i >= n) return;
    uint64_t w = row_edge_off[lo + i], m = edge_mask[lo + i];
    while (m) {  // the surviving pairs of row i in j order = the order of `sub2`'s rows (:157-158)
        edge_i[w] = (uint32_t)i;
        edge_j[w] = (uint32_t)(__ffsll((long long)m) - 1);
        w++;
        m &= m - 1;
    }
}

template <int kPairThreads>
__global__ __launch_bounds__(kPairThreads) void k_pair_count(const uint32_t *list, const uint64_t *read_off, const int64_t *pos, const int64_t *start,
                                                             uint32_t *row_edges, uint8_t *orient, uint8_t *any_kept) {
    __shared__ unsigned long long s_ones, s_zeros;
    const uint32_t r = list[blockIdx.x];
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

template <int kPairThreads>
__global__ __launch_bounds__(kPairThreads) void k_pair_edges(const uint32_t *list, const uint64_t *read_off, const int64_t *pos, const int64_t *start,
                                                             const uint8_t *orient, const uint64_t *row_edge_off,
                                                             uint32_t *edge_i, uint32_t *edge_j) {
    const uint32_t r = list[blockIdx.x];
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

// lock-free union-find over vertex indices: the root of a set is its smallest index (= its earliest vertex)
__device__ __forceinline__ uint32_t uf_root(const uint32_t *parent, uint32_t x) {
    uint32_t p;
    while ((p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != x) x = p;
    return x;
}
__device__ __forceinline__ void uf_union(uint32_t *parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_root(parent, a);
        b = uf_root(parent, b);
        if (a == b) return;
        const uint32_t hi = max(a, b), lo = min(a, b);
        if (atomicCAS(parent + hi, hi, lo) == hi) return;
    }
}

// One workgroup per read; scratch per row (all uint32): deg, rank, vert, parent, size, plus app / vidx / flag below.
// n hits of a read give up to n (n - 1) / 2 edges (n ~ 10^2 .. 10^3 on diploid assemblies, more on ultra-long reads): every
// phase is spread over the workgroup, O(n^2 + E) work per read -- no phase scans the rows more than once per row.
//   deg[i]   occurrences of row i in mergedf = [first endpoints in edge order] ++ [second endpoints] (:160-163)
//   rank[i]  where it first occurs there
//   multi-position clean-up (:167-181), then the graph over ID values with vertices numbered by first appearance
//   (source, then target, edge by edge), its largest component (lowest label on ties), its IDs in vertex order (:189-198)
template <int kGraphThreads>
__global__ __launch_bounds__(kGraphThreads) void k_pair_graph(const uint32_t *list, const uint64_t *read_off, const int64_t *pos,
                                                              const int64_t *id, const uint8_t *any_kept,
                                                              const uint64_t *row_edge_off, const uint32_t *edge_i,
                                                              const uint32_t *edge_j, uint8_t *edge_bad, uint32_t *deg,
                                                              uint32_t *rank, uint32_t *vert, uint32_t *parent, uint32_t *csize,
                                                              uint32_t *app, int64_t *goodpos_of, int64_t *out_ids,
                                                              uint32_t *out_n) {
    __shared__ uint32_t s_nv;
    const uint32_t r = list[blockIdx.x];
    const int tid = threadIdx.x;
    const uint64_t lo = read_off[r];
    const int n = (int)(read_off[r + 1] - lo);
    if (tid == 0) out_n[r] = 0;
    if (n < 2 || !any_kept[r]) return;  // `if sum(mask) < 1: continue` (:148)
    const uint64_t e0 = row_edge_off[lo], e1 = row_edge_off[lo + n];
    const uint64_t E = e1 - e0;
    // The per-row arrays: global scratch for the 256-thread form (reads of any length); LDS for the 64-thread form, whose reads have at
    // most kSmallRead rows -- its phases are O(n^2) loops of DEPENDENT loads over these arrays, bound by their latency (3.3 ms for
    // 3.6e5 reads of ~18 rows with them in global memory).  (Surviving pairs stay in global memory: up to n (n - 1) / 2 per read.)
    __shared__ int64_t s_i64[kGraphThreads == 64 ? 3 : 1][kGraphThreads == 64 ? kSmallRead : 1];
    __shared__ uint32_t s_u32[kGraphThreads == 64 ? 6 : 1][kGraphThreads == 64 ? kSmallRead : 1];
    const int64_t *P = pos + lo, *ID = id + lo;
    uint32_t *dg = deg + lo, *rk = rank + lo, *vt = vert + lo, *pa = parent + lo, *cs = csize + lo, *ap = app + lo;
    int64_t *gp = goodpos_of + lo;
    if constexpr (kGraphThreads == 64) {
        for (int i = tid; i < n; i += kGraphThreads) {
            s_i64[0][i] = P[i];
            s_i64[1][i] = ID[i];
        }
        P = s_i64[0];
        ID = s_i64[1];
        gp = s_i64[2];
        dg = s_u32[0];
        rk = s_u32[1];
        vt = s_u32[2];
        pa = s_u32[3];
        cs = s_u32[4];
        ap = s_u32[5];
        __syncthreads();
    }
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    for (int i = tid; i < n; i += kGraphThreads) {
        dg[i] = 0;
        rk[i] = kNone;
        ap[i] = kNone;
        cs[i] = 0;
    }
    __syncthreads();
    for (uint64_t e = tid; e < E; e += kGraphThreads) {
        const uint32_t a = edge_i[e0 + e], b = edge_j[e0 + e];
        atomicAdd(dg + a, 1u);
        atomicAdd(dg + b, 1u);
        atomicMin(rk + a, (uint32_t)e);
        atomicMin(rk + b, (uint32_t)(E + e));
        edge_bad[e0 + e] = 0;
    }
    __syncthreads();
    // IDs seen with more than one pos keep their most frequent pos (value_counts().index[0]; ties: the pos that appears
    // first in mergedf); gp[row] = that pos for rows of such an ID, else "no constraint" (vt[row] = 0 / 1 marks which).
    // Two passes of one scan over the read's rows each -- O(n^2) per read over the workgroup, like the pair enumeration:
    //   A  per row: does its ID occur with another pos; occurrences of ITS (ID, pos) in mergedf (cs), where that pair first
    //      occurs there (pa), and whether the row is the first one of its (ID, pos) (ap)
    //   B  per row of a multi-position ID: the (ID, pos) with the most occurrences, the earliest on ties
    for (int a = tid; a < n; a += kGraphThreads) {
        vt[a] = 0;
        cs[a] = 0;
        pa[a] = kNone;
        ap[a] = 0;
        if (!dg[a]) continue;
        bool multi = false, first_of_pos = true;
        uint32_t cnt = 0, rnk = kNone;
        const int64_t ida = ID[a], pa_ = P[a];
        for (int b = 0; b < n; b++) {
            if (!dg[b] || ID[b] != ida) continue;
            if (P[b] != pa_) {
                multi = true;
                continue;
            }
            cnt += dg[b];
            rnk = min(rnk, rk[b]);
            if (b < a) first_of_pos = false;
        }
        vt[a] = multi ? 1u : 0u;
        cs[a] = cnt;
        pa[a] = rnk;
        ap[a] = first_of_pos ? 1u : 0u;
    }
    __syncthreads();
    for (int a = tid; a < n; a += kGraphThreads) {
        if (!vt[a]) continue;
        int64_t goodpos = 0;
        uint32_t best_cnt = 0, best_rank = kNone;
        const int64_t ida = ID[a];
        for (int b = 0; b < n; b++) {
            if (!ap[b] || ID[b] != ida) continue;  // (ap is set only on rows that occur in mergedf)
            if (cs[b] > best_cnt || (cs[b] == best_cnt && pa[b] < best_rank)) {
                best_cnt = cs[b];
                best_rank = pa[b];
                goodpos = P[b];
            }
        }
        gp[a] = goodpos;
    }
    __syncthreads();
    for (int a = tid; a < n; a += kGraphThreads) {  // (the scratch goes back to what the next phases expect)
        ap[a] = kNone;
        cs[a] = 0;
    }
    __syncthreads();
    // ... and lose the pairs that start at that ID and touch the good pos on neither side (:175-176); the first appearance of
    // every row among the surviving pairs (source of pair e: 2 e, target: 2 e + 1)
    for (uint64_t e = tid; e < E; e += kGraphThreads) {
        const uint32_t x = edge_i[e0 + e], y = edge_j[e0 + e];
        const bool bad = vt[x] && !(P[x] == gp[x] || P[y] == gp[x]);
        edge_bad[e0 + e] = bad ? 1 : 0;
        if (!bad) {
            atomicMin(ap + x, (uint32_t)(2 * e));
            atomicMin(ap + y, (uint32_t)(2 * e + 1));
        }
    }
    __syncthreads();
    // a vertex = an ID value; its first appearance = the earliest of its rows'; its leader = the row that holds it
    for (int a = tid; a < n; a += kGraphThreads) {
        uint32_t mine = ap[a];
        bool leader = mine != kNone;
        for (int b = 0; b < n && leader; b++)
            if (b != a && ID[b] == ID[a] && (ap[b] < mine || (ap[b] == mine && b < a))) leader = false;
        rk[a] = leader ? 1u : 0u;  // (rank is free now)
    }
    __syncthreads();
    if (tid == 0) s_nv = 0;
    __syncthreads();
    // vertex index = number of leaders that appear earlier; vt[v] = leader row of vertex v; dg[row] = vertex of the row's ID
    for (int a = tid; a < n; a += kGraphThreads) {
        if (!rk[a]) continue;
        uint32_t v = 0;
        for (int b = 0; b < n; b++)
            if (rk[b] && ap[b] < ap[a]) v++;
        cs[a] = v;  // stash
        atomicAdd(&s_nv, 1u);
    }
    __syncthreads();
    const uint32_t nv = s_nv;
    if (!nv) return;
    for (int a = tid; a < n; a += kGraphThreads) vt[a] = 0;
    __syncthreads();
    for (int a = tid; a < n; a += kGraphThreads)
        if (rk[a]) {
            vt[cs[a]] = (uint32_t)a;
            pa[cs[a]] = cs[a];
        }
    __syncthreads();
    for (int a = tid; a < n; a += kGraphThreads) {  // vertex of every row that appears in a surviving pair
        dg[a] = kNone;
        if (ap[a] == kNone) continue;
        for (uint32_t v = 0; v < nv; v++)
            if (ID[vt[v]] == ID[a]) {
                dg[a] = v;
                break;
            }
    }
    __syncthreads();
    for (int a = tid; a < n; a += kGraphThreads) cs[a] = 0;
    for (uint64_t e = tid; e < E; e += kGraphThreads)
        if (!edge_bad[e0 + e]) uf_union(pa, dg[edge_i[e0 + e]], dg[edge_j[e0 + e]]);
    __syncthreads();
    for (uint32_t v = tid; v < nv; v += kGraphThreads) atomicAdd(cs + uf_root(pa, v), 1u);
    __syncthreads();
    if (tid == 0) {  // the first (lowest-label) maximum wins
        uint32_t best_root = 0, best = 0;
        for (uint32_t v = 0; v < nv; v++)
            if (pa[v] == v && cs[v] > best) {
                best = cs[v];
                best_root = v;
            }
        uint32_t m = 0;
        for (uint32_t v = 0; v < nv; v++)
            if (uf_root(pa, v) == best_root) out_ids[lo + m++] = ID[vt[v]];
        out_n[r] = m;
    }
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

namespace {
__global__ void k_widen_u32(const uint32_t *in, uint64_t *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
}  // namespace

// The three kernels over arrays that already are in device memory (gs_chain.hip: the device-resident second half of the
// engine; gs_pair_component below: the host-array form).  d_out_n[n_reads], d_out_ids[n_rows]; synchronous.  The reads are
// taken in chunks whose pair lists (at most n (n - 1) / 2 surviving pairs per read of n rows) stay below kMaxChunkPairs: a
// genome-scale hit list has ~1e10 pairs in all, a chunk's edge arrays a few GB.
int gs::pair_component_device(int device, uint32_t n_reads, uint64_t n_rows, const uint64_t *d_off, const int64_t *d_pos, const int64_t *d_start,
                              const int64_t *d_id, uint32_t *d_outn, int64_t *d_out) {
    if (!n_reads) return GS_OK;
    uint64_t kMaxChunkPairs = 400000000ULL;
    if (const char *e = getenv("GAVISUNK_PAIR_CHUNK")) kMaxChunkPairs = (uint64_t)std::max(1, atoi(e));  // (test knob: many small chunks)
    hipStream_t st = chain_stream(device);
    std::vector<uint64_t> off((size_t)n_reads + 1);
    HIP_TRY(hipMemcpyAsync(off.data(), d_off, ((size_t)n_reads + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    PoolScope db(device);
    int rc;
    uint64_t *d_reoff, *d_rowe64, *d_scan;
    uint32_t *d_rowe, *d_deg, *d_rank, *d_vert, *d_parent, *d_size, *d_app;
    int64_t *d_gp;
    uint8_t *d_orient, *d_any;
    if ((rc = db.get(&d_rowe, n_rows + 1)) || (rc = db.get(&d_reoff, n_rows + 1)) || (rc = db.get(&d_orient, n_reads)) ||
        (rc = db.get(&d_any, n_reads)) || (rc = db.get(&d_deg, n_rows)) || (rc = db.get(&d_rank, n_rows)) || (rc = db.get(&d_vert, n_rows)) ||
        (rc = db.get(&d_parent, n_rows)) || (rc = db.get(&d_size, n_rows)) || (rc = db.get(&d_app, n_rows)) || (rc = db.get(&d_gp, n_rows)))
        return rc;
    HIP_TRY(hipMemsetAsync(d_rowe, 0, (n_rows + 1) * sizeof(uint32_t), st));
    // chunk boundaries and the largest chunk (rows, for the scan buffers)
    std::vector<uint32_t> cuts(1, 0);
    uint64_t acc = 0, max_rows = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        const uint64_t n = off[r + 1] - off[r], pairs = n * (n ? n - 1 : 0) / 2;
        if (acc && acc + pairs > kMaxChunkPairs) {
            cuts.push_back(r);
            acc = 0;
        }
        acc += pairs;
    }
    cuts.push_back(n_reads);
    for (size_t c = 0; c + 1 < cuts.size(); c++) max_rows = std::max(max_rows, off[cuts[c + 1]] - off[cuts[c]]);
    if ((rc = db.get(&d_rowe64, max_rows + 1)) || (rc = db.get(&d_scan, max_rows + 1))) return rc;
    size_t tb = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tb, d_rowe64, d_scan, (uint64_t)0, (size_t)max_rows + 1, rocprim::plus<uint64_t>(), st));
    uint8_t *d_tmp;
    uint32_t *d_list;
    uint64_t *d_emask;
    if ((rc = db.get(&d_tmp, tb)) || (rc = db.get(&d_list, n_reads)) || (rc = db.get(&d_emask, n_rows))) return rc;
    std::vector<uint32_t> lists(n_reads);
    uint32_t *d_ei = nullptr, *d_ej = nullptr;
    uint8_t *d_bad = nullptr;
    uint64_t edge_cap = 0;
    auto free_edges = [&]() {
        pool_free(device, d_ei);
        pool_free(device, d_ej);
        pool_free(device, d_bad);
        d_ei = d_ej = nullptr;
        d_bad = nullptr;
    };
    for (size_t c = 0; c + 1 < cuts.size(); c++) {
        const uint32_t r0 = cuts[c], nr = cuts[c + 1] - r0;
        const uint64_t lo = off[r0], rows = off[cuts[c + 1]] - lo;
        // the chunk's reads by size: [small ... | big ...] (absolute read indices)
        uint32_t n_small = 0;
        for (uint32_t r = r0; r < r0 + nr; r++) n_small += (off[r + 1] - off[r] <= (uint64_t)kSmallRead) ? 1u : 0u;
        {
            uint32_t a = 0, b = n_small;
            for (uint32_t r = r0; r < r0 + nr; r++) lists[(off[r + 1] - off[r] <= (uint64_t)kSmallRead) ? a++ : b++] = r;
        }
        const uint32_t n_big = nr - n_small;
        hipError_t e = hipMemcpyAsync(d_list, lists.data(), (size_t)nr * sizeof(uint32_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // (`lists` is refilled by the next chunk)
        if (e != hipSuccess) {
            free_edges();
            set_error("gs_pair_component: %s", hipGetErrorString(e));
            return GS_E_NODEVICE;
        }
        if (n_small) hipLaunchKernelGGL(k_pair_count_small, dim3(n_small), dim3(64), 0, st, d_list, d_off, d_pos, d_start, d_rowe, d_orient, d_any, d_emask);
        if (n_big) hipLaunchKernelGGL(k_pair_count<256>, dim3(n_big), dim3(256), 0, st, d_list + n_small, d_off, d_pos, d_start, d_rowe, d_orient, d_any);
        // surviving pairs per row -> their offsets in the chunk's edge list (exclusive scan, on the device)
        hipLaunchKernelGGL(k_widen_u32, dim3((unsigned)((rows + 1 + 255) / 256)), dim3(256), 0, st, d_rowe + lo, d_rowe64, rows);
        e = hipMemsetAsync(d_rowe64 + rows, 0, sizeof(uint64_t), st);
        size_t tb2 = tb;
        if (e == hipSuccess) e = rocprim::exclusive_scan(d_tmp, tb2, d_rowe64, d_scan, (uint64_t)0, (size_t)rows + 1, rocprim::plus<uint64_t>(), st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_reoff + lo, d_scan, (rows + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, st);
        uint64_t n_edges = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&n_edges, d_scan + rows, sizeof(uint64_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            free_edges();
            set_error("gs_pair_component: %s", hipGetErrorString(e));
            return GS_E_NODEVICE;
        }
        if (n_edges > edge_cap) {
            free_edges();
            edge_cap = std::max<uint64_t>(n_edges, 1024);
            d_ei = static_cast<uint32_t *>(pool_alloc(device, edge_cap * 4));
            d_ej = static_cast<uint32_t *>(pool_alloc(device, edge_cap * 4));
            d_bad = static_cast<uint8_t *>(pool_alloc(device, edge_cap));
            if (!d_ei || !d_ej || !d_bad) {
                free_edges();
                set_error("gs_pair_component: device allocation of %llu pairs failed", (unsigned long long)edge_cap);
                return GS_E_NOMEM;
            }
        }
        if (n_small) {
            hipLaunchKernelGGL(k_pair_edges_small, dim3(n_small), dim3(64), 0, st, d_list, d_off, d_emask, d_reoff, d_ei, d_ej);
            hipLaunchKernelGGL(k_pair_graph<64>, dim3(n_small), dim3(64), 0, st, d_list, d_off, d_pos, d_id, d_any, d_reoff, d_ei, d_ej, d_bad, d_deg,
                               d_rank, d_vert, d_parent, d_size, d_app, d_gp, d_out, d_outn);
        }
        if (n_big) {
            hipLaunchKernelGGL(k_pair_edges<256>, dim3(n_big), dim3(256), 0, st, d_list + n_small, d_off, d_pos, d_start, d_orient, d_reoff, d_ei, d_ej);
            hipLaunchKernelGGL(k_pair_graph<256>, dim3(n_big), dim3(256), 0, st, d_list + n_small, d_off, d_pos, d_id, d_any, d_reoff, d_ei, d_ej, d_bad,
                               d_deg, d_rank, d_vert, d_parent, d_size, d_app, d_gp, d_out, d_outn);
        }
        e = hipGetLastError();
        if (e == hipSuccess && c + 2 < cuts.size()) e = hipStreamSynchronize(st);  // (the next chunk may replace the edge arrays)
        if (e != hipSuccess) {
            free_edges();
            set_error("gs_pair_component: %s", hipGetErrorString(e));
            return GS_E_NODEVICE;
        }
    }
    free_edges();
    HIP_TRY(hipStreamSynchronize(st));
    return GS_OK;
}

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
    uint64_t *d_off;
    int64_t *d_pos, *d_start, *d_id, *d_out;
    uint32_t *d_outn;
    if ((rc = db.put(&d_off, read_off, (uint64_t)n_reads + 1)) || (rc = db.put(&d_pos, pos, n_rows)) ||
        (rc = db.put(&d_start, start, n_rows)) || (rc = db.put(&d_id, id, n_rows)) ||
        (rc = db.put(&d_out, (const int64_t *)nullptr, n_rows)) || (rc = db.put(&d_outn, (const uint32_t *)nullptr, n_reads)))
        return rc;
    if ((rc = gs::pair_component_device(device, n_reads, n_rows, d_off, d_pos, d_start, d_id, d_outn, d_out))) return rc;
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

// device arrays in, device arrays out (comp_* of n_vertices entries each); synchronous
int gs::union_regions_device(int device, uint64_t n_rows, const uint32_t *d_rr, const uint32_t *d_rv, uint32_t n_vertices, const int64_t *d_vid,
                             uint32_t *d_size, int64_t *d_min, int64_t *d_max) {
    if (!n_vertices) return GS_OK;
    hipStream_t st = chain_stream(device);
    PoolScope db(device);
    int rc;
    uint32_t *d_parent;
    if ((rc = db.get(&d_parent, n_vertices))) return rc;
    HIP_TRY(hipMemsetAsync(d_size, 0, n_vertices * sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(d_min, 0x7F, n_vertices * sizeof(long long), st));  // 0x7F7F...: above every ID
    HIP_TRY(hipMemsetAsync(d_max, 0x80, n_vertices * sizeof(long long), st));  // 0x8080...: below every ID
    hipLaunchKernelGGL(k_uf_init, dim3((n_vertices + 255) / 256), dim3(256), 0, st, d_parent, n_vertices);
    if (n_rows > 1)
        hipLaunchKernelGGL(k_uf_union, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, d_parent, d_rv, d_rr, n_rows);
    hipLaunchKernelGGL(k_uf_stats, dim3((n_vertices + 255) / 256), dim3(256), 0, st, d_parent, (const long long *)d_vid, n_vertices, d_size,
                       (long long *)d_min, (long long *)d_max);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return GS_OK;
}

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
    uint32_t *d_rr, *d_rv, *d_size;
    int64_t *d_vid, *d_min, *d_max;
    if ((rc = db.put(&d_rr, row_read, n_rows)) || (rc = db.put(&d_rv, row_vertex, n_rows)) ||
        (rc = db.put(&d_size, (const uint32_t *)nullptr, n_vertices)) ||
        (rc = db.put(&d_vid, vertex_id, n_vertices)) || (rc = db.put(&d_min, (const int64_t *)nullptr, n_vertices)) ||
        (rc = db.put(&d_max, (const int64_t *)nullptr, n_vertices)))
        return rc;
    if ((rc = gs::union_regions_device(device, n_rows, d_rr, d_rv, n_vertices, d_vid, d_size, d_min, d_max))) return rc;
    HIP_TRY(hipMemcpy(comp_size, d_size, n_vertices * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(comp_min, d_min, n_vertices * sizeof(int64_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(comp_max, d_max, n_vertices * sizeof(int64_t), hipMemcpyDeviceToHost));
    return GS_OK;
}

// diag_filter_v3 over a row set in device memory: ONE WAVEFRONT PER READ, everything between the four row columns and the
// verdict in LDS.
//
// Reference: workflow/src/diag_filter_v3.nim:63-229.  For every read (a run of rows with one name) and every contig it has rows
// on: the median of the diagonals, in both orientations ("-": pos + start, "+": start - pos; fp64 interpolation at (len - 1) / 2,
// truncated toward zero, :86,113), the number of DISTINCT SUNK groups whose diagonal lies within 2500 of it (:88-94,115-123), then
// the selection :95-107,124-136 over the contigs in the slot order of Nim's Table, and `maxgood > 1` (:141).
//
// A read's rows arrive adjacent (rows are in file order), so nothing global has to be sorted (round 4 radix-sorted all rows by
// (read, contig) and then ran one THREAD per (read, contig) pair with a heapsort in global scratch).  A wave
//   1. packs (contig, diagonal, ID) of each row into a 128-bit key and bitonic-sorts the read's keys in LDS: the rows of one
//      contig become one run, ordered by diagonal -- the median is two LDS reads, the rows within the band a contiguous stretch;
//   2. counts the distinct IDs in the band with a hash set in the same LDS (the keys are not needed any more), per contig;
//   3. does 1-2 again for the other orientation, and picks the winner: most distinct groups, then fewest rows, then -- only on
//      an exact tie between two contigs, which needs the Table's slot order -- hands the read's contig list to k_diag_resolve,
//      which replays the Table (capacity carried from read to read, growth and rehash included) for those reads alone.
// Reads of up to 128 / 256 / 512 / 1024 rows: four / four / two / one wave(s) to a workgroup (16-32 KiB of LDS per workgroup);
// longer ones: the same code with its arrays in global scratch.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_reduce.hpp>

#include "gs_chain.h"

using namespace gs;
using namespace gs::chain;

namespace {

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

// Nim's Table: 64 slots to begin with; before a NEW key goes in, the table doubles when dataLen * 2 < counter * 3 or
// dataLen - counter < 4 (counter = entries so far); clear() keeps the capacity (diag_filter_v3.nim:143).  A table of n slots
// therefore takes floor(2 n / 3) + 1 entries without growing: the capacity a read with d distinct contigs needs.
__host__ __device__ inline uint32_t nim_cap_for(uint32_t d) {
    uint32_t n = 64;
    while (d > 0 && (2u * n) / 3u + 1u < d) n *= 2u;
    return n;
}

struct Key128 {
    uint64_t hi, lo;  // hi = contig << 32 | diagonal >> 1 ; lo = (diagonal & 1) << 32 | ID   (diagonal: 33 bits, biased for "+")
};
constexpr uint64_t kEmpty = ~0ULL;
constexpr int kCapS = 128;

struct DiagArgs {
    // rows
    const uint32_t *seg_off;
    const uint32_t *contig, *pos, *start, *id;
    const uint8_t *in_fai;
    uint32_t n_segs;
    int64_t bandwidth;
    // work list (nullptr: segment = wave index) and, for reads that live in global scratch, where
    const uint32_t *list;
    uint32_t n_list;
    const uint64_t *scratch_off;
    uint8_t *scratch;
    // a read with more distinct contigs than a workgroup's LDS has room for goes (back) on the list of the global-scratch kernel
    uint32_t *over_list, *over_count;
    // verdicts per segment
    uint32_t *seg_d, *best_contig, *best_good;
    uint8_t *best_dir;
    // reads whose winner depends on the Table's slot order
    uint32_t *tie_count;          // [0] reads, [1] entries asked for
    uint32_t tie_cap, ent_cap;
    uint32_t *tie_seg, *tie_off, *tie_d;
    uint32_t *ent_contig;
    uint8_t *ent_flag;            // 0 = not among the best, '-' / '+' = among them, with the orientation it would print
};

template <bool GLOBAL>
__device__ inline void wsync() {
    // one wave: its LDS instructions execute in order, what is needed is that the compiler keeps the order (and, for the
    // scratch in global memory, that the stores have landed before another lane loads)
    if (GLOBAL) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
__device__ inline uint32_t lanes_below(uint64_t mask, uint32_t lane) { return __popcll(mask & ((1ULL << lane) - 1ULL)); }
__device__ inline uint64_t wave_max_u64(uint64_t v) {
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}
__device__ inline uint32_t wave_sum_u32(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline bool key_gt(const Key128 &a, const Key128 &b) { return a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo); }
__device__ inline int64_t key_diag(const Key128 &k, int dir) {
    const uint64_t v = ((k.hi & 0xFFFFFFFFULL) << 1) | (k.lo >> 32);
    return dir == 0 ? (int64_t)v : (int64_t)v - (1LL << 32);
}

// EPL > 0: the hash set of step 2 reuses the key array; the per-element hash keys wait in EPL registers per lane meanwhile
// EPL = 0: hash set in an array of its own (global scratch)
// gmax: how many runs (distinct contigs) the run arrays hold; false = the read has more (nothing has been written for it)
template <int EPL, bool GLOBAL>
__device__ bool diag_segment(const DiagArgs &a, uint32_t s, uint32_t lane, uint32_t P, uint32_t gmax, Key128 *K, uint64_t *H, uint32_t *gstart,
                             uint32_t *gcontig, uint32_t *good0, uint32_t *good1) {
    const uint64_t row0 = a.seg_off[s];
    const uint32_t n = a.seg_off[s + 1] - a.seg_off[s];
    const uint32_t HS = 2u * P;
    uint32_t d = 0;
    for (int dir = 0; dir < 2; dir++) {
        // ---- 1. keys, sorted ----
        for (uint32_t j = lane; j < P; j += 64) {
            Key128 k{kEmpty, kEmpty};
            if (j < n) {
                const uint64_t r = row0 + j;
                const uint64_t p = a.pos[r], st = a.start[r];
                const uint64_t v = dir == 0 ? p + st : st - p + (1ULL << 32);
                k.hi = ((uint64_t)a.contig[r] << 32) | (v >> 1);
                k.lo = ((v & 1ULL) << 32) | a.id[r];
            }
            K[j] = k;
        }
        wsync<GLOBAL>();
        for (uint32_t kk = 2; kk <= P; kk <<= 1)
            for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
                for (uint32_t t = lane; t < (P >> 1); t += 64) {
                    const uint32_t i = ((t & ~(jj - 1u)) << 1) | (t & (jj - 1u)), l = i | jj;
                    const bool up = (i & kk) == 0;
                    const Key128 x = K[i], y = K[l];
                    if (key_gt(x, y) == up) {
                        K[i] = y;
                        K[l] = x;
                    }
                }
                wsync<GLOBAL>();
            }
        // ---- the runs of one contig (the same in both orientations: contigs ascend, counts are what they are) ----
        if (dir == 0) {
            uint32_t base = 0;
            for (uint32_t j0 = 0; j0 < n; j0 += 64) {
                const uint32_t j = j0 + lane;
                bool head = false;
                uint32_t c = 0;
                if (j < n) {
                    c = (uint32_t)(K[j].hi >> 32);
                    head = j == 0 || (uint32_t)(K[j - 1].hi >> 32) != c;
                }
                const uint64_t hb = __ballot(head);
                if (head) {
                    const uint32_t q = base + lanes_below(hb, lane);
                    if (q < gmax) {
                        gstart[q] = j;
                        gcontig[q] = c;
                    }
                }
                base += __popcll(hb);
            }
            d = base;
            if (d > gmax) return false;
            if (lane == 0) gstart[d] = n;
            for (uint32_t q = lane; q < d; q += 64) {
                good0[q] = 0;
                good1[q] = 0;
            }
            wsync<GLOBAL>();
        }
        // ---- 2. per element: its run, the run's median, inside the band? -> hash key (run, ID) ----
        uint32_t *good = dir == 0 ? good0 : good1;
        uint64_t hk[EPL > 0 ? EPL : 1];
        uint32_t base = 0;
        auto element_key = [&](uint32_t j, uint32_t &base_io) -> uint64_t {
            bool head = false;
            Key128 me{kEmpty, kEmpty};
            if (j < n) {
                me = K[j];
                head = j == 0 || (uint32_t)(K[j - 1].hi >> 32) != (uint32_t)(me.hi >> 32);
            }
            const uint64_t hb = __ballot(head);
            // run of element j: heads at or below its lane in this chunk, plus the runs of earlier chunks
            const uint32_t lane_ = j & 63u;
            const uint32_t q = base_io + __popcll(hb & ((2ULL << lane_) - 1ULL)) - 1u;
            base_io += __popcll(hb);
            if (j >= n) return kEmpty;
            const uint32_t g0 = gstart[q], m = gstart[q + 1] - g0;
            if (m < 2 || !a.in_fai[(uint32_t)(me.hi >> 32)]) return kEmpty;  // `if p.shape[0] == 1: continue` (:82); rows of other contigs are never stored (:147-150)
            const uint32_t i0 = (m - 1u) >> 1;
            const int64_t va = key_diag(K[g0 + i0], dir);
            double res = (double)va;
            if (!(m & 1u)) {  // (len - 1) * 50 / 100 ends in .5: halfway between the two middle values
                const int64_t vb = key_diag(K[g0 + i0 + 1], dir);
                res = (double)va + 0.5 * ((double)vb - (double)va);
            }
            const int64_t median = (int64_t)res;  // truncation toward zero
            int64_t dl = key_diag(me, dir) - median;
            if (dl < 0) dl = -dl;
            if (dl >= a.bandwidth) return kEmpty;
            return ((uint64_t)q << 32) | (uint32_t)me.lo;
        };
        auto insert = [&](uint64_t key) {
            if (key == kEmpty) return;
            uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> 40) & (HS - 1u);
            for (;;) {
                const uint64_t old = atomicCAS((unsigned long long *)&H[h], (unsigned long long)kEmpty, (unsigned long long)key);
                if (old == kEmpty) {
                    atomicAdd(&good[(uint32_t)(key >> 32)], 1u);  // a group not seen before in this run's band (`goodset.incl`, :93)
                    return;
                }
                if (old == key) return;
                h = (h + 1u) & (HS - 1u);
            }
        };
        if (EPL > 0) {
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                hk[e] = kEmpty;
                if ((uint32_t)e * 64u < n) hk[e] = element_key((uint32_t)e * 64u + lane, base);
            }
            wsync<GLOBAL>();  // (every lane has read what it needs of the keys)
            for (uint32_t h = lane; h < HS; h += 64) H[h] = kEmpty;
            wsync<GLOBAL>();
#pragma unroll
            for (int e = 0; e < EPL; e++)
                if ((uint32_t)e * 64u < n) insert(hk[e]);
        } else {
            for (uint32_t h = lane; h < HS; h += 64) H[h] = kEmpty;
            wsync<GLOBAL>();
            for (uint32_t j0 = 0; j0 < n; j0 += 64) insert(element_key(j0 + lane, base));
        }
        wsync<GLOBAL>();
    }
    // ---- 3. the selection (:95-107,124-136): most groups, then fewest rows, then the first the Table yields; "-" before "+" ----
    uint64_t best = 0;
    for (uint32_t q0 = 0; q0 < d; q0 += 64) {
        const uint32_t q = q0 + lane;
        uint64_t cand = 0;
        if (q < d) {
            const uint32_t m = gstart[q + 1] - gstart[q];
            const uint32_t g = max(good0[q], good1[q]);
            if (g) cand = ((uint64_t)g << 32) | (0xFFFFFFFFu - m);
        }
        best = max(best, wave_max_u64(cand));
    }
    const uint32_t maxgood = (uint32_t)(best >> 32);
    uint32_t n_best = 0, win_q = kNone;
    if (maxgood > 1) {
        for (uint32_t q0 = 0; q0 < d; q0 += 64) {
            const uint32_t q = q0 + lane;
            bool is = false;
            if (q < d) {
                const uint32_t m = gstart[q + 1] - gstart[q];
                const uint32_t g = max(good0[q], good1[q]);
                is = g && ((((uint64_t)g << 32) | (0xFFFFFFFFu - m)) == best);
            }
            const uint64_t b = __ballot(is);
            if (b && win_q == kNone) win_q = q0 + (uint32_t)__ffsll((long long)b) - 1u;
            n_best += __popcll(b);
        }
    }
    if (n_best >= 2) {
        // the Table's slot order decides: contigs in order of first appearance, those among the best marked, to k_diag_resolve
        for (uint32_t q = lane; q < d; q += 64) {
            const uint32_t m = gstart[q + 1] - gstart[q];
            const uint32_t gm = good0[q], gp = good1[q], g = max(gm, gp);
            const bool is = g && ((((uint64_t)g << 32) | (0xFFFFFFFFu - m)) == best);
            good1[q] = is ? (gp > gm ? (uint32_t)'+' : (uint32_t)'-') : 0u;  // (a run's "-" is tried before its "+": "+" only when it is better)
            good0[q] = kNone;                                                 // first row of the contig in the read
        }
        wsync<GLOBAL>();
        for (uint32_t j = lane; j < n; j += 64) {
            const uint32_t c = a.contig[row0 + j];
            uint32_t lo = 0, hi = d;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (gcontig[mid] < c) lo = mid + 1;
                else hi = mid;
            }
            atomicMin(&good0[lo], j);
        }
        wsync<GLOBAL>();
        uint32_t off = 0, t = 0;
        if (lane == 0) {
            t = atomicAdd(&a.tie_count[0], 1u);
            off = atomicAdd(&a.tie_count[1], d);
        }
        t = __shfl(t, 0, 64);
        off = __shfl(off, 0, 64);
        if (t < a.tie_cap && (uint64_t)off + d <= a.ent_cap) {
            for (uint32_t q = lane; q < d; q += 64) {
                const uint32_t f = good0[q];
                uint32_t rank = 0;
                for (uint32_t o = 0; o < d; o++) rank += good0[o] < f ? 1u : 0u;
                a.ent_contig[off + rank] = gcontig[q];
                a.ent_flag[off + rank] = (uint8_t)good1[q];
            }
            if (lane == 0) {
                a.tie_seg[t] = s;
                a.tie_off[t] = off;
                a.tie_d[t] = d;
            }
        }
        if (lane == 0) {
            a.seg_d[s] = d;
            a.best_contig[s] = GS_NONE;  // (k_diag_resolve)
            a.best_good[s] = maxgood;
            a.best_dir[s] = 0;
        }
        return true;
    }
    if (lane == 0) {
        a.seg_d[s] = d;
        uint32_t bc = GS_NONE;
        uint8_t bd = 0;
        if (n_best == 1) {
            bc = gcontig[win_q];
            bd = good1[win_q] > good0[win_q] ? '+' : '-';
        }
        a.best_contig[s] = bc;  // :141 `if maxgood > 1`
        a.best_good[s] = maxgood;
        a.best_dir[s] = bd;
    }
    return true;
}

__device__ inline uint32_t pow2_at_least(uint32_t n) {
    uint32_t p = 2;
    while (p < n) p <<= 1;
    return p;
}

// reads of at most CAP rows, WAVES of them to a workgroup; a longer read is left to the kernel of its class.  A workgroup stays
// below 22.8 KiB of LDS -- what ONE retiring workgroup of the tagger's probe kernel (which fills all 160 KiB of a CU) sets free:
// a workgroup that needs more waits until two of them retire together, which beside a running pass took 30 x as long
// (profiles/r05/trace_overlap_lds_starved.txt).  Hence run arrays for at most 256 distinct contigs per read; more: global scratch.
template <int CAP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_diag_lds(DiagArgs a) {
    constexpr int GMAX = CAP < 256 ? CAP : 256;
    __shared__ Key128 sK[WAVES][CAP];
    __shared__ uint32_t sStart[WAVES][GMAX + 1], sContig[WAVES][GMAX], sGood[WAVES][2][GMAX];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t item = blockIdx.x * WAVES + w;
    uint32_t s = item;
    if (a.list) {
        if (item >= a.n_list) return;
        s = a.list[item];
    } else if (item >= a.n_segs) return;
    const uint32_t n = a.seg_off[s + 1] - a.seg_off[s];
    if (n > (uint32_t)CAP) return;
    if (n == 1) {  // one row: one contig, skipped by :82
        if (lane == 0) {
            a.seg_d[s] = 1;
            a.best_contig[s] = GS_NONE;
            a.best_good[s] = 0;
            a.best_dir[s] = 0;
        }
        return;
    }
    if (!diag_segment<CAP / 64, false>(a, s, lane, pow2_at_least(n), GMAX, sK[w], reinterpret_cast<uint64_t *>(sK[w]), sStart[w], sContig[w],
                                       sGood[w][0], sGood[w][1]) &&
        lane == 0)
        a.over_list[atomicAdd(a.over_count, 1u)] = s;
}
// reads of any length: the same arrays in global scratch (48 P + 64 bytes per read, P = its row count rounded up to a power of two)
__global__ __launch_bounds__(64) void k_diag_global(DiagArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t item = blockIdx.x;
    if (item >= a.n_list) return;
    const uint32_t s = a.list[item];
    const uint32_t n = a.seg_off[s + 1] - a.seg_off[s];
    const uint32_t P = pow2_at_least(n);
    uint8_t *b = a.scratch + a.scratch_off[item];
    Key128 *K = reinterpret_cast<Key128 *>(b);
    uint64_t *H = reinterpret_cast<uint64_t *>(b + 16ULL * P);
    uint32_t *gstart = reinterpret_cast<uint32_t *>(b + 32ULL * P);
    uint32_t *gcontig = gstart + (P + 4);
    uint32_t *good0 = gcontig + P, *good1 = good0 + P;
    (void)diag_segment<0, true>(a, s, lane, P, P, K, H, gstart, gcontig, good0, good1);
}
__host__ __device__ inline uint64_t global_scratch_bytes(uint32_t n) {
    uint64_t P = 2;
    while (P < n) P <<= 1;
    return 48ULL * P + 64ULL;
}

// which kernel takes a read: up to 128 rows (class 0: k_diag_lds<128> runs over all reads and leaves the others alone), up to
// 256 / 512 / 1024 (classes 1-3), longer (class 4: global scratch).  Lists of classes 1-4 (any order), counts[4]; one atomic per
// wave and class that occurs in it
constexpr int kNumLists = 4;
__host__ __device__ inline int seg_class(uint32_t n) { return n <= 128u ? 0 : n <= 256u ? 1 : n <= 512u ? 2 : n <= 1024u ? 3 : 4; }
__global__ void k_seg_class(const uint32_t *seg_off, uint32_t n_segs, uint32_t *lists, uint32_t *counts) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const int cls = s < n_segs ? seg_class(seg_off[s + 1] - seg_off[s]) : 0;
    for (int k = 1; k <= kNumLists; k++) {
        const uint64_t b = __ballot(cls == k);
        if (!b) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&counts[k - 1], (uint32_t)__popcll(b));
        base = __shfl(base, 0, 64);
        if (cls == k) lists[(uint64_t)(k - 1) * n_segs + base + lanes_below(b, lane)] = s;
    }
}
__global__ void k_scratch_need(const uint32_t *list, uint32_t n_list, const uint32_t *seg_off, uint64_t *bytes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_list) bytes[i] = global_scratch_bytes(seg_off[list[i] + 1] - seg_off[list[i]]);
}
__global__ void k_seg_need(const uint32_t *seg_d, uint32_t n_segs, uint32_t *need) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_segs) need[s] = nim_cap_for(seg_d[s]);
}
// the Table capacity a tied read starts from and ends with (the running maximum of what the reads before it needed)
__global__ void k_tie_caps(uint32_t n_ties, const uint32_t *tie_seg, const uint32_t *tie_d, const uint32_t *need_prefmax, uint32_t cap0,
                           uint32_t *cap_before, uint32_t *cap_final) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_ties) return;
    const uint32_t s = tie_seg[t];
    const uint32_t cb = s ? max(cap0, need_prefmax[s - 1]) : cap0;
    cap_before[t] = cb;
    cap_final[t] = max(cb, nim_cap_for(tie_d[t]));
}
// one thread per tied read: the read's contigs go into the Table in order of first appearance, exactly as Nim's tables.nim does it
// (linear probing from hash & mask; `enlarge` re-inserts the old slots in index order), then the slots are walked from 0: the
// first contig among the best wins.  scratch: 4 arrays of the final capacity per read.
__global__ void k_diag_resolve(uint32_t n_ties, const uint32_t *tie_seg, const uint32_t *tie_off, const uint32_t *tie_d, const uint32_t *ent_contig,
                               const uint8_t *ent_flag, const uint32_t *chash, const uint32_t *cap_before, const uint64_t *scratch_off,
                               uint32_t *scratch, uint32_t *best_contig, uint8_t *best_dir) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_ties) return;
    const uint32_t d = tie_d[t], off = tie_off[t];
    uint32_t n = cap_before[t];
    const uint64_t capf = (scratch_off[t + 1] - scratch_off[t]) / 4;
    uint32_t *key = scratch + scratch_off[t], *hc = key + capf, *nk = hc + capf, *nh = nk + capf;
    for (uint32_t i = 0; i < n; i++) key[i] = kNone;
    uint32_t counter = 0;
    for (uint32_t r = 0; r < d; r++) {
        const uint32_t h0 = chash[ent_contig[off + r]];
        if (n * 2u < counter * 3u || n - counter < 4u) {  // enlarge
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
        key[h] = r;
        hc[h] = h0;
        counter++;
    }
    for (uint32_t i = 0; i < n; i++)
        if (key[i] != kNone && ent_flag[off + key[i]]) {
            best_contig[tie_seg[t]] = ent_contig[off + key[i]];
            best_dir[tie_seg[t]] = ent_flag[off + key[i]];
            return;
        }
}

// (a fixed grid walks the rows: one atomic per wave at the END -- a few thousand on one address, not one per 64 rows)
__global__ void k_keep_rows(uint64_t n, const uint32_t *seg, const uint32_t *contig, const uint32_t *best_contig, uint8_t *keep,
                            unsigned long long *n_kept) {
    uint32_t mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = best_contig[seg[i]];
        const bool k = b != GS_NONE && b == contig[i];  // diag_filter_step2.nim:45-64
        keep[i] = k ? 1 : 0;
        mine += k ? 1u : 0u;
    }
    mine = wave_sum_u32(mine);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(n_kept, (unsigned long long)mine);
}
__global__ void k_seg_names(const uint32_t *seg_off, const uint32_t *name, uint32_t n_segs, uint32_t *seg_name) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_segs) seg_name[s] = name[seg_off[s]];
}
__global__ void k_any_ge(const uint32_t *v, uint64_t n, uint32_t bound, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = i < n && v[i] >= bound;
    if (__ballot(bad) && (threadIdx.x & 63u) == 0) atomicOr(flag, 1u);
}

}  // namespace

extern "C" {

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
    GSC_TRY(check_device(rs->device, "gs_rowset_diag_filter"));
    Ctx c(rs->device);
    const uint64_t n = rs->n;
    // contig ids index in_fai / the name hashes: a wrong count must fail loudly, not read out of bounds
    if (n) {
        uint32_t *d_flag, h = 0;
        GSC_TRY(c.mem.get(&d_flag, 1));
        GSC_TRY(c.zero(d_flag, 1));
        hipLaunchKernelGGL(k_any_ge, dim3(blocks_of(n)), dim3(256), 0, c.st, rs->contig, n, n_contigs, d_flag);
        GSC_TRY(c.fetch(&h, d_flag));
        if (h) {
            set_error("gs_rowset_diag_filter: contig: a row's id is not below %u", n_contigs);
            return GS_E_ARG;
        }
    }
    GSC_TRY(compute_segments(c, rs));
    const uint32_t n_segs = rs->n_segs;
    if (rs->keep) pool_free(rs->device, rs->keep);
    rs->keep = static_cast<uint8_t *>(pool_alloc(rs->device, std::max<uint64_t>(n, 1)));
    if (!rs->keep) return GS_E_NOMEM;
    rs->has_keep = true;
    if (n_kept) *n_kept = 0;
    if (!n) return GS_OK;

    DiagArgs a{};
    uint8_t *d_infai;
    GSC_TRY(c.mem.put(&d_infai, in_fai, n_contigs));
    a.seg_off = rs->seg_off;
    a.contig = rs->contig;
    a.pos = rs->pos;
    a.start = rs->start;
    a.id = rs->id;
    a.in_fai = d_infai;
    a.n_segs = n_segs;
    a.bandwidth = 2500;
    GSC_TRY(c.mem.get(&a.seg_d, n_segs));
    GSC_TRY(c.mem.get(&a.best_contig, n_segs));
    GSC_TRY(c.mem.get(&a.best_good, n_segs));
    GSC_TRY(c.mem.get(&a.best_dir, n_segs));
    GSC_TRY(c.mem.get(&a.tie_count, 2));
    // which reads go where
    uint32_t *d_lists, *d_lcount;
    GSC_TRY(c.mem.get(&d_lists, (uint64_t)kNumLists * n_segs));
    GSC_TRY(c.mem.get(&d_lcount, kNumLists));
    GSC_TRY(c.zero(d_lcount, kNumLists));
    hipLaunchKernelGGL(k_seg_class, dim3(blocks_of(n_segs)), dim3(256), 0, c.st, rs->seg_off, n_segs, d_lists, d_lcount);
    uint32_t lcount[kNumLists] = {0};
    GSC_TRY(c.fetch(lcount, d_lcount, kNumLists));
    uint32_t *d_listl = d_lists + (uint64_t)(kNumLists - 1) * n_segs;  // reads for the global-scratch kernel: the long ones, then overflows
    a.over_list = d_listl;
    a.over_count = d_lcount + (kNumLists - 1);
    // ties are rare (two contigs with the same number of groups AND of rows): room for a few, the exact amount on a second go
    uint32_t tie_cap = 1u << 16, ent_cap = 1u << 20;
    uint32_t ties[2] = {0, 0};
    for (int attempt = 0; attempt < 2; attempt++) {
        GSC_TRY(c.mem.get(&a.tie_seg, tie_cap));
        GSC_TRY(c.mem.get(&a.tie_off, tie_cap));
        GSC_TRY(c.mem.get(&a.tie_d, tie_cap));
        GSC_TRY(c.mem.get(&a.ent_contig, ent_cap));
        GSC_TRY(c.mem.get(&a.ent_flag, ent_cap));
        a.tie_cap = tie_cap;
        a.ent_cap = ent_cap;
        GSC_TRY(c.zero(a.tie_count, 2));
        GSC_HIP(hipMemcpyAsync(d_lcount + (kNumLists - 1), &lcount[kNumLists - 1], sizeof(uint32_t), hipMemcpyHostToDevice, c.st));  // (second go: no overflows yet)
        a.list = nullptr;
        a.n_list = 0;
        hipLaunchKernelGGL((k_diag_lds<kCapS, 4>), dim3(blocks_of(n_segs, 4)), dim3(256), 0, c.st, a);
        if (lcount[0]) {
            a.list = d_lists;
            a.n_list = lcount[0];
            hipLaunchKernelGGL((k_diag_lds<256, 2>), dim3(blocks_of(lcount[0], 2)), dim3(128), 0, c.st, a);
        }
        if (lcount[1]) {
            a.list = d_lists + n_segs;
            a.n_list = lcount[1];
            hipLaunchKernelGGL((k_diag_lds<512, 1>), dim3(lcount[1]), dim3(64), 0, c.st, a);
        }
        if (lcount[2]) {
            a.list = d_lists + 2ULL * n_segs;
            a.n_list = lcount[2];
            hipLaunchKernelGGL((k_diag_lds<1024, 1>), dim3(lcount[2]), dim3(64), 0, c.st, a);
        }
        GSC_HIP(hipGetLastError());
        uint32_t n_l = 0;
        GSC_TRY(c.fetch(&n_l, d_lcount + (kNumLists - 1)));
        if (n_l) {  // longer than 1024 rows, or more than 256 contigs: the same code, its arrays in global scratch
            uint64_t *d_bytes, *d_soff;
            uint8_t *d_scratch;
            GSC_TRY(c.mem.get(&d_bytes, (uint64_t)n_l + 1));
            GSC_TRY(c.mem.get(&d_soff, (uint64_t)n_l + 1));
            GSC_TRY(c.zero(d_bytes + n_l, 1));
            hipLaunchKernelGGL(k_scratch_need, dim3(blocks_of(n_l)), dim3(256), 0, c.st, d_listl, n_l, rs->seg_off, d_bytes);
            GSC_TRY(exclusive_sum(c, d_bytes, d_soff, (uint64_t)n_l + 1));
            uint64_t total = 0;
            GSC_TRY(c.fetch(&total, d_soff + n_l));
            GSC_TRY(c.mem.get(&d_scratch, total));
            a.list = d_listl;
            a.n_list = n_l;
            a.scratch_off = d_soff;
            a.scratch = d_scratch;
            hipLaunchKernelGGL(k_diag_global, dim3(n_l), dim3(64), 0, c.st, a);
            GSC_HIP(hipGetLastError());
        }
        GSC_TRY(c.fetch(ties, a.tie_count, 2));
        if (ties[0] <= tie_cap && ties[1] <= ent_cap) break;
        if (attempt == 1) {
            set_error("gs_rowset_diag_filter: internal: the tie list overflowed twice");
            return GS_E_ARG;
        }
        tie_cap = ties[0];
        ent_cap = ties[1];
    }
    // the capacity the Table leaves this row set with: what the read with the most contigs needed
    uint32_t *d_need, *d_prefmax = nullptr;
    GSC_TRY(c.mem.get(&d_need, n_segs));
    hipLaunchKernelGGL(k_seg_need, dim3(blocks_of(n_segs)), dim3(256), 0, c.st, a.seg_d, n_segs, d_need);
    if (ties[0]) {
        GSC_TRY(c.mem.get(&d_prefmax, n_segs));
        GSC_TRY(inclusive_scan(c, d_need, d_prefmax, n_segs, rocprim::maximum<uint32_t>()));
    }
    if (cap_out) {
        uint32_t last = 0;
        if (d_prefmax) {
            GSC_TRY(c.fetch(&last, d_prefmax + n_segs - 1));
        } else {
            uint32_t *d_max;
            uint8_t *tmp;
            size_t tb = 0;
            GSC_TRY(c.mem.get(&d_max, 1));
            GSC_HIP(rocprim::reduce(nullptr, tb, d_need, d_max, 0u, (size_t)n_segs, rocprim::maximum<uint32_t>(), c.st));
            GSC_TRY(c.mem.get(&tmp, std::max<size_t>(tb, 16)));
            GSC_HIP(rocprim::reduce(tmp, tb, d_need, d_max, 0u, (size_t)n_segs, rocprim::maximum<uint32_t>(), c.st));
            GSC_TRY(c.fetch(&last, d_max));
        }
        *cap_out = std::max(cap0, last);
    }
    if (ties[0]) {
        const uint32_t nt = ties[0];
        std::vector<uint32_t> chash(n_contigs);
        for (uint32_t q = 0; q < n_contigs; q++) chash[q] = murmur3_32(contig_names[q], strlen(contig_names[q]));
        uint32_t *d_chash, *d_cb, *d_cf, *d_scr;
        uint64_t *d_so;
        GSC_TRY(c.mem.put(&d_chash, chash.data(), n_contigs));
        GSC_TRY(c.mem.get(&d_cb, nt));
        GSC_TRY(c.mem.get(&d_cf, nt));
        hipLaunchKernelGGL(k_tie_caps, dim3(blocks_of(nt)), dim3(256), 0, c.st, nt, a.tie_seg, a.tie_d, d_prefmax, cap0, d_cb, d_cf);
        std::vector<uint32_t> cf(nt);
        GSC_TRY(c.fetch(cf.data(), d_cf, nt));
        std::vector<uint64_t> so((size_t)nt + 1, 0);
        for (uint32_t t = 0; t < nt; t++) so[t + 1] = so[t] + 4ULL * cf[t];
        GSC_TRY(c.mem.put(&d_so, so.data(), so.size()));
        GSC_TRY(c.mem.get(&d_scr, so[nt]));
        hipLaunchKernelGGL(k_diag_resolve, dim3(blocks_of(nt, 64)), dim3(64), 0, c.st, nt, a.tie_seg, a.tie_off, a.tie_d, a.ent_contig, a.ent_flag,
                           d_chash, d_cb, d_so, d_scr, a.best_contig, a.best_dir);
    }
    // verdict per row
    unsigned long long *d_kept;
    GSC_TRY(c.mem.get(&d_kept, 1));
    GSC_TRY(c.zero(d_kept, 1));
    hipLaunchKernelGGL(k_keep_rows, dim3(std::min(blocks_of(n), 2048u)), dim3(256), 0, c.st, n, rs->seg, rs->contig, a.best_contig, rs->keep, d_kept);
    GSC_HIP(hipGetLastError());
    if (best_contig) GSC_HIP(hipMemcpyAsync(best_contig, a.best_contig, (size_t)n_segs * 4, hipMemcpyDeviceToHost, c.st));
    if (best_good) GSC_HIP(hipMemcpyAsync(best_good, a.best_good, (size_t)n_segs * 4, hipMemcpyDeviceToHost, c.st));
    if (best_dir) GSC_HIP(hipMemcpyAsync(best_dir, a.best_dir, n_segs, hipMemcpyDeviceToHost, c.st));
    if (seg_name) {
        uint32_t *d_sn;
        GSC_TRY(c.mem.get(&d_sn, n_segs));
        hipLaunchKernelGGL(k_seg_names, dim3(blocks_of(n_segs)), dim3(256), 0, c.st, rs->seg_off, rs->name, n_segs, d_sn);
        GSC_HIP(hipMemcpyAsync(seg_name, d_sn, (size_t)n_segs * 4, hipMemcpyDeviceToHost, c.st));
    }
    unsigned long long kept = 0;
    GSC_TRY(c.fetch(&kept, d_kept));
    if (n_kept) *n_kept = kept;
    return GS_OK;
}

}  // extern "C"

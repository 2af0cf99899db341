// Minimiser-addressed table ("minimizer-hbm"): the layout for SUNK tables that are far larger than the caches
// (genome-scale: 10^8 .. 10^9+ keys).  Included by gs_device.hip inside its anonymous namespace.
//
// Why.  A random probe of a table that lives in HBM costs one fabric request whatever it fetches, and the chip moves
// ~47 G such requests per second (profiles/r02/ubench_run_coalescing.txt: 64-byte and 128-byte regions alike) -- one
// request per window is the 0.09-of-roofline wall of the direct layouts.  The texture path, though, coalesces the lanes
// of ONE wave-load that fall into the same 128-byte line into one request: with runs of L adjacent lanes per line the
// same microbenchmark reads 47 G lines/s x L windows (L = 3: 142, L = 4: 186, L = 6: 270 G windows/s).
//
// So (1) a lane is a WINDOW (consecutive lanes = consecutive windows of a read), and (2) the 128-byte region a k-mer lives
// in is chosen by its MINIMISER -- the smallest hash among the canonical m-mers inside it -- which consecutive windows
// share ((w + 1) / 2 of them on average, w = k - m + 1).  Inside the region the k-mer's own hash picks one of 8 pieces
// of 16 bytes = 2 slots, so a window still costs ONE 16-byte load.
//
//   H       = min over the w m-mers of the k-mer of mix_a(min(m-mer, its reverse complement))     strand-symmetric
//   region  = mul_hi(mix_b(H), n_regions)                                                          128 bytes each
//   l | hi  = mix_a(low 32 bits of the canonical code) ^ mix_b(high bits), high bits               a bijection
//   piece   = l >> 29,   tag = (l & 0x1FFFFFFF) << (2k - 32) | hi                                  2k - 3 bits
//   slot    = tag << locbits | loc                    64 bits, all ones = empty (the all-ones loc is never stored)
//
// (region, piece, tag) does not determine the key -- the region comes from the minimiser -- but (piece, tag) alone does,
// so a matching slot is an exact hit.  A key goes to the first piece with a free slot, starting at its own piece and
// walking round the region; a lookup walks the same way and stops at the first piece that is not full.  A key whose
// region is full (16 keys) stays in the generic 16-byte-slot table; a lookup that found all 8 pieces full and no match
// asks that table.  m >= k - 7 (a minimiser owns at most w = 8 adjacent k-mers of a place: half a region) and grows with
// the key count (about log4 of it: each minimiser value should own <= ~1 place in the assembly): human-genome-scale
// tables of 20-mers get m = 16, runs of ~3 windows; small ones m = 13, runs of ~4.5.  m-mers longer than 16 bases
// (k > 23) are hashed down to 32 bits.
//
// Payloads (the loc row a hit reports) are not in the slots: only the emitted rows (1e-4 of the windows) need one, and
// k_emit finds it by binary search in the sorted key list kept beside the table.
#pragma once

struct MzView {
    const uint4 *regions;  // [n_regions][8] pieces of two 64-bit slots
    uint32_t n_regions;
    int k, m, locbits;
    const uint64_t *skeys;     // keys sorted ascending
    const uint32_t *spayload;  // payload of skeys[i]
    uint64_t n_keys;
};

constexpr uint64_t kMzEmpty = ~0ULL;
constexpr int kMzPieces = 8;       // 16-byte pieces per 128-byte region
constexpr int kMzCandLds = 128;    // candidates of a tile kept in LDS; tiles with more are walked a second time
constexpr int kMzU = 4;            // wave-steps (64 windows each) whose table loads are in flight together

// two different bijections on 32-bit values
__host__ __device__ __forceinline__ uint32_t mz_mix_a(uint32_t x) {
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t mz_mix_b(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

struct MzKey {
    uint32_t piece;
    uint64_t tag;
};
// k >= 16: the canonical code has 2k - 32 >= 0 bits above its low word
__host__ __device__ __forceinline__ MzKey mz_key(uint64_t c, int k) {
    const uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
    const uint32_t l = mz_mix_a(lo) ^ mz_mix_b(hi + 0x632BE5ABu);
    MzKey o;
    o.piece = l >> 29;
    o.tag = ((uint64_t)(l & 0x1FFFFFFFu) << (2 * k - 32)) | (uint64_t)hi;  // k = 16: hi = 0, no shift
    return o;
}
__host__ __device__ __forceinline__ uint32_t mz_region(uint32_t H, uint32_t n_regions) {
    return (uint32_t)(((uint64_t)mz_mix_b(H) * (uint64_t)n_regions) >> 32);
}
// hash of a canonical m-mer (m <= 31); for m <= 16 the high word is 0 and mz_mix_b(0) = 0: a bijection of the m-mer
__host__ __device__ __forceinline__ uint32_t mz_mmer_hash(uint64_t cm) { return mz_mix_a((uint32_t)cm ^ mz_mix_b((uint32_t)(cm >> 32))); }
// minimiser hash of a k-mer code (either strand gives the same value); build side and slow paths
__host__ __device__ __forceinline__ uint32_t mz_minimizer(uint64_t c, int k, int m) {
    uint64_t r = 0, x = c;
    for (int i = 0; i < k; i++) {
        r = (r << 2) | (3ULL - (x & 3ULL));
        x >>= 2;
    }
    const uint64_t mmask = (1ULL << (2 * m)) - 1ULL;
    uint32_t H = 0xFFFFFFFFu;
    for (int j = 0; j <= k - m; j++) {
        const uint64_t fm = (c >> (2 * (k - m - j))) & mmask, rm = (r >> (2 * j)) & mmask;
        const uint32_t h = mz_mmer_hash(fm < rm ? fm : rm);
        H = h < H ? h : H;
    }
    return H;
}

// ------------------------------------------------------------------------------------------
// build: one thread per key, first free slot on the walk round the region
// ------------------------------------------------------------------------------------------
__global__ void k_mz_insert(unsigned long long *slots, uint32_t n_regions, int k, int m, int locbits, const uint64_t *keys,
                            const uint32_t *loc, uint64_t n, unsigned long long *overflow, uint32_t *overflow_list,
                            uint64_t overflow_cap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t c = keys[i];
    const uint32_t reg = mz_region(mz_minimizer(c, k, m), n_regions);
    const MzKey key = mz_key(c, k);
    const unsigned long long v = (key.tag << locbits) | (unsigned long long)loc[i];
    unsigned long long *base = slots + (uint64_t)reg * (2 * kMzPieces);
    for (int w = 0; w < kMzPieces; w++) {
        unsigned long long *p = base + 2 * ((key.piece + (uint32_t)w) & (uint32_t)(kMzPieces - 1));
        if (atomicCAS(p, kMzEmpty, v) == kMzEmpty) return;
        if (atomicCAS(p + 1, kMzEmpty, v) == kMzEmpty) return;
    }
    const unsigned long long o = atomicAdd(overflow, 1ULL);
    if (o < overflow_cap) overflow_list[o] = (uint32_t)i;  // goes to the generic table
}

// payload of a key that is known to be in the table (k_emit): binary search in the sorted key list
__device__ __forceinline__ uint32_t mz_payload(const MzView &mz, uint64_t c) {
    uint64_t lo = 0, hi = mz.n_keys;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (mz.skeys[mid] < c) lo = mid + 1;
        else hi = mid;
    }
    return lo < mz.n_keys && mz.skeys[lo] == c ? mz.spayload[lo] : 0u;
}

// ------------------------------------------------------------------------------------------
// probe: one wavefront per tile, 32 steps of 64 consecutive windows (lane = window)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mz_word(uint64_t W0, uint64_t W1, int idx) {  // idx wave-uniform, 0..66
    const int i0 = idx < 64 ? idx : (idx & 1);
    const uint64_t a = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(W0 >> 32), i0) << 32) |
                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)W0, i0);
    const uint64_t b = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(W1 >> 32), i0) << 32) |
                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)W1, i0);
    return idx < 64 ? a : b;
}

struct MzWin {
    uint64_t f, r;  // forward code and reverse-complement code of the window that starts at this lane's base
    uint32_t h;     // hash of the canonical m-mer that starts there
};
// step s (wave-uniform, 0..32) of a tile whose 66 words are held as W0 (word = lane) and W1 (word 64 + (lane & 1))
__device__ __forceinline__ MzWin mz_window(uint64_t W0, uint64_t W1, int s, int lane, int k, int m, uint64_t kmask, uint64_t mmask) {
    const uint64_t A = mz_word(W0, W1, 2 * s), B = mz_word(W0, W1, 2 * s + 1), C = mz_word(W0, W1, 2 * s + 2);
    const bool hsel = lane >= 32;
    const uint64_t a = hsel ? B : A, b = hsel ? C : B;
    const int sh = 2 * (lane & 31);
    const uint64_t x = sh ? ((a << sh) | (b >> (64 - sh))) : a;  // 32 bases from the window's first one
    MzWin o;
    o.f = x >> (64 - 2 * k);
    o.r = ~revpairs(x) & kmask;  // base i of x, complemented, at bits 2i+1:2i
    if (m <= 16) {  // wave-uniform: the m-mer fits a word
        const uint32_t fm = (uint32_t)(o.f >> (2 * (k - m))), rm = (uint32_t)o.r & (uint32_t)mmask;
        o.h = mz_mix_a(fm < rm ? fm : rm);
    } else {
        const uint64_t fm = o.f >> (2 * (k - m)), rm = o.r & mmask;
        o.h = mz_mmer_hash(fm < rm ? fm : rm);
    }
    return o;
}

// lane i <- lane i + 1, lane 63 <- fill (wave-uniform)
__device__ __forceinline__ uint32_t mz_shift_down(uint32_t x, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}

struct MzTileOut {
    uint32_t n_raw, n_cand, first_loc, last_loc;
};

// MODE 0: the first kMzCandLds candidates go to `cbuf` (LDS);  MODE 1: all candidates go straight to cbuf = cand + off
template <int MODE>
__device__ __forceinline__ MzTileOut mz_tile(const MzView &mz, const TableView &tab, const uint64_t *__restrict__ words,
                                            const TileDesc td, uint32_t *cbuf) {
    const int lane = threadIdx.x & 63;
    const int k = mz.k, m = mz.m, w = k - m + 1, locbits = mz.locbits;
    const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
    const uint64_t mmask = (1ULL << (2 * m)) - 1ULL;  // m <= 31
    const uint64_t locmask = (1ULL << locbits) - 1ULL;
    const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    // the tile's 64 words + the two after them (the buffer carries that much slack after the last read)
    const uint64_t W0 = __builtin_nontemporal_load(words + td.wbase + lane);
    const uint64_t W1 = __builtin_nontemporal_load(words + td.wbase + 64 + (lane & 1));

    uint32_t n_raw = 0, n_cand = 0, first_loc = GS_NONE, carry = GS_NONE;  // wave-uniform
    const uint32_t nsteps = min((td.nwin_left + 63u) / 64u, 32u);
    MzWin cur = mz_window(W0, W1, 0, lane, k, m, kmask, mmask);
    for (uint32_t s0 = 0; s0 < nsteps; s0 += kMzU) {
        const uint4 *addr[kMzU];
        uint64_t want[kMzU];
        uint4 bk[kMzU];
        bool valid[kMzU];
        // phase A: minimiser, region, piece of kMzU steps; their loads leave together
#pragma unroll
        for (int u = 0; u < kMzU; u++) {
            const uint32_t s = s0 + (uint32_t)u;
            valid[u] = false;
            addr[u] = mz.regions;
            want[u] = 0;
            bk[u] = make_uint4(~0u, ~0u, ~0u, ~0u);
            if (s >= nsteps) continue;  // wave-uniform
            const MzWin nxt = mz_window(W0, W1, (int)s + 1, lane, k, m, kmask, mmask);
            uint32_t H = cur.h, x = cur.h;
            for (int j = 1; j < w; j++) {
                x = mz_shift_down(x, lane_value(nxt.h, j - 1));
                H = min(H, x);
            }
            const uint64_t c = cur.f < cur.r ? cur.f : cur.r;
            const MzKey key = mz_key(c, k);
            addr[u] = mz.regions + (uint64_t)mz_region(H, mz.n_regions) * kMzPieces + key.piece;
            want[u] = key.tag << locbits;
            valid[u] = s * 64u + (uint32_t)lane < td.nwin_left;
            if (valid[u]) bk[u] = *addr[u];
            cur = nxt;
        }
        // phase B: compare, walk on where a piece is full (rare), ordered bookkeeping
#pragma unroll
        for (int u = 0; u < kMzU; u++) {
            const uint32_t s = s0 + (uint32_t)u;
            if (s >= nsteps) break;  // wave-uniform
            uint64_t s_lo = (uint64_t)bk[u].x | ((uint64_t)bk[u].y << 32), s_hi = (uint64_t)bk[u].z | ((uint64_t)bk[u].w << 32);
            uint64_t e0 = s_lo ^ want[u], e1 = s_hi ^ want[u];
            uint64_t e = e0 < e1 ? e0 : e1;
            bool hit = valid[u] && e < locmask;
            bool more = valid[u] && !hit && s_hi != kMzEmpty;  // the piece is full: the key may sit further round the region
            if (__ballot(more)) {
                const uint4 *rbase = (const uint4 *)((uintptr_t)addr[u] & ~(uintptr_t)127);
                const uint32_t p0 = (uint32_t)(((uintptr_t)addr[u] >> 4) & 7u);
                uint32_t step = 1;
                while (__ballot(more)) {
                    if (more) {
                        if (step == (uint32_t)kMzPieces) {  // 16 keys in the region: the key, if any, is in the generic table
                            const MzWin ww = mz_window(W0, W1, (int)s, lane, k, m, kmask, mmask);
                            const uint64_t c = ww.f < ww.r ? ww.f : ww.r;
                            uint64_t hh = slot_of(c, tab.shift);
                            Slot sl = load_slot(tab.slots + hh);
                            while (sl.key != c && sl.key != kEmptyKey) {
                                hh = (hh + 1) & tab.mask;
                                sl = load_slot(tab.slots + hh);
                            }
                            if (sl.key == c) {
                                hit = true;
                                e = sl.loc;
                            }
                            more = false;
                        } else {
                            const uint4 b2 = rbase[(p0 + step) & 7u];
                            s_lo = (uint64_t)b2.x | ((uint64_t)b2.y << 32);
                            s_hi = (uint64_t)b2.z | ((uint64_t)b2.w << 32);
                            e0 = s_lo ^ want[u];
                            e1 = s_hi ^ want[u];
                            const uint64_t e2 = e0 < e1 ? e0 : e1;
                            if (e2 < locmask) {
                                hit = true;
                                e = e2;
                                more = false;
                            } else if (s_hi == kMzEmpty) {
                                more = false;
                            }
                        }
                    }
                    step++;
                }
            }
            const uint32_t loc = (uint32_t)e;
            const unsigned long long hb = __ballot(hit);
            if (hb) {  // wave-uniform
                const unsigned long long below = hb & lt;
                const int src = below ? 63 - __clzll((long long)below) : 0;
                const uint32_t pl = __shfl(loc, src);
                const uint32_t prev = below ? pl : carry;
                const bool chg = hit && loc != prev;  // the tile's first raw hit always is one (carry = none)
                const unsigned long long cb = __ballot(chg);
                if (chg) {
                    const uint32_t idx = n_cand + (uint32_t)__popcll(cb & lt);
                    const uint32_t rec = ((n_raw + (uint32_t)__popcll(below)) << 16) | (s * 64u + (uint32_t)lane);
                    if (MODE == 1 || idx < (uint32_t)kMzCandLds) cbuf[idx] = rec;
                }
                n_cand += (uint32_t)__popcll(cb);
                if (first_loc == GS_NONE) first_loc = lane_value(loc, __ffsll((unsigned long long)hb) - 1);
                carry = lane_value(loc, 63 - __clzll((long long)hb));
                n_raw += (uint32_t)__popcll(hb);
            }
        }
    }
    return MzTileOut{n_raw, n_cand, first_loc, carry};
}

__global__ __launch_bounds__(256) void k_probe_mz(MzView mz, TableView tab, const uint64_t *__restrict__ words,
                                                  const TileDesc *__restrict__ tdesc, uint32_t n_tiles, TileArrays ta,
                                                  uint32_t *cand, uint32_t cand_cap, uint32_t *g_cand_count) {
    __shared__ uint32_t s_cand[4][kMzCandLds];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t tile = blockIdx.x * 4u + (uint32_t)wave;
    if (tile >= n_tiles) return;
    const TileDesc td = tdesc[tile];
    MzTileOut o = mz_tile<0>(mz, tab, words, td, s_cand[wave]);
    // up to kDirectCand candidates go to the tile's own slots; longer lists are allocated with one atomic
    uint32_t off = tile * (uint32_t)kDirectCand;
    if (o.n_cand > (uint32_t)kDirectCand) {
        if (lane == 0) off = n_tiles * (uint32_t)kDirectCand + atomicAdd(g_cand_count, o.n_cand);
        off = __shfl(off, 0);
    }
    if (o.n_cand && (uint64_t)off + o.n_cand <= cand_cap) {
        if (o.n_cand <= (uint32_t)kMzCandLds) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i = lane; i < o.n_cand; i += 64u) cand[off + i] = s_cand[wave][i];
        } else {
            mz_tile<1>(mz, tab, words, td, cand + off);  // (rare) walk the tile again, candidates straight to their place
        }
    }
    if (lane == 0) {
        ta.n_raw[tile] = o.n_raw;
        ta.n_cand[tile] = o.n_cand;
        ta.cand_off[tile] = off;
        if (o.n_raw) {
            ta.first_loc[tile] = o.first_loc;
            ta.last_loc[tile] = o.last_loc;
        }
    }
}

// Minimiser-addressed table ("minimizer-hbm"): the layout for SUNK tables that are far larger than the caches
// (genome-scale: 10^8 .. 10^9+ keys).  Included by gs_device.hip inside its anonymous namespace.
//
// Why.  A random probe of a table that lives in HBM costs one fabric request whatever it fetches, and the chip moves
// ~47 G such requests per second (profiles/r02/ubench_run_coalescing.txt: 64-byte and 128-byte regions alike) -- one
// request per window is the 0.09-of-roofline wall of the direct layouts.  Lanes of ONE wave-load that fall into the same
// line, though, are one request: with runs of L adjacent lanes per line the same microbenchmark reads
// 47 G lines/s x L windows (L = 3: 142, L = 4: 186, L = 6: 270 G windows/s).  Since requests, not bytes, are what is
// scarce first, a region is 64 bytes: half a line, half the HBM traffic and half the LDS of a 128-byte one.
//
// So (1) a lane is a WINDOW (consecutive lanes = consecutive windows of a read), and (2) the 64-byte region a k-mer lives
// in is chosen by one of the w = k - m + 1 m-mers inside it, picked so that consecutive windows mostly pick the SAME m-mer.
// A region is fetched ONCE per run of windows, whole, into LDS.  The pick is the mod-minimiser rule (Groot Koerkamp &
// Pibiri 2024): look for the smallest short t-mer of the k-mer (t = 5 .. 5 + w - 1, t = m mod w; there are k - t + 1 of them, a
// multiple of w), at position x, and take the m-mer at position x mod w.  The smallest t-mer stays the same over ~k - t
// windows while x mod w walks with the windows: runs of 1 / 0.31 = 3.2 windows for k = 20, m = 17, where the plain
// minimiser (smallest m-mer hash) gives 2.5 -- a fifth fewer requests on the same table geometry.  Everything is canonical
// (a t-mer or m-mer counts as the smaller of itself and its reverse complement) and ties go to the leftmost t-mer of the
// k-mer IN ITS CANONICAL ORIENTATION, so both strands of a k-mer pick the same m-mer (k - t + 1 = 0 mod w makes x mod w
// mirror correctly).
//
//   x       = leftmost argmin over the k - t + 1 t-mers of the canonical k-mer of (t-mer or its rc, the smaller) * c >> 23
//   H       = mix_a(min(m-mer at x mod w, its reverse complement))                                 strand-symmetric
//   region  = mul_hi(H, n_regions)                                                                 64 bytes each
//   RAW slots (2k + locbits <= 64):      tag = the canonical code,  fp = min((low word * c) >> 24, 0xFE)
//   HASHED slots (longer k-mers):        l | hi = mix_a(low word) ^ mix_b(high bits), high bits      a bijection
//                                        fp = min(l >> 24, 0xFE),  tag = (l & 0x1FFFFFFF) << (2k - 32) | hi   (2k - 3 bits)
//
//   region  = 64 bytes = 8 words of 64 bits: word 0 = 8 fingerprint bytes, words 1..7 = slots
//             byte 0      0xFF - (keys of the region), counted up to 8 = "one more than fits": the region OVERFLOWED -- it keeps
//                         its first 6 keys, the others live in the generic table, and word 7 is a filter of those: two bits
//                         per key (mz_filter_bits), a CLEARED bit = seen (~6 % false positives at 8 keys: an overflowed region
//                         answers most of its misses itself)
//             byte i      fingerprint of slot i (i = 1..7), 0xFF = empty
//             word i      tag << locbits | loc            (all ones = empty; the all-ones loc is never stored)
//
// HASHED slots: (fp, tag) determine l and with it the key -- 0xFE stands for both 0xFE and 0xFF, which differ in bit 24 of
// l, and that bit is in the tag too; RAW slots hold the key itself -- either way a slot whose fingerprint and key / tag match
// is an exact hit, wherever the region came from.  A lookup reads the fingerprint word, compares its bytes all at once, and
// reads only the slots whose fingerprint matches (one for a hit, none for 97 % of the misses).  The build is a stable sort by
// region: a region's keys take its slots in key order; the keys an overflowed region cannot hold stay in the generic
// 16-byte-slot table, and a lookup without a match there asks that table unless the region's filter word rules the key out.
// m >= k - 7 (a minimiser owns at most w = 8 adjacent k-mers of one place), starts at k - 4 and grows with the key count (each
// minimiser value should own <= ~1 place of the assembly): 20-mer tables get m = 16 up to ~2.6e8 keys, m = 17 at human-genome
// scale (1.5e9 keys), runs of ~3 windows.  m-mers longer than 16 bases are hashed down to 32 bits.
//
// Payloads (the loc row a hit reports) are not in the slots: only the emitted rows (1e-4 .. 1e-3 of the windows) need one.
// They sit in a parallel array, one 32-bit word per slot, that only k_emit reads.
#pragma once

// ---- compile-time switches (kernel A/B variants: python -m gavisunk_amd.build --variant NAME -D...) ----
#ifndef GS_MZ_PLACE
#define GS_MZ_PLACE 1  // 1: + the place-record format of gs_mzp.h (k <= 21: a record per place, shorter minimisers, longer runs)
#endif
#ifndef GS_MZ_RD
#define GS_MZ_RD GS_MZ_PLACE  // 1: a window's region = the pick of the leftmost smallest t-mer AS THE READ SHOWS THE K-MER (no strand test in
                    // the probe); the table then holds a key under the pick of either strand (two places when they differ)
#endif
#ifndef GS_MZ_RCW
#define GS_MZ_RCW 1  // 1: the tile staged as two dword streams (forward, reverse complement); windows extracted with v_alignbit
#endif

struct MzView {
    const uint4 *regions;  // [n_regions][8] pieces of two 64-bit slots
    uint32_t n_regions;
    int k, m, locbits;
    const uint32_t *pay;  // [n_regions][8]: payload of the key in slot i (word 0 of a region is unused)
    int place, s;         // gs_mzp.h: place-record regions, 2^(2m - s) of them
};

constexpr uint64_t kMzEmpty = ~0ULL;
constexpr int kMzWords = 8;        // 64-bit words per region (64 bytes): word 0 = fingerprint bytes, words 1..7 = slots
constexpr int kMzPieces = kMzWords / 2;  // 16-byte pieces per region (what one lane fetches)
constexpr int kMzSlots = kMzWords - 1;   // keys per region
constexpr int kMzCandLds = 128;    // candidates of a tile kept in LDS; tiles with more are walked a second time (exact form) / done over by it (fast form)
constexpr int kMzU = 4;            // exact form: wave-steps (64 windows each) whose table loads are in flight together

// two different bijections on 32-bit values
__host__ __device__ __forceinline__ uint32_t mz_mix_a(uint32_t x) {
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    return x;
}
__host__ __device__ __forceinline__ uint32_t mz_mix_b(uint32_t x) {  // mz_mix_b(0) = 0
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    return x;
}

struct MzKey {
    uint32_t fp;  // fingerprint byte, 0..0xFE
    uint64_t tag;
};
// Two slot formats.  RAW, when the whole canonical code fits the slot beside the loc (2k + locbits <= 64: k = 20 with any
// loc the fast kernel takes): tag = the code itself, fingerprint = any hash byte of it (one multiply).  HASHED otherwise:
// l = a bijection of the low word, fingerprint = its top byte, tag = its other 29 bits and the code's high bits, so that
// (fingerprint, tag) still determine the key (k >= 16: the code has 2k - 32 >= 0 bits above its low word).
__host__ __device__ constexpr bool mz_raw(int k, int locbits) { return 2 * k + locbits <= 64; }
__host__ __device__ __forceinline__ MzKey mz_key(uint64_t c, int k, int locbits) {
    const uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
    MzKey o;
    if (mz_raw(k, locbits)) {
        o.fp = min((lo * 0x9E3779B1u) >> 24, 0xFEu);
        o.tag = c;
        return o;
    }
    const uint32_t l = mz_mix_a(lo) ^ mz_mix_b(hi + 0x632BE5ABu);
    o.fp = min(l >> 24, 0xFEu);
    o.tag = ((uint64_t)(l & 0x1FFFFFFFu) << (2 * k - 32)) | (uint64_t)hi;  // k = 16: hi = 0, no shift
    return o;
}
// FMT: 1 = RAW, 0 = HASHED (compile-time: the specialised kernels), -1 = decided from (k, locbits) at run time
template <int FMT>
__device__ __forceinline__ MzKey mz_key_fmt(uint64_t c, int k, int locbits) {
    if constexpr (FMT == 1) {
        MzKey o;
        o.fp = min(((uint32_t)c * 0x9E3779B1u) >> 24, 0xFEu);
        o.tag = c;
        return o;
    } else if constexpr (FMT == 0) {
        const uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
        const uint32_t l = mz_mix_a(lo) ^ mz_mix_b(hi + 0x632BE5ABu);
        MzKey o;
        o.fp = min(l >> 24, 0xFEu);
        o.tag = ((uint64_t)(l & 0x1FFFFFFFu) << (2 * k - 32)) | (uint64_t)hi;
        return o;
    } else {
        return mz_key(c, k, locbits);
    }
}
// the two bits of a key in the 64-bit filter word of an overflowed region (from its tag: both slot formats have it at hand)
__host__ __device__ __forceinline__ unsigned long long mz_filter_bits(uint64_t tag) {
    const uint32_t g = ((uint32_t)tag ^ ((uint32_t)(tag >> 32) * 0x7FEB352Du)) * 0x85EBCA6Bu;
    return (1ULL << (g >> 26)) | (1ULL << ((g >> 20) & 63u));
}
// filter word B of an overflowed region: can the key with these bits be among the keys that did not fit?
__host__ __device__ __forceinline__ bool mz_filter_maybe(unsigned long long B, unsigned long long bits) { return (B & bits) == 0ULL; }
__host__ __device__ __forceinline__ uint32_t mz_region(uint32_t H, uint32_t n_regions) {  // H: a hash already
    return (uint32_t)(((uint64_t)H * (uint64_t)n_regions) >> 32);
}
// hash of a canonical m-mer (m <= 31): the region index is mul_hi of it
__host__ __device__ __forceinline__ uint32_t mz_mmer_hash(uint64_t cm) {
    uint32_t x = (uint32_t)cm ^ ((uint32_t)(cm >> 32) * 0x7FEB352Du);
    x ^= x >> 15;  // (the region is taken from the top bits of the product: the high input bits reach them through this)
    return x * 0x9E3779B1u;
}
// the mod-minimiser rule: length of the short t-mers whose minimum anchors the pick (t = m mod w, 5 <= t <= 12), their
// order (9 bits: with the 7-bit position it fills the 16-bit halves the sliding minimum of k_probe_mz works on)
__host__ __device__ constexpr int mz_t_of(int k, int m) { return 5 + (m - 5) % (k - m + 1); }
__host__ __device__ __forceinline__ uint32_t mz_tmer_order(uint32_t ct) { return (ct * 0x9E3779B1u) >> 23; }
// region hash of a k-mer from its CANONICAL code (so either strand of a k-mer gives the same value); build side and slow paths
__host__ __device__ __forceinline__ uint32_t mz_minimizer(uint64_t c, int k, int m) {
    uint64_t r = 0, x = c;
    for (int i = 0; i < k; i++) {
        r = (r << 2) | (3ULL - (x & 3ULL));
        x >>= 2;
    }
    const int w = k - m + 1, t = mz_t_of(k, m), nt = k - t + 1;
    const uint32_t tmask = (1u << (2 * t)) - 1u;
    uint32_t best = 0xFFFFFFFFu;
    int at = 0;
    for (int j = 0; j < nt; j++) {  // c is the canonical strand: leftmost minimum in canonical orientation
        const uint32_t ft = (uint32_t)(c >> (2 * (k - t - j))) & tmask, rt = (uint32_t)(r >> (2 * j)) & tmask;
        const uint32_t o = mz_tmer_order(ft < rt ? ft : rt);
        if (o < best) {
            best = o;
            at = j;
        }
    }
    const int p = at % w;
    const uint64_t mmask = (1ULL << (2 * m)) - 1ULL;
    const uint64_t fm = (c >> (2 * (k - m - p))) & mmask, rm = (r >> (2 * p)) & mmask;
    return mz_mmer_hash(fm < rm ? fm : rm);
}

// ------------------------------------------------------------------------------------------
// build, deterministic: the keys are sorted by region (stable radix sort: inside a region they keep the caller's order), then
// every key knows its rank in its region from its neighbours: the first 7 take the slots in that order, the others stay in the
// generic table.  No atomics on the table, and the same input gives the same bytes and the same overflow keys every time.
// ------------------------------------------------------------------------------------------
__global__ void k_mz_regions(const uint64_t *keys, uint64_t n, int k, int m, uint32_t n_regions, uint64_t *reg) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reg[i] = mz_region(mz_minimizer(keys[i], k, m), n_regions);
}

#if GS_MZ_RD
// GS_MZ_RD build: one entry (region << 32 | key index) per key and strand pick.  A read shows a k-mer in either orientation
// and the probe takes the leftmost smallest t-mer as shown: the key sits in the region of the pick in its canonical
// orientation (entry i) and, where the other strand picks another region (equal t-mer orders at two places of the k-mer: a
// few per cent of the keys), in that one too (entry n + i; ~0 = none).  Sorted on all 64 bits: a region's keys in the caller's
// order, the same bytes every time.
__device__ __forceinline__ uint64_t mz_revcomp(uint64_t c, int k) {
    uint64_t r = 0;
    for (int i = 0; i < k; i++) {
        r = (r << 2) | (3ULL - (c & 3ULL));
        c >>= 2;
    }
    return r;
}
__global__ void k_mz_entries(const uint64_t *keys, uint64_t n, int k, int m, uint32_t n_regions, uint64_t *ent, unsigned long long *n_dup) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t c = keys[i];
    const uint32_t a = mz_region(mz_minimizer(c, k, m), n_regions), b = mz_region(mz_minimizer(mz_revcomp(c, k), k, m), n_regions);
    ent[i] = ((uint64_t)a << 32) | i;
    ent[n + i] = a != b ? ((uint64_t)b << 32) | i : ~0ULL;
    if (a != b) atomicAdd(n_dup, 1ULL);
}
__global__ void k_mz_split_entries(const uint64_t *ent, uint64_t n2, uint64_t *reg, uint32_t *idx) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n2) {
        reg[p] = ent[p] >> 32;
        idx[p] = (uint32_t)ent[p];
    }
}
#endif

__global__ void k_mz_place(unsigned long long *slots, uint32_t *pay, int k, int locbits, const uint64_t *keys, const uint32_t *loc,
                           const uint32_t *payload, const uint64_t *reg_sorted, const uint32_t *idx_sorted, uint64_t n,
                           unsigned long long *overflow, uint32_t *overflow_list, uint64_t overflow_cap) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint64_t reg = reg_sorted[p];
    uint32_t rank = 0;  // keys of this region before this one, counted up to a full region
    while (rank < (uint32_t)kMzSlots && p > rank && reg_sorted[p - 1 - rank] == reg) rank++;
    // the region overflows when an 8th key exists: then it keeps 6 keys, and its last word is a filter of the others
    const bool over = rank >= (uint32_t)kMzSlots || (p + (kMzSlots - rank) < n && reg_sorted[p + (kMzSlots - rank)] == reg);
    const uint32_t i = idx_sorted[p];
    const MzKey key = mz_key(keys[i], k, locbits);
    unsigned long long *base = slots + reg * kMzWords;
    uint8_t *fpb = reinterpret_cast<uint8_t *>(base);
    if (rank == 0) {  // the count byte: 0xFF - keys of the region, counted up to kMzSlots + 1 (= some of them did not fit)
        uint32_t len = 1;
        while (len <= (uint32_t)kMzSlots && p + len < n && reg_sorted[p + len] == reg) len++;
        fpb[0] = (uint8_t)(0xFFu - len);
    }
    if (over && rank >= (uint32_t)kMzSlots - 1u) {  // the key stays in the generic table; the filter learns it
        atomicAnd(base + kMzSlots, ~mz_filter_bits(key.tag));  // (the word starts as all ones: a CLEARED bit = "seen")
        const unsigned long long o = atomicAdd(overflow, 1ULL);
        if (o < overflow_cap) overflow_list[o] = i;
        return;
    }
    const uint32_t slot = 1u + rank;
    base[slot] = (key.tag << locbits) | (unsigned long long)loc[i];
    pay[reg * kMzWords + slot] = payload[i];
    fpb[slot] = (uint8_t)key.fp;
}

// bit 7 of every byte of x that equals the byte replicated in T (exact: no borrow between bytes)
__device__ __forceinline__ uint32_t mz_byte_eq(uint32_t x, uint32_t T) {
    const uint32_t z = x ^ T;
    return ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z | 0x7F7F7F7Fu);
}
struct MzCand {  // fingerprint matches of a lookup: bit 7 of byte b of half j = slot 4 j + b
    uint32_t m0, m1;
    __device__ __forceinline__ bool any() const { return (m0 | m1) != 0u; }
    __device__ __forceinline__ uint32_t pop() {  // lowest matching slot, removed
        const bool lo = m0 != 0u;
        const uint32_t x = lo ? m0 : m1;
        const uint32_t bit = (uint32_t)__ffs((int)x) - 1u;
        const uint32_t y = x & (x - 1u);
        m0 = lo ? y : m0;
        m1 = lo ? m1 : y;
        return (lo ? 0u : 4u) + (bit >> 3);
    }
};
// F = the region's fingerprint word (byte 0: 0xFF - keys stored, byte i: fingerprint of slot i or 0xFF)
__device__ __forceinline__ MzCand mz_candidates(uint64_t F, uint32_t fp) {
    const uint32_t T = __builtin_amdgcn_perm(0u, fp, 0u);  // byte 0 of fp in all four bytes
    MzCand c;
    c.m0 = mz_byte_eq((uint32_t)F, T) & 0x80808000u;  // byte 0 is the count
    c.m1 = mz_byte_eq((uint32_t)(F >> 32), T);
    return c;
}
__device__ __forceinline__ MzCand mz_candidates_T(uint64_t F, uint32_t T) {  // T: the fingerprint in all four bytes
    MzCand c;
    c.m0 = mz_byte_eq((uint32_t)F, T) & 0x80808000u;
    c.m1 = mz_byte_eq((uint32_t)(F >> 32), T);
    return c;
}
// some key of this region lives in the generic table only (the count byte says "more than kMzSlots")
__device__ __forceinline__ bool mz_region_overflowed(uint64_t F) { return ((uint32_t)F & 0xFFu) == 0xFFu - (uint32_t)(kMzSlots + 1); }

// payload of a key that is known to be in the table (k_emit, one emitted row at a time)
__device__ __forceinline__ uint32_t mz_payload(const MzView &mz, const TableView &tab, uint64_t c) {
    const uint32_t reg = mz_region(mz_minimizer(c, mz.k, mz.m), mz.n_regions);
    const MzKey key = mz_key(c, mz.k, mz.locbits);
    const uint64_t want = key.tag << mz.locbits, locmask = (1ULL << mz.locbits) - 1ULL;
    const uint64_t *rp = reinterpret_cast<const uint64_t *>(mz.regions) + (uint64_t)reg * kMzWords;
    MzCand cd = mz_candidates(rp[0], key.fp);
    while (cd.any()) {
        const uint32_t slot = cd.pop();
        if ((rp[slot] ^ want) < locmask) return mz.pay[(uint64_t)reg * kMzWords + slot];
    }
    uint64_t hh = slot_of(c, tab.shift);  // a key of a full region lives in the generic table
    Slot sl = load_slot(tab.slots + hh);
    while (sl.key != c && sl.key != kEmptyKey) {
        hh = (hh + 1) & tab.mask;
        sl = load_slot(tab.slots + hh);
    }
    return sl.payload;
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
    uint32_t o;     // order (mz_tmer_order) of the canonical t-mer that starts there
};
template <bool PLACE = false>
__device__ __forceinline__ MzWin mz_win_from(uint64_t f, uint64_t r, int k, int m, uint64_t mmask);
__host__ __device__ __forceinline__ uint32_t mzp_mmer_word(uint32_t fm, uint32_t rm, int m);
__host__ __device__ constexpr int mzp_t_of(int k, int m);
// the window whose first base is base (lane & 31) of the 64-base string a:b
__device__ __forceinline__ MzWin mz_win_of(uint64_t a, uint64_t b, int lane, int k, int m, uint64_t kmask, uint64_t mmask) {
    const int sh = 2 * (lane & 31);
    const uint64_t x = sh ? ((a << sh) | (b >> (64 - sh))) : a;  // 32 bases from the window's first one
    return mz_win_from<false>(x >> (64 - 2 * k), ~revpairs(x) & kmask, k, m, mmask);  // (revpairs: base i of x, complemented, at bits 2i+1:2i)
}
// ... from the window's forward and reverse-complement codes
template <bool PLACE>
__device__ __forceinline__ MzWin mz_win_from(uint64_t f, uint64_t r, int k, int m, uint64_t mmask) {
    MzWin o;
    o.f = f;
    o.r = r;
#if GS_MZ_PLACE
    if constexpr (PLACE) {  // (m <= 15) bij(canonical m-mer) << 1 | orientation: gs_mzp.h
        o.h = mzp_mmer_word((uint32_t)(o.f >> (2 * (k - m))), (uint32_t)o.r & (uint32_t)mmask, m);
    } else
#endif
    if (m <= 16) {  // wave-uniform: the m-mer fits a word
        const uint32_t fm = (uint32_t)(o.f >> (2 * (k - m))), rm = (uint32_t)o.r & (uint32_t)mmask;
        o.h = mz_mmer_hash((uint64_t)(fm < rm ? fm : rm));
    } else {
        const uint64_t fm = o.f >> (2 * (k - m)), rm = o.r & mmask;
        o.h = mz_mmer_hash(fm < rm ? fm : rm);
    }
#if GS_MZ_PLACE
    const int t = PLACE ? mzp_t_of(k, m) : mz_t_of(k, m);
#else
    const int t = mz_t_of(k, m);
#endif
    const uint32_t tmask = (1u << (2 * t)) - 1u;
    const uint32_t ft = (uint32_t)(o.f >> (2 * (k - t))), rt = (uint32_t)o.r & tmask;
    o.o = mz_tmer_order(ft < rt ? ft : rt);
    return o;
}
// step s (wave-uniform, 0..32) of a tile whose 66 words are held as W0 (word = lane) and W1 (word 64 + (lane & 1))
__device__ __forceinline__ MzWin mz_window(uint64_t W0, uint64_t W1, int s, int lane, int k, int m, uint64_t kmask, uint64_t mmask) {
    const uint64_t A = mz_word(W0, W1, 2 * s), B = mz_word(W0, W1, 2 * s + 1), C = mz_word(W0, W1, 2 * s + 2);
    const bool hsel = lane >= 32;
    return mz_win_of(hsel ? B : A, hsel ? C : B, lane, k, m, kmask, mmask);
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
        uint32_t fp[kMzU];
        uint4 bk[kMzU];
        bool valid[kMzU];
        // phase A: minimiser, region, fingerprint of kMzU steps; their fingerprint loads leave together
#pragma unroll
        for (int u = 0; u < kMzU; u++) {
            const uint32_t s = s0 + (uint32_t)u;
            valid[u] = false;
            addr[u] = mz.regions;
            want[u] = 0;
            fp[u] = 0;
            bk[u] = make_uint4(~0u, ~0u, ~0u, ~0u);
            if (s >= nsteps) continue;  // wave-uniform
            const MzWin nxt = mz_window(W0, W1, (int)s + 1, lane, k, m, kmask, mmask);
            const uint64_t c = cur.f < cur.r ? cur.f : cur.r;
            const MzKey key = mz_key(c, k, locbits);
            addr[u] = mz.regions + (uint64_t)mz_region(mz_minimizer(c, k, m), mz.n_regions) * kMzPieces;  // (the rule, lane by lane)
            want[u] = key.tag << locbits;
            fp[u] = key.fp;
            valid[u] = s * 64u + (uint32_t)lane < td.nwin_left;
            if (valid[u]) bk[u] = *addr[u];
            cur = nxt;
        }
        // phase B: the slots whose fingerprint matches (dependent loads), the generic table for full regions, bookkeeping
#pragma unroll
        for (int u = 0; u < kMzU; u++) {
            const uint32_t s = s0 + (uint32_t)u;
            if (s >= nsteps) break;  // wave-uniform
            const uint64_t Fw = (uint64_t)bk[u].x | ((uint64_t)bk[u].y << 32);  // (the same load brought slot 1 along)
            MzCand cd = mz_candidates(Fw, fp[u]);
            bool hit = false;
            uint64_t e = 0;
            while (__ballot(valid[u] && !hit && cd.any())) {
                if (valid[u] && !hit && cd.any()) {
                    const uint64_t sv = reinterpret_cast<const uint64_t *>(addr[u])[cd.pop()];
                    const uint64_t e2 = sv ^ want[u];
                    if (e2 < locmask) {
                        hit = true;
                        e = e2;
                    }
                }
            }
            // the region overflowed, none of its keys is this one, and its filter word does not rule the key out: the generic table
            if (valid[u] && !hit && mz_region_overflowed(Fw) &&
                mz_filter_maybe(reinterpret_cast<const uint64_t *>(addr[u])[kMzSlots], mz_filter_bits(want[u] >> locbits))) {
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

// The exact per-lane form of a tile (every table load its own request, walks and generic lookups inline): the fallback of
// k_probe_mz for tiles that do not fit its LDS budgets, and the reference the fast form is tested against.
__device__ __forceinline__ void mz_tile_exact(const MzView &mz, const TableView &tab, const uint64_t *__restrict__ words,
                                              const TileDesc td, uint32_t tile, uint32_t n_tiles, const TileArrays &ta, uint32_t *cand,
                                              uint32_t cand_cap, uint32_t *g_cand_count, uint32_t *s_cand) {
    const int lane = threadIdx.x & 63;
    MzTileOut o = mz_tile<0>(mz, tab, words, td, s_cand);
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
            for (uint32_t i = lane; i < o.n_cand; i += 64u) cand[off + i] = s_cand[i];
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

__global__ __launch_bounds__(256) void k_probe_mz_exact(MzView mz, TableView tab, const uint64_t *__restrict__ words,
                                                        const TileDesc *__restrict__ tdesc, uint32_t n_tiles, TileArrays ta,
                                                        uint32_t *cand, uint32_t cand_cap, uint32_t *g_cand_count) {
    __shared__ uint32_t s_cand[4][kMzCandLds];
    const int wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * 4u + (uint32_t)wave;
    if (tile >= n_tiles) return;
    mz_tile_exact(mz, tab, words, tdesc[tile], tile, n_tiles, ta, cand, cand_cap, g_cand_count, s_cand[wave]);
}

#include "gs_mzp.h"

// ------------------------------------------------------------------------------------------
// k_probe_mz: the fast form.  One wavefront per tile, lane = window, and per step of 64 windows:
//
//   stage    minimiser -> region of every window; a RUN = adjacent lanes with the same region (its first lane is the
//            leader).  The whole 64-byte region of every run is fetched ONCE, straight into LDS:
//            run by LDS-DMA (global_load_lds_dwordx4): 4 lanes x 16 bytes per run, 16 runs per wave-instruction (runs
//            past the last one repeat it: same line, no new request).  One request per run is what the fabric sees.
//   consume  (kMzStages - 1 steps later, behind a counted vmcnt) a window reads its region's fingerprint word from LDS,
//            then the slots whose fingerprint matches.  Outcome: hit (loc), miss, or -- the region's count byte says that keys
//            of it did not fit (mz_region_overflowed), none of its 6 is this one and its filter word does not rule the key out
//            -- "ask the generic table", which is deferred: such a window takes a place in the tile's hit list like a
//            hit does.  A step with more than kMzRunCap runs (rare) is consumed in two halves, the second fetched on the spot.
//   flush    when the hit list or the deferred queue might not hold one more step, and after the last step: the deferred
//            lookups go to the generic table 64 at a time (their latencies overlap), then one ordered pass over the dense hit
//            list does the bookkeeping exactly as mz_tile does, and the list starts over (the small list is what lets 24
//            waves share a CU's LDS: 6.5 KB each, 79 VGPRs).
//
// A tile with more candidates (location changes) than the LDS buffer holds (kMzCandLds) is done over by mz_tile_exact:
// exactness never depends on the budgets.
// ------------------------------------------------------------------------------------------
constexpr int kMzRunCap = 32;                  // runs of a step whose regions fit the LDS cache slot (2 KiB)
constexpr int kMzDma = kMzRunCap * kMzPieces / 64;  // LDS-DMA wave-instructions per step
constexpr int kMzStages = 2;                   // steps whose region fetches are in flight together
#if GS_MZ_RCW
constexpr int kMzListCap = 176;                // maybe-hits held before the list is flushed (a step adds up to 64)
constexpr int kMzDeferCap = 80;                // deferred generic-table lookups held before a flush
#else
constexpr int kMzListCap = 240;                // maybe-hits held before the list is flushed (a step adds up to 64)
constexpr int kMzDeferCap = 96;                // deferred generic-table lookups held before a flush
#endif
constexpr int kMzStreamDwords = 2 * (kTileWords + 3);  // a tile's bases as 32-bit pieces: 64 words + 2 a window reaches into + 1 of padding
constexpr int kMzFastCand = 64;                // candidates (location changes) of a tile before it is handed to the exact form
static_assert(kMzListCap >= kMzCandLds, "the exact form borrows the hit list as its candidate buffer");
constexpr int kMzWaves = 2;                    // waves per workgroup
// a list entry = (window offset in the tile: 16 bits, loc + 1: 32 bits, 0 = no hit after all) in two parallel arrays, so that
// tables with any number of location groups the slot format holds (2^27 for k = 20) take this kernel -- a human assembly has
// millions of groups (one per stretch of SUNKs with gaps <= k), a diploid one (BASELINE.json configs[3]) one per SNP
constexpr uint32_t kMzPending = 0xFFFFFFFFu;   // loc of an entry whose generic-table lookup is still to come

struct MzWaveLds {
    uint4 cache[kMzStages][kMzRunCap * kMzPieces];
    uint32_t list[kMzListCap];              // loc + 1 of every maybe-hit (the exact form borrows this as its candidate buffer)
    uint32_t defer[kMzDeferCap];
    uint32_t cand[kMzFastCand];             // the tile's candidates (a tile with more is done over by the exact form)
#if GS_MZ_RCW
  union {
    uint64_t words[kTileWords + 3];         // (instances without compile-time k: the words as they are)
    struct {
    // The tile's packed bases twice, as streams of dwords in BASE order (dword d = bases 16 d .. 16 d + 15, first base most
    // significant): fd = the bases as they are (one dword of padding in front: the window extraction of lane 0 addresses it),
    // rd = the reverse complement of the whole 67-word buffer, so that the reverse complement of the window at base P is the
    // piece of rd at base (67 * 32 - k - P) -- computed ONCE per tile (one revpairs per word) instead of once per window.
    uint32_t fd[1 + kMzStreamDwords];
    uint32_t rd[1 + kMzStreamDwords];
    };
  };
#else
    uint64_t words[kTileWords + 3];         // the tile's packed bases (+ 2 words a window of the last step reaches into, + 1 that
                                            // step's look-ahead read touches and never uses)
#endif
    uint16_t list_w[kMzListCap];            // window offset of every maybe-hit
#ifdef GS_MZ_PAD_LDS
    uint32_t pad_lds[GS_MZ_PAD_LDS / 4];    // (occupancy experiment: fewer waves per CU)
#endif
};

// LDS accesses of the main loop are written as asm: while an LDS-DMA is in flight hipcc puts `s_waitcnt vmcnt(0)` in
// front of every LDS access it knows of (it cannot tell which bytes the DMA writes), which would serialise the steps.
// The counted vmcnt waits in the loop are what orders these reads behind the DMA that fills them.
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
__device__ __forceinline__ uint4 lds_read_b128(uint32_t addr) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ uint64_t lds_read_b64(uint32_t addr) {
    uint64_t v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write_b32(uint32_t addr, uint32_t v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write_b16(uint32_t addr, uint32_t v) {
    asm volatile("ds_write_b16 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// the window that starts at this lane's base in step s, from the tile's 66 words staged in LDS (two 8-byte reads, one
// instruction; lanes 0..31 share one address pair, lanes 32..63 the next)
template <bool PLACE = false>
__device__ __forceinline__ MzWin mz_window_lds(uint32_t words_a, uint32_t s, int lane, int k, int m, uint64_t kmask, uint64_t mmask) {
    uint4 v;
    const uint32_t addr = words_a + (2u * s + (uint32_t)(lane >> 5)) * 8u;
    asm volatile("ds_read2_b64 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    const uint64_t a = (uint64_t)v.x | ((uint64_t)v.y << 32), b = (uint64_t)v.z | ((uint64_t)v.w << 32);
    const int sh = 2 * (lane & 31);
    const uint64_t x = sh ? ((a << sh) | (b >> (64 - sh))) : a;  // 32 bases from the window's first one
    return mz_win_from<PLACE>(x >> (64 - 2 * k), ~revpairs(x) & kmask, k, m, mmask);
}

#if GS_MZ_RCW
// ---- windows from the dword streams ------------------------------------------------------------------------------------------
// The 64 bits that start at bit B of a stream of dwords d[0], d[1], .. are alignbit(d[q], d[q+1], n) : alignbit(d[q+1], d[q+2], n)
// with q = (B - 1) >> 5 and n = (32 - (B - 32 q)) & 31 (v_alignbit_b32 takes 0..31: a piece that starts ON a dword boundary is
// addressed from the dword before it, shift 0 -- hence the dword of padding in front of fd).  For the window that starts at base
// 64 s + lane, q moves by 4 dwords per step and n never changes: both are per-lane constants of the tile.
struct MzLaneGeom {
    uint32_t f_addr, r_addr;  // LDS byte address of d[q] at step 0 (forward: + 16 per step; reverse complement: - 16 per step)
    uint32_t f_n, r_n;        // alignbit shift
};
__device__ __forceinline__ MzLaneGeom mz_lane_geom(uint32_t fd_a, uint32_t rd_a, int lane, int k) {
    MzLaneGeom g;
    const int bf = 2 * lane, qf = lane ? (bf - 1) >> 5 : -1;
    g.f_addr = fd_a + 4u * (uint32_t)(1 + qf);
    g.f_n = (uint32_t)(32 - (bf - 32 * qf)) & 31u;
    const int br = 2 * ((kTileWords + 3) * 32 - k - lane), qr = (br - 1) >> 5;  // >= 1 at every step of a tile
    g.r_addr = rd_a + 4u * (uint32_t)qr;
    g.r_n = (uint32_t)(32 - (br - 32 * qr)) & 31u;
    return g;
}
__device__ __forceinline__ uint32_t mz_alignbit(uint32_t hi, uint32_t lo, uint32_t n) { return __builtin_amdgcn_alignbit(hi, lo, n); }
// three consecutive dwords at a 4-byte aligned LDS address
__device__ __forceinline__ void lds_read3_b32(uint32_t addr, uint32_t &a, uint32_t &b, uint32_t &c) {
    uint2 ab;
    asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read_b32 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ab), "=&v"(c) : "v"(addr) : "memory");
    a = ab.x;
    b = ab.y;
}
// the window that starts at this lane's base in step s (k > 16: the code's high part is 2k - 32 bits of the first dword pair)
template <bool PLACE = false>
__device__ __forceinline__ MzWin mz_window_rcw(const MzLaneGeom &g, uint32_t s, int k, int m, uint64_t mmask) {
    uint32_t a, b, c, ra, rb, rc;
    lds_read3_b32(g.f_addr + 16u * s, a, b, c);
    lds_read3_b32(g.r_addr - 16u * s, ra, rb, rc);
    const uint32_t fh = mz_alignbit(a, b, g.f_n), fl = mz_alignbit(b, c, g.f_n);      // 64 bits from the window's first base
    const uint32_t rh = mz_alignbit(ra, rb, g.r_n), rl = mz_alignbit(rb, rc, g.r_n);  // 64 bits from its reverse complement's
    const uint32_t sh = (uint32_t)(64 - 2 * k);  // 1 .. 31
    const uint64_t f = ((uint64_t)(fh >> sh) << 32) | mz_alignbit(fh, fl, sh);
    const uint64_t r = ((uint64_t)(rh >> sh) << 32) | mz_alignbit(rh, rl, sh);
    return mz_win_from<PLACE>(f, r, k, m, mmask);
}
#endif

// lane i <- lane i - 1, lane 0 <- fill
__device__ __forceinline__ uint32_t mz_shift_up(uint32_t x, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

// ---- the sliding leftmost / rightmost minimum over the NT t-mers of every window -----------------------------------------
// One 32-bit value per t-mer position = two 16-bit halves, order << 7 | position: the low half with the position counted up
// (its minimum = the leftmost smallest t-mer), the high half with it counted down (the rightmost one); v_pk_min_u16 works on
// both at once.  Positions are in the frame of a step: lane + 32, so that the step before (-64) and the step after (+64)
// fit the 7 bits too.  E[T] = minimum over t-mers T - NT + 1 .. T, by doubling (levels of spans 1, 2, 4 ..; lanes are shifted
// by wave_shr:1 or ds_bpermute), the low lanes taking what they need from the same level of the step before.
__device__ __forceinline__ uint32_t mz_pk_min(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
constexpr uint32_t kMzPrevFrame = 0x003FFFC0u;  // + : a value of the step before in this step's frame (low half - 64, high + 64)
constexpr uint32_t kMzNextFrame = 0xFFC00040u;  // + : a value of the step after in this step's frame
__device__ __forceinline__ uint32_t mz_pk_of(uint32_t order, int lane) {
    return order * 0x00800080u + (uint32_t)(lane + 32) + ((uint32_t)(95 - lane) << 16);
}
constexpr int mz_log2_floor(int x) { return x < 2 ? 0 : 1 + mz_log2_floor(x / 2); }

template <int NT>  // NT > 0: compile-time count of t-mers per window;  NT = 0: taken at run time (shifts of one lane)
struct MzSlide {
    static constexpr int L = mz_log2_floor(NT ? NT : 1);
    static constexpr int kLevels = NT ? L + ((1 << L) != NT ? 1 : 0) : 1;
    uint32_t b[kLevels];  // inputs of every level, of the step this was last advanced to
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int l = 0; l < kLevels; l++) b[l] = 0xFFFFFFFFu;  // "no step before": never the minimum
    }
    // advance to the step whose packed t-mer values are n0: returns its E
    __device__ __forceinline__ uint32_t advance(uint32_t n0, int lane, int nt_runtime) {
        uint32_t x = n0;
        if constexpr (NT == 0) {
            const uint32_t prev = b[0] + kMzPrevFrame;
            b[0] = n0;
            uint32_t y = n0;
            for (int j = 1; j < nt_runtime; j++) {
                y = mz_shift_up(y, lane_value(prev, 64 - j));
                x = mz_pk_min(x, y);
            }
        } else {
#pragma unroll
            for (int l = 0; l < kLevels; l++) {
                const int sft = l < L ? (1 << l) : NT - (1 << L);
                const uint32_t prev = b[l] + kMzPrevFrame;
                b[l] = x;
                uint32_t y;
                if (sft == 1) {
                    y = mz_shift_up(x, lane_value(prev, 63));
                } else {
                    const uint32_t comb = lane >= 64 - sft ? prev : x;
                    y = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane - sft) & 63) * 4, (int)comb);
                }
                x = mz_pk_min(x, y);
            }
        }
        return x;
    }
};

// The region hash of the window that starts at this lane in the step `cur` is the window of (e_cur = its E; `nxt` = the
// step after; the slide is advanced to it).  The smallest t-mer of a window ends nt - 1 positions on (in the next step for
// the last lanes), leftmost or rightmost as the canonical strand reads; the m-mer at (its offset mod w) is the one.
template <int WT, int NT>  // WT: w at compile time or 0
__device__ __forceinline__ uint32_t mz_pick(MzSlide<NT> &slide, uint32_t &e_cur, const MzWin &cur, const MzWin &nxt, bool rc_strand,
                                           int lane, int w, int nt, uint32_t *pm_out = nullptr) {
    const uint32_t e_nxt = slide.advance(mz_pk_of(nxt.o, lane), lane, nt);
    const uint32_t ew = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + nt - 1) & 63) * 4,
                                                               (int)(lane < nt - 1 ? e_nxt + kMzNextFrame : e_cur));
    e_cur = e_nxt;
#if GS_MZ_RD
    (void)rc_strand;
    const uint32_t at = (ew & 0x7Fu) - 32u;  // lane of the chosen (leftmost smallest) t-mer (64.. = next step)
#else
    const uint32_t at = rc_strand ? 95u - ((ew >> 16) & 0x7Fu) : (ew & 0x7Fu) - 32u;  // lane of the chosen t-mer (64.. = next step)
#endif
    const uint32_t xoff = at - (uint32_t)lane;                                          // 0 .. nt - 1
    // (xoff < nt: with a compile-time w and at most 2 w t-mers per window one conditional subtraction is the remainder)
    const uint32_t pm = WT == 4 ? (xoff & 3u) : (WT && NT && NT <= 2 * WT) ? xoff - (xoff >= (uint32_t)WT ? (uint32_t)WT : 0u)
                                                                         : xoff % (uint32_t)(WT ? WT : w);
    if (pm_out) *pm_out = pm;
    // the m-mer hash of lane + pm: of this step, or (last lanes) of the next one
    const int src = (int)(((uint32_t)lane + pm) & 63u) * 4;
    const uint32_t h_here = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)cur.h);
    const uint32_t h_next = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)nxt.h);
    return (uint32_t)lane + pm < 64u ? h_here : h_next;
}

// the m-mer hash of lane + pm: of this step, or (last lanes) of the next one
__device__ __forceinline__ uint32_t mz_hash_at(uint32_t pm, const MzWin &cur, const MzWin &nxt, int lane) {
    const int src = (int)(((uint32_t)lane + pm) & 63u) * 4;
    const uint32_t h_here = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)cur.h);
    const uint32_t h_next = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)nxt.h);
    return (uint32_t)lane + pm < 64u ? h_here : h_next;
}
#if GS_MZ_RD
// ---- 16 t-mers per window (w = 4: k = 20 with m = 17): the sliding minimum inside the rows of 16 lanes ----------------------
// A window's t-mers are lanes i .. i + 15 = the rest of its row of 16 lanes and the beginning of the next row.  With P = the
// minimum over a row's lanes up to and including this one, S = over this lane to the end of the row (4 DPP row shifts each:
// no LDS round trip, no wait), the window's minimum is min(S[i], P[i + 15]) -- P of the next row, or, for the last row, of the
// NEXT step's first row.  Value = order << 7 | lane: the smaller lane wins a tie (the leftmost t-mer as the read shows it).
struct MzRowSlide {
    uint32_t P, S;  // of the step the slide was last advanced to
    __device__ __forceinline__ void set(uint32_t order, int lane) {
        uint32_t p = (order << 7) | (uint32_t)lane, q;
        // v_min_u32 with a DPP source, in place: a lane whose source lies outside its row of 16 is switched off for that
        // instruction and keeps its value.  (Written out: the compiler makes a copy, a DPP move and a minimum of every level.
        // A DPP read needs two wait states after the write of its source: the two chains alternate, one s_nop between levels.)
        asm("v_mov_b32 %1, %0\n\t"
            "s_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_min_u32_dpp %1, %1, %1 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_min_u32_dpp %1, %1, %1 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_min_u32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_min_u32_dpp %1, %1, %1 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1"
            : "+v"(p), "=&v"(q));
        P = p;
        S = q;
    }
};
// the m-mer offset (0..3) of the window that starts at this lane in the step the slide stands at, given the slide of the next
// step; lane of the chosen t-mer = e & 0x7F (64.. = in the next step)
template <int W = 4>
__device__ __forceinline__ uint32_t mz_row_pick(const MzRowSlide &cur, const MzRowSlide &nxt, int lane, int w_rt = 4) {
    const uint32_t comb = lane < 16 ? nxt.P + 64u : cur.P;
    const uint32_t q = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + 15) & 63) * 4, (int)comb);
    const uint32_t e = min(cur.S, (lane & 15) ? q : 0xFFFFFFFFu);
    const uint32_t xoff = (e & 0x7Fu) - (uint32_t)lane;  // 0 .. 15
    if constexpr (W == 4) return xoff & 3u;
    else if constexpr (W == 0) return xoff % (uint32_t)w_rt;
    else return xoff - (uint32_t)W * (xoff >= (uint32_t)W) - (uint32_t)W * (xoff >= 2u * (uint32_t)W);  // (W >= 6: at most two wraps)
}
#endif

// self-test: the sliding pick against the rule applied lane by lane (mz_minimizer), on every window of the given tiles
template <int KT, int MT>
__global__ __launch_bounds__(64) void k_mz_rule_check(int k_rt, int m_rt, const uint64_t *__restrict__ words, uint32_t n_tiles,
                                                      unsigned long long *mismatches, uint32_t *first_bad) {
    __shared__ uint64_t s_words[kTileWords + 3];
    const int lane = threadIdx.x;
    const int k = KT ? KT : k_rt, m = MT ? MT : m_rt, w = k - m + 1;
    constexpr int NT = KT ? KT - mz_t_of(KT ? KT : 20, KT ? MT : 17) + 1 : 0;
    const int nt = k - mz_t_of(k, m) + 1;
    const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
    const uint64_t mmask = (1ULL << (2 * m)) - 1ULL;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();
        s_words[lane] = words[(uint64_t)tile * kTileWords + lane];
        if (lane < 3) s_words[64 + lane] = words[(uint64_t)tile * kTileWords + 64 + lane];
        __syncthreads();
        const uint32_t words_a = lds_addr(&s_words[0]);
        MzWin cur = mz_window_lds(words_a, 0, lane, k, m, kmask, mmask);
        constexpr bool ROW = GS_MZ_RD && NT == 16;
        MzSlide<NT> slide;
        slide.reset();
        uint32_t e_cur = 0;
#if GS_MZ_RD
        MzRowSlide row_cur;
        if constexpr (ROW) row_cur.set(cur.o, lane);
        else
#endif
            e_cur = slide.advance(mz_pk_of(cur.o, lane), lane, nt);
        for (uint32_t s = 0; s < 32u; s++) {
            const MzWin nxt = mz_window_lds(words_a, s + 1u, lane, k, m, kmask, mmask);
            const bool rc_strand = cur.r < cur.f;
            uint32_t H;
#if GS_MZ_RD
            if constexpr (ROW) {
                MzRowSlide row_nxt;
                row_nxt.set(nxt.o, lane);
                H = mz_hash_at(mz_row_pick(row_cur, row_nxt, lane), cur, nxt, lane);
                row_cur = row_nxt;
            } else
#endif
                H = mz_pick<KT ? KT - MT + 1 : 0>(slide, e_cur, cur, nxt, rc_strand, lane, w, nt);
            // the rule applied to this one k-mer: in its canonical orientation, or (GS_MZ_RD) as the read shows it
            const uint32_t want = mz_minimizer(GS_MZ_RD ? cur.f : (rc_strand ? cur.r : cur.f), k, m);
            if (H != want) {
                atomicAdd(&first_bad[4 + lane], 1u);  // per-lane histogram
                if (atomicAdd(mismatches, 1ULL) == 0ULL) {
                    first_bad[0] = tile;
                    first_bad[1] = s * 64u + (uint32_t)lane;
                    first_bad[2] = H;
                    first_bad[3] = want;
                }
            }
            cur = nxt;
        }
    }
}

// KT / MT: k and m as compile-time constants (0 = take them from the table); FMT: slot format (mz_key_fmt)
// PLACE: the place-record format of gs_mzp.h (FMT is then unused)
template <int KT, int MT, int FMT, bool PLACE = false>
__global__ __launch_bounds__(64 * kMzWaves, 6) void k_probe_mz(MzView mz, TableView tab, const uint64_t *__restrict__ words,
                                                           const TileDesc *__restrict__ tdesc, uint32_t n_tiles, TileArrays ta,
                                                           uint32_t *cand, uint32_t cand_cap, uint32_t *g_cand_count,
                                                           uint32_t *g_fallback_tiles) {
    __shared__ MzWaveLds s_lds[kMzWaves];
    // (the wave index through readfirstlane: the compiler then knows that the tile, its descriptor and every loop bound
    // derived from it are wave-uniform, and keeps them and the control flow on the scalar unit)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t tile = blockIdx.x * (uint32_t)kMzWaves + (uint32_t)wave;
    if (tile >= n_tiles) return;
    MzWaveLds &L = s_lds[wave];
    const TileDesc td = tdesc[tile];
    const int k = KT ? KT : mz.k, m = MT ? MT : mz.m, w = k - m + 1, locbits = mz.locbits;
#if GS_MZ_PLACE
    constexpr int NT = !KT ? 0 : PLACE ? KT - mzp_t_of(KT ? KT : 20, KT ? MT : 15) + 1 : KT - mz_t_of(KT ? KT : 20, KT ? MT : 17) + 1;
    const int nt = k - (PLACE ? mzp_t_of(k, m) : mz_t_of(k, m)) + 1;  // t-mers per window
#else
    constexpr int NT = KT ? KT - mz_t_of(KT ? KT : 20, KT ? MT : 17) + 1 : 0;
    const int nt = k - mz_t_of(k, m) + 1;  // t-mers per window
#endif
    const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
    const uint64_t mmask = (1ULL << (2 * m)) - 1ULL;
    const uint64_t locmask = (1ULL << locbits) - 1ULL;
    const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    // the tile's 64 words + the two after them (the buffer carries that much slack after the last read), staged in LDS
    constexpr bool RCW = GS_MZ_RCW && KT > 16;
    if constexpr (RCW) {
#if GS_MZ_RCW
        for (int j = lane; j < kTileWords + 3; j += 64) {  // (one more round for lanes 0..2)
            const uint64_t wv = j < kTileWords + 2 ? __builtin_nontemporal_load(words + td.wbase + j) : 0ULL;
            const uint64_t rv = ~revpairs(wv);  // the word's reverse complement: word (66 - j) of the reversed stream
            L.fd[1 + 2 * j] = (uint32_t)(wv >> 32);
            L.fd[2 + 2 * j] = (uint32_t)wv;
            L.rd[2 * (kTileWords + 2 - j)] = (uint32_t)(rv >> 32);
            L.rd[2 * (kTileWords + 2 - j) + 1] = (uint32_t)rv;
        }
        if (lane == 0) {
            L.fd[0] = 0u;
            L.rd[kMzStreamDwords] = 0u;
        }
#endif
    } else {
        L.words[lane] = __builtin_nontemporal_load(words + td.wbase + lane);
        if (lane < 3) L.words[64 + lane] = lane < 2 ? __builtin_nontemporal_load(words + td.wbase + 64 + lane) : 0ULL;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t nsteps = min((td.nwin_left + 63u) / 64u, 32u);
    const uint32_t list_a = lds_addr(&L.list[0]), listw_a = lds_addr(&L.list_w[0]), defer_a = lds_addr(&L.defer[0]), words_a = lds_addr(&L.words[0]);

    // per in-flight step (slot = step % kMzStages): the fingerprint to look for, where the lane's region is in the cache slot,
    // the key / tag to look for, lane r = region of run r, and the run count
    // (PLACE: fpT = the fp word (fp << (32 - s)), want = tag word | mask word << 32)
    uint32_t fpT[kMzStages], regoff[kMzStages], runreg[kMzStages], nruns[kMzStages];  // fpT: fingerprint in all four bytes, bit 31 of regoff: invalid lane
    uint64_t want[kMzStages];
#if GS_MZ_RCW
    const MzLaneGeom geom = mz_lane_geom(lds_addr(&L.fd[0]), lds_addr(&L.rd[0]), lane, k);
    auto window = [&](uint32_t s) {
        if constexpr (RCW) return mz_window_rcw<PLACE>(geom, s, k, m, mmask);
        else return mz_window_lds<PLACE>(words_a, s, lane, k, m, kmask, mmask);
    };
#else
    auto window = [&](uint32_t s) { return mz_window_lds<PLACE>(words_a, s, lane, k, m, kmask, mmask); };
#endif
    MzWin cur = window(0u);
    constexpr bool ROW = GS_MZ_RD && NT == 16;  // (the sliding minimum inside the rows of 16 lanes)
    MzSlide<NT> slide;
    slide.reset();
    uint32_t e_cur = 0;
#if GS_MZ_RD
    MzRowSlide row_cur;
    if constexpr (ROW) row_cur.set(cur.o, lane);
    else
#endif
        e_cur = slide.advance(mz_pk_of(cur.o, lane), lane, nt);  // E of the step `cur` is the window of

    // LDS-DMA wave-instructions that fetch the regions of (the first kMzRunCap of) R runs: 16 runs each
    constexpr uint32_t kRunsPerDma = 64 / kMzPieces;
    auto dma_count = [](uint32_t R) { return min((R + kRunsPerDma - 1u) / kRunsPerDma, (uint32_t)kMzDma); };
    // the regions of runs [r0, r0 + kMzRunCap) of a step into cache slot `slot`
    auto fetch = [&](int slot, uint32_t rr_lane, uint32_t R, uint32_t r0) {
        const uint32_t ni = dma_count(R - r0);
#pragma unroll
        for (int it = 0; it < kMzDma; it++) {
            if ((uint32_t)it >= ni) break;  // wave-uniform
            const uint32_t q = min(r0 + (uint32_t)(it * (int)kRunsPerDma + lane / kMzPieces), R - 1u);
            const uint32_t rr = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(q * 4u), (int)rr_lane);
            const uint4 *src = mz.regions + (uint64_t)rr * kMzPieces + (lane % kMzPieces);
            __builtin_amdgcn_global_load_lds(src, &L.cache[slot][it * 64], 16, 0, 0);
        }
    };
    // FULL (compile-time): all 2048 windows of the tile exist (every tile of a read but its last), no lane is ever invalid
    auto stage = [&](uint32_t s, int slot, auto full_tile) {
        constexpr bool FULL = decltype(full_tile)::value;
        const MzWin nxt = window(s + 1u);
        const bool rc_strand = cur.r < cur.f;
        uint32_t H, pm = 0;
#if GS_MZ_RD
        if constexpr (ROW) {
            MzRowSlide row_nxt;
            row_nxt.set(nxt.o, lane);
            pm = mz_row_pick<(KT && KT - MT + 1 >= 6) || KT - MT + 1 == 4 ? KT - MT + 1 : 0>(row_cur, row_nxt, lane, w);
            H = mz_hash_at(pm, cur, nxt, lane);
            row_cur = row_nxt;
        } else
#endif
            H = mz_pick<KT ? KT - MT + 1 : 0>(slide, e_cur, cur, nxt, rc_strand, lane, w, nt, &pm);
        uint32_t reg, fpw;
        uint64_t wantw;
#if GS_MZ_PLACE
        if constexpr (PLACE) {
            // H = bij(canonical m-mer) << 1 | orientation of the m-mer at lane + pm; the k-mer oriented like its m-mer, the
            // m-mer's offset in it, the flank bases this window knows (gs_mzp.h)
            const uint32_t y = H >> 1;
            const bool rev = H & 1u;
            const uint32_t p = rev ? (uint32_t)(w - 1) - pm : pm;
            uint32_t tw, mk;
            mzp_tag(rev ? cur.r : cur.f, p, k, m, tw, mk);
            reg = y >> mz.s;
            fpw = mz.s ? (y << (32 - mz.s)) : 0u;
            wantw = (uint64_t)tw | ((uint64_t)mk << 32);
            cur = nxt;
        } else
#endif
        {
            const uint64_t c = rc_strand ? cur.r : cur.f;
            cur = nxt;
            const MzKey key = mz_key_fmt<FMT>(c, k, locbits);
            reg = mz_region(H, mz.n_regions);
            fpw = key.fp * 0x01010101u;
            wantw = key.tag << locbits;
        }
        const bool valid = FULL || s * 64u + (uint32_t)lane < td.nwin_left;
        const bool leader = valid && (lane == 0 || reg != mz_shift_up(reg, ~reg));
        const unsigned long long lb = __ballot(leader);
        const uint32_t R = (uint32_t)__popcll(lb);  // >= 1: lane 0 of an existing step is valid
        const uint32_t myrun = (uint32_t)__popcll(lb & (lt | (1ULL << lane))) - 1u;
        fpT[slot] = fpw;
        regoff[slot] = myrun * (uint32_t)(kMzWords * 8) | (valid ? 0u : 0x80000000u);  // byte offset of the run's region in the cache slot
        want[slot] = wantw;
        nruns[slot] = R;
        // lane r <- region of run r (leaders push; the others push to lane 63, which holds a run only when all 64 lanes lead)
        runreg[slot] = (uint32_t)__builtin_amdgcn_ds_permute((int)((leader ? myrun : 63u) * 4u), (int)reg);
        fetch(slot, runreg[slot], R, 0u);
    };

    uint32_t lcount = 0, dcount = 0;     // wave-uniform: entries of the list / of the deferred queue
    // the windows of step s whose run is in [r0, r0 + kMzRunCap): their regions are in cache slot u
    // BIG (compile-time): the step has more runs than a cache slot holds and is looked up in two halves
    auto lookups = [&](uint32_t s, int u, uint32_t r0, auto full_tile, auto big_step) {
        constexpr bool FULL = decltype(full_tile)::value, BIG = decltype(big_step)::value;
        const uint32_t off = regoff[u] & 0x7FFFFFFFu, myrun = off / (uint32_t)(kMzWords * 8);
        const bool mine = (FULL || !(regoff[u] >> 31)) && (!BIG || (myrun >= r0 && myrun < r0 + (uint32_t)kMzRunCap));
        const uint32_t reg_a = lds_addr(&L.cache[u][0]) + (FULL ? regoff[u] : off) - r0 * (uint32_t)(kMzWords * 8);
        bool hit = false, defer = false;
        uint64_t e = 0;
#if GS_MZ_PLACE
        if constexpr (PLACE) {
            // the region's 7 tag words against this window's (TW, MK): the record that matches (at most one can), its loc word
            const uint32_t tw = (uint32_t)want[u], mk = (uint32_t)(want[u] >> 32);
            uint4 q0 = make_uint4(0u, 0u, 0u, 0u), q1 = q0, q2 = q0, q3 = q0;
            if ((FULL && !BIG) || mine) {
                q0 = lds_read_b128(reg_a);
                q1 = lds_read_b128(reg_a + 16u);
                q2 = lds_read_b128(reg_a + 32u);
                q3 = lds_read_b128(reg_a + 48u);
            }
            uint32_t sel = 0xFFFFFFFFu;  // (tw has no bit outside mk)
            sel = (q0.z & mk) == tw ? q2.y : sel;
            sel = (q0.w & mk) == tw ? q2.z : sel;
            sel = (q1.x & mk) == tw ? q2.w : sel;
            sel = (q1.y & mk) == tw ? q3.x : sel;
            sel = (q1.z & mk) == tw ? q3.y : sel;
            sel = (q1.w & mk) == tw ? q3.z : sel;
            sel = (q2.x & mk) == tw ? q3.w : sel;
            const int sft = 32 - mz.s;
            hit = mine && sel != 0xFFFFFFFFu && (!mz.s || ((sel ^ fpT[u]) >> sft) == 0u);
            e = mz.s ? (uint64_t)(sel & ((1u << sft) - 1u)) : (uint64_t)sel;
            defer = mine && !hit && (q0.x & kMzpOverflow);  // keys of this region in the generic table
        } else
#endif
        {
        uint64_t F = ~0ULL;
        if (FULL && !BIG)
            F = lds_read_b64(reg_a);
        else if (mine)
            F = lds_read_b64(reg_a);
        MzCand cd = mz_candidates_T(F, fpT[u]);
        // the first slot whose fingerprint matches: the hit, nearly always (a second candidate needs a fingerprint collision)
        {
            const bool lo = cd.m0 != 0u;
            const uint32_t x = lo ? cd.m0 : cd.m1;
            const bool any = mine && x != 0u;
            if (any) {
                const uint32_t slot = (lo ? 0u : 4u) + (((uint32_t)__ffs((int)x) - 1u) >> 3);
                const uint64_t e2 = lds_read_b64(reg_a + slot * 8u) ^ want[u];
                if (e2 < locmask) {
                    hit = true;
                    e = e2;
                }
            }
            const bool more = any && !hit && (((x & (x - 1u)) | (lo ? cd.m1 : 0u)) != 0u);
            if (__ballot(more)) {  // (wave-uniform, rare) walk the other candidates
                if (any) cd.pop();
                while (__ballot(mine && !hit && cd.any())) {
                    if (mine && !hit && cd.any()) {
                        const uint64_t e2 = lds_read_b64(reg_a + cd.pop() * 8u) ^ want[u];
                        if (e2 < locmask) {
                            hit = true;
                            e = e2;
                        }
                    }
                }
            }
        }
        defer = mine && !hit && mz_region_overflowed(F);  // generic table, later -- unless the region's filter word says no
        if (defer) defer = mz_filter_maybe(lds_read_b64(reg_a + (uint32_t)kMzSlots * 8u), mz_filter_bits(want[u] >> locbits));
        }
        const bool mb_mine = hit || defer;
        const unsigned long long mb = __ballot(mb_mine), db = __ballot(defer);
        if (mb) {
            // (a big step calls this twice, lower lanes first: its entries still land in lane order)
            const uint32_t idx = lcount + (uint32_t)__popcll(mb & lt);
            if (mb_mine) {
                lds_write_b32(list_a + idx * 4u, hit ? (uint32_t)e + 1u : kMzPending);
                lds_write_b16(listw_a + idx * 2u, s * 64u + (uint32_t)lane);
            }
            if (defer) lds_write_b32(defer_a + (dcount + (uint32_t)__popcll(db & lt)) * 4u, (s * 64u + (uint32_t)lane) | (idx << 16));
            lcount += (uint32_t)__popcll(mb);
            dcount += (uint32_t)__popcll(db);
        }
    };

    // flush: the deferred lookups held so far go to the generic table 64 at a time (their latencies overlap), then one ordered
    // pass over the (dense) hit list does the bookkeeping exactly as mz_tile does -- 64 maybe-hits at a time, whatever steps
    // they came from -- and the list starts over.  Runs when a step might not fit any more, and after the last step.
    uint32_t n_raw = 0, n_cand = 0, first_loc = GS_NONE, carry = GS_NONE;  // of the tile so far
    auto flush = [&]() {
#ifdef GS_MZ_COUNT_DEFER  // (experiment: the fallback-tile counter counts deferred generic-table lookups instead)
        if (lane == 0 && dcount) atomicAdd(g_fallback_tiles, dcount);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j0 = 0; j0 < dcount; j0 += 64u) {
            if (j0 + (uint32_t)lane < dcount) {
                const uint32_t ent = L.defer[j0 + lane];
                const uint32_t wi = ent & 0xFFFFu;
                uint64_t a, b2;
                if constexpr (RCW) {
#if GS_MZ_RCW
                    a = ((uint64_t)L.fd[1 + 2 * (wi >> 5)] << 32) | L.fd[2 + 2 * (wi >> 5)];
                    b2 = ((uint64_t)L.fd[3 + 2 * (wi >> 5)] << 32) | L.fd[4 + 2 * (wi >> 5)];
#endif
                } else {
                    a = L.words[wi >> 5];
                    b2 = L.words[(wi >> 5) + 1];
                }
                const uint64_t c = canon_at(a, b2, ~revpairs(a), ~revpairs(b2), (int)(wi & 31u), k, kmask);
                uint64_t hh = slot_of(c, tab.shift);
                Slot sl = load_slot(tab.slots + hh);
                while (sl.key != c && sl.key != kEmptyKey) {
                    hh = (hh + 1) & tab.mask;
                    sl = load_slot(tab.slots + hh);
                }
                L.list[ent >> 16] = sl.key == c ? sl.loc + 1u : 0u;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i0 = 0; i0 < lcount; i0 += 64u) {
            const bool have = i0 + (uint32_t)lane < lcount;
            const uint32_t lv = have ? L.list[i0 + (uint32_t)lane] : 0u, woff = have ? (uint32_t)L.list_w[i0 + (uint32_t)lane] : 0u;
            const bool hit = lv != 0u;
            const uint32_t loc = lv - 1u;
            const unsigned long long hb = __ballot(hit);
            if (!hb) continue;
            const unsigned long long below = hb & lt;
            const int src = below ? 63 - __clzll((long long)below) : 0;
            const uint32_t pl = __shfl(loc, src);
            const uint32_t prev = below ? pl : carry;
            const bool chg = hit && loc != prev;
            const unsigned long long cb = __ballot(chg);
            const uint32_t ci = n_cand + (uint32_t)__popcll(cb & lt);
            if (chg && ci < (uint32_t)kMzFastCand) L.cand[ci] = ((n_raw + (uint32_t)__popcll(below)) << 16) | woff;
            n_cand += (uint32_t)__popcll(cb);
            if (first_loc == GS_NONE) first_loc = lane_value(loc, __ffsll((unsigned long long)hb) - 1);
            carry = lane_value(loc, 63 - __clzll((long long)hb));
            n_raw += (uint32_t)__popcll(hb);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        lcount = dcount = 0;
    };

    static_assert(kMzStages == 2, "the prologue below stages steps 0 and 1 by hand (slot indices must be compile-time)");
    auto steps = [&](auto full_tile) {
        constexpr std::false_type small_step{};
        constexpr std::true_type big_step{};
        stage(0, 0, full_tile);
        if (nsteps > 1) stage(1, 1, full_tile);
        for (uint32_t s0 = 0; s0 < nsteps; s0 += kMzStages) {
#pragma unroll
            for (int u = 0; u < kMzStages; u++) {
                const uint32_t s = s0 + (uint32_t)u;
                if (s >= nsteps) break;  // wave-uniform
                if (lcount + 64u > (uint32_t)kMzListCap || dcount + 64u > (uint32_t)kMzDeferCap) flush();  // (wave-uniform) room for this step
                // the fetches of step s have landed once at most those of the next step (1 to kMzDma instructions) are outstanding
                static_assert(kMzDma == 2, "the waits below count at most two instructions of the next step");
                switch (s + 1 < nsteps ? dma_count(nruns[u ^ 1]) : 0u) {  // wave-uniform
                    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                }
                if (nruns[u] <= (uint32_t)kMzRunCap) {
                    lookups(s, u, 0u, full_tile, small_step);
                } else {
                    // a big step: its windows in lane order are runs 0..31 then 32..63, and a hit list entry is written at its
                    // rank among the step's maybe-hits -- so the first half is looked up first, and the ranks line up
                    lookups(s, u, 0u, full_tile, big_step);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    fetch(u, runreg[u], nruns[u], (uint32_t)kMzRunCap);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    lookups(s, u, (uint32_t)kMzRunCap, full_tile, big_step);
                }
                if (s + kMzStages < nsteps) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads of this slot are done before it is refilled
                    stage(s + kMzStages, u, full_tile);
                }
            }
        }
    };
    if (td.nwin_left >= (uint32_t)kTileWindows)  // wave-uniform
        steps(std::true_type{});
    else
        steps(std::false_type{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    flush();
    if (n_cand > (uint32_t)kMzFastCand) {  // (wave-uniform, rare) more location changes than the candidate buffer holds
        if (lane == 0) atomicAdd(g_fallback_tiles, 1u);
#if GS_MZ_PLACE
        if constexpr (PLACE)
            mzp_tile_exact(MzpView{mz.regions, mz.pay, k, m, mz.s}, tab, words, td, tile, n_tiles, ta, cand, cand_cap, g_cand_count, L.list);
        else
#endif
            mz_tile_exact(mz, tab, words, td, tile, n_tiles, ta, cand, cand_cap, g_cand_count, L.list);
        return;
    }
    const uint32_t *cbuf = L.cand;
    const MzTileOut o{n_raw, n_cand, first_loc, carry};
    uint32_t off = tile * (uint32_t)kDirectCand;
    if (o.n_cand > (uint32_t)kDirectCand) {
        if (lane == 0) off = n_tiles * (uint32_t)kDirectCand + atomicAdd(g_cand_count, o.n_cand);
        off = __shfl(off, 0);
    }
    if (o.n_cand && (uint64_t)off + o.n_cand <= cand_cap) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < o.n_cand; i += 64u) cand[off + i] = cbuf[i];
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

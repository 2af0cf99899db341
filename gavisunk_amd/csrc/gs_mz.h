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
// minimiser (smallest m-mer hash) gives 2.5 -- a fifth fewer requests on the same table geometry.  t-mers and m-mers are
// canonical (the smaller of itself and its reverse complement), and ties go to the leftmost t-mer of the k-mer AS THE READ
// SHOWS IT: the probe never asks which strand of the k-mer is the canonical one before it picks, and the table holds a key
// under the pick of either strand (two entries when they differ: ~1 % of the keys).
//
//   x       = leftmost argmin over the k - t + 1 t-mers of the k-mer as shown of (t-mer or its rc, the smaller) * c >> 8
//   H       = mix_a(min(m-mer at x mod w, its reverse complement))
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
#define GS_MZ_PLACE 1  // 1: + the place-record format of gs_mzp.h (k <= 27: a record per place, shorter minimisers, longer runs)
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
// order (24 bits of a multiplicative hash: with the lane in the low byte it is the 32-bit value the sliding minimum of
// k_probe_mz works on -- one v_and_or from the product)
__host__ __device__ constexpr int mz_t_of(int k, int m) { return 5 + (m - 5) % (k - m + 1); }
__host__ __device__ __forceinline__ uint32_t mz_tmer_order(uint32_t ct) { return (ct * 0x9E3779B1u) >> 8; }
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

// One entry (region << 32 | key index) per key and strand pick.  A read shows a k-mer in either orientation
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
    uint32_t h;     // hash of the canonical m-mer that ENDS the window (the m-mer at offset w - 1)
    uint32_t hx;    // (place format, m > 15) its bits 32.. and, in bit 31, its orientation (m <= 15: both are in h)
    uint32_t o;     // order (mz_tmer_order) of the canonical t-mer that ENDS the window (the t-mer at offset k - t)
    uint32_t fa, ra;  // (place format, dword-stream windows only) the 16 bases from the window's first one / from its reverse complement's
};
// Why the LAST t-mer and m-mer: the window at lane i owns the t-mers that start at lanes i .. i + k - t; when lane j holds the
// t-mer that starts at base j + (k - t), those are the values of lanes i - (k - t) .. i -- every pull of the sliding minimum
// and of the pick reaches DOWN the lanes, i.e. into this step or the one before, and a step needs no look-ahead.
template <int PLACE = 0>
__device__ __forceinline__ MzWin mz_win_from(uint64_t f, uint64_t r, int k, int m, uint64_t mmask);
__host__ __device__ __forceinline__ uint32_t mzp_mmer_word(uint32_t fm, uint32_t rm, int m);
__host__ __device__ __forceinline__ uint64_t mzp_bij64(uint64_t cm, int m);
__host__ __device__ constexpr int mzp_t_of(int k, int m);
// the window whose first base is base (lane & 31) of the 64-base string a:b
__device__ __forceinline__ MzWin mz_win_of(uint64_t a, uint64_t b, int lane, int k, int m, uint64_t kmask, uint64_t mmask) {
    const int sh = 2 * (lane & 31);
    const uint64_t x = sh ? ((a << sh) | (b >> (64 - sh))) : a;  // 32 bases from the window's first one
    return mz_win_from<false>(x >> (64 - 2 * k), ~revpairs(x) & kmask, k, m, mmask);  // (revpairs: base i of x, complemented, at bits 2i+1:2i)
}
// ... from the window's forward and reverse-complement codes
template <int PLACE>
__device__ __forceinline__ MzWin mz_win_from(uint64_t f, uint64_t r, int k, int m, uint64_t mmask) {
    MzWin o;
    o.f = f;
    o.r = r;
    o.fa = o.ra = 0u;
    o.hx = 0u;
#if GS_MZ_PLACE
    if constexpr (PLACE == 2) {  // bij(canonical m-mer) on 2m <= 42 bits: low word | high bits, orientation in bit 31 of the second
        const uint64_t fm = o.f & mmask, rm = o.r >> (2 * (k - m));
        const bool rev = rm < fm;
        const uint64_t y = mzp_bij64(rev ? rm : fm, m);
        o.h = (uint32_t)y;
        o.hx = (uint32_t)(y >> 32) | (rev ? 0x80000000u : 0u);
    } else if constexpr (PLACE == 1) {  // (m <= 15) orientation << 31 | bij(canonical m-mer): gs_mzp.h
        o.h = mzp_mmer_word((uint32_t)o.f & (uint32_t)mmask, (uint32_t)(o.r >> (2 * (k - m))), m);
    } else
#endif
    if (m <= 16) {  // wave-uniform: the m-mer fits a word
        const uint32_t fm = (uint32_t)o.f & (uint32_t)mmask, rm = (uint32_t)(o.r >> (2 * (k - m)));
        o.h = mz_mmer_hash((uint64_t)(fm < rm ? fm : rm));
    } else {
        const uint64_t fm = o.f & mmask, rm = o.r >> (2 * (k - m));
        o.h = mz_mmer_hash(fm < rm ? fm : rm);
    }
#if GS_MZ_PLACE
    const int t = PLACE ? mzp_t_of(k, m) : mz_t_of(k, m);
#else
    const int t = mz_t_of(k, m);
#endif
    const uint32_t tmask = (1u << (2 * t)) - 1u;
    const uint32_t ft = (uint32_t)o.f & tmask, rt = (uint32_t)(o.r >> (2 * (k - t)));
    o.o = mz_tmer_order(ft < rt ? ft : rt);
    return o;
}
// step s (wave-uniform, 0..32) of a tile whose 66 words are held as W0 (word = lane) and W1 (word 64 + (lane & 1))
__device__ __forceinline__ MzWin mz_window(uint64_t W0, uint64_t W1, int s, int lane, int k, int m, uint64_t kmask, uint64_t mmask) {
    const uint64_t A = mz_word(W0, W1, 2 * s), B = mz_word(W0, W1, 2 * s + 1), C = mz_word(W0, W1, 2 * s + 2);
    const bool hsel = lane >= 32;
    return mz_win_of(hsel ? B : A, hsel ? C : B, lane, k, m, kmask, mmask);
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
//   stage    the window's pieces from the tile's two dword streams in LDS (forward, reverse complement: v_alignbit); the
//            sliding minimum over its t-mers and the pick (MzPicker: every pull reaches down the lanes, no look-ahead);
//            region of every window; a RUN = adjacent lanes with the same region (its first lane is the leader).  The whole
//            64-byte region of every run is fetched ONCE, straight into LDS, by LDS-DMA (global_load_lds_dwordx4): 4 lanes x
//            16 bytes per run, 16 runs per wave-instruction (runs past the last one repeat it: same line, no new request).
//            One request per run is what the fabric sees.
//   consume  (kMzStages - 1 steps later, behind a counted vmcnt) a window reads its region from LDS.  Slot format: the
//            fingerprint word, then the slots whose fingerprint matches.  Place format (gs_mzp.h): all 64 bytes, the 7 tag
//            words against the window's (tag, mask).  Outcome: hit (loc), miss, or -- the region says that keys of it live in
//            the generic table and its filter does not rule this one out -- "ask the generic table", which is deferred:
//            such a window takes a place in the tile's hit list like a hit does.  A step with more than kMzRunCap runs
//            (rare) is consumed a slot-full of runs at a time, the later ones fetched on the spot.
//   flush    when the hit list might not hold one more step, and after the last step: one ordered pass over the dense hit
//            list, 64 entries at a time -- the deferred ones among them go to the generic table together (their latencies
//            overlap) -- does the bookkeeping exactly as mz_tile does, and the list starts over.  The small list, one cache
//            slot of 28 runs and 68 VGPRs are what lets 28 waves share a CU (5.7 KB of LDS each, four to a workgroup).
//
// A tile with more candidates (location changes) than the LDS buffer holds (kMzFastCand) is done over by mz_tile_exact:
// exactness never depends on the budgets.
// ------------------------------------------------------------------------------------------
#ifndef GS_MZ_RUN_CAP
#define GS_MZ_RUN_CAP 28
#endif
#ifndef GS_MZ_FAST_CAND
#define GS_MZ_FAST_CAND 64
#endif
#ifndef GS_MZ_MIN_WAVES
#define GS_MZ_MIN_WAVES 7  // waves per SIMD the register allocation leaves room for (the LDS per wave decides how many run)
#endif
constexpr int kMzRunCap = GS_MZ_RUN_CAP;       // runs of a step whose regions fit the LDS cache slot (64 bytes each)
constexpr int kMzDma = (kMzRunCap * kMzPieces + 63) / 64;  // LDS-DMA wave-instructions per step (the last one may fill only part of its 16 runs)
constexpr int kMzStages = 2;                   // steps whose region fetches are in flight together
#ifndef GS_MZ_LIST_CAP
#define GS_MZ_LIST_CAP 136
#endif
constexpr int kMzListCap = GS_MZ_LIST_CAP;     // maybe-hits (hits and deferred lookups) held before the list is flushed (a step adds up to 64)
constexpr int kMzStreamDwords = 2 * (kTileWords + 3);  // a tile's bases as 32-bit pieces: 64 words + 2 the last windows reach into + 1 of padding
constexpr int kMzFastCand = GS_MZ_FAST_CAND;   // candidates (location changes) of a tile before it is handed to the exact form
static_assert(kMzListCap >= kMzCandLds, "the exact form borrows the hit list as its candidate buffer");
#ifndef GS_MZ_DMA_AUX
#define GS_MZ_DMA_AUX 0  // cache policy bits of the region fetches (A/B: 2 = nt, 1 = sc0, 16 = sc1)
#endif
#ifndef GS_MZ_WG_WAVES
#define GS_MZ_WG_WAVES 4
#endif
constexpr int kMzWaves = GS_MZ_WG_WAVES;       // waves per workgroup (each with its own MzWaveLds: what decides how many fit a CU's 160 KB)
// a list entry = (window offset in the tile: 16 bits; loc: 32 bits, kMzPending = no hit after all) in two parallel arrays, so that
// tables with any number of location groups the slot format holds (2^27 for k = 20) take this kernel -- a human assembly has
// millions of groups (one per stretch of SUNKs with gaps <= k), a diploid one (BASELINE.json configs[3]) one per SNP
constexpr uint32_t kMzPending = 0xFFFFFFFFu;   // an entry whose generic-table lookup is still to come, or that turned out no hit

struct MzWaveLds {
    uint4 cache[kMzStages][kMzRunCap * kMzPieces];
    uint32_t list[kMzListCap];              // loc of every maybe-hit (the exact form borrows this as its candidate buffer)
    uint32_t cand[kMzFastCand];             // the tile's candidates (a tile with more is done over by the exact form)
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
// (the lanes where p holds; HIP's __ballot takes an int: the compiler turns the flag into 0 / 1 and compares again)
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// lane is set in `mask` ? a : b
__device__ __forceinline__ uint32_t select_by_mask(unsigned long long mask, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}
// lanes below this one that are set in `mask`, + acc
__device__ __forceinline__ uint32_t mbcnt(unsigned long long mask, uint32_t acc = 0u) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, acc));
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

// the window that starts at this lane's base in step s (-1 .. 31), from the tile's words staged in LDS (two 8-byte reads, one
// instruction; lanes 0..31 share one address pair, lanes 32..63 the next).  Step -1 = the 64 bases before the tile: its
// high lanes' windows end inside the tile (what they read in front of the words is never looked at).
template <int PLACE = 0>
__device__ __forceinline__ MzWin mz_window_lds(uint32_t words_a, int s, int lane, int k, int m, uint64_t kmask, uint64_t mmask) {
    uint4 v;
    const uint32_t addr = words_a + (uint32_t)((2 * s + (lane >> 5)) * 8);
    asm volatile("ds_read2_b64 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    const uint64_t a = (uint64_t)v.x | ((uint64_t)v.y << 32), b = (uint64_t)v.z | ((uint64_t)v.w << 32);
    const int sh = 2 * (lane & 31);
    const uint64_t x = sh ? ((a << sh) | (b >> (64 - sh))) : a;  // 32 bases from the window's first one
    return mz_win_from<PLACE>(x >> (64 - 2 * k), ~revpairs(x) & kmask, k, m, mmask);
}

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
// the window that starts at this lane's base in step s (-1 .. 31; k > 16: the code's high part is 2k - 32 bits of the first dword pair)
template <int PLACE = 0>
__device__ __forceinline__ MzWin mz_window_rcw(const MzLaneGeom &g, int s, int k, int m, uint64_t mmask) {
    uint2 ab, rab;
    uint32_t c, rc;
    // (both streams' reads leave together; one wait)
    asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read_b32 %1, %4 offset:8\n\t"
                 "ds_read2_b32 %2, %5 offset1:1\n\tds_read_b32 %3, %5 offset:8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(ab), "=&v"(c), "=&v"(rab), "=&v"(rc)
                 : "v"(g.f_addr + (uint32_t)(16 * s)), "v"(g.r_addr - (uint32_t)(16 * s))
                 : "memory");
    const uint32_t fh = mz_alignbit(ab.x, ab.y, g.f_n), fl = mz_alignbit(ab.y, c, g.f_n);      // 64 bits from the window's first base
    const uint32_t rh = mz_alignbit(rab.x, rab.y, g.r_n), rl = mz_alignbit(rab.y, rc, g.r_n);  // 64 bits from its reverse complement's
    const uint32_t sh = (uint32_t)(64 - 2 * k);  // 1 .. 31
#if GS_MZ_PLACE
    if constexpr (PLACE == 1) {
        // (k <= 21, m <= 15, t <= 12: everything the stage needs is a piece of one dword -- the forward code's low word holds
        //  the t-mer and the m-mer that end the window, the reverse complement's first dword their reverse complements)
        MzWin o;
        o.hx = 0u;
        const uint32_t flo = mz_alignbit(fh, fl, sh);
        o.f = (uint64_t)flo;  // (low word only: what the tag takes its right flank from)
        o.r = (uint64_t)mz_alignbit(rh, rl, sh);
        o.fa = fh;
        o.ra = rh;
        o.h = mzp_mmer_word(flo & (uint32_t)mmask, rh >> (32 - 2 * m), m);
        const int t = mzp_t_of(k, m);
        const uint32_t ft = flo & ((1u << (2 * t)) - 1u), rt = rh >> (32 - 2 * t);
        o.o = mz_tmer_order(ft < rt ? ft : rt);
        return o;
    }
#endif
    const uint64_t f = ((uint64_t)(fh >> sh) << 32) | mz_alignbit(fh, fl, sh);
    const uint64_t r = ((uint64_t)(rh >> sh) << 32) | mz_alignbit(rh, rl, sh);
    return mz_win_from<PLACE>(f, r, k, m, mmask);
}

// lane i <- lane i - 1, lane 0 <- fill
__device__ __forceinline__ uint32_t mz_shift_up(uint32_t x, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

// ---- the sliding leftmost minimum over the NT t-mers of every window -------------------------------------------------------
// Lane j of a step holds V[j] = order << 8 | (j + 32) of the t-mer that starts at base j + NT - 1 of the step (MzWin::o), so
// the window at lane i owns V[i - NT + 1 .. i] and the smaller lane wins a tie: the leftmost smallest t-mer as the read shows
// the k-mer.  Lanes below 0 are the step before: its values, kept from the last call, are the same numbers - 64.
//   NT = 16 (w = 4): inside the rows of 16 lanes.  P = minimum over a row's lanes up to this one, S = from this one to the end
//            of the row (4 DPP row shifts each: no LDS round trip, no wait); E[i] = min(P[i], S[i - 15]) -- one lane pull.
//   other compile-time NT: X2 = pairs (wave_shr:1), X4 = fours (one pull by 2), then E[i] = min of X4[i - d] over d = 0, 4, 8 ..
//            and NT - 4: pulls that do not depend on each other, one wait.
//   NT = 0 (k, m only known at run time): NT - 1 shifts of one lane.
constexpr uint32_t kMzInf = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t mz_value_of(uint32_t order, int lane) { return (order << 8) | (uint32_t)(lane + 32); }  // (the lane in a byte of its own: one SDWA operand)
// lane i <- x of lane i - d; lanes i < d <- `before` (already in this step's frame) of lane 64 + i - d
__device__ __forceinline__ uint32_t mz_pull(uint32_t x, uint32_t before, int d, int lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(((lane - d) & 63) * 4, (int)(lane >= 64 - d ? before : x));
}
__device__ __forceinline__ void mz_row_minima(uint32_t v, uint32_t &P, uint32_t &S) {
    uint32_t p = v, q;
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
template <int NT>
struct MzSlide {
    uint32_t a, b, c;  // of the step before: (NT = 16) c = S;  (else) a = V, b = X2, c = X4;  (NT = 0) a = V
    __device__ __forceinline__ void reset() { a = b = c = kMzInf; }  // "no step before": never the minimum
    // V of this step in, E of this step out
    __device__ __forceinline__ uint32_t advance(uint32_t v, int lane, int nt_runtime) {
        if constexpr (NT == 16) {
            uint32_t P, S;
            mz_row_minima(v, P, S);
            const uint32_t e = min(P, mz_pull(S, c - 64u, 15, lane));  // (lane 15 of a row pulls its own row's whole minimum: harmless)
            c = S;
            return e;
        } else if constexpr (NT >= 8) {
            const uint32_t x2 = min(v, mz_shift_up(v, lane_value(a, 63) - 64u));
            a = v;
            const uint32_t x4 = min(x2, mz_pull(x2, b - 64u, 2, lane));
            b = x2;
            const uint32_t before = c - 64u;
            c = x4;
            constexpr int n4 = NT / 4, tail = NT % 4 ? NT - 4 : 0;
            uint32_t y[5];
#pragma unroll
            for (int j = 1; j < n4; j++) y[j] = mz_pull(x4, before, 4 * j, lane);
            if constexpr (tail != 0) y[0] = mz_pull(x4, before, tail, lane);
            uint32_t e = x4;
#pragma unroll
            for (int j = 1; j < n4; j++) e = min(e, y[j]);
            if constexpr (tail != 0) e = min(e, y[0]);
            return e;
        } else {
            static_assert(NT == 0, "compile-time NT: 8 or more t-mers per window");
            const uint32_t before = a - 64u;
            a = v;
            uint32_t x = v, y = v;
            for (int j = 1; j < nt_runtime; j++) {
                y = mz_shift_up(y, lane_value(before, 64 - j));
                x = min(x, y);
            }
            return x;
        }
    }
};

// What a step's windows pick.  `before`: the step before this one (for step 0 of a tile: mz_picker_start), `win`: this step's
// windows.  -> the m-mer's offset in the window (0 .. w - 1) and the m-mer hash / word (MzWin::h) of the m-mer there.
template <int WT, int NT, bool WIDE = false>  // WT: w at compile time or 0; WIDE: the m-mer word is two registers (MzWin::h, hx)
struct MzPicker {
    MzSlide<NT> slide;
    uint32_t h_before, hx_before;
    uint32_t c_xoff, c_src4;  // per-lane constants (kept opaque: the compiler would rather re-derive them from the lane every step)
    // the step before a tile's first one: only the lanes whose t-mer / m-mer lies inside the tile count
    __device__ __forceinline__ void start(const MzWin &before, int lane, int w, int nt) {
        slide.reset();
        (void)slide.advance(lane >= 64 - (nt - 1) ? mz_value_of(before.o, lane) : kMzInf, lane, nt);
        h_before = before.h;
        hx_before = before.hx;
        c_xoff = (uint32_t)(lane + 33 - nt);
        c_src4 = 4u * (uint32_t)(lane + 64 - (w - 1));
        asm volatile("" : "+v"(c_xoff), "+v"(c_src4));
    }
    __device__ __forceinline__ uint32_t pick(const MzWin &win, int lane, int w, int nt, uint32_t &pm_out, uint32_t *hx_out = nullptr) {
        const uint32_t e = slide.advance(mz_value_of(win.o, lane), lane, nt);
        // the chosen t-mer is V of lane (e & 255) - 32 = the t-mer at offset that + (nt - 1) - lane of this lane's window
        const uint32_t xoff = (e & 0xFFu) - c_xoff;  // 0 .. nt - 1
        uint32_t pm;
        if constexpr (WT == 4) pm = xoff & 3u;
        else if constexpr (WT != 0 && NT != 0 && NT <= 2 * WT) pm = min(xoff, xoff - (uint32_t)WT);  // (unsigned: the wrapped difference is large)
        else if constexpr (WT != 0 && NT != 0 && NT <= 3 * WT) pm = min(min(xoff, xoff - (uint32_t)WT), xoff - 2u * (uint32_t)WT);
        else pm = xoff % (uint32_t)(WT ? WT : w);
        pm_out = pm;
        // its m-mer starts at offset pm: the m-mer that ENDS the window of lane + pm - (w - 1), of this step or the one before
        const uint32_t src4 = c_src4 + 4u * pm;  // 4 (lane + 64 - (w - 1) + pm): + 64, never negative
        // (ds_bpermute takes the lane from address bits 7:2: no mask needed)
        const uint32_t h_here = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src4, (int)win.h);
        const uint32_t h_prev = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src4, (int)h_before);
        h_before = win.h;
        if constexpr (WIDE) {
            const uint32_t x_here = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src4, (int)win.hx);
            const uint32_t x_prev = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src4, (int)hx_before);
            hx_before = win.hx;
            *hx_out = src4 >= 256u ? x_here : x_prev;
        }
        return src4 >= 256u ? h_here : h_prev;
    }
};

// self-test: the sliding pick against the rule applied lane by lane (mz_minimizer), on every window of the given tiles
template <int KT, int MT>
__global__ __launch_bounds__(64) void k_mz_rule_check(int k_rt, int m_rt, const uint64_t *__restrict__ words, uint32_t n_tiles,
                                                      unsigned long long *mismatches, uint32_t *first_bad) {
    __shared__ uint64_t s_words[2 + kTileWords + 3];  // (two words in front: what step -1 reads and never looks at)
    const int lane = threadIdx.x;
    const int k = KT ? KT : k_rt, m = MT ? MT : m_rt, w = k - m + 1;
    constexpr int NT = KT ? KT - mz_t_of(KT ? KT : 20, KT ? MT : 17) + 1 : 0;
    const int nt = k - mz_t_of(k, m) + 1;
    const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
    const uint64_t mmask = (1ULL << (2 * m)) - 1ULL;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();
        s_words[2 + lane] = words[(uint64_t)tile * kTileWords + lane];
        if (lane < 3) s_words[2 + 64 + lane] = words[(uint64_t)tile * kTileWords + 64 + lane];
        if (lane < 2) s_words[lane] = 0x5555555555555555ULL * (uint64_t)(lane + 1);
        __syncthreads();
        const uint32_t words_a = lds_addr(&s_words[2]);
        MzPicker<KT ? KT - MT + 1 : 0, NT> picker;
        picker.start(mz_window_lds(words_a, -1, lane, k, m, kmask, mmask), lane, w, nt);
        for (int s = 0; s < 32; s++) {
            const MzWin cur = mz_window_lds(words_a, s, lane, k, m, kmask, mmask);
            uint32_t pm;
            const uint32_t H = picker.pick(cur, lane, w, nt, pm);
            // the rule applied to this one k-mer as the read shows it
            const uint32_t want = mz_minimizer(cur.f, k, m);
            if (H != want) {
                atomicAdd(&first_bad[4 + lane], 1u);  // per-lane histogram
                if (atomicAdd(mismatches, 1ULL) == 0ULL) {
                    first_bad[0] = tile;
                    first_bad[1] = (uint32_t)s * 64u + (uint32_t)lane;
                    first_bad[2] = H;
                    first_bad[3] = want;
                }
            }
        }
    }
}

#ifdef GS_TUNING
// (tuning build only: tools/probe_census.py) what the region fetches of k_probe_mz's place format find -- per step of a
// non-big step: [0] steps, [1] runs (= region fetches), [2] runs in which some window hits, [3] runs without a hit in which some
// window is deferred to the generic table, [4] runs whose region holds no record and no overflow mark at all, [5] runs whose
// m-mer is absent (no record of the region carries its fp; hot regions not counted), [6] runs that meet a hot region,
// [7] windows deferred, [8] windows that hit, [9] valid windows
__device__ unsigned long long g_mz_census[16];
#endif
// KT / MT: k and m as compile-time constants (0 = take them from the table); FMT: slot format (mz_key_fmt)
// PLACE: 0 = per-key slots; the place-record format of gs_mzp.h (FMT is then unused): 1 = m <= 15 (the minimiser's value and
// orientation in one register), 2 = longer minimisers (k = 22 .. 27: two registers)
template <int KT, int MT, int FMT, int PLACE = 0>
__global__ __launch_bounds__(64 * kMzWaves, GS_MZ_MIN_WAVES) void k_probe_mz(MzView mz, TableView tab, const uint64_t *__restrict__ words,
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
    // the tile's 64 words + the two after them (the buffer carries that much slack after the last read), staged in LDS
    constexpr bool RCW = KT > 16;  // (compile-time k: the tile staged as two dword streams; else the words as they are)
    if constexpr (RCW) {
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
    } else {
        L.words[lane] = __builtin_nontemporal_load(words + td.wbase + lane);
        if (lane < 3) L.words[64 + lane] = lane < 2 ? __builtin_nontemporal_load(words + td.wbase + 64 + lane) : 0ULL;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t nsteps = min((td.nwin_left + 63u) / 64u, 32u);
    const uint32_t list_a = lds_addr(&L.list[0]), listw_a = lds_addr(&L.list_w[0]), words_a = lds_addr(&L.words[0]);

    // per in-flight step (slot = step % kMzStages): the fingerprint to look for, where the lane's region is in the cache slot,
    // the key / tag to look for, lane r = region of run r, and the run count
    // (PLACE: fpT = the fp word (fp << (32 - s)), want = tag word | mask word << 32)
    uint32_t fpT[kMzStages], regoff[kMzStages], runreg[kMzStages], nruns[kMzStages];  // fpT: fingerprint in all four bytes, bit 31 of regoff: invalid lane
    uint64_t want[kMzStages];
#ifdef GS_TUNING
    unsigned long long leaders[kMzStages];
#endif
    const MzLaneGeom geom = mz_lane_geom(lds_addr(&L.fd[0]), lds_addr(&L.rd[0]), lane, k);
    auto window = [&](int s) {
        if constexpr (RCW) return mz_window_rcw<PLACE>(geom, s, k, m, mmask);
        else return mz_window_lds<PLACE>(words_a, s, lane, k, m, kmask, mmask);
    };
    MzPicker<KT ? KT - MT + 1 : 0, NT, PLACE == 2> picker;
    picker.start(window(-1), lane, w, nt);
    uint32_t c_regbits = (uint32_t)(31 - mz.s);  // (place format) width of the region index in H; in a VGPR: the bit-field extract takes one scalar operand
    asm volatile("" : "+v"(c_regbits));

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
            if constexpr (kMzRunCap % (int)kRunsPerDma != 0) {  // (a slot that ends inside the last instruction's 16 runs: its other lanes stay out)
                if (it < kMzDma - 1 || lane < (kMzRunCap % (int)kRunsPerDma) * kMzPieces)
                    __builtin_amdgcn_global_load_lds(src, &L.cache[slot][it * 64], 16, 0, GS_MZ_DMA_AUX);
            } else {
                __builtin_amdgcn_global_load_lds(src, &L.cache[slot][it * 64], 16, 0, GS_MZ_DMA_AUX);
            }
        }
    };
    // FULL (compile-time): all 2048 windows of the tile exist (every tile of a read but its last), no lane is ever invalid
    auto stage = [&](uint32_t s, int slot, auto full_tile) {
        constexpr bool FULL = decltype(full_tile)::value;
        const MzWin win = window((int)s);
        uint32_t pm, Hx = 0u;
        const uint32_t H = picker.pick(win, lane, w, nt, pm, &Hx);
        uint32_t reg, fpw;
        uint64_t wantw;
#if GS_MZ_PLACE
        if constexpr (PLACE != 0) {
            // H (, Hx) = orientation and bij(canonical m-mer) of the m-mer at offset pm; the k-mer oriented like its m-mer, the
            // m-mer's offset in it, the flank bases this window knows (gs_mzp.h)
            const bool rev = PLACE == 2 ? (int32_t)Hx < 0 : (int32_t)H < 0;
            const uint32_t p = rev ? (uint32_t)(w - 1) - pm : pm;
            uint32_t tw, mk;
            if constexpr (RCW && PLACE == 1) mzp_tag_parts((rev ? win.ra : win.fa) >> (32 - 2 * (k - m)), (uint32_t)(rev ? win.r : win.f), p, k, m, tw, mk);
            else mzp_tag(rev ? win.r : win.f, p, k, m, tw, mk);
            if constexpr (PLACE == 2) reg = (uint32_t)((((uint64_t)(Hx & 0x7FFFFFFFu) << 32) | H) >> mz.s);  // y >> s
            else reg = __builtin_amdgcn_ubfe(H, (uint32_t)mz.s, c_regbits);  // y >> s, y = H less its orientation bit
            fpw = H << (32 - mz.s);  // y << (32 - s) (s >= 1: gs_device.hip, mzp_choose)
            wantw = (uint64_t)tw | ((uint64_t)mk << 32);
        } else
#endif
        {
            const uint64_t c = win.r < win.f ? win.r : win.f;
            const MzKey key = mz_key_fmt<FMT>(c, k, locbits);
            reg = mz_region(H, mz.n_regions);
            fpw = key.fp * 0x01010101u;
            wantw = key.tag << locbits;
        }
        const bool valid = FULL || s * 64u + (uint32_t)lane < td.nwin_left;
        // leaders: lanes whose region is another than the lane's below (lane 0 sees 0 there: it leads whatever its region,
        // being valid in every step that exists)
        const uint32_t below = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)reg, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
        const unsigned long long lb = ballot(valid && reg != below) | 1ULL;
        const uint32_t R = (uint32_t)__popcll(lb);
        // the leaders at or below this lane, less one = those among lanes 1 .. lane (v_mbcnt: the lanes below, of a mask)
        const uint32_t myrun = mbcnt(lb >> 1);
#ifdef GS_TUNING
        leaders[slot] = lb;
#endif
        fpT[slot] = fpw;
        regoff[slot] = myrun * (uint32_t)(kMzWords * 8) | (valid ? 0u : 0x80000000u);  // byte offset of the run's region in the cache slot
        want[slot] = wantw;
        nruns[slot] = R;
        // lane r <- region of run r (leaders push; the others push to lane 63, which holds a run only when all 64 lanes lead)
        runreg[slot] = (uint32_t)__builtin_amdgcn_ds_permute((int)select_by_mask(lb, myrun * 4u, 63u * 4u), (int)reg);
        fetch(slot, runreg[slot], R, 0u);
    };

    uint32_t lcount = 0;                 // wave-uniform: entries of the list
    // the windows of step s whose run is in [r0, r0 + kMzRunCap): their regions are in cache slot u
    // BIG (compile-time): the step has more runs than a cache slot holds and is looked up in two halves
    auto lookups = [&](uint32_t s, int u, uint32_t r0, auto full_tile, auto big_step) {
        constexpr bool FULL = decltype(full_tile)::value, BIG = decltype(big_step)::value;
        const uint32_t off = regoff[u] & 0x7FFFFFFFu, myrun = off / (uint32_t)(kMzWords * 8);
        const bool mine = (FULL || !(regoff[u] >> 31)) && (!BIG || (myrun >= r0 && myrun < r0 + (uint32_t)kMzRunCap));
        const uint32_t reg_a = lds_addr(&L.cache[u][0]) + (FULL ? regoff[u] : off) - r0 * (uint32_t)(kMzWords * 8);
        bool hit = false, defer = false;
        unsigned long long hitb, deferb;  // the same as wave masks (the scalar unit combines them)
        uint64_t e = 0;
#if GS_MZ_PLACE
        if constexpr (PLACE != 0) {
            // the region's 7 tag words against this window's (TW, MK): the record that matches (at most one can), its loc word
            const uint32_t tw = (uint32_t)want[u], mk = (uint32_t)(want[u] >> 32);
            uint4 q0 = make_uint4(0u, 0u, 0u, 0u), q1 = q0, q2 = q0, q3 = q0;
            if ((FULL && !BIG) || mine) {  // (the four reads leave together; one wait)
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                             "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(reg_a) : "memory");
            }
            // (tw has no bit outside mk.)  Without a match the loc word stays one whose fp bits are NOT this window's: the fp
            // test below then fails by itself (s >= 1)
            uint32_t sel = ~fpT[u];
            sel = (q0.z & mk) == tw ? q2.y : sel;
            sel = (q0.w & mk) == tw ? q2.z : sel;
            sel = (q1.x & mk) == tw ? q2.w : sel;
            sel = (q1.y & mk) == tw ? q3.x : sel;
            sel = (q1.z & mk) == tw ? q3.y : sel;
            sel = (q1.w & mk) == tw ? q3.z : sel;
            sel = (q2.x & mk) == tw ? q3.w : sel;
            // (lanes that are not `mine` read nothing: no record, no header -- neither a hit nor deferred)
            hitb = ballot(((sel ^ fpT[u]) >> (32 - mz.s)) == 0u);
            e = (uint64_t)sel;  // (fp bits and all: the flush takes them off)
            // header word 1: the filter of the region's keys in the generic table (0: it has none there)
            const uint32_t fh = mzp_filter_hash(tw, fpT[u]);
            deferb = ballot((int32_t)(q0.y << (fh >> 27)) < 0) & ~hitb;
            // hot regions (header word 0 = kMzpOverflow | kMzpHot): no records, 15 filter words -- what looked like tags is none
            const unsigned long long hotb = ballot(q0.x > kMzpOverflow);
            if (hotb) {  // wave-uniform; rare outside variant-dominated assemblies
                uint32_t wv = 0u, wv2 = 0u;
                const uint32_t fh2 = mzp_hot_second(fh);
                if (__builtin_amdgcn_inverse_ballot_w64(hotb)) {
                    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(wv), "=&v"(wv2) : "v"(reg_a + 4u * mzp_hot_word(fh)), "v"(reg_a + 4u * mzp_hot_word(fh2)) : "memory");
                }
                hitb &= ~hotb;
                deferb = (deferb & ~hotb) | (ballot((wv >> mzp_hot_bit(fh)) & (wv2 >> mzp_hot_bit(fh2)) & 1u) & hotb);
            }
#ifdef GS_TUNING
            if constexpr (!BIG) {
                const unsigned long long validb = ballot(FULL || !(regoff[u] >> 31)), lb = leaders[u] & validb;
                // this lane's run = lanes [lane, next leader): does any of them hit / get deferred
                const unsigned long long above = lane == 63 ? 0ULL : (leaders[u] & ~((2ULL << lane) - 1ULL));
                const int nxt = above ? __ffsll(above) - 1 : 64;
                const unsigned long long rmask = (nxt == 64 ? ~0ULL : ((1ULL << nxt) - 1ULL)) & ~((1ULL << lane) - 1ULL);
                const bool run_hit = (hitb & rmask) != 0ULL, run_defer = (deferb & rmask) != 0ULL;
                const uint32_t fpbits = fpT[u] >> (32 - mz.s);
                const bool none = (q0.x | q0.z | q0.w | q1.x | q1.y | q1.z | q1.w | q2.x) == 0u;
                const bool present = (q0.z && (q2.y >> (32 - mz.s)) == fpbits) || (q0.w && (q2.z >> (32 - mz.s)) == fpbits) ||
                                     (q1.x && (q2.w >> (32 - mz.s)) == fpbits) || (q1.y && (q3.x >> (32 - mz.s)) == fpbits) ||
                                     (q1.z && (q3.y >> (32 - mz.s)) == fpbits) || (q1.w && (q3.z >> (32 - mz.s)) == fpbits) ||
                                     (q2.x && (q3.w >> (32 - mz.s)) == fpbits);
                const unsigned long long c2 = ballot(run_hit) & lb, c3 = ballot(!run_hit && run_defer) & lb, c4 = ballot(none) & lb,
                                         c5 = ballot(!present) & lb & ~hotb, c6 = hotb & lb;
                if (lane == 0) {
                    atomicAdd(&g_mz_census[0], 1ULL);
                    atomicAdd(&g_mz_census[1], (unsigned long long)__popcll(lb));
                    atomicAdd(&g_mz_census[2], (unsigned long long)__popcll(c2));
                    atomicAdd(&g_mz_census[3], (unsigned long long)__popcll(c3));
                    atomicAdd(&g_mz_census[4], (unsigned long long)__popcll(c4));
                    atomicAdd(&g_mz_census[5], (unsigned long long)__popcll(c5));
                    atomicAdd(&g_mz_census[6], (unsigned long long)__popcll(c6));
                    atomicAdd(&g_mz_census[7], (unsigned long long)__popcll(deferb));
                    atomicAdd(&g_mz_census[8], (unsigned long long)__popcll(hitb));
                    atomicAdd(&g_mz_census[9], (unsigned long long)__popcll(validb));
                }
            }
#endif
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
            if (ballot(more)) {  // (wave-uniform, rare) walk the other candidates
                if (any) cd.pop();
                while (ballot(mine && !hit && cd.any())) {
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
        if constexpr (PLACE == 0) {
            hitb = ballot(hit);
            deferb = ballot(defer);
        }
        const unsigned long long mb = hitb | deferb;
        if (mb) {
            // (a big step calls this twice, lower lanes first: its entries still land in lane order)
            const uint32_t rank = mbcnt(mb);  // (the list's fill level joins the address on the scalar side)
            if (__builtin_amdgcn_inverse_ballot_w64(mb)) {
                lds_write_b32(list_a + lcount * 4u + rank * 4u, select_by_mask(hitb, (uint32_t)e, kMzPending));
                lds_write_b16(listw_a + lcount * 2u + rank * 2u, s * 64u + (uint32_t)lane);
            }
            lcount += (uint32_t)__popcll(mb);
        }
    };

    // flush: the deferred lookups held so far go to the generic table 64 at a time (their latencies overlap), then one ordered
    // pass over the (dense) hit list does the bookkeeping exactly as mz_tile does -- 64 maybe-hits at a time, whatever steps
    // they came from -- and the list starts over.  Runs when a step might not fit any more, and after the last step.
    uint32_t n_raw = 0, n_cand = 0, first_loc = GS_NONE, carry = GS_NONE;  // of the tile so far
    auto flush = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
        for (uint32_t i0 = 0; i0 < lcount; i0 += 64u) {
            const bool have = i0 + (uint32_t)lane < lcount;
            uint32_t lv = have ? L.list[i0 + (uint32_t)lane] : kMzPending;
            const uint32_t woff = have ? (uint32_t)L.list_w[i0 + (uint32_t)lane] : 0u;
            // the deferred lookups among these entries go to the generic table together (their latencies overlap)
            const bool pend = have && lv == kMzPending;
            const unsigned long long pb = ballot(pend);
            if (pb) {  // wave-uniform
#ifdef GS_MZ_COUNT_DEFER  // (experiment: the fallback-tile counter counts deferred generic-table lookups instead)
                if (lane == 0) atomicAdd(g_fallback_tiles, (uint32_t)__popcll(pb));
#endif
                if (pend) {
                    const uint32_t wi = woff;
                    uint64_t c;
                    if constexpr (RCW) {
                        // the window's code and its reverse complement's from the two dword streams (mz_lane_geom, for window wi)
                        const int bf = 2 * (int)wi, qf = (bf - 1) >> 5;  // (wi = 0: -1, the dword of padding)
                        const uint32_t nf = (uint32_t)(32 - (bf - 32 * qf)) & 31u;
                        const int br = 2 * ((kTileWords + 3) * 32 - k - (int)wi), qr = (br - 1) >> 5;
                        const uint32_t nr = (uint32_t)(32 - (br - 32 * qr)) & 31u;
                        const uint32_t fh = mz_alignbit(L.fd[1 + qf], L.fd[2 + qf], nf), fl = mz_alignbit(L.fd[2 + qf], L.fd[3 + qf], nf);
                        const uint32_t rh = mz_alignbit(L.rd[qr], L.rd[qr + 1], nr), rl = mz_alignbit(L.rd[qr + 1], L.rd[qr + 2], nr);
                        const uint64_t f = (((uint64_t)fh << 32) | fl) >> (64 - 2 * k), r = (((uint64_t)rh << 32) | rl) >> (64 - 2 * k);
                        c = f < r ? f : r;
                    } else {
                        const uint64_t a = L.words[wi >> 5], b2 = L.words[(wi >> 5) + 1];
                        c = canon_at(a, b2, ~revpairs(a), ~revpairs(b2), (int)(wi & 31u), k, kmask);
                    }
                    uint64_t hh = slot_of(c, tab.shift);
                    Slot sl = load_slot(tab.slots + hh);
                    while (sl.key != c && sl.key != kEmptyKey) {
                        hh = (hh + 1) & tab.mask;
                        sl = load_slot(tab.slots + hh);
                    }
                    lv = sl.key == c ? sl.loc : kMzPending;
                }
            }
            const bool hit = lv != kMzPending;
            const uint32_t loc = lv & (uint32_t)locmask;  // (place format: the loc word's fp bits go)
            const unsigned long long hb = ballot(hit);
            if (!hb) continue;
            const unsigned long long below = hb & lt;
            const int src = below ? 63 - __clzll((long long)below) : 0;
            const uint32_t pl = __shfl(loc, src);
            const uint32_t prev = below ? pl : carry;
            const bool chg = hit && loc != prev;
            const unsigned long long cb = ballot(chg);
            const uint32_t ci = mbcnt(cb, n_cand);
            if (chg && ci < (uint32_t)kMzFastCand) L.cand[ci] = (mbcnt(hb, n_raw) << 16) | woff;
            n_cand += (uint32_t)__popcll(cb);
            if (first_loc == GS_NONE) first_loc = lane_value(loc, __ffsll((unsigned long long)hb) - 1);
            carry = lane_value(loc, 63 - __clzll((long long)hb));
            n_raw += (uint32_t)__popcll(hb);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        lcount = 0;
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
                if (lcount + 64u > (uint32_t)kMzListCap) flush();  // (wave-uniform) room for this step
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
                    // a big step: its windows in lane order are the runs in order, and a hit list entry is written at its rank
                    // among the step's maybe-hits -- so a slot-full of runs at a time is looked up, lowest first, and the ranks line up
                    lookups(s, u, 0u, full_tile, big_step);
                    for (uint32_t r0 = (uint32_t)kMzRunCap; r0 < nruns[u]; r0 += (uint32_t)kMzRunCap) {  // wave-uniform
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        fetch(u, runreg[u], nruns[u], r0);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        lookups(s, u, r0, full_tile, big_step);
                    }
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
        if constexpr (PLACE != 0)
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

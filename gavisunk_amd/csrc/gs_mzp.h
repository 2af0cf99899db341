// "minimizer-place": the minimiser-addressed table with one record per PLACE instead of one slot per key.  Included by
// gs_device.hip after gs_mz.h (same anonymous namespace).
//
// Why.  k_probe_mz runs at the rate the memory system answers random requests (~60 G/s: profiles/r02/ubench_*.txt; the same
// wall for a table in HBM and one that fits the 256 MB last-level cache), so what is left is the number of requests per window
// = the density of the minimiser scheme, ~1 / w.  A longer run of windows per request needs a shorter minimiser (k = 20:
// m = 17 gives w = 4, density 0.31; m = 14 gives w = 7, density 0.2) -- but the shorter the m-mer, the more k-mers share it:
// a 14-mer occurs ~20 times in a human genome, and 64 bytes hold 7 keys.  They do hold 7 PLACES, though: the up to w SUNKs
// that one occurrence of a minimiser owns are substrings of one (m + 2 (w - 1))-base context, they share their location group
// (they lie within w - 1 < k bases: one merged interval, workflow/rules/defineSUNKs.smk:124-126), and their kmer.loc rows are
// consecutive.  So:
//
//   y        = bij(canonical m-mer)                      a bijection on 2m bits (k <= 21: m <= 15, one register in k_probe_mz;
//                                                        k = 22 .. 27: m = k - 6 <= 21, w = 7 -- the widest context a tag holds -- two)
//   region   = y >> s,  fp = y & (2^s - 1)               64 bytes each; s from the table size (2^(2m-s) regions, at most 2^31)
//   region   = 8 header bytes | 7 tag words | 7 loc words
//     tag    [context: 2 (w-1) bases, bits 8..] [sign: bit 7] [mask: bits 0..w-1]
//              context = the w - 1 bases left of the place's m-mer, then the w - 1 bases right of it, in the orientation in
//              which the m-mer is canonical; mask bit p = the k-mer whose m-mer sits at offset p is a SUNK; context bases no
//              SUNK of the place covers are 0 and never looked at
//     loc    fp << (32 - s) | loc id of the place
//     header word 0 bit 0 = keys of this region live in the generic side table (more than 7 places, or a place the format
//              does not take: keys of one place with different locs or rows that are not consecutive); word 1 = a 32-bit
//              filter of those keys (mzp_filter_shift).  Bit 1 (kMzpHot): the region keeps NO record, all its keys live in
//              the generic table and words 1..15 are one 480-bit filter of them (crowded minimisers of variant-dominated
//              assemblies: k_mzp_place, hot_min)
//   payload  (k_emit only) base row of the place, in a parallel array; the row of the key at offset p = base + sign * (number
//              of mask bits below p)
//
// A window (as the read shows it: the leftmost smallest t-mer) knows p, the p bases left of the m-mer and the
// w - 1 - p bases right of it -- a contiguous stretch of the context: it builds (TW, MK) once and a record matches iff
// ((tag ^ TW) & MK) == 0 -- mask bit p included.
// The builder makes sure no two records of a region can match the same window (records of different m-mers that share a region
// AND agree on a window's flank bases: one of them goes to the generic table), so the first match is the match and its fp is
// checked afterwards.  Exactness never depends on how well places merge: a record is a set of (offset, k-mer) pairs.
#pragma once

#ifndef GS_MZ_PLACE
#define GS_MZ_PLACE 1
#endif

#if GS_MZ_PLACE

constexpr int kMzpRecs = 7;            // places per region
constexpr int kMzpTagWord0 = 2;        // 32-bit word of the region where the tags start (after 8 header bytes)
constexpr int kMzpLocWord0 = 2 + kMzpRecs;
constexpr uint32_t kMzpOverflow = 1u;  // header word 0, bit 0: keys of this region live in the generic table
constexpr uint32_t kMzpHot = 2u;       // header word 0, bit 1: ALL of them do, and words 1..15 are a 480-bit filter of them (a crowded
                                       // minimiser of a variant-dominated assembly: k_mzp_place, hot_min)
constexpr int kMzpHotWords = 15;

// a bijection on 2m-bit values (multiplication by an odd constant): the region is taken from its HIGH bits, which every bit of
// the m-mer reaches; the fp = its low s bits.  m <= 15: one 32-bit multiply (the narrow form the k <= 21 kernels carry in one
// register); longer m-mers (k = 22 .. 27, w = 7: m = k - 6 <= 21): the same on 64 bits.
__host__ __device__ __forceinline__ uint32_t mzp_bij(uint32_t cm, int m) {
    const uint32_t M = m >= 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
    return (cm * 0x9E3779B1u) & M;
}
__host__ __device__ __forceinline__ uint64_t mzp_bij64(uint64_t cm, int m) {
    if (m <= 15) return (uint64_t)mzp_bij((uint32_t)cm, m);  // (one definition for every m: the narrow kernels' values are these)
    return (cm * 0x9E3779B97F4A7C15ULL) & ((1ULL << (2 * m)) - 1ULL);
}
// length of the short t-mers whose minimum anchors the pick: the mod-minimiser's (gs_mz.h: mz_t_of).  (Tried: t = k - 15, i.e. 16
// t-mers per window whatever w, so that the sliding minimum runs inside the rows of 16 lanes -- the density lost by a t-mer
// count that is no multiple of w cost more than the cheaper minimum gave: 172.7 against 211.2 Gbase/s at m = 14.)
__host__ __device__ constexpr int mzp_t_of(int k, int m) { return mz_t_of(k, m); }
// per lane: bit 31 = the canonical form is the reverse complement of the m-mer as shown, bits 0 .. 2m - 1 = bij(canonical m-mer)
// (m <= 15: the difference of the two 30-bit codes carries the comparison in its sign bit)
__host__ __device__ __forceinline__ uint32_t mzp_mmer_word(uint32_t fm, uint32_t rm, int m) {
    const uint32_t d = rm - fm;  // negative: rm < fm
    return (mzp_bij(rm < fm ? rm : fm, m) & 0x7FFFFFFFu) | (d & 0x80000000u);
}

struct MzpProbe {   // what a window looks for
    uint64_t y;     // bij(canonical m-mer) (2m bits): region = y >> s, fp = y & (2^s - 1)
    uint32_t tw, mk;
    uint32_t p;     // offset of the m-mer in the oriented k-mer
};
// (TW, MK) of the oriented k-mer X (its m-mer canonical) with the m-mer at offset p.  Bits 8.. of a tag are the place's context
// with the m-mer cut out -- the w - 1 bases left of it, then the w - 1 bases right of it -- and a window knows a CONTIGUOUS
// stretch of that: its k-mer minus the m-mer (the last p bases of the left flank, the first w - 1 - p of the right one), which
// is the k-mer's top 2 (w - 1) bits with their low 2 (w - 1 - p) bits replaced by the k-mer's own lowest bits (one v_bfi).
// xtop = X >> 2 m;  xlow = X's low 32 bits (bits above the right flank are not looked at).
__host__ __device__ __forceinline__ void mzp_tag_parts(uint32_t xtop, uint32_t xlow, uint32_t p, int k, int m, uint32_t &tw, uint32_t &mk) {
    const uint32_t nf2 = 2u * (uint32_t)(k - m);  // flank bits per side = 2 (w - 1) (<= 12)
    const uint32_t sr = 8u + 2u * p;
    const uint32_t ones_r = (1u << (nf2 + 8u - sr)) - 1u;
    const uint32_t cut = (xlow & ones_r) | (xtop & ~ones_r);
    const uint32_t bit = 1u << p;
    tw = (cut << sr) | bit;
    mk = (((1u << nf2) - 1u) << sr) | bit;
}
__host__ __device__ __forceinline__ void mzp_tag(uint64_t X, uint32_t p, int k, int m, uint32_t &tw, uint32_t &mk) {
    mzp_tag_parts((uint32_t)(X >> (2 * m)), (uint32_t)X, p, k, m, tw, mk);
}
// Header word 1 of a region whose keys did not all fit = a 32-bit filter of the ones in the generic table: the bit of a window
// (from what it looks for: tag word and fp word).  A window that finds no record asks the generic table only when its bit is
// set -- a crowded region is one that reads visit often (a minimiser of a repeat family), nearly always for k-mers that are no
// SUNKs at all.  The shift is the bit's distance from bit 31 (the lookup shifts it there and tests the sign).
__host__ __device__ __forceinline__ uint32_t mzp_filter_hash(uint32_t tw, uint32_t fpw) { return (tw ^ fpw) * 0x9E3779B1u; }
__host__ __device__ __forceinline__ uint32_t mzp_filter_shift(uint32_t tw, uint32_t fpw) { return mzp_filter_hash(tw, fpw) >> 27; }
// hot regions: word (1..15) and bit of a window in the 480-bit filter; every key sets TWO bits (h and mzp_hot_second(h)): with
// the dozens of keys such a region has, two probes of 480 bits pass a third as many strangers as one
__host__ __device__ __forceinline__ uint32_t mzp_hot_word(uint32_t h) { return 1u + (((h >> 27) * (uint32_t)kMzpHotWords) >> 5); }
__host__ __device__ __forceinline__ uint32_t mzp_hot_bit(uint32_t h) { return (h >> 22) & 31u; }
__host__ __device__ __forceinline__ uint32_t mzp_hot_second(uint32_t h) { return (h ^ (h >> 15)) * 0x85EBCA6Bu; }
__host__ __device__ __forceinline__ uint64_t mzp_revcomp(uint64_t c, int k) {
    uint64_t r = 0;
    for (int i = 0; i < k; i++) {
        r = (r << 2) | (3ULL - (c & 3ULL));
        c >>= 2;
    }
    return r;
}
// the rule applied to ONE k-mer as shown (f; its reverse complement r): build side, emit, exact form
__host__ __device__ __forceinline__ MzpProbe mzp_probe_of(uint64_t f, uint64_t r, int k, int m) {
    const int w = k - m + 1, t = mzp_t_of(k, m), nt = k - t + 1;
    const uint32_t tmask = (1u << (2 * t)) - 1u;
    uint32_t best = 0xFFFFFFFFu;
    int at = 0;
    for (int j = 0; j < nt; j++) {  // the leftmost smallest canonical t-mer of the k-mer as shown
        const uint32_t ft = (uint32_t)(f >> (2 * (k - t - j))) & tmask, rt = (uint32_t)(r >> (2 * j)) & tmask;
        const uint32_t o = mz_tmer_order(ft < rt ? ft : rt);
        if (o < best) {
            best = o;
            at = j;
        }
    }
    const int x = at % w;
    const uint64_t mmask = (1ULL << (2 * m)) - 1ULL;
    const uint64_t fm = (f >> (2 * (k - m - x))) & mmask, rm = (r >> (2 * x)) & mmask;
    const bool rev = rm < fm;
    MzpProbe o;
    o.y = mzp_bij64(rev ? rm : fm, m);
    o.p = rev ? (uint32_t)(w - 1 - x) : (uint32_t)x;
    mzp_tag(rev ? r : f, o.p, k, m, o.tw, o.mk);
    return o;
}

// ------------------------------------------------------------------------------------------
// build
// ------------------------------------------------------------------------------------------
// one entry (region << 33 | key index << 1 | which strand's pick) per key and pick; the reverse strand's pick only when it is
// another (region, offset, orientation) than the canonical strand's.  Sorted on all 64 bits.
__global__ void k_mzp_entries(const uint64_t *keys, uint64_t n, int k, int m, int s, uint64_t *ent, unsigned long long *n_dup) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t c = keys[i], rc = mzp_revcomp(c, k);
    const MzpProbe a = mzp_probe_of(c, rc, k, m), b = mzp_probe_of(rc, c, k, m);
    ent[i] = ((uint64_t)(a.y >> s) << 33) | (i << 1);
    const bool dup = a.y != b.y || a.p != b.p || a.tw != b.tw;
    ent[n + i] = dup ? ((uint64_t)(b.y >> s) << 33) | (i << 1) | 1ULL : ~0ULL;
    if (dup) atomicAdd(n_dup, 1ULL);
}

// One thread per region (the thread of the region's first entry): its entries, in key order, become records -- an entry joins
// the first record it is compatible with (same m-mer, same loc, agreeing flank bases, its offset still free), else opens a new
// one.  An entry that a record of the region would already match (its offset taken there and the flank bases its windows know
// equal: a record of ANOTHER m-mer that shares the region, since the keys are distinct) goes to the generic table, like what
// does not fit 7 records: so no window can match two records, and the first match is the match.  Finally every record's rows
// must be consecutive in offset order (base + sign * rank: what k_emit derives a row from); a record whose are not keeps its
// first member only.
__global__ void k_mzp_place(uint4 *regions, uint32_t *pay, int k, int m, int s, const uint64_t *keys, const uint32_t *loc,
                            const uint32_t *payload, const uint64_t *ent, uint64_t n2, unsigned long long *overflow,
                            uint32_t *overflow_list, uint64_t overflow_cap, uint32_t hot_min) {
    const uint64_t e0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e0 >= n2) return;
    const uint64_t reg = ent[e0] >> 33;
    if (e0 && (ent[e0 - 1] >> 33) == reg) return;  // not the region's first entry
    const int w = k - m + 1;
    const uint32_t fpmask = s ? (1u << s) - 1u : 0u, flank = 0xFFFFFF00u;
    uint32_t tag[kMzpRecs], km[kMzpRecs], locw[kMzpRecs], rows[kMzpRecs][7], who[kMzpRecs][7], twm[kMzpRecs][7];
    uint32_t nrec = 0;
    bool over = false;
    // (overflow[1..5]: why -- 1 loc too wide for the loc word, 2 the "no record" value, 3 a record would match its windows
    //  already, 4 the region's 7 records are taken, 5 a record whose rows are not consecutive)
    uint32_t filter = 0u, n_spilled = 0u;
    uint32_t hotf[kMzpHotWords + 1] = {0u};  // (hot_min) the 480-bit filter of EVERY entry of the region, in case it turns out hot
    auto spill = [&](uint32_t key_idx, int why, uint32_t tw, uint32_t fpw) {
        const unsigned long long o = atomicAdd(overflow, 1ULL);
        atomicAdd(overflow + why, 1ULL);
        if (o < overflow_cap) overflow_list[o] = key_idx;
        over = true;
        n_spilled++;
        filter |= 0x80000000u >> mzp_filter_shift(tw, fpw);
    };
    uint32_t n_ent = 0;
    for (uint64_t e = e0; e < n2 && (ent[e] >> 33) == reg; e++) {
        n_ent++;
        const uint32_t i = (uint32_t)(ent[e] >> 1);
        const uint64_t c = keys[i], rc = mzp_revcomp(c, k);
        const MzpProbe q = (ent[e] & 1ULL) ? mzp_probe_of(rc, c, k, m) : mzp_probe_of(c, rc, k, m);
        const uint32_t fpw = s ? ((uint32_t)q.y & fpmask) << (32 - s) : 0u;
        {
            const uint32_t h = mzp_filter_hash(q.tw, fpw), h2 = mzp_hot_second(h);
            hotf[mzp_hot_word(h)] |= 1u << mzp_hot_bit(h);
            hotf[mzp_hot_word(h2)] |= 1u << mzp_hot_bit(h2);
        }
        if (s && (loc[i] >> (32 - s))) {  // (a loc id that needs the fp's bits: not in this format)
            spill(i, 1, q.tw, fpw);
            continue;
        }
        const uint32_t lw = fpw | loc[i];
        if (lw == 0xFFFFFFFFu) {  // (never a loc word: kept out of the way of the lookups' sentinels)
            spill(i, 2, q.tw, fpw);
            continue;
        }
        int home = -1;
        bool clash = false;
        for (uint32_t r = 0; r < nrec; r++) {
            const bool taken = (tag[r] >> q.p) & 1u;
            if (taken) {
                if (((tag[r] ^ q.tw) & q.mk & flank) == 0u) clash = true;  // this key's windows would match record r
            } else if (home < 0 && locw[r] == lw && ((tag[r] ^ q.tw) & km[r] & q.mk & flank) == 0u) {
                home = (int)r;
            }
        }
        if (clash || (home < 0 && nrec == (uint32_t)kMzpRecs)) {
            spill(i, clash ? 3 : 4, q.tw, fpw);
            continue;
        }
        if (home < 0) {
            home = (int)nrec++;
            tag[home] = 0u;
            km[home] = 0u;
            locw[home] = lw;
        }
        tag[home] |= q.tw;
        km[home] |= q.mk;
        rows[home][q.p] = payload[i];
        who[home][q.p] = i;
        twm[home][q.p] = q.tw;
    }
    // overflow[6]: keys of HOT regions (dozens of places under one minimiser: an m-mer of a repeat family that SUNKs of many of
    // its copies share).  Reads visit such a region in proportion to the copies, nearly always without a match, and every such
    // window becomes a lookup in the generic table: the host weighs this count when it picks the layout.
    if (n_ent > 32u) atomicAdd(overflow + 7, (unsigned long long)n_ent);
    uint32_t base[kMzpRecs];
    for (uint32_t r = 0; r < nrec; r++) {
        const uint32_t mask = tag[r] & ((1u << w) - 1u);
        int first = -1, second = -1;
        for (int p = 0; p < w; p++)
            if ((mask >> p) & 1u) {
                if (first < 0) first = p;
                else if (second < 0) second = p;
            }
        base[r] = rows[r][first];
        const bool down = second >= 0 && rows[r][second] + 1u == rows[r][first];
        bool ok = true;
        uint32_t rank = 0;
        for (int p = 0; p < w; p++)
            if ((mask >> p) & 1u) {
                if (rows[r][p] != (down ? base[r] - rank : base[r] + rank)) ok = false;
                rank++;
            }
        if (!ok) {  // rows that are not consecutive: the record keeps its first member
            for (int p = first + 1; p < w; p++)
                if ((mask >> p) & 1u) spill(who[r][p], 5, twm[r][p], s ? locw[r] & ~((1u << (32 - s)) - 1u) : 0u);
            tag[r] &= ~(((1u << w) - 1u) & ~(1u << first));
        } else if (down) {
            tag[r] |= 0x80u;
        }
    }
    uint32_t wd[16];
    wd[0] = over ? kMzpOverflow : 0u;
    wd[1] = filter;
    for (int r = 0; r < kMzpRecs; r++) {
        wd[kMzpTagWord0 + r] = (uint32_t)r < nrec ? tag[r] : 0u;
        wd[kMzpLocWord0 + r] = (uint32_t)r < nrec ? locw[r] : 0u;
    }
    // A HOT region (hot_min > 0: variant-dominated assemblies): so many of its keys did not fit that the 32-bit filter is
    // mostly ones -- a minimiser that dozens of copies of a repeat share, each copy with SUNKs of its own over its variants.
    // Reads visit it in proportion to the copies, nearly always for the k-mers the copies have in COMMON (no SUNKs).  All its
    // keys go to the generic table, and the region's 60 bytes become one filter of them.
    if (hot_min && n_spilled >= hot_min) {
        for (uint32_t r = 0; r < nrec; r++)
            for (int p = 0; p < w; p++)
                if ((tag[r] >> p) & 1u) spill(who[r][p], 6, 0u, 0u);
        wd[0] = kMzpOverflow | kMzpHot;
        for (int j = 1; j <= kMzpHotWords; j++) wd[j] = hotf[j];
        nrec = 0;
    }
    uint4 *dst = regions + reg * 4;
    for (int q = 0; q < 4; q++) dst[q] = make_uint4(wd[4 * q], wd[4 * q + 1], wd[4 * q + 2], wd[4 * q + 3]);
    for (int r = 0; r < kMzpRecs; r++) pay[reg * 8 + r] = (uint32_t)r < nrec ? base[r] : 0u;
}

// ------------------------------------------------------------------------------------------
// lookups from global memory (exact form, emit)
// ------------------------------------------------------------------------------------------
struct MzpView {
    const uint4 *regions;
    const uint32_t *pay;
    int k, m, s;
};
// -> record index 0..6 of the match, -1 none; *locw = its loc word, *hdr = the region's header word
__device__ __forceinline__ int mzp_find(const MzpView &v, const MzpProbe &q, uint32_t *locw, uint32_t *hdr) {
    const uint4 *rp = v.regions + (uint64_t)(q.y >> v.s) * 4;
    const uint4 a = rp[0], b = rp[1], c = rp[2], d = rp[3];
    const uint32_t wd[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    *hdr = wd[0];
    if (wd[0] & kMzpHot) return -1;  // (no records: a filter; the caller goes to the generic table)
    const uint32_t fp = v.s ? ((uint32_t)q.y & ((1u << v.s) - 1u)) : 0u;
    for (int r = 0; r < kMzpRecs; r++)
        if (((wd[kMzpTagWord0 + r] ^ q.tw) & q.mk) == 0u) {
            const uint32_t lw = wd[kMzpLocWord0 + r];
            if (v.s && (lw >> (32 - v.s)) != fp) return -1;  // (the only record that can match this window: another m-mer's)
            *locw = lw;
            return r;
        }
    return -1;
}
__device__ __forceinline__ uint32_t mzp_payload(const MzpView &v, const TableView &tab, uint64_t c) {
    const MzpProbe q = mzp_probe_of(c, mzp_revcomp(c, v.k), v.k, v.m);
    uint32_t lw = 0, hdr = 0;
    const int r = mzp_find(v, q, &lw, &hdr);
    if (r >= 0) {
        const uint32_t *wp = reinterpret_cast<const uint32_t *>(v.regions + (uint64_t)(q.y >> v.s) * 4);
        const uint32_t tg = wp[kMzpTagWord0 + r];
        const uint32_t rank = (uint32_t)__popc(tg & ((1u << q.p) - 1u));
        const uint32_t base = v.pay[(uint64_t)(q.y >> v.s) * 8 + r];
        return (tg & 0x80u) ? base - rank : base + rank;
    }
    uint64_t hh = slot_of(c, tab.shift);  // a key the region could not take lives in the generic table
    Slot sl = load_slot(tab.slots + hh);
    while (sl.key != c && sl.key != kEmptyKey) {
        hh = (hh + 1) & tab.mask;
        sl = load_slot(tab.slots + hh);
    }
    return sl.payload;
}
// one window, from global memory: hit and loc
__device__ __forceinline__ bool mzp_lookup(const MzpView &v, const TableView &tab, uint64_t f, uint64_t r, uint32_t *loc) {
    const MzpProbe q = mzp_probe_of(f, r, v.k, v.m);
    uint32_t lw = 0, hdr = 0;
    if (mzp_find(v, q, &lw, &hdr) >= 0) {
        *loc = v.s ? (lw & ((1u << (32 - v.s)) - 1u)) : lw;
        return true;
    }
    if (!(hdr & kMzpOverflow)) return false;
    const uint64_t c = f < r ? f : r;
    uint64_t hh = slot_of(c, tab.shift);
    Slot sl = load_slot(tab.slots + hh);
    while (sl.key != c && sl.key != kEmptyKey) {
        hh = (hh + 1) & tab.mask;
        sl = load_slot(tab.slots + hh);
    }
    if (sl.key != c) return false;
    *loc = sl.loc;
    return true;
}

// the exact per-lane form of a tile: every window its own lookups (the fallback of the fast form, and the "minimizer-exact" layout)
template <int MODE>
__device__ __forceinline__ MzTileOut mzp_tile(const MzpView &v, const TableView &tab, const uint64_t *__restrict__ words,
                                             const TileDesc td, uint32_t *cbuf) {
    const int lane = threadIdx.x & 63;
    const int k = v.k, m = v.m;
    const uint64_t kmask = (1ULL << (2 * k)) - 1ULL, mmask = (1ULL << (2 * m)) - 1ULL;
    const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint64_t W0 = __builtin_nontemporal_load(words + td.wbase + lane);
    const uint64_t W1 = __builtin_nontemporal_load(words + td.wbase + 64 + (lane & 1));
    uint32_t n_raw = 0, n_cand = 0, first_loc = GS_NONE, carry = GS_NONE;  // wave-uniform
    const uint32_t nsteps = min((td.nwin_left + 63u) / 64u, 32u);
    for (uint32_t s = 0; s < nsteps; s++) {
        const MzWin ww = mz_window(W0, W1, (int)s, lane, k, m, kmask, mmask);
        uint32_t loc = 0;
        const bool hit = s * 64u + (uint32_t)lane < td.nwin_left && mzp_lookup(v, tab, ww.f, ww.r, &loc);
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
    return MzTileOut{n_raw, n_cand, first_loc, carry};
}

__device__ __forceinline__ void mzp_tile_exact(const MzpView &v, const TableView &tab, const uint64_t *__restrict__ words,
                                               const TileDesc td, uint32_t tile, uint32_t n_tiles, const TileArrays &ta, uint32_t *cand,
                                               uint32_t cand_cap, uint32_t *g_cand_count, uint32_t *s_cand) {
    const int lane = threadIdx.x & 63;
    MzTileOut o = mzp_tile<0>(v, tab, words, td, s_cand);
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
            mzp_tile<1>(v, tab, words, td, cand + off);  // (rare) walk the tile again, candidates straight to their place
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

__global__ __launch_bounds__(256) void k_probe_mzp_exact(MzpView v, TableView tab, const uint64_t *__restrict__ words,
                                                         const TileDesc *__restrict__ tdesc, uint32_t n_tiles, TileArrays ta,
                                                         uint32_t *cand, uint32_t cand_cap, uint32_t *g_cand_count) {
    __shared__ uint32_t s_cand[4][kMzCandLds];
    const int wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * 4u + (uint32_t)wave;
    if (tile >= n_tiles) return;
    mzp_tile_exact(v, tab, words, tdesc[tile], tile, n_tiles, ta, cand, cand_cap, g_cand_count, s_cand[wave]);
}

#endif  // GS_MZ_PLACE

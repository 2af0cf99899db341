// Device half of libgavisunk.so: the SUNK tagger of GAVISUNK (workflow/src/kmerpos_annot3.nim)
// as hand-written HIP for gfx950 (MI355X, wave64).  Pure integer / indexing work; see DESIGN.md section 4.
//
// Data layout in HBM
//   reads   2 bits/base, 32 bases per uint64 (first base most significant); every read starts
//           on a word boundary and is followed by >= 1 zero word.
//   tiles   a read is cut into tiles of 2048 windows = 64 words; a wavefront lane owns one word
//           = windows [32 l, 32 l + 32) of a tile.
//   tables  (1) minimiser-addressed (gs_mz.h): 64-byte regions picked by a mod-minimiser of the k-mer, fetched once per run of
//               windows -- the default for k >= 16;
//           (2) compact: exact, 16 partitions x 2^qb buckets x 4 x 32-bit slots [flags|alt|remainder|loc], + one payload word
//               per slot -- tables (1) does not take;
//           (3) generic: open-addressed, linear probing, 16-byte slots {key u64, loc u32, payload u32},
//               load <= 0.5 -- everything when neither applies, else only the keys (1) / (2) could not take.
//
// One `gs_tag` pass:
//   phase 1, one of
//     k_probe_mz       (1): one wave per tile, lane = window, a region fetched once per run by LDS-DMA (gs_mz.h)
//     k_probe<true>    (2): one 16-byte bucket load (one 64-byte line) per window
//     k_probe<false>   (3): one wave per tile, 8 probes in flight per lane
//   -> per tile {n_raw, first loc, last loc} and the ordered candidates (hits whose loc differs from the
//      previous raw hit of the tile; the tile's first raw hit is always a candidate)
//   scan A     over tiles: raw-hit prefix and the loc carried into each tile (`prevLoc` of
//              kmerpos_annot3.nim:82-84 crosses reads, so the scan is global); drops a tile's first
//              candidate when it equals the carried loc
//   scan B     emitted-hit prefix
//   k_emit     pos = window - (raw hits before it in the read - emitted hits before it in the read); rows dealt out over the lanes
//              [`continue` at kmerpos_annot3.nim:92 skips `inc i` at :96], payload by re-probing the
//              (rare) emitted windows; writes gs_hit in stream order
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <type_traits>
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

struct __align__(16) Slot {
    uint64_t key;
    uint32_t loc;
    uint32_t payload;
};
constexpr uint64_t kEmptyKey = ~0ULL;  // never a canonical code (k < 32: too wide; k = 32: poly-T is not canonical)
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ULL;
constexpr int kDirectCand = 4;  // candidate slots every tile owns in the candidate buffer
constexpr uint64_t kMirrorRows = 1u << 20;  // hit lists up to this long are written to the pinned host mirror by k_emit itself

struct TableView {
    const Slot *slots;
    uint64_t mask;
    int shift;
    int k;
};

struct ReadsView {
    const uint64_t *words;
    const uint64_t *read_word_off;  // per read
    const uint32_t *read_nwin;      // per read: max(len - k + 1, 1)   (Q6: short reads give one window)
    const uint32_t *read_tile0;     // per read: first tile
    const uint32_t *tile_read;      // per tile
    uint32_t n_tiles;
};

struct TileArrays {
    uint32_t *n_raw, *n_cand, *first_loc, *last_loc, *cand_off, *n_emit, *carry_loc;
    uint64_t *raw_prefix, *emit_prefix;
    // scatter-chunk emulation (gs_reads_set_seeds): seed[t] != kNoSeed forces the loc carried INTO tile t (GS_NONE = the fresh
    // prevLoc of a new chunk file); seg_carry[t] then receives the loc that was carried up to there.  NULL = no seeds.
    const uint32_t *seed;
    uint32_t *seg_carry;
};
constexpr uint32_t kNoSeed = 0xFFFFFFFDu;
constexpr uint64_t kReadSlackWords = 66;  // zero words after the last read of a batch (a tile loads 64 + 2 words whatever its length)

__device__ __forceinline__ uint64_t revpairs(uint64_t x) {
    x = __brevll(x);
    return ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
}

// canonical code of the window that starts at base `b` (0..31) of the 64-base string (a:b2)
__device__ __forceinline__ uint64_t canon_at(uint64_t a, uint64_t b2, uint64_t ra, uint64_t rb, int b, int k,
                                             uint64_t kmask) {
    const int s = 2 * b;
    const uint64_t f = (s ? ((a << s) | (b2 >> (64 - s))) : a) >> (64 - 2 * k);
    const uint64_t r = (s ? ((ra >> s) | (rb << (64 - s))) : ra) & kmask;
    return f < r ? f : r;
}

__device__ __forceinline__ uint64_t slot_of(uint64_t key, int shift) { return (key * kGolden) >> shift; }

__device__ __forceinline__ Slot load_slot(const Slot *p) {
    const uint4 v = *reinterpret_cast<const uint4 *>(p);
    Slot s;
    s.key = (uint64_t)v.x | ((uint64_t)v.y << 32);
    s.loc = v.z;
    s.payload = v.w;
    return s;
}

// Exclusive prefix sum over the 64 lanes with DPP adds (6 VALU instructions: shifts inside the rows of 16 lanes, then the
// two row broadcasts of gfx9); `total` comes back wave-uniform.  All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, int lane, uint32_t &total) {
    (void)lane;
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    total = (uint32_t)__builtin_amdgcn_readlane(x, 63);
    return (uint32_t)x - v;
}

constexpr int kNPart = 16;  // key-space partitions of the compact table (strand-symmetric: part_of)
constexpr uint32_t kEmptySlot32 = 0xFFFFFFFFu;

// Bijection on `kbits`-bit keys, cheap in 32-bit arithmetic, returning the pair (bucket, remainder)
// with bucket in [0, 2^qb) taken from the best-mixed bits.  kbits = 2k - 4 in [28, 60], qb <= 20.
struct BucketRem {
    uint32_t bucket;
    uint64_t rem;  // kbits - qb bits (fits 32 bits whenever the 32-bit slot format is in use)
};
__host__ __device__ __forceinline__ BucketRem hmix2(uint64_t key, int kbits, int qb) {
    const int lbits = kbits < 32 ? kbits : 32;  // low part that gets mixed
    const uint32_t lmask = lbits == 32 ? 0xFFFFFFFFu : ((1u << lbits) - 1u);
    const uint32_t hi = (uint32_t)(key >> lbits);  // kbits - lbits bits, passes through
    uint32_t l = (uint32_t)key & lmask;
    l = (l * 0x9E3779B1u) & lmask;
    l ^= l >> (lbits / 2);
    l = (l * 0x85EBCA6Bu) & lmask;
    l ^= l >> (lbits / 2 + 1);
    l = (l ^ (hi * 0xC2B2AE35u)) & lmask;
    BucketRem o;
    o.bucket = l >> (lbits - qb);
    o.rem = ((uint64_t)(l & ((1u << (lbits - qb)) - 1u)) << (kbits - lbits)) | hi;
    return o;
}
// second-chance bucket of a key (same partition): bucket ^ (odd, non-zero function of the remainder).  Only the low
// bits change, so both buckets of a key lie in the same 1/16 of the partition (the slice one CU stages in LDS).
__host__ __device__ __forceinline__ int alt_bits(int qb) { return qb - 4 < 1 ? 1 : (qb - 4 > 9 ? 9 : qb - 4); }
__host__ __device__ __forceinline__ uint32_t alt_bucket(uint32_t bucket, uint64_t rem, int qb) {
    const uint32_t r = (uint32_t)rem ^ (uint32_t)(rem >> 32);
    return bucket ^ ((((r + 1u) * 0x9E3779B1u) >> (32 - alt_bits(qb))) | 1u);
}

// partition of a canonical (or not: it is strand-symmetric) 2k-bit code
__host__ __device__ __forceinline__ uint32_t part_of(uint64_t c, int k) {
    const uint32_t x0 = (uint32_t)(c >> (2 * k - 2)) & 3u, x1 = (uint32_t)(c >> (2 * k - 4)) & 3u;
    const uint32_t xl = (uint32_t)c & 3u, xl2 = (uint32_t)(c >> 2) & 3u;
    return ((x0 ^ xl) << 2) | (x1 ^ xl2);
}

// n low bits set; n may be 32 (a 31-bit remainder with its alt bit position, in tables whose slots have no spare bit)
__host__ __device__ __forceinline__ uint32_t low_bits32(int n) { return n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u; }

// slot = [ flags(2, slot 0 only) | alt(1) | rem | loc ];  flags: bit 31 = a key of this bucket lives in its
// second-chance bucket, bit 30 = a key of this bucket lives in the generic table only
struct CompactView {
    const uint4 *buckets;     // [kNPart << qb] buckets of 4 slots: one uint4 each (32-bit slots) or two (64-bit slots)
    const uint32_t *payload;  // one per slot (what a hit on that slot reports)
    int qb, rembits, locbits, has_flag, w64;
};


// Exact lookup in the compact table (any size: resident in L2 slices or out in HBM).  Returns hit / loc; a key
// that overflowed both of its buckets at build time is found in the generic table.
__device__ __forceinline__ bool compact_lookup(const CompactView &ct, const TableView &tab, uint64_t c, int k,
                                               uint32_t &loc, uint32_t *payload = nullptr) {
    const int kbits = 2 * k - 4;
    const uint64_t keymask = (1ULL << kbits) - 1ULL;
    const BucketRem br = hmix2((c >> 4) & keymask, kbits, ct.qb);
    if (ct.w64) {  // 64-bit slots [flags(2) | alt | rem | loc]; always with flags
        const uint64_t locmask = (1ULL << ct.locbits) - 1ULL, tagmask = (1ULL << (ct.rembits + 1)) - 1ULL;
        const uint64_t pbase = (uint64_t)part_of(c, k) << ct.qb;
        const ulonglong2 *b64 = reinterpret_cast<const ulonglong2 *>(ct.buckets);
        bool full_out = false;
        auto scan = [&](uint64_t bucket, uint64_t want, uint64_t &s0) -> bool {
            const ulonglong2 lo = b64[(pbase + bucket) * 2], hi2 = b64[(pbase + bucket) * 2 + 1];
            const uint64_t sl[4] = {lo.x, lo.y, hi2.x, hi2.y};
            s0 = sl[0];
            bool h = false;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (sl[i] != ~0ULL && ((sl[i] >> ct.locbits) & tagmask) == want) {
                    h = true;
                    loc = (uint32_t)(sl[i] & locmask);
                    if (payload) *payload = ct.payload[(pbase + bucket) * 4 + i];
                }
            full_out = sl[0] != ~0ULL && sl[1] != ~0ULL && sl[2] != ~0ULL && sl[3] != ~0ULL;
            return h;
        };
        uint64_t s0 = 0;
        if (scan(br.bucket, br.rem, s0)) return true;
        if (!full_out) return false;
        const bool generic = ((s0 >> 62) & 1ULL) != 0;
        if (s0 >> 63) {
            uint64_t s1;
            if (scan(alt_bucket(br.bucket, br.rem, ct.qb), br.rem | (1ULL << ct.rembits), s1)) return true;
        }
        if (!generic) return false;
        uint64_t hh = slot_of(c, tab.shift);
        Slot cur = load_slot(tab.slots + hh);
        while (cur.key != c && cur.key != kEmptyKey) {
            hh = (hh + 1) & tab.mask;
            cur = load_slot(tab.slots + hh);
        }
        if (cur.key == c) {
            loc = cur.loc;
            if (payload) *payload = cur.payload;
            return true;
        }
        return false;
    }
    const uint32_t locmask = (1u << ct.locbits) - 1u, tagmask = low_bits32(ct.rembits + 1);
    const uint4 *pb = ct.buckets + ((uint64_t)part_of(c, k) << ct.qb);
    const uint4 b = pb[br.bucket];
    const uint32_t sl[4] = {b.x, b.y, b.z, b.w};
    bool hit = false, full = true;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (sl[i] == kEmptySlot32) full = false;
        else if (((sl[i] >> ct.locbits) & tagmask) == (uint32_t)br.rem) {
            hit = true;
            loc = sl[i] & locmask;
            if (payload) *payload = ct.payload[(uint64_t)(pb - ct.buckets + br.bucket) * 4 + i];
        }
    }
    if (hit || !full) return hit;
    bool generic = !ct.has_flag;
    if (ct.has_flag) {
        generic = ((sl[0] >> 30) & 1u) != 0;
        if (sl[0] >> 31) {
            const uint32_t ab = alt_bucket(br.bucket, br.rem, ct.qb);
            const uint4 b2 = pb[ab];
            const uint32_t s2[4] = {b2.x, b2.y, b2.z, b2.w};
            const uint32_t want = (uint32_t)br.rem | (1u << ct.rembits);
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (s2[i] != kEmptySlot32 && ((s2[i] >> ct.locbits) & tagmask) == want) {
                    hit = true;
                    loc = s2[i] & locmask;
                    if (payload) *payload = ct.payload[(uint64_t)(pb - ct.buckets + ab) * 4 + i];
                }
            if (hit) return true;
        }
    }
    if (!generic) return false;
    uint64_t hh = slot_of(c, tab.shift);
    Slot cur = load_slot(tab.slots + hh);
    while (cur.key != c && cur.key != kEmptyKey) {
        hh = (hh + 1) & tab.mask;
        cur = load_slot(tab.slots + hh);
    }
    if (cur.key == c) {
        loc = cur.loc;
        if (payload) *payload = cur.payload;
        return true;
    }
    return false;
}

// Per-lane results of one tile (lane = 32 consecutive windows: raw-hit mask, "loc differs from the lane's previous
// raw hit" mask, loc of the lane's first and last raw hit) -> the tile's {n_raw, n_cand, first/last loc, candidates}.
__device__ __forceinline__ void tile_summary(uint32_t raw, uint32_t chg, uint32_t first_loc, uint32_t prev, uint32_t tile,
                                             uint32_t n_tiles, const TileArrays &ta, uint32_t *cand, uint32_t cand_cap,
                                             uint32_t *g_cand_count) {
    const int lane = threadIdx.x & 63;
    // loc of the nearest raw hit in a lower lane of this tile
    const unsigned long long ne = __ballot(raw != 0);
    const unsigned long long below = ne & ((1ULL << lane) - 1ULL);
    const int src = below ? 63 - __clzll((long long)below) : -1;
    const uint32_t prev_lane_loc = __shfl(prev, src < 0 ? 0 : src);
    if (raw && (src < 0 || first_loc != prev_lane_loc)) chg |= 1u << (__ffs(raw) - 1);

    uint32_t tot_raw, tot_cand;
    const uint32_t ex_raw = wave_excl_scan(__popc(raw), lane, tot_raw);
    const uint32_t ex_cand = wave_excl_scan(__popc(chg), lane, tot_cand);

    // up to kDirectCand candidates go to the tile's own slots; longer lists (rare) are allocated with one atomic
    // (a single counter word sustains only ~90 returning atomics per microsecond on this chip)
    uint32_t off = tile * kDirectCand;
    if (tot_cand > (uint32_t)kDirectCand) {
        if (lane == 0) off = n_tiles * kDirectCand + atomicAdd(g_cand_count, tot_cand);
        off = __shfl(off, 0);
    }
    if (tot_cand && (uint64_t)off + tot_cand <= cand_cap) {
        uint32_t m = chg, i = 0;
        while (m) {
            const int bit = __ffs(m) - 1;
            m &= m - 1;
            const uint32_t woff = lane * 32 + bit;
            const uint32_t rank = ex_raw + __popc(raw & ((1u << bit) - 1u));
            cand[off + ex_cand + i] = (rank << 16) | woff;
            i++;
        }
    }
    if (ne) {
        const int lo = __ffsll((unsigned long long)ne) - 1, hi = 63 - __clzll((long long)ne);
        const uint32_t fl = __shfl(first_loc, lo), ll = __shfl(prev, hi);
        if (lane == 0) {
            ta.first_loc[tile] = fl;
            ta.last_loc[tile] = ll;
        }
    }
    if (lane == 0) {
        ta.n_raw[tile] = tot_raw;
        ta.n_cand[tile] = tot_cand;
        ta.cand_off[tile] = off;
    }
}

// ------------------------------------------------------------------------------------------
// probe_tile / k_probe: one wavefront per tile (generic 16-byte-slot table in HBM)
// ------------------------------------------------------------------------------------------
template <bool COMPACT>
__device__ __forceinline__ void probe_tile(const TableView &tab, const CompactView &ct, const ReadsView &rd,
                                           const TileArrays &ta, uint32_t *cand, uint32_t cand_cap,
                                           uint32_t *g_cand_count, unsigned long long *g_err_pos, uint32_t tile) {
    const int lane = threadIdx.x & 63;
    const uint32_t r = rd.tile_read[tile];
    const uint32_t tin = tile - rd.read_tile0[r];
    const uint64_t wbase = rd.read_word_off[r] + (uint64_t)tin * kTileWords;
    const uint32_t nwin = rd.read_nwin[r];
    const uint32_t w0 = tin * kTileWindows + lane * 32;
    const uint32_t nvalid = nwin > w0 ? min(nwin - w0, 32u) : 0u;
    const uint32_t vmask = nvalid >= 32 ? 0xFFFFFFFFu : ((1u << nvalid) - 1u);

    const int k = tab.k;
    const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
    // unconditional: the buffer carries kTileWords + 2 words of slack, windows past the read are masked
    const uint64_t a = rd.words[wbase + lane];
    const uint64_t b2 = rd.words[wbase + lane + 1];
    const uint64_t ra = ~revpairs(a), rb = ~revpairs(b2);

    uint32_t raw = 0, chg = 0;  // raw-hit mask, "loc differs from the previous raw hit of this lane" mask
    uint32_t first_loc = GS_NONE, prev = GS_NONE;
    bool missing = false;
    uint32_t missing_bit = 0;

    if (COMPACT && ct.w64) {
        // 64-bit slots: one 32-byte bucket (two 16-byte loads of one 64-byte line) per window, 4 windows in flight
        const int kbits = 2 * k - 4;
        const uint64_t keymask = (1ULL << kbits) - 1ULL;
        const uint64_t locmask = (1ULL << ct.locbits) - 1ULL, tagmask = (1ULL << (ct.rembits + 1)) - 1ULL;
        const ulonglong2 *b64 = reinterpret_cast<const ulonglong2 *>(ct.buckets);
#pragma unroll
        for (int g = 0; g < 8; g++) {
            uint64_t c[4], tag[4];
            ulonglong2 lo[4], hi[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                c[j] = canon_at(a, b2, ra, rb, g * 4 + j, k, kmask);
                const BucketRem br = hmix2((c[j] >> 4) & keymask, kbits, ct.qb);
                tag[j] = br.rem;
                const uint64_t bi = (((uint64_t)part_of(c[j], k) << ct.qb) + br.bucket) * 2;
                lo[j] = b64[bi];
                hi[j] = b64[bi + 1];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int bb = g * 4 + j;
                const uint64_t sl[4] = {lo[j].x, lo[j].y, hi[j].x, hi[j].y};
                bool hit = false, full = true;
                uint64_t l = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (sl[i] == ~0ULL) full = false;
                    else if (((sl[i] >> ct.locbits) & tagmask) == tag[j]) {
                        hit = true;
                        l = sl[i];
                    }
                }
                uint32_t loc = (uint32_t)(l & locmask);
                if (!hit && full && ((vmask >> bb) & 1u)) hit = compact_lookup(ct, tab, c[j], k, loc);  // second chance / generic
                if (hit && ((vmask >> bb) & 1u)) {
                    raw |= 1u << bb;
                    if (prev == GS_NONE) first_loc = loc;
                    else if (loc != prev) chg |= 1u << bb;
                    prev = loc;
                }
            }
        }
    } else if (COMPACT) {
        // compact table out in HBM: one 16-byte bucket load per window (one 64-byte line from the fabric), 8 in flight
        const int kbits = 2 * k - 4;
        const uint64_t keymask = (1ULL << kbits) - 1ULL;
        const uint32_t locmask = (1u << ct.locbits) - 1u, tagmask = low_bits32(ct.rembits + 1);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            uint64_t c[8];
            uint32_t tag[8];
            uint4 bk[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                c[j] = canon_at(a, b2, ra, rb, g * 8 + j, k, kmask);
                const BucketRem br = hmix2((c[j] >> 4) & keymask, kbits, ct.qb);
                tag[j] = (uint32_t)br.rem;
                bk[j] = ct.buckets[((uint64_t)part_of(c[j], k) << ct.qb) + br.bucket];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int bb = g * 8 + j;
                const uint32_t sl[4] = {bk[j].x, bk[j].y, bk[j].z, bk[j].w};
                uint32_t h = 0, l = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t eq = (((sl[i] >> ct.locbits) & tagmask) == tag[j]) & (sl[i] != kEmptySlot32);
                    h |= eq;
                    l = eq ? sl[i] : l;
                }
                uint32_t loc = l & locmask;
                bool hit = h != 0;
                const bool full = max(max(sl[0], sl[1]), max(sl[2], sl[3])) != kEmptySlot32;
                if (!hit && full && ((vmask >> bb) & 1u)) hit = compact_lookup(ct, tab, c[j], k, loc);  // second chance / generic
                if (hit && ((vmask >> bb) & 1u)) {
                    raw |= 1u << bb;
                    if (prev == GS_NONE) first_loc = loc;
                    else if (loc != prev) chg |= 1u << bb;
                    prev = loc;
                }
            }
        }
    } else {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            uint64_t c[8], h[8];
            Slot s[8];
    #pragma unroll
            for (int j = 0; j < 8; j++) {
                c[j] = canon_at(a, b2, ra, rb, g * 8 + j, k, kmask);
                h[j] = slot_of(c[j], tab.shift);
            }
    #pragma unroll
            for (int j = 0; j < 8; j++) s[j] = load_slot(tab.slots + h[j]);
    #pragma unroll
            for (int j = 0; j < 8; j++) {
                const int bb = g * 8 + j;
                Slot cur = s[j];
                uint64_t hh = h[j];
                while (cur.key != c[j] && cur.key != kEmptyKey) {
                    hh = (hh + 1) & tab.mask;
                    cur = load_slot(tab.slots + hh);
                }
                if (cur.key == c[j] && ((vmask >> bb) & 1u)) {
                    raw |= 1u << bb;
                    if (prev == GS_NONE) {
                        first_loc = cur.loc;
                    } else if (cur.loc != prev) {
                        chg |= 1u << bb;
                    }
                    prev = cur.loc;
                    if (cur.loc == GS_MISSING && !missing) {
                        missing = true;
                        missing_bit = bb;
                    }
                }
            }
        }
    }
    if (missing) atomicMin(g_err_pos, ((unsigned long long)tile << 32) | (uint32_t)(lane * 32 + missing_bit));

    tile_summary(raw, chg, first_loc, prev, tile, rd.n_tiles, ta, cand, cand_cap, g_cand_count);
}


template <bool COMPACT>
__global__ __launch_bounds__(256) void k_probe(TableView tab, CompactView ct, ReadsView rd, TileArrays ta, uint32_t *cand,
                                               uint32_t cand_cap, uint32_t *g_cand_count,
                                               unsigned long long *g_err_pos, const uint32_t *tile_list,
                                               const uint32_t *tile_list_n) {
    // whole batch (tile_list == nullptr: one wave per tile) or a list of tiles (grid-stride)
    const uint32_t n_items = tile_list ? *tile_list_n : rd.n_tiles;
    for (uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6); item < n_items; item += gridDim.x * 4)
        probe_tile<COMPACT>(tab, ct, rd, ta, cand, cand_cap, g_cand_count, g_err_pos, tile_list ? tile_list[item] : item);
}


struct __align__(16) TileDesc {
    uint64_t wbase;      // first word of the tile
    uint32_t nwin_left;  // windows of the read from the tile's first window on
    uint32_t read;
};

__device__ __forceinline__ uint32_t lane_value(uint32_t v, int uniform_lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, uniform_lane);
}

#include "gs_mz.h"

// ------------------------------------------------------------------------------------------
// tile scans (3 launches each): exclusive scan of {sum, last non-empty loc}
// ------------------------------------------------------------------------------------------
struct Agg {
    uint64_t sum;
    uint32_t loc;
    uint32_t has;  // bit 0: a loc is carried out of this stretch; bit 1: the stretch contains a seed (what came before it is cut off)
};
__device__ __forceinline__ Agg combine(const Agg &l, const Agg &r) {
    Agg o;
    o.sum = l.sum + r.sum;
    if (r.has & 2u) {  // r starts over at its seed: nothing of l's carry survives it
        o.has = r.has;
        o.loc = r.loc;
    } else {
        o.has = (l.has | r.has);
        o.loc = (r.has & 1u) ? r.loc : l.loc;
    }
    return o;
}
__device__ __forceinline__ Agg agg_shfl_up(const Agg &a, int d) {
    Agg o;
    o.sum = __shfl_up((unsigned long long)a.sum, d);
    o.loc = __shfl_up(a.loc, d);
    o.has = __shfl_up(a.has, d);
    return o;
}

// Scalars of a pass, in one device block: reset with one copy, fetched with one copy into pinned host memory.
struct PassStatus {
    unsigned long long err_pos;
    Agg totals[2];  // [0] scan A grand total, [1] scan B grand total
    uint32_t cand_count, complex_n, carry_in, pad;
};

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanBlock = kScanThreads * kScanItems;

// mode 0: element = {n_raw[t], last_loc[t], n_raw[t] > 0};  mode 1: element = {n_emit[t], -, 0}
__device__ __forceinline__ Agg scan_elem(const TileArrays &ta, uint32_t t, uint32_t n, int mode) {
    Agg e{0, GS_NONE, 0};
    if (t < n) {
        if (mode == 0) {
            const uint32_t nr = ta.n_raw[t];
            e.sum = nr;
            e.has = nr ? 1u : 0u;
            e.loc = nr ? ta.last_loc[t] : GS_NONE;
            if (ta.seed && ta.seed[t] != kNoSeed) {  // the tile starts from its seed
                e.has |= 2u;
                if (!nr && ta.seed[t] != GS_NONE) {
                    e.has |= 1u;
                    e.loc = ta.seed[t];
                }
            }
        } else {
            e.sum = ta.n_emit[t];
        }
    }
    return e;
}

// Exclusive block scan of per-thread aggregates; returns the thread's exclusive prefix and the block total.
__device__ __forceinline__ Agg block_excl_scan(const Agg &mine, Agg &block_total) {
    __shared__ Agg wave_tot[kScanThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Agg inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        Agg y = agg_shfl_up(inc, d);
        if (lane >= d) inc = combine(y, inc);
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    Agg pre{0, GS_NONE, 0};
    for (int w = 0; w < wave; w++) pre = combine(pre, wave_tot[w]);
    Agg tot{0, GS_NONE, 0};
    for (int w = 0; w < kScanThreads / 64; w++) tot = combine(tot, wave_tot[w]);
    block_total = tot;
    Agg prev = agg_shfl_up(inc, 1);
    if (lane == 0) prev = Agg{0, GS_NONE, 0};
    __syncthreads();
    return combine(pre, prev);
}

__global__ __launch_bounds__(kScanThreads) void k_scan_reduce(TileArrays ta, uint32_t n, int mode, Agg *block_agg) {
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * kScanItems;
    Agg a{0, GS_NONE, 0};
#pragma unroll
    for (int i = 0; i < kScanItems; i++) a = combine(a, scan_elem(ta, base + i, n, mode));
    Agg tot;
    block_excl_scan(a, tot);
    if (threadIdx.x == 0) block_agg[blockIdx.x] = tot;
}

// single block: exclusive scan of block aggregates in place, seeded with `seed_loc` (the carry-in)
__global__ __launch_bounds__(kScanThreads) void k_scan_blocks(Agg *block_agg, uint32_t nblocks, const uint32_t *seed_loc,
                                                              Agg *grand_total) {
    Agg run{0, GS_NONE, 0};
    if (seed_loc && *seed_loc != GS_NONE) {
        run.loc = *seed_loc;
        run.has = 1;
    }
    for (uint32_t base = 0; base < nblocks; base += kScanThreads) {
        const uint32_t i = base + threadIdx.x;
        Agg mine = i < nblocks ? block_agg[i] : Agg{0, GS_NONE, 0};
        Agg tot;
        Agg ex = block_excl_scan(mine, tot);
        if (i < nblocks) block_agg[i] = combine(run, ex);
        run = combine(run, tot);
    }
    if (threadIdx.x == 0) *grand_total = run;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_apply(TileArrays ta, uint32_t n, int mode, const Agg *block_agg) {
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * kScanItems;
    Agg e[kScanItems];
    Agg a{0, GS_NONE, 0};
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        e[i] = scan_elem(ta, base + i, n, mode);
        a = combine(a, e[i]);
    }
    Agg tot;
    Agg run = combine(block_agg[blockIdx.x], block_excl_scan(a, tot));
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        const uint32_t t = base + i;
        if (t < n) {
            if (mode == 0) {
                ta.raw_prefix[t] = run.sum;
                uint32_t carry = (run.has & 1u) ? run.loc : GS_NONE;
                if (ta.seed && ta.seed[t] != kNoSeed) {
                    ta.seg_carry[t] = carry;  // what the reads before this seed carried out
                    carry = ta.seed[t];
                }
                ta.carry_loc[t] = carry;
                // k_resolve fused: the tile's first candidate is suppressed when it repeats the carried loc
                const uint32_t nr = ta.n_raw[t];
                const uint32_t supp = (nr && carry != GS_NONE && ta.first_loc[t] == carry) ? 1u : 0u;
                ta.n_emit[t] = ta.n_cand[t] - supp;
            } else {
                ta.emit_prefix[t] = run.sum;
            }
        }
        run = combine(run, e[i]);
    }
}

// ------------------------------------------------------------------------------------------
// k_emit: a wave takes 64 tiles and deals their emitted rows out over its lanes (a tile has 0 .. dozens of them: one thread per
// tile would run the row loop as often as the wave's busiest tile needs, with most lanes idle)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_emit(TableView tab, CompactView ct, MzView mz, int compact, ReadsView rd, TileArrays ta,
                                              const uint32_t *cand, uint32_t cand_cap, gs_hit *hits, gs_hit *hits_host,
                                              uint64_t hit_cap, const Agg *emit_total) {
    // short lists (<= kMirrorRows) also go straight to pinned host memory, row by row: no copy after the pass; long ones
    // are fetched in one bulk copy instead (scattered 12-byte writes across PCIe are slow)
    const bool mirror = emit_total->sum <= kMirrorRows;
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // this lane's tile (none, no candidates, or an overflowed pass -- the host re-runs it with a larger buffer: no rows)
    uint32_t ne = 0, supp = 0, off = 0, r = 0, tin = 0;
    uint64_t raw_before = 0, emit_before = 0, out0 = 0, rwoff = 0;
    if (tile < rd.n_tiles) {
        const uint32_t nc = ta.n_cand[tile];
        off = ta.cand_off[tile];
        if (nc && (uint64_t)off + nc <= cand_cap) {
            ne = ta.n_emit[tile];
            supp = nc - ne;
            r = rd.tile_read[tile];
            const uint32_t t0 = rd.read_tile0[r];
            tin = tile - t0;
            raw_before = ta.raw_prefix[tile] - ta.raw_prefix[t0];
            emit_before = ta.emit_prefix[tile] - ta.emit_prefix[t0];
            out0 = ta.emit_prefix[tile];
            rwoff = rd.read_word_off[r];
        }
    }
    // rows of the lanes up to and including this one
    uint32_t inc = ne;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(inc, d);
        if (lane >= d) inc += y;
    }
    const uint32_t total = __shfl(inc, 63);
    const int k = tab.k;
    const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
    for (uint32_t q0 = 0; q0 < total; q0 += 64u) {  // wave-uniform
        const uint32_t q = q0 + (uint32_t)lane;  // the wave's q-th row
        // its tile = the first lane whose running count exceeds q (the counts never decrease: a bisection over the lanes)
        int own = 0;
#pragma unroll
        for (int step = 32; step; step >>= 1)
            if (__shfl(inc, own + step - 1) <= q) own += step;
        own = min(own, 63);
        const uint32_t o_ne = __shfl(ne, own), o_inc = __shfl(inc, own), o_supp = __shfl(supp, own), o_off = __shfl(off, own);
        const uint32_t o_r = __shfl(r, own), o_tin = __shfl(tin, own);
        const uint64_t o_raw = __shfl(raw_before, own), o_emit = __shfl(emit_before, own), o_out0 = __shfl(out0, own), o_rwoff = __shfl(rwoff, own);
        const uint32_t jj = q - (o_inc - o_ne);  // the row's index among its tile's emitted candidates
        const uint64_t o = o_out0 + jj;
        if (q >= total || o >= hit_cap) continue;
        const uint32_t j = o_supp + jj;
        const uint32_t cv = cand[o_off + j];
        const uint32_t woff = cv & 0xFFFFu, rank = cv >> 16;
        const uint32_t w = o_tin * kTileWindows + woff;
        const uint64_t suppressed_before = (o_raw + rank) - (o_emit + jj);
        // payload: re-probe this one window
        const uint64_t wi = o_rwoff + (w >> 5);
        const uint64_t a = rd.words[wi], b2 = rd.words[wi + 1];
        const uint64_t c = canon_at(a, b2, ~revpairs(a), ~revpairs(b2), (int)(w & 31u), k, kmask);
        uint32_t payload = 0, loc_unused = 0;
#if GS_MZ_PLACE
        if (compact == 4 && mz.place) {
            payload = mzp_payload(MzpView{mz.regions, mz.pay, mz.k, mz.m, mz.s}, tab, c);
        } else
#endif
        if (compact == 4) {
            payload = mz_payload(mz, tab, c);
        } else if (compact) {
            compact_lookup(ct, tab, c, k, loc_unused, &payload);
        } else {
            uint64_t hh = slot_of(c, tab.shift);
            Slot cur = load_slot(tab.slots + hh);
            while (cur.key != c && cur.key != kEmptyKey) {
                hh = (hh + 1) & tab.mask;
                cur = load_slot(tab.slots + hh);
            }
            payload = cur.payload;
        }
        gs_hit hrec;
        hrec.read = o_r;
        hrec.pos = (uint32_t)(w - suppressed_before);
        hrec.payload = payload;
        hits[o] = hrec;
        if (mirror) hits_host[o] = hrec;
    }
}

__global__ void k_selftest_nop() {}

// ------------------------------------------------------------------------------------------
// table build
// ------------------------------------------------------------------------------------------
__global__ void k_table_fill(Slot *slots, uint64_t cap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) {
        uint4 v;
        v.x = 0xFFFFFFFFu;
        v.y = 0xFFFFFFFFu;
        v.z = GS_NONE;
        v.w = 0;
        *reinterpret_cast<uint4 *>(slots + i) = v;
    }
}

__global__ void k_table_insert(Slot *slots, uint64_t mask, int shift, const uint64_t *keys, const uint32_t *loc,
                               const uint32_t *payload, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    uint64_t h = slot_of(key, shift);
    for (;;) {
        unsigned long long *kp = reinterpret_cast<unsigned long long *>(&slots[h].key);
        const unsigned long long old = atomicCAS(kp, (unsigned long long)kEmptyKey, (unsigned long long)key);
        if (old == kEmptyKey || old == key) {  // keys are distinct; `old == key` only on a re-run
            slots[h].loc = loc[i];
            slots[h].payload = payload[i];
            return;
        }
        h = (h + 1) & mask;
    }
}

// Compact-table build on the GPU: first free slot of the home bucket, else of the second-chance bucket (alt bit),
// else the key stays in the generic table only; the home bucket's slot 0 carries the two overflow flags.
__global__ void k_compact_insert(uint32_t *slots, uint32_t *slot_payload, int k, int qb, int rembits, int locbits,
                                 int has_flag, const uint64_t *keys, const uint32_t *loc, const uint32_t *payload, uint64_t n,
                                 unsigned long long *overflow, uint32_t *overflow_list) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t c = keys[i];
    const int kbits = 2 * k - 4;
    const BucketRem br = hmix2((c >> 4) & ((1ULL << kbits) - 1ULL), kbits, qb);
    const uint64_t pbase = (uint64_t)part_of(c, k) << qb;
    const uint32_t v = (br.rem << locbits) | loc[i];
    uint32_t *home = slots + (pbase + br.bucket) * 4;
    for (int j = 0; j < 4; j++)
        if (atomicCAS(home + j, kEmptySlot32, v) == kEmptySlot32) {
            slot_payload[(home - slots) + j] = payload[i];
            return;
        }
    if (has_flag) {
        uint32_t *alt = slots + (pbase + alt_bucket(br.bucket, br.rem, qb)) * 4;
        const uint32_t va = v | (1u << (rembits + locbits));
        for (int j = 0; j < 4; j++)
            if (atomicCAS(alt + j, kEmptySlot32, va) == kEmptySlot32) {
                slot_payload[(alt - slots) + j] = payload[i];
                atomicOr(home, 0x80000000u);
                return;
            }
        atomicOr(home, 0x40000000u);
    }
    overflow_list[atomicAdd(overflow, 1ULL)] = (uint32_t)i;  // goes to the (small) generic table
}

// the same with 64-bit slots [flags(2) | alt | rem | loc] (tables whose remainder + loc exceed 32 bits)
__global__ void k_compact_insert64(unsigned long long *slots, uint32_t *slot_payload, int k, int qb, int rembits, int locbits,
                                   const uint64_t *keys, const uint32_t *loc, const uint32_t *payload, uint64_t n,
                                   unsigned long long *overflow, uint32_t *overflow_list) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t c = keys[i];
    const int kbits = 2 * k - 4;
    const BucketRem br = hmix2((c >> 4) & ((1ULL << kbits) - 1ULL), kbits, qb);
    const uint64_t pbase = (uint64_t)part_of(c, k) << qb;
    const unsigned long long v = (br.rem << locbits) | loc[i];
    unsigned long long *home = slots + (pbase + br.bucket) * 4;
    for (int j = 0; j < 4; j++)
        if (atomicCAS(home + j, ~0ULL, v) == ~0ULL) {
            slot_payload[(home - slots) + j] = payload[i];
            return;
        }
    unsigned long long *alt = slots + (pbase + alt_bucket(br.bucket, br.rem, qb)) * 4;
    const unsigned long long va = v | (1ULL << (rembits + locbits));
    for (int j = 0; j < 4; j++)
        if (atomicCAS(alt + j, ~0ULL, va) == ~0ULL) {
            slot_payload[(alt - slots) + j] = payload[i];
            atomicOr(home, 1ULL << 63);
            return;
        }
    atomicOr(home, 1ULL << 62);
    overflow_list[atomicAdd(overflow, 1ULL)] = (uint32_t)i;
}

// Deterministic compact-table build (what gs_table_create runs): the keys are sorted by home bucket (stable radix sort, so the
// caller's order breaks ties), a key's rank in its bucket comes from its neighbours, and the second-chance buckets are filled
// in a second sorted pass with whatever the home keys left free.  No race decides which keys are displaced: the same input
// gives the same table and the same overflow keys every time (the CAS kernels above remain as the concurrent form).
struct CompactGeom {
    int k, qb, rembits, locbits, has_flag;
};
__device__ __forceinline__ uint64_t compact_home(uint64_t c, const CompactGeom &g, BucketRem &br) {
    const int kbits = 2 * g.k - 4;
    br = hmix2((c >> 4) & ((1ULL << kbits) - 1ULL), kbits, g.qb);
    return ((uint64_t)part_of(c, g.k) << g.qb) + br.bucket;
}
__global__ void k_compact_home_ids(const uint64_t *keys, uint64_t n, CompactGeom g, uint64_t *home) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BucketRem br;
    home[i] = compact_home(keys[i], g, br);
}
// pass 1 over the keys sorted by home bucket: ranks 0..3 take the bucket's slots, the others are left over (flag = 1)
template <typename SlotT>
__global__ void k_compact_place_home(SlotT *slots, uint32_t *slot_payload, uint8_t *home_count, CompactGeom g, const uint64_t *keys,
                                     const uint32_t *loc, const uint32_t *payload, const uint64_t *home_sorted,
                                     const uint32_t *idx_sorted, uint64_t n, uint8_t *left_flag) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint64_t b = home_sorted[p];
    uint32_t rank = 0;
    while (rank < 4u && p > rank && home_sorted[p - 1 - rank] == b) rank++;
    left_flag[p] = rank >= 4u ? 1 : 0;
    if (rank == 0) {
        uint32_t len = 1;
        while (len < 4u && p + len < n && home_sorted[p + len] == b) len++;
        home_count[b] = (uint8_t)len;
    }
    if (rank >= 4u) return;
    const uint32_t i = idx_sorted[p];
    BucketRem br;
    compact_home(keys[i], g, br);
    slots[b * 4 + rank] = ((SlotT)br.rem << g.locbits) | (SlotT)loc[i];
    slot_payload[b * 4 + rank] = payload[i];
}
// the second-chance bucket of every left-over key (keys without one sort last)
__global__ void k_compact_alt_ids(const uint64_t *keys, const uint32_t *left_idx, uint64_t n_left, CompactGeom g, uint64_t *alt) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_left) return;
    BucketRem br;
    const uint64_t home = compact_home(keys[left_idx[j]], g, br);
    alt[j] = g.has_flag ? (home - br.bucket) + alt_bucket(br.bucket, br.rem, g.qb) : 0xFFFFFFFFull;
}
// pass 2 over the left-over keys sorted by second-chance bucket: they take the slots the home keys left free, in order
template <typename SlotT>
__global__ void k_compact_place_alt(SlotT *slots, uint32_t *slot_payload, const uint8_t *home_count, CompactGeom g,
                                    const uint64_t *keys, const uint32_t *loc, const uint32_t *payload, const uint64_t *alt_sorted,
                                    const uint32_t *idx_sorted, uint64_t n_left, unsigned long long *overflow, uint32_t *overflow_list) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_left) return;
    const uint64_t a = alt_sorted[p];
    const uint32_t i = idx_sorted[p];
    BucketRem br;
    const uint64_t home = compact_home(keys[i], g, br);
    constexpr int kTop = 8 * (int)sizeof(SlotT) - 1;
    if (g.has_flag) {
        uint32_t rank = 0;
        while (rank < 4u && p > rank && alt_sorted[p - 1 - rank] == a) rank++;
        const uint32_t used = home_count[a];  // 0 for a bucket without home keys (the array starts zeroed)
        if (used + rank < 4u) {
            const SlotT v = ((SlotT)br.rem << g.locbits) | (SlotT)loc[i] | ((SlotT)1 << (g.rembits + g.locbits));
            slots[a * 4 + used + rank] = v;
            slot_payload[a * 4 + used + rank] = payload[i];
            atomicOr(slots + home * 4, (SlotT)1 << kTop);  // "a key of this bucket lives in its second-chance bucket"
            return;
        }
        atomicOr(slots + home * 4, (SlotT)1 << (kTop - 1));  // "... lives in the generic table only"
    }
    overflow_list[atomicAdd(overflow, 1ULL)] = i;
}

// generic-table insert of a subset of the keys (those the compact table could not take)
__global__ void k_table_insert_list(Slot *slots, uint64_t mask, int shift, const uint64_t *keys, const uint32_t *loc,
                                    const uint32_t *payload, const uint32_t *list, uint64_t n_list) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_list) return;
    const uint32_t i = list[t];
    const uint64_t key = keys[i];
    uint64_t h = slot_of(key, shift);
    for (;;) {
        unsigned long long *kp = reinterpret_cast<unsigned long long *>(&slots[h].key);
        const unsigned long long old = atomicCAS(kp, (unsigned long long)kEmptyKey, (unsigned long long)key);
        if (old == kEmptyKey || old == key) {
            slots[h].loc = loc[i];
            slots[h].payload = payload[i];
            return;
        }
        h = (h + 1) & mask;
    }
}

template <typename T>
int dev_alloc(T **p, uint64_t n) {
    *p = nullptr;
    HIP_TRY(hipMalloc((void **)p, std::max<uint64_t>(n, 1) * sizeof(T)));
    return GS_OK;
}

}  // namespace

// ==========================================================================================
// handles
// ==========================================================================================
struct gs_table {
    int device = 0;
    int k = 0;
    uint64_t n = 0, cap = 0;
    int shift = 0;
    Slot *slots = nullptr;
    // exact compact copy ("compact-hbm"): tables the minimiser-addressed layout does not take
    int compact = 0;  // 0 none, 2 compact
    uint4 *cbuckets = nullptr;
    uint32_t *cpayload = nullptr;  // payload per compact slot
    int c_qb = 0, c_rembits = 0, c_locbits = 0, c_has_flag = 0, c_w64 = 0;
    uint64_t c_overflow = 0;
    // minimiser-addressed table (gs_mz.h): large tables
    int mz = 0;
    uint4 *mz_regions = nullptr;
    uint32_t mz_n_regions = 0;
    int mz_m = 0, mz_locbits = 0;
    uint32_t *mz_pay = nullptr;  // payload per slot
    uint64_t mz_overflow = 0, mz_dup = 0, mz_hot = 0;  // keys in the generic side table; keys stored under both strands' picks; (place format) placed keys that hot regions gave up
    int mz_place = 0, mz_s = 0;            // place-record regions (gs_mzp.h): 2^(2 mz_m - mz_s) of them
    hipStream_t stream = nullptr;       // every pass over this table runs here, in launch order
    hipStream_t copy_stream = nullptr;  // hit lists to the host: overlaps the next pass
    gs_counters ctr{};
};

// State of ONE tagging pass.  It belongs to the read batch, not to the table: several batches can be in flight against one
// table (launched back to back on the table's stream, `prevLoc` handed from pass to pass in device memory), and each is
// settled on its own.
struct PassState {
    PassStatus *d_status = nullptr;       // the five scalars below live in this one block
    PassStatus *h_status = nullptr;       // pinned: filled by an async copy at the end of a pass
    PassStatus *h_init = nullptr;         // pinned: template the block is reset from
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // pass start, probe done, emit done, status on the host
    // scalars on device
    uint32_t *d_cand_count = nullptr;
    unsigned long long *d_err_pos = nullptr;
    uint32_t *d_carry_in = nullptr;
    Agg *d_totals = nullptr;  // [0] scan A grand total, [1] scan B grand total
    uint32_t *d_complex_n = nullptr;
    bool pass_pending = false;
    mutable bool was_chained_to = false;  // some later pass took (or may still take) its carry from this batch's device memory
    const gs_table *table = nullptr;      // of the pending / settled pass
    const struct gs_reads *after = nullptr;  // the batch whose pass hands its carry to this one on the device (or NULL)
    uint32_t last_carry_in = GS_NONE;
    bool settled = false;
    // outcome of the last settled pass
    uint64_t res_n = 0;
    uint32_t res_carry = GS_NONE;
    bool res_keyerr = false;
};

struct gs_reads {
    int device = 0;
    int k = 0;
    uint32_t n_reads = 0;
    uint32_t n_tiles = 0;
    uint64_t n_bases = 0, n_windows = 0, n_words = 0;
    uint64_t *words = nullptr;
    uint64_t *read_word_off = nullptr;
    uint32_t *read_nwin = nullptr, *read_tile0 = nullptr, *tile_read = nullptr;
    TileArrays ta{};
    Agg *block_agg = nullptr;
    uint32_t n_scan_blocks = 0;
    uint32_t *cand = nullptr;
    uint32_t cand_cap = 0;
    gs_hit *hits = nullptr;
    gs_hit *hits_host = nullptr;       // pinned mirror written by k_emit itself: no device-to-host copy after the pass
    uint64_t hit_cap = 0;
    uint32_t *d_seed = nullptr, *d_seg_carry = nullptr;  // [n_tiles], scatter-chunk emulation (gs_reads_set_seeds)
    std::vector<uint32_t> seed_tiles;                    // tiles that carry a seed, in the order they were given
    std::vector<uint32_t> read_tile0_host;               // first tile of every read
    TileDesc *tile_desc = nullptr;     // [n_tiles]
    PassState ps;
    // what the arrays hold room for (reads_alloc); a complete handle whose pass has settled is kept for the next batch of
    // its size instead of being freed (reads_pool below)
    uint64_t cap_words = 0, cap_tiles = 0;
    uint32_t cap_reads = 0;
    bool complete = false;
};

namespace {

TableView view(const gs_table *t) { return TableView{t->slots, t->cap - 1, t->shift, t->k}; }
MzView mz_view(const gs_table *t) {
    return MzView{t->mz_regions, t->mz_n_regions, t->k, t->mz_m, t->mz_locbits, t->mz_pay, t->mz_place, t->mz_s};
}
ReadsView view(const gs_reads *r) {
    return ReadsView{r->words, r->read_word_off, r->read_nwin, r->read_tile0, r->tile_read, r->n_tiles};
}

int ensure_out_buffers(gs_reads *r, uint32_t cand_cap) {
    if (r->cand && r->cand_cap >= cand_cap) return GS_OK;
    if (r->cand) (void)hipFree(r->cand);
    if (r->hits) (void)hipFree(r->hits);
    if (r->hits_host) (void)hipHostFree(r->hits_host);
    r->cand = nullptr;
    r->hits = nullptr;
    r->hits_host = nullptr;
    int rc;
    if ((rc = dev_alloc(&r->cand, cand_cap))) return rc;
    if ((rc = dev_alloc(&r->hits, cand_cap))) return rc;
    HIP_TRY(hipHostMalloc((void **)&r->hits_host, (size_t)cand_cap * sizeof(gs_hit), hipHostMallocDefault));
    r->cand_cap = cand_cap;
    r->hit_cap = cand_cap;
    return GS_OK;
}

int launch_phase1(gs_table *t, gs_reads *r) {
    HIP_TRY(hipSetDevice(t->device));
    memset(r->ps.h_init, 0, sizeof(PassStatus));
    r->ps.h_init->err_pos = ~0ULL;
    r->ps.h_init->totals[0].loc = r->ps.h_init->totals[1].loc = GS_NONE;
    r->ps.h_init->carry_in = GS_NONE;
    HIP_TRY(hipMemcpyAsync(r->ps.d_status, r->ps.h_init, sizeof(PassStatus), hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipEventRecord(r->ps.ev[0], t->stream));
    if (r->n_tiles) {
        const uint32_t blocks = (r->n_tiles + 3) / 4;
        const CompactView cv{t->cbuckets, t->cpayload, t->c_qb, t->c_rembits, t->c_locbits, t->c_has_flag, t->c_w64};
        // GAVISUNK_TABLE=minimizer-exact: every tile by the per-lane form (the fast form's fallback)
#if GS_MZ_PLACE
        if (t->mz == 2 && t->mz_place) {
            hipLaunchKernelGGL(k_probe_mzp_exact, dim3(blocks), dim3(256), 0, t->stream, MzpView{t->mz_regions, t->mz_pay, t->k, t->mz_m, t->mz_s},
                               view(t), (const uint64_t *)r->words, (const TileDesc *)r->tile_desc, r->n_tiles, r->ta, r->cand, r->cand_cap,
                               r->ps.d_cand_count);
        } else if (t->mz && t->mz_place) {
#define GS_LAUNCH_MZP(KT, MT, PL)                                                                                                     \
    hipLaunchKernelGGL((k_probe_mz<KT, MT, 1, PL>), dim3((r->n_tiles + kMzWaves - 1) / kMzWaves), dim3(64 * kMzWaves), 0, t->stream,   \
                       mz_view(t), view(t), (const uint64_t *)r->words, (const TileDesc *)r->tile_desc, r->n_tiles, r->ta,             \
                       r->cand, r->cand_cap, r->ps.d_cand_count, r->ps.d_complex_n)
            if (t->k == 20 && t->mz_m == 14)
                GS_LAUNCH_MZP(20, 14, 1);
            else if (t->k == 20 && t->mz_m == 15)
                GS_LAUNCH_MZP(20, 15, 1);
            else if (t->k == 21 && t->mz_m == 15)
                GS_LAUNCH_MZP(21, 15, 1);  // BASELINE.json configs[4], k = 21
            else if (t->k == 25 && t->mz_m == 19)
                GS_LAUNCH_MZP(25, 19, 2);  // BASELINE.json configs[4], k = 25 (a 38-bit minimiser: two registers)
            else if (t->mz_m > 15)
                GS_LAUNCH_MZP(0, 0, 2);
            else
                GS_LAUNCH_MZP(0, 0, 1);
#undef GS_LAUNCH_MZP
        } else
#endif
        if (t->mz == 2) {
            hipLaunchKernelGGL(k_probe_mz_exact, dim3(blocks), dim3(256), 0, t->stream, mz_view(t), view(t), (const uint64_t *)r->words,
                               (const TileDesc *)r->tile_desc, r->n_tiles, r->ta, r->cand, r->cand_cap, r->ps.d_cand_count);
        } else if (t->mz) {
#define GS_LAUNCH_MZ(KT, MT, FMT)                                                                                                  \
    hipLaunchKernelGGL((k_probe_mz<KT, MT, FMT>), dim3((r->n_tiles + kMzWaves - 1) / kMzWaves), dim3(64 * kMzWaves), 0, t->stream,  \
                       mz_view(t), view(t), (const uint64_t *)r->words, (const TileDesc *)r->tile_desc, r->n_tiles, r->ta,          \
                       r->cand, r->cand_cap, r->ps.d_cand_count, r->ps.d_complex_n)
            // GAVISUNK's default SUNK length and the minimiser lengths its tables get, the slot format that goes with them
            // (k = 20: the key itself up to 2^24 location groups), and BASELINE.json configs[4] (k = 21 and 25): constants
            const bool raw = mz_raw(t->k, t->mz_locbits);
            if (t->k == 20 && t->mz_m == 16 && raw)
                GS_LAUNCH_MZ(20, 16, 1);
            else if (t->k == 20 && t->mz_m == 17 && raw)
                GS_LAUNCH_MZ(20, 17, 1);
            else if (t->k == 20 && t->mz_m == 18 && raw)
                GS_LAUNCH_MZ(20, 18, 1);
            else if (t->k == 21 && t->mz_m == 17 && raw)
                GS_LAUNCH_MZ(21, 17, 1);
            else if (t->k == 25 && t->mz_m == 21)
                GS_LAUNCH_MZ(25, 21, -1);
            else
                GS_LAUNCH_MZ(0, 0, -1);
#undef GS_LAUNCH_MZ
        } else if (t->compact) {
            hipLaunchKernelGGL(k_probe<true>, dim3(blocks), dim3(256), 0, t->stream, view(t), cv, view(r), r->ta, r->cand,
                               r->cand_cap, r->ps.d_cand_count, r->ps.d_err_pos, (const uint32_t *)nullptr,
                               (const uint32_t *)nullptr);
        } else {
            hipLaunchKernelGGL(k_probe<false>, dim3(blocks), dim3(256), 0, t->stream, view(t), cv, view(r), r->ta, r->cand,
                               r->cand_cap, r->ps.d_cand_count, r->ps.d_err_pos, (const uint32_t *)nullptr,
                               (const uint32_t *)nullptr);
        }
    }
    HIP_TRY(hipEventRecord(r->ps.ev[1], t->stream));
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

// a batch without tiles: its totals are "nothing", or what it was handed
__global__ void k_empty_totals(const uint32_t *carry_in, Agg *total, int mode) {
    Agg z{0, GS_NONE, 0};
    if (mode == 0 && *carry_in != GS_NONE) {
        z.loc = *carry_in;
        z.has = 1;
    }
    *total = z;
}
// prevLoc handed from one pass to the next in device memory (kmerpos_annot3.nim:82-84: the state crosses reads, and batches)
__global__ void k_chain_carry(const Agg *prev_total, uint32_t *carry_in) { *carry_in = (prev_total->has & 1u) ? prev_total->loc : GS_NONE; }

int launch_scan(gs_table *t, gs_reads *r, int mode) {
    const uint32_t nb = r->n_scan_blocks;
    if (!r->n_tiles) {
        hipLaunchKernelGGL(k_empty_totals, dim3(1), dim3(1), 0, t->stream, (const uint32_t *)r->ps.d_carry_in, r->ps.d_totals + mode, mode);
        return GS_OK;
    }
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(kScanThreads), 0, t->stream, r->ta, r->n_tiles, mode, r->block_agg);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(kScanThreads), 0, t->stream, r->block_agg, nb,
                       mode == 0 ? r->ps.d_carry_in : (const uint32_t *)nullptr, r->ps.d_totals + mode);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(kScanThreads), 0, t->stream, r->ta, r->n_tiles, mode, r->block_agg);
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

// `after` != NULL: the carry is what the pass of that batch (launched before this one on the same table) carries out
int launch_phase2(gs_table *t, gs_reads *r, uint32_t carry_in, const gs_reads *after = nullptr) {
    HIP_TRY(hipSetDevice(t->device));
    r->ps.last_carry_in = carry_in;
    r->ps.after = after;
    if (after) {
        after->ps.was_chained_to = true;  // (its device memory is read by this pass: such a handle is never recycled, only freed behind a device synchronisation)
        hipLaunchKernelGGL(k_chain_carry, dim3(1), dim3(1), 0, t->stream, (const Agg *)after->ps.d_totals, r->ps.d_carry_in);
    }
    else
        HIP_TRY(hipMemcpyAsync(r->ps.d_carry_in, &r->ps.last_carry_in, sizeof(uint32_t), hipMemcpyHostToDevice, t->stream));
    int rc;
    if ((rc = launch_scan(t, r, 0))) return rc;
    if ((rc = launch_scan(t, r, 1))) return rc;
    if (r->n_tiles) {
        const CompactView cv{t->cbuckets, t->cpayload, t->c_qb, t->c_rembits, t->c_locbits, t->c_has_flag, t->c_w64};
        hipLaunchKernelGGL(k_emit, dim3((r->n_tiles + 255) / 256), dim3(256), 0, t->stream, view(t), cv, mz_view(t), t->mz ? 4 : t->compact, view(r), r->ta,
                           r->cand, r->cand_cap, r->hits, r->hits_host, r->hit_cap, (const Agg *)(r->ps.d_totals + 1));
    }
    HIP_TRY(hipEventRecord(r->ps.ev[2], t->stream));
    HIP_TRY(hipMemcpyAsync(r->ps.h_status, r->ps.d_status, sizeof(PassStatus), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipEventRecord(r->ps.ev[3], t->stream));
    HIP_TRY(hipGetLastError());
    r->ps.pass_pending = true;
    r->ps.settled = false;
    r->ps.table = t;
    return GS_OK;
}

}  // namespace

namespace {

uint64_t revcomp_code(uint64_t c, int k) {
    uint64_t r = 0;
    for (int i = 0; i < k; i++) {
        r = (r << 2) | (3 - (c & 3));
        c >>= 2;
    }
    return r;
}

// Layout choice for the exact compact copy of the table:
//   2 = "compact-hbm": 32-bit slots; one 16-byte bucket load per window                                       -> k_probe<true>
//   3 = "compact-hbm": 64-bit slots (remainder + loc do not fit 32 bits: long k, many groups, huge tables)    -> k_probe<true>
//   0 = generic 16-byte slots only (k < 16, or a key without loc row)
int choose_compact(int k, uint64_t n, uint64_t generic_bytes, uint32_t maxloc, bool any_missing, const char *env, int &qb,
                   int &rembits, int &locbits, int &has_flag) {
    if (env && !strcmp(env, "generic")) return 0;
    if (k < 16 || n == 0 || any_missing) return 0;
    locbits = 1;
    while ((1ull << locbits) < (uint64_t)maxloc + 2) locbits++;  // the all-ones loc stays unused
    const int kbits = 2 * k - 4;
    int qb_nat = 8;
    while (((uint64_t)(4 * kNPart) << qb_nat) < (uint64_t)(2.2 * (double)n)) qb_nat++;
    const int lbits = kbits < 32 ? kbits : 32;
    auto valid_qb = [&](int q) { return q >= 4 && q <= 28 && q < lbits && kbits - q >= 1; };
    // 32-bit slots: with the natural bucket count, or with more buckets (a shorter remainder) while the table stays small
    qb = qb_nat;
    while (kbits - qb + locbits > 32) qb++;
    if (const char *fq = getenv("GAVISUNK_COMPACT_QB")) {  // test knob: an undersized table makes every overflow path busy
        const int v = atoi(fq);
        if (valid_qb(v) && kbits - v + locbits <= 32) qb = v;
    }
    const bool fits32 = valid_qb(qb) && ((uint64_t)(16 * kNPart) << qb) <= std::max<uint64_t>(generic_bytes, 32ull << 20);
    if (fits32) {
        rembits = kbits - qb;
        has_flag = rembits + 1 + locbits <= 30;
        return 2;
    }
    // 64-bit slots with the natural bucket count
    qb = qb_nat;
    rembits = kbits - qb;
    if (!valid_qb(qb) || rembits + 1 + locbits > 62) return 0;
    has_flag = 1;
    return 3;
}

#if GS_MZ_PLACE
// Place-record regions (gs_mzp.h): k - 6 <= m <= 15 (a tag word holds two flanks of w - 1 <= 6 bases; regions are addressed by
// a bijection of the 2m-bit m-mer).  The shortest such m whose regions stay below ~2.5 places on average (7 fit); s from the
// number of places: 2^(2m - s) regions, about one per place.
// An assembly whose SUNKs come from point variants inside otherwise repeated sequence (a diploid assembly: a SNP in one copy
// of a repeat family makes the ~k k-mers over it unique) has them in many small location groups, and under a short minimiser
// dozens of copies' variants share one m-mer: such a region overflows many times over, reads visit it in proportion to the
// copies, nearly always for the k-mers the copies have in common, and a 32-bit filter of that many keys is all ones (measured
// on BASELINE.json configs[3]: 6 % of all windows went on to the generic table).  Tables with fewer than 2000 keys per
// location group are built with HOT regions (k_mzp_place): a region with that many misfits keeps no record at all, its keys
// live in the generic table and its 60 bytes are one filter of them.
uint32_t mzp_hot_min(uint64_t n, uint32_t maxloc) {
    if (const char *e = getenv("GAVISUNK_MZ_HOT")) return (uint32_t)std::max(0, atoi(e));  // test knob (0: never, n: regions with n misfits)
    return (double)n < 2000.0 * ((double)maxloc + 1.0) ? 6u : 0u;
}
int mzp_choose(int k, uint64_t n, uint32_t maxloc, bool any_missing, int &m, int &s, int m_from = 0, bool forced = false) {
    (void)forced;
    // k <= 21: m = k - 6 .. 15 by size; k = 22 .. 27: m = k - 6 only (w = 7: a 32-bit tag holds the 2 x 6 flank bases, the 7
    // offset bits and the sign; the minimiser's value then is 2m = 32 .. 42 bits: two registers in k_probe_mz)
    if (k < 16 || k > 27 || n == 0 || any_missing) return 0;
    int m_lo = std::max(std::max(k - 6, 10), m_from), m_hi = k > 21 ? k - 6 : std::min(15, k - 1);
    if (m_lo > m_hi) return 0;
    if (const char *e = getenv("GAVISUNK_MZ_M")) {  // tuning / test knob, read once per table
        const int v = atoi(e);
        if (v >= m_lo && v <= m_hi) m_lo = m_hi = v;
    }
    for (m = m_lo; m <= m_hi; m++) {
        const double w = k - m + 1, places = 1.1 * (double)n / std::max(1.0, 0.6 * w);
        // regions = the power of two next above places / 1.5 (s >= 1: half of the 2m-bit values are no canonical m-mers)
        // (m > 15: s <= 2m - 14, at least 2^14 regions; s >= 2m - 31: the region index is 31 bits)
        const int s_hi = m <= 15 ? 16 : std::min(2 * m - 14, 31), s_lo = std::max(1, 2 * m - 31);
        s = s_lo;
        while (s < s_hi && std::ldexp(1.0, 2 * m - (s + 1)) >= places / 1.5) s++;
        const double load = places / std::ldexp(1.0, 2 * m - s);
        if (const char *e = getenv("GAVISUNK_MZ_LOAD")) {  // test knob: crowded regions on purpose
            const double v = atof(e);
            if (v >= 0.25 && v <= 64.0) {
                s = s_lo;
                while (s < s_hi && std::ldexp(1.0, 2 * m - (s + 1)) >= places / v) s++;
            }
        } else if ((load > 3.0 || (double)n > 3.0e8 * std::ldexp(1.0, 2 * (m - 14))) && m < m_hi) {
            // (the m-mers a mod-minimiser samples are a biased subset of the 4^m values -- those that begin or end with a small
            // t-mer -- so the places crowd into a fraction of the regions: measured, a 3e8-key table is what m = 14 takes)
            continue;
        }
        while (s > s_lo && ((uint64_t)maxloc + 2) >> (32 - s)) s--;  // the loc word holds fp and loc (s >= 1: k_probe_mz shifts by 32 - s)
        if (((uint64_t)maxloc + 2) >> (32 - s)) return 0;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
        if (std::ldexp(96.0, 2 * m - s) > 0.5 * (double)free_b) continue;
        return 1;
    }
    return 0;
}
#endif

// Minimiser-addressed layout (gs_mz.h): applies when a slot [2k-3 tag bits | loc] fits 64 bits.  m from the key count: the
// places of the assembly that own a minimiser (about n / (0.65 (w + 1) / 2)) should be few against the 4^m / 2 values.
int mz_choose(int k, uint64_t n, uint32_t maxloc, bool any_missing, int &m, uint32_t &n_regions, int &locbits) {
    if (k < 16 || n == 0 || any_missing) return 0;
    locbits = 1;
    while ((1ull << locbits) < (uint64_t)maxloc + 2) locbits++;
    if (2 * k - 3 + locbits > 64) return 0;
    const int m_hi = std::min(31, k - 1);
    m = std::max(8, k - 4);  // w <= 5: the adjacent k-mers one minimiser owns are a third of a region (measured: the probe
                             // kernel is instruction-bound, so shorter runs cost nothing while crowded regions do)
    for (; m < m_hi; m++) {
        const double w = k - m + 1, places = (double)n / (0.65 * (w + 1) / 2);
        if (places <= 0.06 * std::pow(4.0, m)) break;  // (0.25 left 13 % of a 1.5e9-key table's keys in full regions)
    }
    if (const char *e = getenv("GAVISUNK_MZ_M")) {  // tuning / test knob, read once per table
        const int v = atoi(e);
        if (v >= 4 && v <= m_hi) m = v;
    }
    double per_region = 1.25;  // keys per 7-slot region (measured on the 100 Mb rehearsal: 1.0-1.25 tag at 133 Gbase/s, 1.5 at 124, 2 at 98)
    bool forced_load = false;
    if (const char *e = getenv("GAVISUNK_MZ_LOAD")) {  // test knob: an undersized table makes the walk and the overflow path busy
        const double v = atof(e);
        if (v >= 0.25 && v <= 16.0) {
            per_region = v;
            forced_load = true;
        }
    }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    uint64_t regions = std::max<uint64_t>((uint64_t)((double)n / per_region), 64);
    regions = std::min<uint64_t>(regions, (uint64_t)(0.5 * (double)free_b) / (kMzWords * 12));
    regions = std::min<uint64_t>(regions, 0xFFFFFFF0ull);
    if (regions < 64 || (!forced_load && (double)n / (double)regions > 5.0)) return 0;
    n_regions = (uint32_t)regions;
    return 1;
}

__global__ void k_loc_stats(const uint32_t *loc, uint64_t n, uint32_t *out) {  // out[0] max loc below GS_MISSING, out[1] any >= GS_MISSING
    uint32_t mx = 0, miss = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = loc[i];
        if (l >= GS_MISSING) miss = 1;
        else mx = max(mx, l);
    }
    atomicMax(out, mx);
    if (miss) atomicMax(out + 1, 1u);
}

__global__ void k_gather_list(const uint64_t *keys, const uint32_t *loc, const uint32_t *list, uint64_t n, uint64_t *ok, uint32_t *ol) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        ok[i] = keys[list[i]];
        ol[i] = loc[list[i]];
    }
}

__global__ void k_iota(uint32_t *p, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}

struct DevBufs {  // temporary device buffers of a build, freed on every return path
    std::vector<void *> p;
    ~DevBufs() {
        for (void *x : p) (void)hipFree(x);
    }
    template <typename T>
    int get(T **d, uint64_t n) {
        int rc = dev_alloc(d, n);
        if (!rc) p.push_back(*d);
        return rc;
    }
};

int table_init(gs_table *t, int device, int k, uint64_t n) {
    t->device = device;
    t->k = k;
    t->n = n;
    HIP_TRY(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&t->copy_stream, hipStreamNonBlocking));
    return GS_OK;
}

int pass_state_init(PassState &ps) {
    int rc;
    for (auto &e : ps.ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDefault));
    if ((rc = dev_alloc(&ps.d_status, 1))) return rc;
    HIP_TRY(hipHostMalloc((void **)&ps.h_status, sizeof(PassStatus), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&ps.h_init, sizeof(PassStatus), hipHostMallocDefault));
    ps.d_err_pos = &ps.d_status->err_pos;
    ps.d_totals = ps.d_status->totals;
    ps.d_cand_count = &ps.d_status->cand_count;
    ps.d_complex_n = &ps.d_status->complex_n;
    ps.d_carry_in = &ps.d_status->carry_in;
    return GS_OK;
}

void pass_state_free(PassState &ps) {
    (void)hipFree(ps.d_status);
    if (ps.h_status) (void)hipHostFree(ps.h_status);
    if (ps.h_init) (void)hipHostFree(ps.h_init);
    for (auto &e : ps.ev)
        if (e) (void)hipEventDestroy(e);
    ps = PassState{};
}

// Builds the device tables from key / loc / payload arrays that already are in device memory (not owned).
int table_build(gs_table *t, const uint64_t *dk, const uint32_t *dl, const uint32_t *dp, uint64_t n) {
    const int k = t->k;
    int rc;
    DevBufs tmp;
    uint32_t *dlist = nullptr, *dstat = nullptr;
    uint32_t stat[2] = {0, 0};
    if ((rc = tmp.get(&dstat, 2))) return rc;
    HIP_TRY(hipMemsetAsync(dstat, 0, 2 * sizeof(uint32_t), t->stream));
    if (n) hipLaunchKernelGGL(k_loc_stats, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, t->stream, dl, n, dstat);
    HIP_TRY(hipMemcpyAsync(stat, dstat, sizeof stat, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    const uint32_t maxloc = stat[0];
    const bool any_missing = stat[1] != 0;
    auto pow2_cap = [](uint64_t want, int &lg) {
        uint64_t c = 1024;
        lg = 10;
        while (c < want) {
            c <<= 1;
            lg++;
        }
        return c;
    };
    int lg_full;
    const uint64_t cap_full = pow2_cap(2 * n, lg_full);
    // Layout (GAVISUNK_TABLE forces one; read once per table).  The minimiser-addressed layout whenever it applies (k >= 16, the
    // slot fits 64 bits): its kernel is the fastest on every table size, cache-resident tables included (the round-1 layouts for
    // small tables -- an LDS join, XCD-partitioned L2 probing -- lost to it on their home ground and were retired in round 3).
    // Otherwise an exact compact copy, one bucket load per window ("compact-hbm"); the generic 16-byte-slot table holds what
    // those could not take, or everything ("generic": k < 16, a db k-mer without loc row).
    const char *env = getenv("GAVISUNK_TABLE");
    if (env && !strcmp(env, "auto")) env = nullptr;
    const bool force_mz = env && (!strcmp(env, "minimizer") || !strcmp(env, "minimizer-exact") || !strcmp(env, "minimizer-slots"));
#if GS_MZ_PLACE
    const bool want_slots = env && !strcmp(env, "minimizer-slots");  // the per-key-slot regions (the only form for k > 21)
    if ((force_mz || !env) && !want_slots && mzp_choose(k, n, maxloc, any_missing, t->mz_m, t->mz_s, 0, force_mz)) {
        t->mz = force_mz && !strcmp(env, "minimizer-exact") ? 2 : 1;
        t->mz_place = 1;
        t->mz_n_regions = 1u << (2 * t->mz_m - t->mz_s);
        t->mz_locbits = 32 - t->mz_s;
        t->compact = 0;
    } else if ((force_mz || !env) && mz_choose(k, n, maxloc, any_missing, t->mz_m, t->mz_n_regions, t->mz_locbits)) {
        t->mz = force_mz && !strcmp(env, "minimizer-exact") ? 2 : 1;
        t->compact = 0;
    } else {
#else
    if ((force_mz || !env) && mz_choose(k, n, maxloc, any_missing, t->mz_m, t->mz_n_regions, t->mz_locbits)) {
        t->mz = force_mz && !strcmp(env, "minimizer-exact") ? 2 : 1;
        t->compact = 0;
    } else {
#endif
        t->compact = choose_compact(k, n, cap_full * sizeof(Slot), maxloc, any_missing, force_mz ? nullptr : env, t->c_qb, t->c_rembits, t->c_locbits, t->c_has_flag);
    }
    uint64_t n_generic = n;
    // A minimiser length whose regions cannot hold the keys (more than an eighth of them overflow: clustered minimisers, a
    // repeat-rich assembly) is no reason to fail: the next length is tried (shorter runs, emptier regions), and when none
    // does, the layouts below take the table -- only a FORCED layout reports the misfit.
#if GS_MZ_PLACE
    while (t->mz && t->mz_place) {
        const uint64_t nreg = t->mz_n_regions;
        const bool knob = getenv("GAVISUNK_MZ_M") || getenv("GAVISUNK_MZ_LOAD");
        const uint32_t hot_min = mzp_hot_min(n, maxloc);
        const uint64_t ocap = std::max<uint64_t>(knob && force_mz ? 2 * n : hot_min ? n / 4 : n / 16, 1u << 16);
        unsigned long long *dov = nullptr;
        if ((rc = dev_alloc((unsigned long long **)&t->mz_regions, nreg * kMzWords)) || (rc = dev_alloc(&t->mz_pay, nreg * 8)) ||
            (rc = tmp.get(&dov, 8)) || (rc = tmp.get(&dlist, ocap)))
            return rc;
        HIP_TRY(hipMemsetAsync(t->mz_regions, 0, nreg * kMzWords * sizeof(unsigned long long), t->stream));
        HIP_TRY(hipMemsetAsync(dov, 0, 8 * sizeof(unsigned long long), t->stream));
        uint64_t n2 = n;
        {
            DevBufs st;
            uint64_t *ent1 = nullptr;
            {
                DevBufs se;
                uint64_t *ent0 = nullptr;
                unsigned long long *dnd = nullptr;
                if ((rc = se.get(&ent0, 2 * n)) || (rc = st.get(&ent1, 2 * n)) || (rc = se.get(&dnd, 1))) return rc;
                HIP_TRY(hipMemsetAsync(dnd, 0, sizeof(unsigned long long), t->stream));
                hipLaunchKernelGGL(k_mzp_entries, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, t->stream, dk, n, k, t->mz_m, t->mz_s, ent0, dnd);
                if ((rc = sort_keys_u64(ent0, ent1, 2 * n, t->stream))) return rc;
                unsigned long long nd = 0;
                HIP_TRY(hipMemcpyAsync(&nd, dnd, sizeof nd, hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipStreamSynchronize(t->stream));
                n2 = n + nd;
                t->mz_dup = nd;
            }
            hipLaunchKernelGGL(k_mzp_place, dim3((unsigned)((n2 + 127) / 128)), dim3(128), 0, t->stream, const_cast<uint4 *>(t->mz_regions), t->mz_pay, k,
                               t->mz_m, t->mz_s, dk, dl, dp, (const uint64_t *)ent1, n2, dov, dlist, ocap, hot_min);
            HIP_TRY(hipStreamSynchronize(t->stream));
        }
        unsigned long long ovs[8] = {0};
        HIP_TRY(hipMemcpyAsync(ovs, dov, sizeof ovs, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        HIP_TRY(hipGetLastError());
        const unsigned long long ov = ovs[0];
        if (getenv("GAVISUNK_DEBUG_BUILD"))
            fprintf(stderr, "place records m = %d s = %d: %llu of %llu keys to the generic table (loc too wide %llu, reserved value %llu, clash %llu, region full %llu, rows %llu, placed keys of hot regions %llu), %llu entries in regions of more than 32, %llu under both picks\n",
                    t->mz_m, t->mz_s, ov, (unsigned long long)n, ovs[1], ovs[2], ovs[3], ovs[4], ovs[5], ovs[6], ovs[7], (unsigned long long)t->mz_dup);
        if (ov <= ocap) {
            t->mz_overflow = ov;
            t->mz_hot = ovs[6];
            n_generic = ov;
            break;
        }
        // more than a sixteenth of the keys outside their regions: the next minimiser length (more, emptier regions), then the
        // per-key-slot regions, then the compact / generic layouts -- only a forced geometry reports the misfit
        if (knob && force_mz) {
            set_error("gs_table_create: %llu of %llu keys do not fit the place records (m = %d, s = %d); force another layout with GAVISUNK_TABLE",
                      ov, (unsigned long long)n, t->mz_m, t->mz_s);
            return GS_E_NOMEM;
        }
        (void)hipFree(t->mz_regions);
        (void)hipFree(t->mz_pay);
        t->mz_regions = nullptr;
        t->mz_pay = nullptr;
        dlist = nullptr;
        const int tried = t->mz_m;
        if (!getenv("GAVISUNK_MZ_M") && mzp_choose(k, n, maxloc, any_missing, t->mz_m, t->mz_s, tried + 1, force_mz)) {
            t->mz_n_regions = 1u << (2 * t->mz_m - t->mz_s);
            t->mz_locbits = 32 - t->mz_s;
            continue;
        }
        t->mz_place = 0;
        if (!mz_choose(k, n, maxloc, any_missing, t->mz_m, t->mz_n_regions, t->mz_locbits)) {
            t->mz = 0;
            t->compact = choose_compact(k, n, cap_full * sizeof(Slot), maxloc, any_missing, nullptr, t->c_qb, t->c_rembits, t->c_locbits, t->c_has_flag);
        }
    }
    while (t->mz && !t->mz_place) {
#else
    while (t->mz) {
#endif
        const uint64_t nslots = (uint64_t)t->mz_n_regions * kMzWords;
        const bool knob = getenv("GAVISUNK_MZ_M") || getenv("GAVISUNK_MZ_LOAD");
        const uint64_t ocap = std::max<uint64_t>(knob && force_mz ? 2 * n : n / 8, 1u << 16);
        unsigned long long *dov = nullptr;
        DevBufs ot;
        if ((rc = dev_alloc((unsigned long long **)&t->mz_regions, nslots)) || (rc = dev_alloc(&t->mz_pay, nslots)) ||
            (rc = ot.get(&dov, 1)) || (rc = ot.get(&dlist, ocap)))
            return rc;
        HIP_TRY(hipMemsetAsync(t->mz_regions, 0xFF, nslots * sizeof(unsigned long long), t->stream));
        HIP_TRY(hipMemsetAsync(dov, 0, sizeof(unsigned long long), t->stream));
        {  // one entry per key and strand pick (k_mz_entries), sorted by (region, key index)
            uint64_t n2 = n;
            uint64_t *reg1 = nullptr;
            uint32_t *idx1 = nullptr;
            DevBufs st;
            {
                DevBufs se;
                uint64_t *ent0 = nullptr, *ent1 = nullptr;
                unsigned long long *dnd = nullptr;
                if ((rc = se.get(&ent0, 2 * n)) || (rc = st.get(&ent1, 2 * n)) || (rc = se.get(&dnd, 1))) return rc;
                HIP_TRY(hipMemsetAsync(dnd, 0, sizeof(unsigned long long), t->stream));
                hipLaunchKernelGGL(k_mz_entries, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, t->stream, dk, n, k, t->mz_m, t->mz_n_regions, ent0, dnd);
                if ((rc = sort_keys_u64(ent0, ent1, 2 * n, t->stream))) return rc;
                unsigned long long nd = 0;
                HIP_TRY(hipMemcpyAsync(&nd, dnd, sizeof nd, hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipStreamSynchronize(t->stream));
                n2 = n + nd;
                t->mz_dup = nd;
                if ((rc = st.get(&reg1, n2)) || (rc = st.get(&idx1, n2))) return rc;
                hipLaunchKernelGGL(k_mz_split_entries, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, t->stream, (const uint64_t *)ent1, n2, reg1, idx1);
                HIP_TRY(hipStreamSynchronize(t->stream));
            }
            hipLaunchKernelGGL(k_mz_place, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, t->stream, reinterpret_cast<unsigned long long *>(t->mz_regions),
                               t->mz_pay, k, t->mz_locbits, dk, dl, dp, (const uint64_t *)reg1, (const uint32_t *)idx1, n2, dov, dlist, ocap);
            HIP_TRY(hipStreamSynchronize(t->stream));
        }
        unsigned long long ov = 0;
        HIP_TRY(hipMemcpyAsync(&ov, dov, sizeof ov, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        HIP_TRY(hipGetLastError());
        if (ov <= ocap) {
            t->mz_overflow = ov;
            n_generic = ov;
            tmp.p.insert(tmp.p.end(), ot.p.begin(), ot.p.end());  // (the overflow list is read once more below)
            ot.p.clear();
            break;
        }
        if (force_mz) {
            set_error("gs_table_create: %llu of %llu keys do not fit their minimiser regions (m = %d); force another layout with GAVISUNK_TABLE",
                      ov, (unsigned long long)n, t->mz_m);
            return GS_E_NOMEM;
        }
        (void)hipFree(t->mz_regions);
        (void)hipFree(t->mz_pay);
        t->mz_regions = nullptr;
        t->mz_pay = nullptr;
        dlist = nullptr;
        if (!getenv("GAVISUNK_MZ_M") && t->mz_m < std::min(31, k - 1) && t->mz_m < k - 2) {
            t->mz_m++;
            continue;
        }
        t->mz = 0;
        t->compact = choose_compact(k, n, cap_full * sizeof(Slot), maxloc, any_missing, nullptr, t->c_qb, t->c_rembits, t->c_locbits, t->c_has_flag);
    }
    if (t->mz) {
    } else if (t->compact) {
        t->c_w64 = t->compact == 3;
        if (t->compact == 3) t->compact = 2;  // same probe kernel; the view says which slot width
        const uint64_t nslots = (uint64_t)(4 * kNPart) << t->c_qb;
        const uint64_t slot_words = t->c_w64 ? 2 * nslots : nslots;  // in uint32 units
        uint32_t *d = nullptr;
        unsigned long long *dov = nullptr;
        if ((rc = dev_alloc(&d, slot_words))) return rc;
        t->cbuckets = reinterpret_cast<uint4 *>(d);
        if ((rc = dev_alloc(&t->cpayload, nslots)) || (rc = tmp.get(&dov, 1)) || (rc = tmp.get(&dlist, n))) return rc;
        HIP_TRY(hipMemsetAsync(d, 0xFF, slot_words * sizeof(uint32_t), t->stream));
        HIP_TRY(hipMemsetAsync(dov, 0, sizeof(unsigned long long), t->stream));
        {
            const CompactGeom g{k, t->c_qb, t->c_rembits, t->c_locbits, t->c_has_flag};
            const uint64_t nbuckets = (uint64_t)kNPart << t->c_qb;
            DevBufs st;
            uint64_t *id0 = nullptr, *id1 = nullptr;
            uint32_t *ix0 = nullptr, *ix1 = nullptr, *left = nullptr;
            uint8_t *hcount = nullptr, *lflag = nullptr;
            unsigned long long *dnl = nullptr;
            if ((rc = st.get(&id0, n)) || (rc = st.get(&id1, n)) || (rc = st.get(&ix0, n)) || (rc = st.get(&ix1, n)) ||
                (rc = st.get(&left, n)) || (rc = st.get(&hcount, nbuckets)) || (rc = st.get(&lflag, n)) || (rc = st.get(&dnl, 1)))
                return rc;
            const unsigned nb = (unsigned)((n + 255) / 256);
            HIP_TRY(hipMemsetAsync(hcount, 0, nbuckets, t->stream));
            hipLaunchKernelGGL(k_compact_home_ids, dim3(nb), dim3(256), 0, t->stream, dk, n, g, id0);
            hipLaunchKernelGGL(k_iota, dim3(nb), dim3(256), 0, t->stream, ix0, n);
            if ((rc = sort_pairs_u64_u32(id0, id1, ix0, ix1, n, 4 + t->c_qb, t->stream))) return rc;
            if (t->c_w64)
                hipLaunchKernelGGL((k_compact_place_home<unsigned long long>), dim3(nb), dim3(256), 0, t->stream,
                                   reinterpret_cast<unsigned long long *>(d), t->cpayload, hcount, g, dk, dl, dp, (const uint64_t *)id1,
                                   (const uint32_t *)ix1, n, lflag);
            else
                hipLaunchKernelGGL((k_compact_place_home<uint32_t>), dim3(nb), dim3(256), 0, t->stream, d, t->cpayload, hcount, g, dk, dl,
                                   dp, (const uint64_t *)id1, (const uint32_t *)ix1, n, lflag);
            // the left-over keys, in sorted order (stable selection), then sorted by their second-chance bucket
            uint64_t n_left = 0;
            if ((rc = select_flagged_u32(ix1, lflag, left, dnl, n, &n_left, t->stream))) return rc;
            if (n_left) {
                const unsigned nbl = (unsigned)((n_left + 255) / 256);
                hipLaunchKernelGGL(k_compact_alt_ids, dim3(nbl), dim3(256), 0, t->stream, dk, (const uint32_t *)left, n_left, g, id0);
                if ((rc = sort_pairs_u64_u32(id0, id1, left, ix0, n_left, 33, t->stream))) return rc;
                if (t->c_w64)
                    hipLaunchKernelGGL((k_compact_place_alt<unsigned long long>), dim3(nbl), dim3(256), 0, t->stream,
                                       reinterpret_cast<unsigned long long *>(d), t->cpayload, (const uint8_t *)hcount, g, dk, dl, dp,
                                       (const uint64_t *)id1, (const uint32_t *)ix0, n_left, dov, dlist);
                else
                    hipLaunchKernelGGL((k_compact_place_alt<uint32_t>), dim3(nbl), dim3(256), 0, t->stream, d, t->cpayload,
                                       (const uint8_t *)hcount, g, dk, dl, dp, (const uint64_t *)id1, (const uint32_t *)ix0, n_left, dov, dlist);
            }
            HIP_TRY(hipStreamSynchronize(t->stream));
        }
        unsigned long long ov = 0;
        HIP_TRY(hipMemcpyAsync(&ov, dov, sizeof ov, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        HIP_TRY(hipGetLastError());
        t->c_overflow = ov;
        n_generic = ov;
    }
    int lg;
    // (behind a minimiser-addressed table the generic slots are probed by deferred lookups, most of which miss: a load of 1/4
    // keeps such a probe at ~1.2 requests)
    t->cap = pow2_cap((t->mz ? 4 : 2) * n_generic, lg);
    t->shift = 64 - lg;
    if ((rc = dev_alloc(&t->slots, t->cap))) return rc;
    hipLaunchKernelGGL(k_table_fill, dim3((unsigned)((t->cap + 255) / 256)), dim3(256), 0, t->stream, t->slots, t->cap);
    if (n_generic) {
        if (t->compact || t->mz)
            hipLaunchKernelGGL(k_table_insert_list, dim3((unsigned)((n_generic + 255) / 256)), dim3(256), 0, t->stream,
                               t->slots, t->cap - 1, t->shift, dk, dl, dp, dlist, n_generic);
        else
            hipLaunchKernelGGL(k_table_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, t->stream, t->slots,
                               t->cap - 1, t->shift, dk, dl, dp, n);
    }
    HIP_TRY(hipStreamSynchronize(t->stream));
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

}  // namespace

extern "C" {

int gs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gs_device_synchronize(int device) {
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_device_synchronize: HIP device %d not available", device);
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return GS_OK;
}

// How this process waits for the device: HIP's default is to SPIN (lowest latency: what the engine and bench.py want).  A command line
// whose host side is the bottleneck -- kmerpos_annot3 on a .fq.gz: every CPU of the box's quota inflates, while the main, staging
// and writer threads each burn one waiting for a copy or a kernel -- asks for blocking waits instead.  Call before anything else
// touches the device.
int gs_device_blocking_sync(int device, int on) {
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_device_blocking_sync: HIP device %d not available", device);
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipSetDeviceFlags(on ? hipDeviceScheduleBlockingSync : hipDeviceScheduleSpin));
    return GS_OK;
}

#ifdef GS_TUNING
// (libgavisunk_tuning.so only; not in include/gavisunk.h) the census counters of k_probe_mz (gs_mz.h: g_mz_census) of `device`
int gs_tuning_census(int device, uint64_t out[16], int reset) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    if (out) HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mz_census), 16 * sizeof(uint64_t)));
    if (reset) {
        const uint64_t z[16] = {0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_mz_census), z, sizeof(z)));
    }
    return GS_OK;
}
#endif

int gs_selftest_minimizer_rule(int device, int k, int m, uint64_t seed, uint32_t n_tiles, uint64_t *mismatches, uint32_t first_bad[4],
                               uint32_t lane_hist[64]) {
    if (!mismatches || !first_bad || k < 16 || k > 32 || m < 8 || m > k || m > 31 || k - m > 7 || !n_tiles || n_tiles > (1u << 20)) {
        set_error("gs_selftest_minimizer_rule: bad argument (16 <= k <= 32, max(8, k - 7) <= m <= min(k, 31), 1 .. 2^20 tiles)");
        return GS_E_ARG;
    }
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_selftest_minimizer_rule: HIP device %d not available", device);
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    const size_t nw = (size_t)n_tiles * kTileWords + 4;
    std::vector<uint64_t> h(nw);
    uint64_t x = seed;
    for (size_t i = 0; i < nw; i++) {  // splitmix64; every fourth tile is low-complexity (many equal t-mers: the tie rule)
        x += 0x9E3779B97F4A7C15ULL;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        if ((i / kTileWords) % 4 == 3) z = (z & 0x0303030303030303ULL) * 0x15ULL;  // runs of four equal bases
        h[i] = z;
    }
    uint64_t *d_words = nullptr;
    unsigned long long *d_mis = nullptr;
    uint32_t *d_bad = nullptr;
    HIP_TRY(hipMalloc(&d_words, nw * 8));
    HIP_TRY(hipMalloc(&d_mis, 8));
    HIP_TRY(hipMalloc(&d_bad, 4 * 68));
    HIP_TRY(hipMemcpy(d_words, h.data(), nw * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_mis, 0, 8));
    HIP_TRY(hipMemset(d_bad, 0, 4 * 68));
    const dim3 grid(std::min<uint32_t>(n_tiles, 4096u)), block(64);
#define GS_CHECK_MZ(KT, MT) hipLaunchKernelGGL((k_mz_rule_check<KT, MT>), grid, block, 0, 0, k, m, d_words, n_tiles, d_mis, d_bad)
    if (k == 20 && m == 16)
        GS_CHECK_MZ(20, 16);
    else if (k == 20 && m == 17)
        GS_CHECK_MZ(20, 17);
    else if (k == 20 && m == 18)
        GS_CHECK_MZ(20, 18);
    else if (k == 21 && m == 17)
        GS_CHECK_MZ(21, 17);
    else if (k == 25 && m == 21)
        GS_CHECK_MZ(25, 21);
    else
        GS_CHECK_MZ(0, 0);
#undef GS_CHECK_MZ
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long mis = 0;
    HIP_TRY(hipMemcpy(&mis, d_mis, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(first_bad, d_bad, 16, hipMemcpyDeviceToHost));
    if (lane_hist) HIP_TRY(hipMemcpy(lane_hist, d_bad + 4, 4 * 64, hipMemcpyDeviceToHost));
    *mismatches = mis;
    HIP_TRY(hipFree(d_words));
    HIP_TRY(hipFree(d_mis));
    HIP_TRY(hipFree(d_bad));
    return GS_OK;
}

int gs_table_create(int device, int k, uint64_t n, const uint64_t *keys, const uint32_t *loc, const uint32_t *payload,
                    gs_table **out) {
    if (!out || k < 1 || k > 32 || (n && (!keys || !loc || !payload)) || n > 0xFFFFFFF0ull) {
        set_error("gs_table_create: bad argument (k must be 1..32, n below 2^32)");
        return GS_E_ARG;
    }
    *out = nullptr;
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_table_create: HIP device %d not available (%d devices) -- there is no CPU fallback", device,
                  gs_device_count());
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    auto *t = new gs_table();
    std::unique_ptr<gs_table, void (*)(gs_table *)> guard(t, gs_table_destroy);  // frees everything on an early return
    int rc;
    if ((rc = table_init(t, device, k, n))) return rc;
    DevBufs in;
    uint64_t *dk = nullptr;
    uint32_t *dl = nullptr, *dp = nullptr;
    if ((rc = in.get(&dk, n)) || (rc = in.get(&dl, n)) || (rc = in.get(&dp, n))) return rc;
    if (n) {
        HIP_TRY(hipMemcpyAsync(dk, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(dl, loc, n * sizeof(uint32_t), hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(dp, payload, n * sizeof(uint32_t), hipMemcpyHostToDevice, t->stream));
    }
    if ((rc = table_build(t, dk, dl, dp, n))) return rc;
    *out = guard.release();
    return GS_OK;
}

int gs_table_create_device(int device, int k, uint64_t n, const void *d_keys, const void *d_loc, const void *d_payload,
                           gs_table **out) {
    if (!out || k < 1 || k > 32 || (n && (!d_keys || !d_loc)) || n > 0xFFFFFFF0ull) {
        set_error("gs_table_create_device: bad argument (k must be 1..32, n below 2^32)");
        return GS_E_ARG;
    }
    *out = nullptr;
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_table_create_device: HIP device %d not available -- there is no CPU fallback", device);
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());  // the arrays may have been written on another stream
    auto *t = new gs_table();
    std::unique_ptr<gs_table, void (*)(gs_table *)> guard(t, gs_table_destroy);
    int rc;
    if ((rc = table_init(t, device, k, n))) return rc;
    DevBufs in;
    const uint32_t *dp = static_cast<const uint32_t *>(d_payload);
    if (!dp) {  // payload = index of the key
        uint32_t *iota = nullptr;
        if ((rc = in.get(&iota, n))) return rc;
        if (n) hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, t->stream, iota, n);
        dp = iota;
    }
    if ((rc = table_build(t, static_cast<const uint64_t *>(d_keys), static_cast<const uint32_t *>(d_loc), dp, n))) return rc;
    *out = guard.release();
    return GS_OK;
}

const char *gs_table_layout(const gs_table *t) {
    static thread_local char buf[448];
    if (!t) return "";
    if (t->mz && t->mz_place)
        snprintf(buf, sizeof buf, "minimizer-hbm: %u regions of 64 B (7 place records: flank bases of the minimiser's context + offset mask | %d-bit fp + loc), mod-minimiser of length %d (w = %d, t = %d), %llu keys under both strands' picks, %llu overflow keys in a generic 16 B-slot side table%s",
                 t->mz_n_regions, t->mz_s, t->mz_m, t->k - t->mz_m + 1, mz_t_of(t->k, t->mz_m), (unsigned long long)t->mz_dup, (unsigned long long)t->mz_overflow,
                 t->mz_hot ? " (crowded regions kept as filters of their keys)" : "");
    else if (t->mz)
        snprintf(buf, sizeof buf, "minimizer-hbm: %u regions of 64 B (8 fingerprint bytes + 7 x u64 [%d-bit %s | %d-bit loc]), mod-minimiser of length %d (w = %d, t = %d), %llu overflow keys in a generic 16 B-slot side table",
                 t->mz_n_regions, mz_raw(t->k, t->mz_locbits) ? 2 * t->k : 2 * t->k - 3, mz_raw(t->k, t->mz_locbits) ? "key" : "tag", t->mz_locbits,
                 t->mz_m, t->k - t->mz_m + 1, mz_t_of(t->k, t->mz_m), (unsigned long long)t->mz_overflow);
    else if (t->compact)
        snprintf(buf, sizeof buf, "%s: 16 partitions x 2^%d buckets x 4 x %s (%d-bit remainder, %d-bit loc%s), %llu overflow keys in a generic 16 B-slot side table",
                 "compact-hbm", t->c_qb, t->c_w64 ? "u64" : "u32", t->c_rembits, t->c_locbits,
                 t->c_has_flag ? ", second-chance buckets" : "", (unsigned long long)t->c_overflow);
    else
        snprintf(buf, sizeof buf, "generic: open-addressed 16 B slots in HBM");
    return buf;
}

void gs_table_destroy(gs_table *t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    (void)hipFree(t->slots);
    (void)hipFree(t->cbuckets);
    (void)hipFree(t->cpayload);
    (void)hipFree(t->mz_regions);
    (void)hipFree(t->mz_pay);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    if (t->copy_stream) (void)hipStreamDestroy(t->copy_stream);
    delete t;
}

uint64_t gs_table_bytes(const gs_table *t) {
    if (!t) return 0;
    if (t->mz && t->mz_place) return t->cap * sizeof(Slot) + (uint64_t)t->mz_n_regions * 96;  // regions + 8 base rows each
    if (t->mz) return t->cap * sizeof(Slot) + (uint64_t)t->mz_n_regions * kMzWords * 12;  // regions + payload words
    return t->cap * sizeof(Slot) + (t->compact ? ((uint64_t)((t->c_w64 ? 48 : 32) * kNPart) << t->c_qb) : 0);  // slots + payloads
}

// ---- a read batch on the device: host-side index (per read: first word, window count, first tile), device arrays at a given
// capacity, tile descriptors built by a kernel.  Shared by the one-shot upload entry points and the streaming engine's slots.
struct ReadIndexTotals {
    uint64_t tiles = 0, n_bases = 0, n_windows = 0;
};
// `order` (may be NULL = identity): the batch reads in the order they are tagged; n_reads = its length.  start / nwin / tile0
// receive n_reads (+ 1 for start) entries.
static int read_index_host(int k, const uint64_t *word_off, const uint32_t *len, uint32_t n_reads, const uint32_t *order,
                           uint32_t n_batch_reads, uint64_t *start, uint32_t *nwin, uint32_t *tile0, ReadIndexTotals &tot) {
    tot = ReadIndexTotals{};
    for (uint32_t i = 0; i < n_reads; i++) {
        const uint32_t src = order ? order[i] : i;
        if (order && src >= n_batch_reads) {
            set_error("gs_reads_upload_ordered: order[%u] = %u is not a read of the batch", i, src);
            return GS_E_ARG;
        }
        const uint32_t l = len[src];
        start[i] = word_off[src];
        // the kernel reads words [w, w+1] of a window's first base: check the packing promise
        const uint64_t have = word_off[src + 1] - word_off[src];
        if (have < (uint64_t)(l + 31) / 32 + 1) {
            set_error("gs_reads_upload: read %u has %llu words for %u bases; need ceil(len/32)+1", i,
                      (unsigned long long)have, l);
            return GS_E_ARG;
        }
        nwin[i] = l >= (uint32_t)k ? l - (uint32_t)k + 1 : 1u;  // Q6: a short read still yields one window
        tile0[i] = (uint32_t)tot.tiles;
        tot.tiles += (nwin[i] + kTileWindows - 1) / kTileWindows;
        if (tot.tiles > 0x20000000ull) {
            set_error("gs_reads_upload: batch too large (%llu tiles); split it", (unsigned long long)tot.tiles);
            return GS_E_ARG;
        }
        tot.n_bases += l;
        tot.n_windows += nwin[i];
    }
    start[n_reads] = 0;
    return GS_OK;
}

// tile t of the batch: its read (the last one whose first tile is <= t), its words, the windows left in the read
__global__ void k_tile_index(const uint64_t *__restrict__ read_word_off, const uint32_t *__restrict__ read_nwin,
                             const uint32_t *__restrict__ read_tile0, uint32_t n_reads, uint32_t n_tiles, uint32_t *tile_read,
                             TileDesc *tdesc) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    uint32_t lo = 0, hi = n_reads;
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (read_tile0[mid] <= t) lo = mid;
        else hi = mid;
    }
    const uint32_t q = t - read_tile0[lo];
    tile_read[t] = lo;
    tdesc[t] = TileDesc{read_word_off[lo] + (uint64_t)q * kTileWords, read_nwin[lo] - q * kTileWindows, lo};
}

// Batch handles that were destroyed with nothing in flight wait here for the next batch that fits them: a batch owns ~22 device
// arrays, a pinned mirror of its hit list, pinned status blocks and four events, and every hipFree / hipHostFree / hipMalloc
// synchronises the whole device -- in the command-line pipeline that means waiting for the copy of the NEXT block, which another
// thread has in flight.  A file is tagged block after block, the blocks alike: from the third block on nothing is allocated.
struct ReadsPool {
    std::mutex m;
    std::vector<gs_reads *> idle;
};
static ReadsPool &reads_pool() {
    static ReadsPool *p = new ReadsPool;  // (never torn down: the HIP runtime may be gone before static destructors run)
    return *p;
}
static void reads_free_now(gs_reads *r);

// every device array of a batch, for up to cap_words packed words, cap_reads reads, cap_tiles tiles
static int reads_alloc(int device, int k, uint64_t cap_words, uint32_t cap_reads, uint64_t cap_tiles, uint64_t cand_rows, gs_reads **out) {
    HIP_TRY(hipSetDevice(device));
    const uint32_t cand_cap = (uint32_t)std::min<uint64_t>(cand_rows, 0xFFFFFFFFull);
    {
        ReadsPool &pool = reads_pool();
        std::lock_guard<std::mutex> lk(pool.m);
        size_t best = pool.idle.size();
        for (size_t i = 0; i < pool.idle.size(); i++) {
            const gs_reads *c = pool.idle[i];
            if (c->device == device && c->cap_words >= cap_words && c->cap_reads >= cap_reads && c->cap_tiles >= cap_tiles && c->cand_cap >= cand_cap &&
                c->cap_words <= 2 * cap_words + (1u << 20) && (best == pool.idle.size() || c->cap_words < pool.idle[best]->cap_words))
                best = i;
        }
        if (best < pool.idle.size()) {
            gs_reads *r = pool.idle[best];
            pool.idle.erase(pool.idle.begin() + (long)best);
            r->k = k;
            *out = r;
            return GS_OK;
        }
    }
    auto *r = new gs_reads();
    std::unique_ptr<gs_reads, void (*)(gs_reads *)> guard(r, gs_reads_destroy);
    r->device = device;
    r->k = k;
    int rc;
    if ((rc = pass_state_init(r->ps))) return rc;
    // zero slack after the last read: every lane of a tile loads words [l, l+1] unconditionally
    if ((rc = dev_alloc(&r->words, cap_words + kReadSlackWords))) return rc;
    if ((rc = dev_alloc(&r->read_word_off, (uint64_t)cap_reads + 1))) return rc;
    if ((rc = dev_alloc(&r->read_nwin, cap_reads)) || (rc = dev_alloc(&r->read_tile0, cap_reads)) ||
        (rc = dev_alloc(&r->tile_read, cap_tiles)) || (rc = dev_alloc(&r->tile_desc, cap_tiles)))
        return rc;
    TileArrays &ta = r->ta;
    if ((rc = dev_alloc(&ta.n_raw, cap_tiles)) || (rc = dev_alloc(&ta.n_cand, cap_tiles)) || (rc = dev_alloc(&ta.first_loc, cap_tiles)) ||
        (rc = dev_alloc(&ta.last_loc, cap_tiles)) || (rc = dev_alloc(&ta.cand_off, cap_tiles)) || (rc = dev_alloc(&ta.n_emit, cap_tiles)) ||
        (rc = dev_alloc(&ta.carry_loc, cap_tiles)) || (rc = dev_alloc(&ta.raw_prefix, cap_tiles)) ||
        (rc = dev_alloc(&ta.emit_prefix, cap_tiles)))
        return rc;
    if ((rc = dev_alloc(&r->block_agg, (cap_tiles + kScanBlock - 1) / kScanBlock))) return rc;
    if ((rc = ensure_out_buffers(r, cand_cap))) return rc;
    r->cap_words = cap_words;
    r->cap_reads = cap_reads;
    r->cap_tiles = cap_tiles;
    r->complete = true;
    *out = guard.release();
    return GS_OK;
}

// the per-read index to the device and the tile descriptors built there, on `st` (the host arrays must outlive the copies:
// pinned staging for the streaming engine, a synchronised stream for the one-shot path)
static int reads_index_upload(gs_reads *r, const uint64_t *start, const uint32_t *nwin, const uint32_t *tile0, uint32_t n_reads,
                              const ReadIndexTotals &tot, hipStream_t st) {
    r->n_reads = n_reads;
    r->n_tiles = (uint32_t)tot.tiles;
    r->n_bases = tot.n_bases;
    r->n_windows = tot.n_windows;
    r->n_scan_blocks = (uint32_t)((tot.tiles + kScanBlock - 1) / kScanBlock);
    if (n_reads) {
        HIP_TRY(hipMemcpyAsync(r->read_word_off, start, ((uint64_t)n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(r->read_nwin, nwin, n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(r->read_tile0, tile0, n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    }
    if (tot.tiles)
        hipLaunchKernelGGL(k_tile_index, dim3((unsigned)((tot.tiles + 255) / 256)), dim3(256), 0, st, (const uint64_t *)r->read_word_off,
                           (const uint32_t *)r->read_nwin, (const uint32_t *)r->read_tile0, n_reads, r->n_tiles, r->tile_read, r->tile_desc);
    HIP_TRY(hipGetLastError());
    return GS_OK;
}

// index building + allocations shared by the upload entry points; the packed words are zeroed, not filled
static int reads_setup(int device, uint64_t n_words, const uint64_t *word_off, const uint32_t *len, uint32_t n_reads, int k,
                       gs_reads **out, const uint32_t *order = nullptr, uint32_t n_batch_reads = 0) {
    if (!out || k < 1 || k > 32 || (n_reads && (!word_off || !len))) {
        set_error("gs_reads_upload: bad argument");
        return GS_E_ARG;
    }
    *out = nullptr;
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_reads_upload: HIP device %d not available -- there is no CPU fallback", device);
        return GS_E_NODEVICE;
    }
    std::vector<uint32_t> nwin(n_reads), tile0(n_reads);
    std::vector<uint64_t> start((size_t)n_reads + 1, 0);
    ReadIndexTotals tot;
    int rc;
    if ((rc = read_index_host(k, word_off, len, n_reads, order, n_batch_reads, start.data(), nwin.data(), tile0.data(), tot))) return rc;
    gs_reads *r = nullptr;
    if ((rc = reads_alloc(device, k, n_words, n_reads, tot.tiles, (kDirectCand + 2) * tot.tiles + 65536, &r))) return rc;
    std::unique_ptr<gs_reads, void (*)(gs_reads *)> guard(r, gs_reads_destroy);
    r->n_words = n_words;
    r->read_tile0_host = tile0;
    HIP_TRY(hipMemsetAsync(r->words, 0, (n_words + kReadSlackWords) * sizeof(uint64_t), nullptr));
    if ((rc = reads_index_upload(r, start.data(), nwin.data(), tile0.data(), n_reads, tot, nullptr))) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    *out = guard.release();
    return GS_OK;
}

int gs_reads_upload(int device, const uint64_t *words, uint64_t n_words, const uint64_t *word_off, const uint32_t *len,
                    uint32_t n_reads, int k, gs_reads **out) {
    if (n_reads && !words) {
        set_error("gs_reads_upload: bad argument");
        return GS_E_ARG;
    }
    int rc = reads_setup(device, n_words, word_off, len, n_reads, k, out);
    if (rc) return rc;
    if (n_words) {
        hipError_t e = hipMemcpy((*out)->words, words, n_words * sizeof(uint64_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("hipMemcpy of the packed reads failed: %s", hipGetErrorString(e));
            gs_reads_destroy(*out);
            *out = nullptr;
            return GS_E_NODEVICE;
        }
    }
    return GS_OK;
}

int gs_reads_upload_ordered(int device, const uint64_t *words, uint64_t n_words, const uint64_t *word_off, const uint32_t *len,
                            uint32_t n_reads, const uint32_t *order, uint32_t n_order, int k, gs_reads **out) {
    if ((n_reads && !words) || (n_order && !order)) {
        set_error("gs_reads_upload_ordered: bad argument");
        return GS_E_ARG;
    }
    int rc = reads_setup(device, n_words, word_off, len, n_order, k, out, order ? order : &n_order /* unused when 0 */, n_reads);
    if (rc) return rc;
    if (n_words) {
        hipError_t e = hipMemcpy((*out)->words, words, n_words * sizeof(uint64_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("hipMemcpy of the packed reads failed: %s", hipGetErrorString(e));
            gs_reads_destroy(*out);
            *out = nullptr;
            return GS_E_NODEVICE;
        }
    }
    return GS_OK;
}

int gs_reads_set_seeds(gs_reads *r, const uint32_t *seed_at, const uint32_t *seed, uint32_t n) {
    if (!r || (n && (!seed_at || !seed))) {
        set_error("gs_reads_set_seeds: bad argument");
        return GS_E_ARG;
    }
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint32_t> h(std::max<uint32_t>(r->n_tiles, 1), kNoSeed);
    r->seed_tiles.clear();
    for (uint32_t j = 0; j < n; j++) {
        if (seed_at[j] >= r->n_reads) {
            set_error("gs_reads_set_seeds: read %u of %u", seed_at[j], r->n_reads);
            return GS_E_ARG;
        }
        if (seed[j] == kNoSeed) {
            set_error("gs_reads_set_seeds: 0x%X is not a location id", kNoSeed);
            return GS_E_ARG;
        }
        const uint32_t t = r->read_tile0_host[seed_at[j]];
        h[t] = seed[j];
        r->seed_tiles.push_back(t);
    }
    int rc;
    if (!r->d_seed && ((rc = dev_alloc(&r->d_seed, r->n_tiles)) || (rc = dev_alloc(&r->d_seg_carry, r->n_tiles)))) return rc;
    if (r->n_tiles) HIP_TRY(hipMemcpy(r->d_seed, h.data(), (size_t)r->n_tiles * sizeof(uint32_t), hipMemcpyHostToDevice));
    r->ta.seed = n ? r->d_seed : nullptr;
    r->ta.seg_carry = r->d_seg_carry;
    return GS_OK;
}

// 2-bit packing on the device: one thread per output word, 32 source bytes each,
// nim-kmer's `encode` table (A/a 0, C/c 1, G/g 2, T/t/U/u 3, everything else 0)
__global__ void k_pack_reads(const uint8_t *seq, const uint64_t *byte_off, const uint64_t *word_off, uint32_t n_reads,
                             uint64_t n_words, uint64_t *words) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t lo = 0, hi = n_reads;  // read of this word: last r with word_off[r] <= w
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (word_off[mid] <= w) lo = mid;
        else hi = mid;
    }
    const uint64_t len = byte_off[lo + 1] - byte_off[lo];
    const uint64_t b0 = (w - word_off[lo]) * 32;
    uint64_t v = 0;
    if (b0 < len) {
        const uint8_t *s = seq + byte_off[lo] + b0;
        const int n = (int)min((uint64_t)32, len - b0);
        for (int i = 0; i < n; i++) {
            const uint32_t c = s[i] & 0xDFu;  // upper case
            const uint64_t code = c == 'C' ? 1u : c == 'G' ? 2u : (c == 'T' || c == 'U') ? 3u : 0u;
            v |= code << (62 - 2 * i);
        }
    }
    words[w] = v;
}

// shared by the two ASCII entry points: index + allocations, then the packing kernel over a device copy of the bases
// d_boff_scratch != NULL (the on-device record parser, whose kernels all ran on the null stream): n_reads + 1 words of device
// memory to put the offsets in, and the call waits for the NULL STREAM only -- a device-wide synchronisation (and every hipMalloc
// / hipFree is one) would wait for the copy of the NEXT text block, which another thread of the command line has in flight
static int reads_from_ascii(int device, const uint8_t *seq, bool seq_on_device, const uint64_t *off, uint32_t n_reads, int k,
                            gs_reads **out, const char *who, uint64_t *d_boff_scratch = nullptr) {
    if (!out || (n_reads && (!off || (off[n_reads] && !seq)))) {
        set_error("%s: bad argument", who);
        return GS_E_ARG;
    }
    std::vector<uint64_t> word_off((size_t)n_reads + 1, 0);
    std::vector<uint32_t> len(n_reads);
    for (uint32_t i = 0; i < n_reads; i++) {
        const uint64_t l = off[i + 1] - off[i];
        if (l > 0xFFFFFFFFull) {
            set_error("%s: read %u is longer than 2^32 bases", who, i);
            return GS_E_ARG;
        }
        len[i] = (uint32_t)l;
        word_off[i + 1] = word_off[i] + (l + 31) / 32 + 1;
    }
    const uint64_t n_words = word_off[n_reads];
    int rc = reads_setup(device, n_words, word_off.data(), len.data(), n_reads, k, out);
    if (rc) return rc;
    if (!n_reads) return GS_OK;
    std::unique_ptr<gs_reads, void (*)(gs_reads *)> guard(*out, gs_reads_destroy);
    *out = nullptr;
    uint8_t *d_seq = nullptr;
    uint64_t *d_boff = d_boff_scratch;
    const uint64_t nbytes = off[n_reads];
    if (!seq_on_device && (rc = dev_alloc(&d_seq, nbytes))) return rc;
    if (!d_boff_scratch && (rc = dev_alloc(&d_boff, (uint64_t)n_reads + 1))) {
        (void)hipFree(d_seq);
        return rc;
    }
    hipError_t e = hipSuccess;
    if (!seq_on_device && nbytes) e = hipMemcpy(d_seq, seq, nbytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_boff, off, ((uint64_t)n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, 0, seq_on_device ? seq : d_seq, d_boff,
                           guard->read_word_off, n_reads, n_words, guard->words);
        e = d_boff_scratch ? hipStreamSynchronize(nullptr) : hipDeviceSynchronize();
    }
    if (d_seq) (void)hipFree(d_seq);
    if (!d_boff_scratch) (void)hipFree(d_boff);
    if (e != hipSuccess) {
        set_error("%s: %s", who, hipGetErrorString(e));
        return GS_E_NODEVICE;
    }
    *out = guard.release();
    return GS_OK;
}

int gs_reads_upload_ascii(int device, const uint8_t *seq, const uint64_t *off, uint32_t n_reads, int k, gs_reads **out) {
    return reads_from_ascii(device, seq, false, off, n_reads, k, out, "gs_reads_upload_ascii");
}

int gs_reads_from_device_ascii(int device, const void *d_seq, const uint64_t *off, uint32_t n_reads, int k, gs_reads **out) {
    if (gs_device_count() > device && device >= 0) {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipDeviceSynchronize());  // the bases may have been written on another stream
    }
    return reads_from_ascii(device, static_cast<const uint8_t *>(d_seq), true, off, n_reads, k, out, "gs_reads_from_device_ascii");
}

}  // extern "C"
// (gs_ingest.hip) the bases were written by kernels on the null stream; d_boff_scratch: n_reads + 1 words of device memory
int gs::reads_from_parsed_text(int device, const void *d_seq, const uint64_t *off, uint32_t n_reads, int k, uint64_t *d_boff_scratch,
                               gs_reads **out) {
    HIP_TRY(hipSetDevice(device));
    return reads_from_ascii(device, static_cast<const uint8_t *>(d_seq), true, off, n_reads, k, out, "gs_reads_from_text", d_boff_scratch);
}
extern "C" {

void gs_reads_destroy(gs_reads *r) {
    if (!r) return;
    // a complete handle with nothing in flight (its last pass settled, or it never had one) goes to the pool
    if (r->complete && !r->ps.pass_pending && !r->ps.was_chained_to && !r->d_seed && !getenv("GAVISUNK_NO_READS_POOL")) {
        r->n_reads = r->n_tiles = 0;
        r->n_bases = r->n_windows = r->n_words = 0;
        r->n_scan_blocks = 0;
        r->seed_tiles.clear();
        r->read_tile0_host.clear();
        r->ps.table = nullptr;
        r->ps.after = nullptr;
        r->ps.last_carry_in = GS_NONE;
        r->ps.settled = false;
        r->ps.res_n = 0;
        r->ps.res_carry = GS_NONE;
        r->ps.res_keyerr = false;
        ReadsPool &pool = reads_pool();
        gs_reads *evict = nullptr;
        {
            std::lock_guard<std::mutex> lk(pool.m);
            pool.idle.push_back(r);
            if (pool.idle.size() > 3) {  // (the oldest goes)
                evict = pool.idle.front();
                pool.idle.erase(pool.idle.begin());
            }
        }
        if (evict) reads_free_now(evict);
        return;
    }
    reads_free_now(r);
}

static void reads_free_now(gs_reads *r) {
    (void)hipSetDevice(r->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(r->words);
    (void)hipFree(r->read_word_off);
    (void)hipFree(r->read_nwin);
    (void)hipFree(r->read_tile0);
    (void)hipFree(r->tile_read);
    TileArrays &ta = r->ta;
    (void)hipFree(ta.n_raw);
    (void)hipFree(ta.n_cand);
    (void)hipFree(ta.first_loc);
    (void)hipFree(ta.last_loc);
    (void)hipFree(ta.cand_off);
    (void)hipFree(ta.n_emit);
    (void)hipFree(ta.carry_loc);
    (void)hipFree(ta.raw_prefix);
    (void)hipFree(ta.emit_prefix);
    (void)hipFree(r->block_agg);
    (void)hipFree(r->cand);
    (void)hipFree(r->hits);
    if (r->hits_host) (void)hipHostFree(r->hits_host);
    (void)hipFree(r->tile_desc);
    (void)hipFree(r->d_seed);
    (void)hipFree(r->d_seg_carry);
    pass_state_free(r->ps);
    delete r;
}

uint64_t gs_reads_n_bases(const gs_reads *r) { return r ? r->n_bases : 0; }
uint64_t gs_reads_n_windows(const gs_reads *r) { return r ? r->n_windows : 0; }

static int check_pair(const gs_table *t, const gs_reads *r, const char *who) {
    if (!t || !r) {
        set_error("%s: null handle", who);
        return GS_E_ARG;
    }
    if (t->device != r->device || t->k != r->k) {
        set_error("%s: table (device %d, k %d) and reads (device %d, k %d) do not match", who, t->device, t->k,
                  r->device, r->k);
        return GS_E_ARG;
    }
    return GS_OK;
}

int gs_tag_launch(gs_table *t, gs_reads *r, uint32_t carry_in) {
    int rc;
    if ((rc = check_pair(t, r, "gs_tag_launch"))) return rc;
    if ((rc = launch_phase1(t, r))) return rc;
    return launch_phase2(t, r, carry_in);
}

int gs_tag_launch_after(gs_table *t, gs_reads *r, const gs_reads *prev) {
    int rc;
    if ((rc = check_pair(t, r, "gs_tag_launch_after"))) return rc;
    if (!prev || prev == r || prev->ps.table != t || !(prev->ps.pass_pending || prev->ps.settled)) {
        set_error("gs_tag_launch_after: `prev` must be another batch whose pass was launched on this table");
        return GS_E_ARG;
    }
    if ((rc = launch_phase1(t, r))) return rc;
    return launch_phase2(t, r, GS_NONE, prev);
}

static hipError_t wait_stream(hipStream_t st);
static hipError_t wait_event(hipEvent_t ev);

int gs_tag_phase1_launch(gs_table *t, gs_reads *r) {
    int rc;
    if ((rc = check_pair(t, r, "gs_tag_phase1"))) return rc;
    if ((rc = launch_phase1(t, r))) return rc;
    // summary = scan A with no carry-in; cheap, and phase 2 redoes it with the true carry
    r->ps.last_carry_in = GS_NONE;
    HIP_TRY(hipMemcpyAsync(r->ps.d_carry_in, &r->ps.last_carry_in, sizeof(uint32_t), hipMemcpyHostToDevice, t->stream));
    if ((rc = launch_scan(t, r, 0))) return rc;
    HIP_TRY(hipMemcpyAsync(r->ps.h_status, r->ps.d_status, sizeof(PassStatus), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipEventRecord(r->ps.ev[3], t->stream));
    return GS_OK;
}

int gs_tag_phase1_wait(gs_table *t, gs_reads *r, uint32_t *has_raw, uint32_t *last_loc) {
    int rc;
    if ((rc = check_pair(t, r, "gs_tag_phase1"))) return rc;
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(wait_event(r->ps.ev[3]));
    const Agg tot = r->ps.h_status->totals[0];
    if (has_raw) *has_raw = tot.has & 1u;
    if (last_loc) *last_loc = (tot.has & 1u) ? tot.loc : GS_NONE;
    return GS_OK;
}

int gs_tag_phase1(gs_table *t, gs_reads *r, uint32_t *has_raw, uint32_t *last_loc) {
    int rc;
    if ((rc = gs_tag_phase1_launch(t, r))) return rc;
    return gs_tag_phase1_wait(t, r, has_raw, last_loc);
}

int gs_tag_phase2(gs_table *t, gs_reads *r, uint32_t carry_in) {
    int rc;
    if ((rc = check_pair(t, r, "gs_tag_phase2"))) return rc;
    return launch_phase2(t, r, carry_in);
}

// A pass over a batch is a millisecond or two: poll first (the wake-up of a blocking wait can cost a third of that),
// block only when the pass turns out to be long.
static hipError_t wait_stream(hipStream_t st) {
    hipError_t q = hipErrorNotReady;
    const auto t_poll = std::chrono::steady_clock::now();
    while ((q = hipStreamQuery(st)) == hipErrorNotReady)
        if (std::chrono::steady_clock::now() - t_poll > std::chrono::milliseconds(20)) break;
    return q == hipErrorNotReady ? hipStreamSynchronize(st) : q;
}
static hipError_t wait_event(hipEvent_t ev) {
    hipError_t q = hipErrorNotReady;
    const auto t_poll = std::chrono::steady_clock::now();
    while ((q = hipEventQuery(ev)) == hipErrorNotReady)
        if (std::chrono::steady_clock::now() - t_poll > std::chrono::milliseconds(20)) break;
    return q == hipErrorNotReady ? hipEventSynchronize(ev) : q;
}

// Waits for the pass of THIS batch (later passes on the same table keep running) and caches its outcome in the batch handle
// (idempotent until the next launch).  A candidate-buffer overflow (more location changes than budgeted) grows the buffer and
// re-runs.
static int settle(gs_table *t, gs_reads *r) {
    PassState &ps = r->ps;
    if (!ps.pass_pending) {
        if (ps.settled && ps.table == t) return GS_OK;
        set_error("no tagging pass was launched for this (table, reads) pair");
        return GS_E_ARG;
    }
    if (ps.table != t) {
        set_error("the pending pass of these reads was launched on another table");
        return GS_E_ARG;
    }
    HIP_TRY(hipSetDevice(t->device));
    for (int attempt = 0; attempt < 8; attempt++) {
        HIP_TRY(wait_event(ps.ev[3]));
        const uint32_t ncand = ps.h_status->cand_count;
        const unsigned long long err = ps.h_status->err_pos;
        const Agg tot[2] = {ps.h_status->totals[0], ps.h_status->totals[1]};
        if ((uint64_t)r->n_tiles * kDirectCand + ncand > r->cand_cap) {
            int rc;
            const uint32_t carry_in = ps.h_status->carry_in;  // (what the pass was handed, by the host or by its predecessor)
            HIP_TRY(hipStreamSynchronize(t->stream));  // nothing may still read the buffers that are about to be replaced
            if ((rc = ensure_out_buffers(r, (uint32_t)std::min<uint64_t>((uint64_t)r->n_tiles * kDirectCand + ncand + 65536, 0xFFFFFFFFull)))) return rc;
            if ((rc = launch_phase1(t, r))) return rc;
            if ((rc = launch_phase2(t, r, carry_in))) return rc;
            continue;
        }
        float ms01 = 0, ms02 = 0;
        HIP_TRY(hipEventElapsedTime(&ms01, ps.ev[0], ps.ev[1]));
        HIP_TRY(hipEventElapsedTime(&ms02, ps.ev[0], ps.ev[2]));
        t->ctr.probe_ms += ms01;
        t->ctr.total_ms += ms02;
        t->ctr.launches += 1;
        t->ctr.windows = r->n_windows;
        t->ctr.tiles = r->n_tiles;
        t->ctr.fallback_tiles = ps.h_status->complex_n;
        t->ctr.raw_hits = tot[0].sum;
        t->ctr.emitted = tot[1].sum;
        ps.res_n = tot[1].sum;
        ps.res_carry = (tot[0].has & 1u) ? tot[0].loc : GS_NONE;
        ps.res_keyerr = err != ~0ULL;
        if (ps.res_keyerr) {
            // hits emitted before the offending window = emit_prefix of its tile + its earlier candidates
            const uint32_t tile = (uint32_t)(err >> 32), woff = (uint32_t)err;
            uint64_t ep = 0;
            uint32_t nc = 0, ne = 0, off = 0;
            HIP_TRY(hipMemcpy(&ep, r->ta.emit_prefix + tile, sizeof ep, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(&nc, r->ta.n_cand + tile, sizeof nc, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(&ne, r->ta.n_emit + tile, sizeof ne, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(&off, r->ta.cand_off + tile, sizeof off, hipMemcpyDeviceToHost));
            std::vector<uint32_t> cv(nc);
            if (nc) HIP_TRY(hipMemcpy(cv.data(), r->cand + off, nc * sizeof(uint32_t), hipMemcpyDeviceToHost));
            uint64_t before = 0;
            for (uint32_t j = nc - ne; j < nc; j++)
                if ((cv[j] & 0xFFFFu) < woff) before++;
            ps.res_n = ep + before;
        }
        // long list (k_emit mirrors only lists up to kMirrorRows to the host itself -- it decides by the same total): one bulk
        // copy into the pinned mirror, on the copy stream: the next pass on this table keeps the kernels' stream busy meanwhile
        if (tot[1].sum > kMirrorRows && ps.res_n) {
            HIP_TRY(hipMemcpyAsync(r->hits_host, r->hits, std::min<uint64_t>(ps.res_n, r->hit_cap) * sizeof(gs_hit), hipMemcpyDeviceToHost, t->copy_stream));
            HIP_TRY(hipStreamSynchronize(t->copy_stream));
        }
        ps.pass_pending = false;
        ps.settled = true;
        return GS_OK;
    }
    set_error("gs_tag: candidate buffer kept overflowing");
    return GS_E_NOMEM;
}

static int copy_hits(gs_table *t, gs_reads *r, void *out, uint64_t cap, uint64_t *n_out, hipMemcpyKind kind) {
    int rc;
    if ((rc = settle(t, r))) return rc;
    if (n_out) *n_out = r->ps.res_n;
    if (out) {
        if (r->ps.res_n > cap) {
            set_error("%llu hits do not fit in cap = %llu", (unsigned long long)r->ps.res_n, (unsigned long long)cap);
            return GS_E_CAPACITY;
        }
        if (r->ps.res_n) {
            if (kind == hipMemcpyDeviceToHost) memcpy(out, r->hits_host, r->ps.res_n * sizeof(gs_hit));  // k_emit wrote them there too
            else HIP_TRY(hipMemcpy(out, r->hits, r->ps.res_n * sizeof(gs_hit), kind));
        }
    }
    if (r->ps.res_keyerr) {
        set_error("key not found: a read window is in the SUNK db but has no row in the loc table [KeyError]");
        return GS_E_KEYERROR;
    }
    return GS_OK;
}

int gs_tag_finish(gs_table *t, gs_reads *r, gs_hit *out, uint64_t cap, uint64_t *n_out, uint32_t *carry_out,
                  gs_counters *ctr) {
    int rc;
    if ((rc = check_pair(t, r, "gs_tag_finish"))) return rc;
    if ((rc = settle(t, r))) return rc;
    if (carry_out) *carry_out = r->ps.res_carry;
    if (ctr) *ctr = t->ctr;
    return copy_hits(t, r, out, cap, n_out, hipMemcpyDeviceToHost);
}

int gs_tag(gs_table *t, gs_reads *r, uint32_t carry_in, gs_hit *out, uint64_t cap, uint64_t *n_out, uint32_t *carry_out,
           gs_counters *ctr) {
    int rc = gs_tag_launch(t, r, carry_in);
    if (rc) return rc;
    return gs_tag_finish(t, r, out, cap, n_out, carry_out, ctr);
}

int gs_tag_hits_to_host(gs_table *t, gs_reads *r, gs_hit *out, uint64_t cap, uint64_t *n_out) {
    int rc;
    if ((rc = check_pair(t, r, "gs_tag_hits_to_host"))) return rc;
    return copy_hits(t, r, out, cap, n_out, hipMemcpyDeviceToHost);
}

int gs_tag_segment_carries(gs_table *t, gs_reads *r, uint32_t *out) {
    int rc;
    if ((rc = check_pair(t, r, "gs_tag_segment_carries"))) return rc;
    if ((rc = settle(t, r))) return rc;
    for (size_t j = 0; j < r->seed_tiles.size(); j++)
        HIP_TRY(hipMemcpy(out + j, r->d_seg_carry + r->seed_tiles[j], sizeof(uint32_t), hipMemcpyDeviceToHost));
    return GS_OK;
}

int gs_tag_hits_view(gs_table *t, gs_reads *r, const gs_hit **out, uint64_t *n_out) {
    int rc;
    if ((rc = check_pair(t, r, "gs_tag_hits_view"))) return rc;
    if ((rc = settle(t, r))) return rc;
    if (out) *out = r->hits_host;
    if (n_out) *n_out = r->ps.res_n;
    if (r->ps.res_keyerr) {
        set_error("key not found: a read window is in the SUNK db but has no row in the loc table [KeyError]");
        return GS_E_KEYERROR;
    }
    return GS_OK;
}

int gs_tag_hits_to_device(gs_table *t, gs_reads *r, void *dst, uint64_t cap, uint64_t *n_out) {
    int rc;
    if ((rc = check_pair(t, r, "gs_tag_hits_to_device"))) return rc;
    return copy_hits(t, r, dst, cap, n_out, hipMemcpyDeviceToDevice);
}

int gs_tag_hits_device(gs_table *t, gs_reads *r, const void **d_hits, uint64_t *n_out) {
    int rc;
    if ((rc = check_pair(t, r, "gs_tag_hits_device"))) return rc;
    if (!d_hits || !n_out) {
        set_error("gs_tag_hits_device: null argument");
        return GS_E_ARG;
    }
    if ((rc = settle(t, r))) return rc;
    *d_hits = r->hits;
    *n_out = r->ps.res_n;
    if (r->ps.res_keyerr) {
        set_error("key not found: a read window is in the SUNK db but has no row in the loc table [KeyError]");
        return GS_E_KEYERROR;
    }
    return GS_OK;
}

// (tests) a launch that fails -- a block of 4096 threads -- and leaves its error in this thread's last-error slot, as a failed launch
// of ours would: the next entry point that checks the slot must report it (gs_comm_*: checked BEFORE a collective, csrc/gs_comm.h)
int gs_selftest_inject_launch_error(int device) {
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_selftest_inject_launch_error: HIP device %d not available", device);
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    hipLaunchKernelGGL(k_selftest_nop, dim3(1), dim3(4096), 0, 0);
    return GS_OK;
}

void gs_counters_reset(gs_table *t) {
    if (t) t->ctr = gs_counters{};
}

void gs_table_counters(const gs_table *t, gs_counters *out) {
    if (t && out) *out = t->ctr;
}

int gs_reads_copy_words(const gs_reads *r, uint64_t word_lo, uint64_t n_words, uint64_t *dst) {
    if (!r || (n_words && !dst) || word_lo + n_words > r->n_words) {
        set_error("gs_reads_copy_words: bad argument (the batch has %llu words)", (unsigned long long)(r ? r->n_words : 0));
        return GS_E_ARG;
    }
    HIP_TRY(hipSetDevice(r->device));
    if (n_words) HIP_TRY(hipMemcpy(dst, r->words + word_lo, n_words * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return GS_OK;
}

}  // extern "C"

#include "gs_stream.h"
#include "gs_comm.h"
#include "gs_sidecar.h"

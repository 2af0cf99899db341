// Device half of libgavisunk.so: the SUNK tagger of GAVISUNK (workflow/src/kmerpos_annot3.nim)
// as hand-written HIP for gfx950 (MI355X, wave64).  Pure integer / indexing work; see DESIGN.md section 4.
//
// Data layout in HBM
//   reads   2 bits/base, 32 bases per uint64 (first base most significant); every read starts
//           on a word boundary and is followed by >= 1 zero word.
//   tiles   a read is cut into tiles of 2048 windows = 64 words; a wavefront lane owns one word
//           = windows [32 l, 32 l + 32) of a tile.
//   tables  (1) compact: exact, 16 partitions x 2^qb buckets x 4 x 32-bit slots [flags|alt|remainder|loc],
//               built on the GPU, + one payload word per slot;
//           (2) generic: open-addressed, linear probing, 16-byte slots {key u64, loc u32, payload u32},
//               load <= 0.5 -- everything when (1) does not apply, else only the keys (1) could not take.
//
// One `gs_tag` pass:
//   phase 1, one of
//     k_probe<false>   generic table, one wave per tile, 8 probes in flight per lane
//     k_probe<true>    compact table out in HBM: one 16-byte bucket load (one 64-byte line) per window
//     k_probe_part + k_merge_part (+ k_probe<true> on the few tiles that cannot be merged)
//                      compact table resident in the XCD-partitioned L2s
//     k_join_bin + k_join_segT + k_join_probe + k_join_merge (k = 20, the default for small tables)
//                      windows routed to the CU that holds their 1/256 slice of the compact table in LDS
//   -> per tile {n_raw, first loc, last loc} and the ordered candidates (hits whose loc differs from the
//      previous raw hit of the tile; the tile's first raw hit is always a candidate)
//   scan A     over tiles: raw-hit prefix and the loc carried into each tile (`prevLoc` of
//              kmerpos_annot3.nim:82-84 crosses reads, so the scan is global); drops a tile's first
//              candidate when it equals the carried loc
//   scan B     emitted-hit prefix
//   k_emit     pos = window - (raw hits before it in the read - emitted hits before it in the read)
//              [`continue` at kmerpos_annot3.nim:92 skips `inc i` at :96], payload by re-probing the
//              (rare) emitted windows; writes gs_hit in stream order
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
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

#ifndef GS_PART_WAVES
#define GS_PART_WAVES 6  // waves per SIMD k_probe_part is compiled for (register budget); 6 x pipe 2 measured best
#endif
constexpr int kMaxRuns = 3;
constexpr int kNPart = 16;  // key-space partitions; 8 are live at a time (one per XCD), kNPart / 8 passes
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


// ------------------------------------------------------------------------------------------
// Small-table fast path: an exact compact table that lives in the L2 caches.
//
// MI355X has 8 XCDs with a private 4 MiB L2 each, and workgroup b is dispatched to XCD b % 8
// (observed, used for speed only: any other placement gives the same results, slower).  Random
// probes that hit L2 run at ~270 G/s chip-wide; past L2 they run at ~55 G/s whatever the table
// size (profiles/r01/ubench_*.txt).  So the key space is cut into 16 partitions, 8 of them live
// at a time (two passes over the tiles), partition p is only probed by workgroups with
// blockIdx % 8 == p % 8, and each XCD's L2 holds one 1/16 slice of the table at a time.
//
// The partition function is strand-symmetric, costs a few bit-parallel operations for all 32
// windows of a lane at once, and is "free" in key bits.  With x_i the i-th base of a window,
//
//   part = (x_0 ^ x_{k-1}) << 2 | (x_1 ^ x_{k-2})
//
// is the same for a k-mer and its reverse complement, and given `part` the last two bases follow
// from the first two, so the table only has to store the canonical code without them:
//
//   key~      = canonical code >> 4                       (2k-4 bits)
//   y         = hmix(key~)                                a bijection on (2k-4)-bit values
//   bucket    = y >> rembits                              2^qb buckets of 4 x 32-bit slots per partition
//   slot      = flag << 31 | (y & remmask) << locbits | loc        0xFFFFFFFF = empty
//
// (partition, bucket, remainder) identifies the key exactly, so a match is a true hit.  A key whose
// home bucket is full at build time stays in the generic table only and the bucket gets its
// overflow flag (bit 31 of slot 0); a lookup that finds no match in a flagged bucket asks the
// generic table, every other lookup without a match is an exact miss.
//
// k_probe_part  grid = 16 x ceil(tiles/4): wave (tile, p) finds which of the tile's 2048 windows
//               belong to partition p with ~40 bit-parallel operations per lane, compacts them
//               through an LDS queue so the probe loop runs with full waves and four bucket loads
//               in flight per lane, and writes a run-length summary of its hits: up to 3 runs
//               {loc, first window, last window, count}.
// k_merge_part  per tile, merges the 16 summaries.  If runs of different loc never overlap in
//               window range the interleaving of the partitions does not matter and the tile's
//               {n_raw, first/last loc, candidates} follow exactly; otherwise the tile is queued
//               for the generic probe_tile (k_probe with a tile list).
// ------------------------------------------------------------------------------------------
constexpr uint32_t kEmptySlot = kEmptySlot32;

struct __align__(16) PartRec {
    uint32_t head;  // nruns | complex << 8 | n_hits << 16
    uint32_t run[kMaxRuns][3];  // loc, first | last << 16, count
    uint32_t pad[2];
};

struct __align__(16) TileDesc {
    uint64_t wbase;      // first word of the tile
    uint32_t nwin_left;  // windows of the read from the tile's first window on
    uint32_t read;
};

__device__ __forceinline__ uint32_t lane_value(uint32_t v, int uniform_lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, uniform_lane);
}

// wave-uniform run-length state of one (tile, partition); completed runs go to LDS
struct PartState {
    uint32_t nruns, complex, have, hits, cur_loc, cur_first, cur_last, cur_cnt;
    __device__ __forceinline__ void reset() { nruns = complex = have = hits = cur_loc = cur_first = cur_last = cur_cnt = 0; }
    __device__ __forceinline__ void push(uint32_t (*runs)[3], int lane) {
        if (nruns < (uint32_t)kMaxRuns) {
            if (lane == 0) {
                runs[nruns][0] = cur_loc;
                runs[nruns][1] = cur_first | (cur_last << 16);
                runs[nruns][2] = cur_cnt;
            }
            nruns++;
        } else {
            complex = 1;
        }
    }
    // fold one ordered group of <= 64 items (one per lane) into the run list
    __device__ __forceinline__ void step(bool hit, uint32_t loc, uint32_t woff, int lane, uint32_t (*runs)[3]) {
        const unsigned long long hb = __ballot(hit);
        if (!hb) return;
        const int first_lane = __ffsll(hb) - 1, last_lane = 63 - __clzll((long long)hb);
        const uint32_t loc0 = lane_value(loc, first_lane);
        hits += __popcll(hb);
        if (!__ballot(hit && loc != loc0)) {  // the usual case: every hit of the group has the same loc
            if (!(have && cur_loc == loc0)) {
                if (have) push(runs, lane);
                cur_loc = loc0;
                cur_first = lane_value(woff, first_lane);
                cur_cnt = 0;
                have = 1;
            }
            cur_cnt += __popcll(hb);
            cur_last = lane_value(woff, last_lane);
            return;
        }
        const unsigned long long below = hb & (lane ? (~0ULL >> (64 - lane)) : 0ULL);
        const int src = below ? 63 - __clzll((long long)below) : -1;
        const uint32_t ploc = __shfl(loc, src < 0 ? 0 : src);
        const bool start = hit && (src < 0 ? (!have || loc != cur_loc) : (loc != ploc));
        unsigned long long sb = __ballot(start);
        int lo = 0;
        while (sb) {  // wave-uniform
            const int sl = __ffsll(sb) - 1;
            sb &= sb - 1;
            const unsigned long long seg = hb & ((1ULL << sl) - 1ULL) & ~((1ULL << lo) - 1ULL);
            if (have) {
                if (seg) {
                    cur_cnt += __popcll(seg);
                    cur_last = lane_value(woff, 63 - __clzll((long long)seg));
                }
                push(runs, lane);
            }
            cur_loc = lane_value(loc, sl);
            cur_first = lane_value(woff, sl);
            cur_last = cur_first;
            cur_cnt = 0;
            have = 1;
            lo = sl;
        }
        const unsigned long long seg = hb & ~((1ULL << lo) - 1ULL);
        cur_cnt += __popcll(seg);
        cur_last = lane_value(woff, 63 - __clzll((long long)seg));
    }
};

// A wave works on NT consecutive tiles of one partition per pass over its phases, so that every
// phase has NT (or more) independent memory requests in flight per lane: descriptors, words,
// first-chance buckets, second-chance buckets.  DBG: timing experiments (1 = no bucket loads,
// 2 = stop after the queue); results are wrong for DBG != 0.
template <int KT, int NT, int kPipe, int DBG>
__global__ __launch_bounds__(256, GS_PART_WAVES) void k_probe_part(CompactView ct, TableView tab, const uint64_t *__restrict__ words,
                                                    const TileDesc *__restrict__ tdesc, uint32_t n_tiles, PartRec *recs) {
    static_assert(NT == 2 || NT == 4, "tile index takes 2 bits of a queue entry");
    constexpr int QCAP = 640;  // queue window (items); more items are handled window by window
    constexpr int PCAP = 64;   // deferred (second-chance / generic) lookups per flush
    // queue entry: woff[0:11) | tile[11:13) | hit[13] | loc[14:32)
    __shared__ uint32_t sA[4][NT][2 * (kTileWords + 2)];  // the tiles' 2-bit streams as 32-bit words in stream order
    __shared__ uint32_t sQ[4][QCAP];
    __shared__ uint32_t sP[4][PCAP][3];
    __shared__ uint32_t sRun[4][kMaxRuns][3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nbatch = (n_tiles + NT - 1) / NT;
    const uint32_t groups = (nbatch + 3) / 4;
    // blocks [pass * 8 * groups, (pass + 1) * 8 * groups) work on partitions pass * 8 + (b % 8): dispatch order
    // keeps one 1/kNPart slice of the table hot in each XCD's L2 at a time
    const uint32_t pass = blockIdx.x / (8u * groups);
    const uint32_t bip = blockIdx.x - pass * 8u * groups;
    const uint32_t p = pass * 8u + (bip & 7u);
    const uint32_t batch = (bip >> 3) * 4 + wave;
    if (batch >= nbatch) return;  // waves are independent: no workgroup barrier below
    const uint32_t tile0 = batch * NT;
    const int k = KT ? KT : tab.k;
    const uint32_t wlane = (uint32_t)lane * 32u;

    uint64_t m[NT];
    uint32_t ex[NT], tot[NT];
    {
        TileDesc td[NT];
#pragma unroll
        for (int q = 0; q < NT; q++) {
            td[q] = tdesc[min(tile0 + q, n_tiles - 1)];
            if (tile0 + q >= n_tiles) td[q].nwin_left = 0;
        }
        uint64_t a[NT], b2[NT];
#pragma unroll
        for (int q = 0; q < NT; q++) {
            a[q] = __builtin_nontemporal_load(words + td[q].wbase + lane);
            b2[q] = __builtin_nontemporal_load(words + td[q].wbase + lane + 1);
        }
        // Which of my 32 windows of each tile belong to partition p, all at once.  Base j of the 64-base string
        // (a:b2) sits at bits 127-2j, 126-2j.  W1 field b = x_b ^ x_{b+k-1}; W2 field b+1 = x_{b+1} ^ x_{b+k-2}.
        const int s1 = 2 * (k - 1), s2 = 2 * (k - 3);
        const uint64_t t1 = (uint64_t)(p >> 2) * 0x5555555555555555ULL, t2 = (uint64_t)(p & 3u) * 0x5555555555555555ULL;
        uint32_t cnt[NT];
#pragma unroll
        for (int q = 0; q < NT; q++) {
            reinterpret_cast<uint2 *>(sA[wave][q])[lane] = make_uint2((uint32_t)(a[q] >> 32), (uint32_t)a[q]);
            if (lane == 63) reinterpret_cast<uint2 *>(sA[wave][q])[64] = make_uint2((uint32_t)(b2[q] >> 32), (uint32_t)b2[q]);
            const uint32_t nvalid = td[q].nwin_left > wlane ? min(td[q].nwin_left - wlane, 32u) : 0u;
            const uint64_t w1 = a[q] ^ ((a[q] << s1) | (b2[q] >> (64 - s1)));
            const uint64_t w2h = a[q] ^ ((a[q] << s2) | (b2[q] >> (64 - s2)));
            const uint64_t w2l = b2[q] ^ (b2[q] << s2);
            const uint64_t w2s = (w2h << 2) | (w2l >> 62);
            const uint64_t e = (w1 ^ t1) | (w2s ^ t2);
            uint64_t mm = ~(e | (e >> 1)) & 0x5555555555555555ULL;  // bit 62-2b set <=> window b is in partition p
            mm &= nvalid >= 32 ? ~0ULL : ~(~0ULL >> (2 * nvalid));
            m[q] = mm;
            cnt[q] = __popcll(mm);
        }
        // per-tile exclusive prefix over lanes, two tiles per 32-bit scan (counts < 2^16)
#pragma unroll
        for (int q = 0; q < NT; q += 2) {
            const uint32_t packed = cnt[q] | (cnt[q + 1] << 16);
            uint32_t t;
            const uint32_t e2 = wave_excl_scan(packed, lane, t);
            ex[q] = e2 & 0xFFFFu;
            tot[q] = t & 0xFFFFu;
            ex[q + 1] = e2 >> 16;
            tot[q + 1] = t >> 16;
        }
    }
    uint32_t vbase[NT + 1];  // the tiles' item lists back to back
    vbase[0] = 0;
#pragma unroll
    for (int q = 0; q < NT; q++) vbase[q + 1] = vbase[q] + tot[q];
    const uint32_t V = vbase[NT];
    if (DBG == 2) {
        if (lane == 0)
            for (int q = 0; q < NT; q++)
                if (tile0 + q < n_tiles) recs[(uint64_t)(tile0 + q) * kNPart + p].head = 0;
        return;
    }

    const int kbits = 2 * k - 4;
    const uint64_t keymask = (1ULL << kbits) - 1ULL;
    const uint32_t locmask = (1u << ct.locbits) - 1u;
    const uint32_t tagmask = low_bits32(ct.rembits + 1);  // remainder + alt bit
    const uint4 *part_buckets = ct.buckets + ((uint64_t)p << ct.qb);
    uint32_t (*runs)[3] = sRun[wave];
    PartState st;
    st.reset();
    int cur_q = 0;

    auto canon_of = [&](uint32_t entry) -> uint64_t {  // canonical code of the window a queue entry names
        const uint32_t q = (entry >> 11) & (uint32_t)(NT - 1), woff = entry & 2047u;
        // the window's 2k <= 64 bits start at bit `ob` (< 32) of three consecutive 32-bit words of the stream
        const uint32_t i32 = woff >> 4, ob = 2u * (woff & 15u);
        const uint32_t h0 = sA[wave][q][i32], h1 = sA[wave][q][i32 + 1], h2 = sA[wave][q][i32 + 2];
        const uint32_t xhi = (uint32_t)(((((uint64_t)h0 << 32) | h1) << ob) >> 32);
        const uint32_t xlo = (uint32_t)(((((uint64_t)h1 << 32) | h2) << ob) >> 32);
        const uint64_t f = (((uint64_t)xhi << 32) | xlo) >> (64 - 2 * k);
        const uint64_t rr = (~revpairs(f)) >> (64 - 2 * k);  // reverse complement of the window itself
        return f < rr ? f : rr;
    };
    // Branch-free bucket compare.  slot ^ (want << locbits) is zero in the tag field exactly for a match.  With the
    // flag layout an empty slot (all ones) carries alt = 1, rem = all ones: it can only equal a second-chance tag
    // whose remainder is all ones, so the explicit empty test is needed for second-chance lookups only.
    const uint32_t fieldmask = tagmask << ct.locbits;
    auto match4 = [&](const uint4 &b, uint32_t want, uint32_t &l, bool check_empty) -> uint32_t {
        const uint32_t sl[4] = {b.x, b.y, b.z, b.w};
        const uint32_t ws = want << ct.locbits;
        bool h = false;
        uint32_t pick = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool eq = (((sl[i] ^ ws) & fieldmask) == 0u) && (!check_empty || sl[i] != kEmptySlot);
            h = h || eq;
            pick = eq ? sl[i] : pick;
        }
        l = h ? (pick & locmask) : l;
        return h ? 1u : 0u;
    };
    auto finish_tile = [&](int q) {  // wave-uniform
        if (st.have) st.push(runs, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (tile0 + q < n_tiles && lane < 3) {
            const uint32_t nr = st.nruns;
            uint4 v;
            if (lane == 0)
                v = make_uint4(nr | (st.complex << 8) | (st.hits << 16), nr > 0 ? runs[0][0] : 0u, nr > 0 ? runs[0][1] : 0u,
                               nr > 0 ? runs[0][2] : 0u);
            else if (lane == 1)
                v = make_uint4(nr > 1 ? runs[1][0] : 0u, nr > 1 ? runs[1][1] : 0u, nr > 1 ? runs[1][2] : 0u,
                               nr > 2 ? runs[2][0] : 0u);
            else
                v = make_uint4(nr > 2 ? runs[2][1] : 0u, nr > 2 ? runs[2][2] : 0u, 0u, 0u);
            reinterpret_cast<uint4 *>(recs + (uint64_t)(tile0 + q) * kNPart + p)[lane] = v;
        }
        __builtin_amdgcn_wave_barrier();
        st.reset();
    };

    for (uint32_t vlo = 0; vlo < V; vlo += QCAP) {
        const uint32_t nitems = min(V - vlo, (uint32_t)QCAP);
        // queue window [vlo, vlo + nitems): one entry per matching window, in (tile, window) order
#pragma unroll
        for (int q = 0; q < NT; q++) {
            uint64_t mm = m[q];
            uint32_t v = vbase[q] + ex[q] - vlo;  // wraps below the window: the unsigned compare rejects it
            while (mm) {
                const int pb = __clzll((long long)mm);  // 1, 3, 5, ... (window b at bit 62-2b)
                mm &= ~(0x8000000000000000ULL >> pb);
                if (v < nitems) sQ[wave][v] = ((uint32_t)q << 11) | (wlane + (uint32_t)(pb >> 1));
                v++;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        const uint32_t ngroups = DBG == 6 ? 0u : (nitems + 63u) / 64u;
        uint32_t npend = 0;  // wave-uniform
        auto flush_pending = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j0 = 0; j0 < npend; j0 += 64) {
                const uint32_t j = j0 + lane;
                const bool on = j < npend;
                const uint32_t meta = sP[wave][on ? j : 0][0], tg = sP[wave][on ? j : 0][1], bkt = sP[wave][on ? j : 0][2];
                const uint32_t idx = meta & 1023u;  // QCAP <= 1024
                uint32_t h = 0, l = 0;
                if (on && (meta & 1024u)) {  // second-chance bucket
                    const uint4 b = part_buckets[alt_bucket(bkt, tg, ct.qb)];
                    h = match4(b, tg | (1u << ct.rembits), l, true);
                }
                if (on && !h && (meta & 2048u)) {  // rare: both buckets were full at build time
                    const uint64_t c = canon_of(sQ[wave][idx]);
                    uint64_t hh = slot_of(c, tab.shift);
                    Slot cur = load_slot(tab.slots + hh);
                    while (cur.key != c && cur.key != kEmptyKey) {
                        hh = (hh + 1) & tab.mask;
                        cur = load_slot(tab.slots + hh);
                    }
                    if (cur.key == c) {
                        h = 1;
                        l = cur.loc;
                    }
                }
                if (on && h) sQ[wave][idx] |= (1u << 13) | (l << 14);
            }
            npend = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };

        auto chunk = [&](uint32_t g0, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;  // all kPipe groups exist: no guards, so the compiler can interleave them
            // Straight-line per-lane code: lanes past the end compute on harmless values (every LDS and bucket
            // index they form is in range) and are masked out at the end.
            uint32_t entry[kPipe], tag[kPipe], bkt[kPipe];
            uint4 bk[kPipe];
#pragma unroll
            for (int u = 0; u < kPipe; u++) {
                if (FULL || g0 + u < ngroups) {  // wave-uniform
                    entry[u] = sQ[wave][(g0 + u) * 64 + lane];
                    const BucketRem br = hmix2((canon_of(entry[u]) >> 4) & keymask, kbits, ct.qb);
                    tag[u] = (uint32_t)br.rem;
                    bkt[u] = br.bucket;
                    bk[u] = DBG == 1 ? make_uint4(kEmptySlot, kEmptySlot, kEmptySlot, br.rem) : part_buckets[br.bucket];
                }
            }
#pragma unroll
            for (int u = 0; u < kPipe; u++) {
                if (!FULL && g0 + u >= ngroups) break;  // wave-uniform
                const uint32_t idx = (g0 + u) * 64 + lane;
                if (DBG == 4) {
                    if ((bk[u].x ^ bk[u].y ^ bk[u].z ^ bk[u].w) == 0x12345u) sQ[wave][idx] = 1;
                    continue;
                }
                const uint32_t live = idx < nitems;
                uint32_t l = 0;
                const uint32_t h = match4(bk[u], tag[u], l, !ct.has_flag) & live;
                const uint32_t full = max(max(bk[u].x, bk[u].y), max(bk[u].z, bk[u].w)) != kEmptySlot;
                const uint32_t miss_full = live & (h ^ 1u) & full;
                const uint32_t again = ct.has_flag ? (miss_full & (bk[u].x >> 31)) : 0u;
                const uint32_t generic = ct.has_flag ? (miss_full & ((bk[u].x >> 30) & 1u)) : miss_full;
                if (h) sQ[wave][idx] = entry[u] | (1u << 13) | (l << 14);
                const unsigned long long pb = __ballot((again | generic) != 0);
                if (pb) {  // a few lanes per group at most: park them, resolve them together later
                    const uint32_t np = __popcll(pb);
                    if (npend + np > (uint32_t)PCAP) flush_pending();
                    if (again | generic) {
                        const uint32_t slot = npend + __popcll(pb & (lane ? (~0ULL >> (64 - lane)) : 0ULL));
                        sP[wave][slot][0] = idx | (again << 10) | (generic << 11);
                        sP[wave][slot][1] = tag[u];
                        sP[wave][slot][2] = bkt[u];
                    }
                    npend += np;
                }
            }
        };
        {
            uint32_t g0 = 0;
            for (; g0 + kPipe <= ngroups; g0 += kPipe) chunk(g0, std::true_type{});
            if (g0 < ngroups) chunk(g0, std::false_type{});
        }
        if (npend) flush_pending();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        if (V <= (uint32_t)QCAP && DBG < 3) {
            // Usual case, the whole batch sits in this window: run-length pass tile by tile, three groups of 64
            // entries at a time.  The group descriptors (hit ballot, first/last hit, "all the same loc") are
            // independent of each other, only their merge into the run list is sequential and it is scalar.
            for (int q = 0; q < NT; q++) {
                const uint32_t n = tot[q], base = vbase[q];
                for (uint32_t j0 = 0; j0 < n; j0 += 192) {
                    uint32_t e[3];
                    unsigned long long hb[3], nb[3];
                    uint32_t floc[3], fw[3], lw[3];
#pragma unroll
                    for (int u = 0; u < 3; u++) {
                        const uint32_t j = j0 + u * 64 + lane;
                        e[u] = j < n ? sQ[wave][base + j] : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 3; u++) {
                        const bool hit = ((e[u] >> 13) & 1u) != 0;
                        hb[u] = __ballot(hit);
                        const int fl = hb[u] ? __ffsll(hb[u]) - 1 : 0, ll = hb[u] ? 63 - __clzll((long long)hb[u]) : 0;
                        floc[u] = lane_value(e[u] >> 14, fl);
                        fw[u] = lane_value(e[u] & 2047u, fl);
                        lw[u] = lane_value(e[u] & 2047u, ll);
                        nb[u] = __ballot(hit && (e[u] >> 14) != floc[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 3; u++) {
                        if (!hb[u]) continue;
                        if (nb[u]) {  // a location change inside the group: the general fold
                            st.step(((e[u] >> 13) & 1u) != 0, e[u] >> 14, e[u] & 2047u, lane, runs);
                            continue;
                        }
                        const uint32_t c = (uint32_t)__popcll(hb[u]);
                        if (!(st.have && st.cur_loc == floc[u])) {
                            if (st.have) st.push(runs, lane);
                            st.cur_loc = floc[u];
                            st.cur_first = fw[u];
                            st.cur_cnt = 0;
                            st.have = 1;
                        }
                        st.cur_cnt += c;
                        st.cur_last = lw[u];
                        st.hits += c;
                    }
                }
                finish_tile(q);
            }
            cur_q = NT;
        } else {
        // run-length pass over the resolved entries, tile by tile (a group of 64 may straddle tiles)
        for (uint32_t g = 0; g < (DBG >= 3 ? 0u : ngroups); g++) {
            const uint32_t idx = g * 64 + lane;
            const uint32_t e = idx < nitems ? sQ[wave][idx] : 0u;
            const bool hit = ((e >> 13) & 1u) != 0;
            const int myq = (int)((e >> 11) & (uint32_t)(NT - 1));
            const uint32_t vfirst = vlo + g * 64u, vlast = vlo + min(g * 64u + 63u, nitems - 1u);
            int qlo = 0, qhi = 0;
#pragma unroll
            for (int i = 1; i < NT; i++) {
                qlo += vfirst >= vbase[i] ? 1 : 0;
                qhi += vlast >= vbase[i] ? 1 : 0;
            }
            for (int q = qlo; q <= qhi; q++) {  // wave-uniform
                while (cur_q < q) finish_tile(cur_q++);
                st.step(hit && myq == q, e >> 14, e & 2047u, lane, runs);
            }
        }
        }
    }
    while (cur_q < NT) finish_tile(cur_q++);
}

// One wave merges 4 tiles: 16 lanes per tile, one partition summary (<= 3 runs) per lane.  The runs are
// peeled loc by loc in order of first window: for the loc of the earliest remaining run, its hull
// [min first, max last] and hit count over all remaining runs with that loc (three 16-lane reductions).
// If the hulls are pairwise disjoint the order of hits across partitions is decided without knowing it:
// the tile's raw hits are hull after hull, one candidate (location change) at each hull's first window,
// with rank = hits of the earlier hulls.  Anything else (overlapping hulls, > 4 locs, a summary that
// overflowed) is queued for the exact generic probe.
constexpr int kMaxHulls = 4;
__global__ __launch_bounds__(256) void k_merge_part(const PartRec *recs, uint32_t n_tiles, TileArrays ta, uint32_t *cand,
                                                    uint32_t cand_cap, uint32_t *g_cand_count, uint32_t *complex_list,
                                                    uint32_t *complex_n) {
    const int lane = threadIdx.x & 63, part = lane & 15;
    const uint32_t tile = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool on = tile < n_tiles;
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0;
    if (on) {
        const uint4 *rp = reinterpret_cast<const uint4 *>(recs + (uint64_t)tile * kNPart + part);
        r0 = rp[0];
        r1 = rp[1];
        r2 = rp[2];
    }
    const uint32_t nr = r0.x & 0xFFu;
    uint32_t rloc[kMaxRuns] = {r0.y, r1.x, r1.w};
    uint32_t rfl[kMaxRuns] = {r0.z, r1.y, r2.x};  // first | last << 16
    uint32_t rcnt[kMaxRuns] = {r0.w, r1.z, r2.y};
    uint32_t alive = nr >= 3 ? 7u : (nr == 2 ? 3u : (nr == 1 ? 1u : 0u));
    uint32_t bad = (r0.x >> 8) & 1u, total = r0.x >> 16;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        bad |= __shfl_xor(bad, d, 16);
        total += __shfl_xor(total, d, 16);
    }
    uint32_t hull_loc[kMaxHulls], hull_first[kMaxHulls], hull_rank[kMaxHulls];
    uint32_t nh = 0, seen = 0, max_last = 0;
#pragma unroll
    for (int it = 0; it <= kMaxHulls; it++) {
        // earliest remaining run of the tile -> its loc
        uint32_t key = 0xFFFFFFFFu;  // first << 16 | lane-local tie breaker is irrelevant: equal firsts share... see below
        uint32_t kloc = 0;
#pragma unroll
        for (int i = 0; i < kMaxRuns; i++) {
            const uint32_t f = rfl[i] & 0xFFFFu;
            if (((alive >> i) & 1u) && f < key) {
                key = f;
                kloc = rloc[i];
            }
        }
        // argmin over 16 lanes of (first, loc): two windows are never equal across partitions, so `first` alone decides
        uint32_t best = key, bloc = kloc;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const uint32_t ob = __shfl_xor(best, d, 16), ol = __shfl_xor(bloc, d, 16);
            if (ob < best) {
                best = ob;
                bloc = ol;
            }
        }
        if (best == 0xFFFFFFFFu) continue;  // nothing left in this tile (other tiles of the wave may go on)
        if (it == kMaxHulls) {
            bad = 1;
            continue;
        }
        uint32_t hl = 0, hc = 0;  // hull end and hits of loc `bloc`
#pragma unroll
        for (int i = 0; i < kMaxRuns; i++)
            if (((alive >> i) & 1u) && rloc[i] == bloc) {
                hl = max(hl, rfl[i] >> 16);
                hc += rcnt[i];
                alive &= ~(1u << i);
            }
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            hl = max(hl, (uint32_t)__shfl_xor(hl, d, 16));
            hc += __shfl_xor(hc, d, 16);
        }
        if (nh && best <= max_last) bad = 1;  // this hull starts inside an earlier one
        max_last = max(max_last, hl);
#pragma unroll
        for (int i = 0; i < kMaxHulls; i++)
            if ((uint32_t)i == nh) {
                hull_loc[i] = bloc;
                hull_first[i] = best;
                hull_rank[i] = seen;
            }
        nh++;
        seen += hc;
    }
    if (on && part == 0) {
        if (bad) {
            complex_list[atomicAdd(complex_n, 1u)] = tile;
        } else {
            ta.n_raw[tile] = total;
            ta.n_cand[tile] = nh;
            static_assert(kMaxHulls <= kDirectCand, "a merged tile's candidates live in its own slots");
            const uint32_t off = tile * kDirectCand;
            if (nh) {
#pragma unroll
                for (int i = 0; i < kMaxHulls; i++)
                    if ((uint32_t)i < nh) cand[off + i] = (hull_rank[i] << 16) | hull_first[i];
                ta.first_loc[tile] = hull_loc[0];
                uint32_t ll = hull_loc[0];
#pragma unroll
                for (int i = 1; i < kMaxHulls; i++) ll = (uint32_t)i < nh ? hull_loc[i] : ll;
                ta.last_loc[tile] = ll;
            }
            ta.cand_off[tile] = off;
        }
    }
}

// ------------------------------------------------------------------------------------------
// LDS join (k = 20, 32-bit compact table): the same exact table, probed from LDS.
//
// One L2 request per window is itself a wall (~270 G requests/s chip-wide), and finding "the windows of my
// partition" inside every tile costs more instructions than the probes.  So the windows are routed instead,
// like the probe side of a radix-partitioned hash join:
//
//   k_join_bin    a workgroup takes a batch of 4 tiles (8192 windows): canonical code, hash, and the
//                 table slice (1 of 256: partition x top 4 bucket bits) of every window; a counting sort
//                 in LDS groups the 32-bit hashed keys by slice; the batch goes out densely packed:
//                 item[batch][8192] (+ the window of each item, + start/count per slice).
//   k_join_segT   transposes the start/count table to [slice][batch] (so that a slice reads it contiguously).
//   k_join_probe  one workgroup per slice stages its 2^(qb-4) buckets (<= 128 KiB) in LDS and streams the
//                 slice's items of all batches: one ds_read_b128 per window instead of an L2 request.  A
//                 wave takes 64 batches at a time and appends its hits {item position, loc + 1} to one
//                 contiguous stream per (slice, 64 batches), with the start of every merge group noted.  Lookups
//                 that need the second-chance bucket or the overflow list are parked and resolved a wave-full at
//                 a time; their hits go to a short unordered side list.
//   k_join_merge  a workgroup per 4 batches (16 tiles) scatters the hits of the 256 streams into window
//                 order (LDS, 16 bits per window) and writes the same per-tile summary as probe_tile:
//                 nothing downstream changes.
//
// All traffic is streaming and nothing is written in place (partial-sector writes cost a read-modify-write).
// Exactness never depends on placement.  Degenerate input (more than 4096 windows of 64 consecutive batches in
// one slice: homopolymer reads) sends those batches' tiles to the generic exact probe.
// ------------------------------------------------------------------------------------------
constexpr int kJoinTiles = 4;
constexpr int kJoinWin = kJoinTiles * kTileWindows;  // windows per batch
constexpr int kJoinSlices = 256;                     // kNPart x 16
constexpr int kJoinSliceBuckets = 8192;              // LDS image of a slice: <= 128 KiB
constexpr int kJoinOvfMax = 512;                     // overflow keys of a slice kept in LDS
constexpr uint32_t kJoinChunkMax = 32768;            // batches per queue buffer (item indices stay below 2^28)
constexpr uint32_t kJoinHitCap = 4096;               // hit records per (slice, 64 batches): the fast path cannot exceed it
constexpr uint32_t kJoinLateCap = 256;                // late hit records per (slice, 64 batches); more = degenerate group
constexpr int kJoinMergeBatches = 4;                 // batches per k_join_merge workgroup (64 KiB of LDS: two per CU)
constexpr int kJoinSub = 64 / kJoinMergeBatches;     // such groups per hit stream

struct JoinQ {
    uint32_t *item;   // [batches][kJoinWin]     hashed key (bucket in slice | remainder), grouped by slice
    uint16_t *woff;   // [batches][kJoinWin]     window of the item inside its batch
    uint32_t *seg;    // [batches][kJoinSlices]  start | count << 16 of a slice's items inside the batch
    uint32_t *segT;   // [kJoinSlices][bpad]     the same, transposed (bpad = batches rounded up to 64)
    uint32_t *hits;   // [kJoinSlices][bpad / 64][kJoinHitCap]  window in its merge group (15 bits: batch & 3, window) | (loc + 1) << 16
    uint32_t *hsub;   // [bpad / kJoinMergeBatches][kJoinSlices] start | count << 16 of a merge group inside its hit stream
    uint32_t *gflag;  // [bpad / 64]             1 = degenerate group: its tiles go to the generic probe
    uint2 *late;      // [kJoinSlices][bpad / 64][kJoinLateCap]  hits found in a second-chance bucket or the overflow list, in no
                      //                          order: {batch & 63, window in the batch (19 bits), loc + 1}
    uint32_t *nlate;  // [bpad / 64][kJoinSlices]
    uint32_t bpad;
};

template <int KT>
__global__ __launch_bounds__(512, 6) void k_join_bin(CompactView ct, const uint64_t *__restrict__ words,
                                                      const TileDesc *__restrict__ tdesc, uint32_t n_tiles, uint32_t batch0,
                                                      JoinQ jq) {
    __shared__ __align__(16) uint32_t dkey[kJoinWin];
    __shared__ __align__(16) uint16_t dwoff[kJoinWin];
    __shared__ __align__(16) uint32_t cnt[kJoinSlices];
    __shared__ __align__(16) uint32_t off[kJoinSlices];
    constexpr int k = KT;
    constexpr int kbits = 2 * k - 4;
    static_assert(8 + KT - 1 <= 32, "8 consecutive windows must fit one 64-bit word");
    constexpr uint64_t kmask = (1ULL << (2 * k)) - 1ULL, keymask = (1ULL << kbits) - 1ULL;
    const int tid = threadIdx.x;
    const uint32_t bl = blockIdx.x;
    if (tid < kJoinSlices) cnt[tid] = 0;
    // thread = 16 consecutive windows: word (tid >> 1) of the batch's 256 words, half (tid & 1) of it; three workgroups per CU
    const int q = __builtin_amdgcn_readfirstlane(tid >> 7), l = (tid >> 1) & 63, h = tid & 1;  // q is the same for a whole wave
    const uint32_t tile = (batch0 + bl) * kJoinTiles + q;
    TileDesc td = tdesc[min(tile, n_tiles - 1)];  // scalar load
    if (tile >= n_tiles) td.nwin_left = 0;
    const uint64_t a = __builtin_nontemporal_load(words + td.wbase + l);
    const uint64_t b2 = __builtin_nontemporal_load(words + td.wbase + l + 1);
    __syncthreads();
    const uint32_t w0 = (uint32_t)l * 32u + (uint32_t)h * 16u;
    const uint32_t nvalid = td.nwin_left > w0 ? min(td.nwin_left - w0, 16u) : 0u;
    const int sb = ct.qb - 4;  // bucket bits inside a slice; sb + rembits = 32
    uint32_t item[16], sr[16];
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int s = 32 * h + 16 * half;
        const uint64_t X = s ? ((a << s) | (b2 >> (64 - s))) : a;  // 8 windows start at bases 0..7 of X (8 + k - 1 <= 32 bases)
        const uint64_t R = ~revpairs(X);                            // base i of X, complemented, at bits 2i+1:2i
        // the partition of every window at once (it is the same for a k-mer and its reverse complement, so the forward
        // strand will do): field b of W1 = x_b ^ x_{b+k-1}, field b+1 of W2 = x_{b+1} ^ x_{b+k-2}; fields 0..8 sit in the high word
        const uint32_t w1 = (uint32_t)((X ^ (X << (2 * (k - 1)))) >> 32), w2 = (uint32_t)((X ^ (X << (2 * (k - 3)))) >> 32);
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {
            const int j = half * 8 + jj;
            const uint64_t f = (X << (2 * jj)) >> (64 - 2 * k);
            const uint64_t r = (R >> (2 * jj)) & kmask;
            const uint64_t c = f < r ? f : r;
            const BucketRem br = hmix2((c >> 4) & keymask, kbits, ct.qb);
            item[j] = ((br.bucket & ((1u << sb) - 1u)) << ct.rembits) | (uint32_t)br.rem;
            const uint32_t part = (((w1 >> (30 - 2 * jj)) & 3u) << 2) | ((w2 >> (28 - 2 * jj)) & 3u);
            const uint32_t slice = (part << 4) | (br.bucket >> sb);
            // (an unconditional add to a dummy counter, so that the atomics overlap, measured slower)
            sr[j] = slice | (((uint32_t)j < nvalid ? atomicAdd(&cnt[slice], 1u) : 0u) << 9);
        }
    }
    __syncthreads();
    if (tid < 64) {  // exclusive prefix over the 256 counters, 4 per lane
        const uint4 c4 = reinterpret_cast<const uint4 *>(cnt)[tid];
        uint32_t total;
        const uint32_t ex = wave_excl_scan(c4.x + c4.y + c4.z + c4.w, tid, total);
        const uint4 o4 = make_uint4(ex, ex + c4.x, ex + c4.x + c4.y, ex + c4.x + c4.y + c4.z);
        reinterpret_cast<uint4 *>(off)[tid] = o4;
        reinterpret_cast<uint4 *>(jq.seg + (uint64_t)bl * kJoinSlices)[tid] =
            make_uint4(o4.x | (c4.x << 16), o4.y | (c4.y << 16), o4.z | (c4.z << 16), o4.w | (c4.w << 16));
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++)
        if ((uint32_t)j < nvalid) {
            const uint32_t pos = off[sr[j] & 255u] + (sr[j] >> 9);
            dkey[pos] = item[j];
            dwoff[pos] = (uint16_t)(((uint32_t)q << 11) | (w0 + (uint32_t)j));
        }
    __syncthreads();
    const uint32_t total = off[kJoinSlices - 1] + cnt[kJoinSlices - 1];
    uint4 *gk = reinterpret_cast<uint4 *>(jq.item + (uint64_t)bl * kJoinWin);
    uint4 *gw = reinterpret_cast<uint4 *>(jq.woff + (uint64_t)bl * kJoinWin);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t e4 = (uint32_t)tid + 512u * (uint32_t)i;
        if (e4 * 4u < total) gk[e4] = reinterpret_cast<const uint4 *>(dkey)[e4];
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t e8 = (uint32_t)tid + 512u * (uint32_t)i;
        if (e8 * 8u < total) gw[e8] = reinterpret_cast<const uint4 *>(dwoff)[e8];
    }
}

// seg [nb][256] -> segT [256][bpad] in 64 x 64 pieces (batches past nb read as empty); also clears the group flags
__global__ __launch_bounds__(256) void k_join_segT(JoinQ jq, uint32_t nb) {
    __shared__ uint32_t tl[64][65];
    const uint32_t b0 = blockIdx.x * 64u, s0 = blockIdx.y * 64u;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (blockIdx.y == 0 && threadIdx.x == 0) jq.gflag[blockIdx.x] = 0;
    for (int i = ty; i < 64; i += 4) tl[i][tx] = b0 + i < nb ? jq.seg[(uint64_t)(b0 + i) * kJoinSlices + s0 + tx] : 0u;
    __syncthreads();
    for (int i = ty; i < 64; i += 4) jq.segT[(uint64_t)(s0 + i) * jq.bpad + b0 + tx] = tl[tx][i];
}

template <int KT, int PIPE, bool HASFLAG, int DBG>
__global__ __launch_bounds__(1024) void k_join_probe(CompactView ct, JoinQ jq, uint32_t nb, uint32_t reps,
                                                      const uint64_t *__restrict__ ovf_code,
                                                      const uint32_t *__restrict__ ovf_loc,
                                                      const uint32_t *__restrict__ ovf_start) {
    constexpr int PCAP = 96;  // parked (second-chance / overflow) lookups; resolved 64 at a time
    __shared__ uint4 sT[kJoinSliceBuckets];
    __shared__ uint32_t sBase[16][65];
    __shared__ unsigned long long sEnd[16][64];
    __shared__ uint32_t sPend[16][PCAP][2];
    __shared__ uint32_t sG[16][kJoinSub + 1];
    __shared__ uint32_t sOvfKey[kJoinOvfMax], sOvfLoc[kJoinOvfMax];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t s = blockIdx.x % (uint32_t)kJoinSlices, rep = blockIdx.x / (uint32_t)kJoinSlices;
    const int sb = ct.qb - 4;
    const uint32_t n_ovf = ovf_start[s + 1] - ovf_start[s];
    {
        const uint4 *src = ct.buckets + ((uint64_t)(s >> 4) << ct.qb) + ((uint64_t)(s & 15u) << sb);
        for (uint32_t i = tid; i < (1u << sb); i += blockDim.x) sT[i] = src[i];
        if (n_ovf <= (uint32_t)kJoinOvfMax)
            for (uint32_t i = tid; i < n_ovf; i += blockDim.x) {
                sOvfKey[i] = (uint32_t)ovf_code[ovf_start[s] + i];  // the item part; the slice part is s
                sOvfLoc[i] = ovf_loc[ovf_start[s] + i];
            }
    }
    __syncthreads();
    const int rembits = ct.rembits, locbits = ct.locbits;
    const uint32_t remmask = (1u << rembits) - 1u;
    const uint32_t locmask = (1u << locbits) - 1u;
    constexpr uint32_t slot0mask = HASFLAG ? 0x3FFFFFFFu : 0xFFFFFFFFu;
    const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint32_t ng64 = jq.bpad / 64u;

    // second lookup of a parked item: its second-chance bucket (same slice, alt bit set in the slot), then the slice's
    // overflow keys; returns loc + 1 or 0
    auto second = [&](uint32_t key, uint32_t flags) -> uint32_t {
        const uint32_t bucket = key >> rembits, rem = key & remmask;
        uint32_t res = 0;
        if (flags & 1u) {
            const uint4 b = sT[alt_bucket(bucket, rem, ct.qb)];
            const uint32_t wa = (rem | (1u << rembits)) << locbits;
            const uint32_t sl[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t e = (sl[i] ^ wa) & (i == 0 ? slot0mask : 0xFFFFFFFFu);
                if (sl[i] != kEmptySlot32 && e < locmask) res = e + 1u;
            }
        }
        if (!res && (flags & 2u)) {  // rare: the key overflowed both buckets at build time
            if (n_ovf <= (uint32_t)kJoinOvfMax) {
                uint32_t lo = 0, hi = n_ovf;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (sOvfKey[mid] < key) lo = mid + 1;
                    else hi = mid;
                }
                if (lo < n_ovf && sOvfKey[lo] == key) res = sOvfLoc[lo] + 1u;
            } else {  // more overflow keys than the LDS list holds (tables without second-chance buckets): search them in HBM
                const uint64_t code = ((uint64_t)s << 32) | key;
                uint32_t lo = ovf_start[s], hi = ovf_start[s + 1];
                const uint32_t end = hi;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (ovf_code[mid] < code) lo = mid + 1;
                    else hi = mid;
                }
                if (lo < end && ovf_code[lo] == code) res = ovf_loc[lo] + 1u;
            }
        }
        return res;
    };

    const uint32_t ngroups = (nb + 63u) / 64u;
    const uint32_t nwaves = blockDim.x >> 6;  // 16; fewer only in occupancy experiments
    const uint32_t gfirst = rep * nwaves + (uint32_t)wave, gstep = reps * nwaves;
    uint32_t sg_next = gfirst < ngroups ? jq.segT[(uint64_t)s * jq.bpad + gfirst * 64u + (uint32_t)lane] : 0u;
    for (uint32_t g = gfirst; g < ngroups; g += gstep) {
        const uint32_t b = g * 64u + (uint32_t)lane;
        const uint32_t sg = sg_next;  // batches past nb are empty
        // the next group's start/count table is on its way while this group is probed
        if (g + gstep < ngroups) sg_next = jq.segT[(uint64_t)s * jq.bpad + (g + gstep) * 64u + (uint32_t)lane];
        const uint32_t cnt = sg >> 16, start = sg & 0xFFFFu;
        uint32_t T;
        const uint32_t ex = wave_excl_scan(cnt, lane, T);
        uint32_t *hsub = jq.hsub + (uint64_t)g * kJoinSub * kJoinSlices + s;  // [merge groups of the 64 batches][slices]
        if (T == 0 || T > kJoinHitCap) {
            if (lane < kJoinSub) hsub[(uint32_t)lane * kJoinSlices] = 0u;
            if (lane == 0) jq.nlate[(uint64_t)g * kJoinSlices + s] = 0u;
            if (T && lane == 0) jq.gflag[g] = 1u;
            continue;
        }
        // the items of the 64 batches as one stream: position -> batch through a bit vector of segment ends
        const unsigned long long nz = __ballot(cnt != 0);
        sEnd[wave][lane] = 0ULL;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) sBase[wave][__popcll(nz)] = 0u - T;  // positions past T (last step) name this entry: item[0..63]
        if (cnt) {
            sBase[wave][__popcll(nz & lt)] = b * (uint32_t)kJoinWin + start - ex;
            const uint32_t e = ex + cnt - 1u;
            atomicOr(&sEnd[wave][e >> 6], 1ULL << (e & 63u));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long word = sEnd[wave][lane];
        uint32_t unused;
        const uint32_t before = wave_excl_scan((uint32_t)__popcll(word), lane, unused);
        const uint32_t wlo = (uint32_t)word, whi = (uint32_t)(word >> 32);
        const uint32_t nsteps = (T + 63u) / 64u;
        uint32_t *hout = jq.hits + ((uint64_t)s * ng64 + g) * kJoinHitCap;
        uint32_t cursor = 0;  // hits written so far (wave-uniform)
        // hit-stream position where each merge group starts: noted when the item stream crosses its first item
        int jn = 0;         // next group whose start is not noted yet
        uint32_t jpos = 0;  // its first item position (= ex of its first batch)

        // two sets of window buffers {where a hit would be, item, home bucket}: one is probed while the other is loaded
        uint32_t idx_a[PIPE], key_a[PIPE], idx_b[PIPE], key_b[PIPE];
        uint4 bk_a[PIPE], bk_b[PIPE];
        auto fetch = [&](uint32_t t0, uint32_t (&idx_n)[PIPE], uint32_t (&key_n)[PIPE]) {
#pragma unroll
            for (int u = 0; u < PIPE; u++) {
                if (t0 + (uint32_t)u >= nsteps) break;  // wave-uniform: the last window of a group is usually short
                const uint32_t t = (t0 + (uint32_t)u) & 63u, p = (t0 + (uint32_t)u) * 64u + (uint32_t)lane;
                const uint32_t sl = lane_value(wlo, (int)t), sh = lane_value(whi, (int)t), sbv = lane_value(before, (int)t);
                // lanes past T may name one segment past the last: sBase has a spare entry
                const uint32_t ci = __builtin_amdgcn_mbcnt_hi(sh, __builtin_amdgcn_mbcnt_lo(sl, sbv));
                const uint32_t ix = sBase[wave][ci] + p;
                key_n[u] = jq.item[ix];  // lanes past T read some valid word (sBase's spare entry) and are masked below
                // where a hit of this item is: batch & 63 | window inside the batch
                idx_n[u] = (ix & 0x7E000u) | (uint32_t)jq.woff[ix];
            }
        };
#pragma unroll
        for (int u = 0; u < PIPE; u++) key_a[u] = key_b[u] = 0u;  // steps a short window does not have still name a valid bucket
        fetch(0, idx_a, key_a);
#pragma unroll
        for (int u = 0; u < PIPE; u++) bk_a[u] = sT[(key_a[u] >> rembits) & (uint32_t)(kJoinSliceBuckets - 1)];
        uint2 *lout = jq.late + ((uint64_t)s * ng64 + g) * kJoinLateCap;
        uint32_t npend = 0, nlate = 0;  // wave-uniform: parked lookups, late hits written
        // resolve the parked lookups (at most one per lane); their hits go to the side list, in no particular order
        auto flush_parked = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j0 = 0; j0 < npend; j0 += 64u) {
                uint32_t res = 0, where = 0;
                if (j0 + (uint32_t)lane < npend) {
                    where = sPend[wave][j0 + lane][1];
                    res = second(sPend[wave][j0 + lane][0], where >> 19);
                }
                const unsigned long long hb = __ballot(res != 0);
                const uint32_t nh = __popcll(hb);
                if (nlate + nh > kJoinLateCap) {
                    if (lane == 0) jq.gflag[g] = 1u;  // more late hits than the list holds: exact generic probe for this group
                } else if (res) {  // {batch & 63, window inside the batch}, loc + 1
                    lout[nlate + (uint32_t)__popcll(hb & lt)] = make_uint2(where & 0x7FFFFu, res);
                }
                nlate = min(nlate + nh, kJoinLateCap);
            }
            npend = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        auto window = [&](uint32_t t0, auto full_tag, uint32_t (&idx)[PIPE], uint32_t (&key)[PIPE], uint4 (&bk)[PIPE],
                          uint32_t (&idx_n)[PIPE], uint32_t (&key_n)[PIPE], uint4 (&bk_n)[PIPE]) {
            constexpr bool FULL = decltype(full_tag)::value;  // all PIPE steps exist: no guards
            uint32_t rp[PIPE];  // loc + 1 of a hit in the home bucket, or 0
            const bool more = t0 + PIPE < nsteps;  // wave-uniform
            if (more) fetch(t0 + PIPE, idx_n, key_n);
            if (npend > (uint32_t)PCAP - 64u) flush_parked();  // the one place inside a group: room for 64 more lookups
#pragma unroll
            for (int u = 0; u < PIPE; u++) {
                if (!FULL && t0 + (uint32_t)u >= nsteps) {  // wave-uniform
                    rp[u] = 0u;
                    continue;
                }
                const bool live = (t0 + (uint32_t)u) * 64u + (uint32_t)lane < T;
                const uint32_t ws = (key[u] & remmask) << locbits;
                const uint32_t e0 = (bk[u].x ^ ws) & slot0mask, e1 = bk[u].y ^ ws, e2 = bk[u].z ^ ws, e3 = bk[u].w ^ ws;
                const uint32_t m = min(min(e0, e1), min(e2, e3));
                // a matching slot leaves its loc (< locmask: the all-ones loc is never stored); an empty slot leaves at least locmask
                const bool hit = m < locmask;
                rp[u] = live && hit ? m + 1u : 0u;
                // bits 31:30 of slot 0: a key of this bucket lives in its second-chance bucket / in the overflow list only
                // (set on full buckets only; an empty slot 0 has them set too, hence the test on slot 3: slots fill in order).
                // Such lookups are parked: bit 0 try the second-chance bucket, bit 1 try the overflow list.
                const uint32_t pf = HASFLAG ? ((bk[u].x >> 31) | ((bk[u].x >> 29) & 2u)) : 2u;
                const bool pend = DBG != 2 && live && !hit && bk[u].w != kEmptySlot32 && pf != 0u;
                const unsigned long long pb = __ballot(pend);
                if (pb) {  // wave-uniform
                    const uint32_t np = __popcll(pb);
                    if (npend + np > (uint32_t)PCAP) {  // no room (a table with very many displaced keys): exact generic probe
                        if (lane == 0) jq.gflag[g] = 1u;
                        continue;
                    }
                    if (pend) {
                        const uint32_t slot = npend + (uint32_t)__popcll(pb & lt);
                        sPend[wave][slot][0] = key[u];
                        sPend[wave][slot][1] = (pf << 19) | (idx[u] & 0x7FFFFu);  // batch & 63, index inside the batch
                    }
                    npend += np;
                }
            }
            if (more) {  // the next window's bucket reads are in flight while this window is finished
#pragma unroll
                for (int u = 0; u < PIPE; u++)
                    bk_n[u] = DBG == 3 ? make_uint4(key_n[u], ~0u, ~0u, ~0u) : sT[(key_n[u] >> rembits) & (uint32_t)(kJoinSliceBuckets - 1)];
            }
#pragma unroll
            for (int u = 0; u < PIPE; u++) {
                if (!FULL && t0 + (uint32_t)u >= nsteps) break;  // wave-uniform
                const unsigned long long hb = __ballot(rp[u] != 0);
                const uint32_t p0 = (t0 + (uint32_t)u) * 64u;
                // merge groups that start inside [p0, p0 + 64): their hits start after the hits of the lanes before
                while (jn < kJoinSub && jpos < p0 + 64u) {  // wave-uniform
                    const uint32_t d = jpos > p0 ? jpos - p0 : 0u;
                    const uint32_t v = DBG ? 0u : cursor + (uint32_t)__popcll(hb & (d ? (~0ULL >> (64 - d)) : 0ULL));
                    if (lane == 0) sG[wave][jn] = v;
                    jn++;
                    jpos = jn < kJoinSub ? lane_value(ex, jn * kJoinMergeBatches) : 0xFFFFFFFFu;
                }
                if (DBG) {  // timing experiments: an empty hit stream keeps everything downstream well-defined
                    if (DBG == 1 ? rp[u] == 0x12345u : (rp[u] != 0 && cursor == 0xFFFFFFF0u)) hout[(uint32_t)__popcll(hb & lt)] = idx[u];
                    if (DBG == 1) continue;
                    cursor += (uint32_t)__popcll(hb);
                    continue;
                }
                // a hit record names its window inside the 4-batch merge group: (batch & 3) << 13 | window in the batch
                if (rp[u]) hout[cursor + (uint32_t)__popcll(hb & lt)] = (idx[u] & 0x7FFFu) | (rp[u] << 16);
                cursor += (uint32_t)__popcll(hb);
            }
        };
        {
            uint32_t t0 = 0;
            bool in_a = true;  // wave-uniform: which set holds the window at t0
            for (; t0 + 2 * PIPE <= nsteps; t0 += 2 * PIPE) {
                window(t0, std::true_type{}, idx_a, key_a, bk_a, idx_b, key_b, bk_b);
                window(t0 + PIPE, std::true_type{}, idx_b, key_b, bk_b, idx_a, key_a, bk_a);
            }
            if (t0 + PIPE <= nsteps) {
                window(t0, std::true_type{}, idx_a, key_a, bk_a, idx_b, key_b, bk_b);
                t0 += PIPE;
                in_a = false;
            }
            if (t0 < nsteps) {
                if (in_a) window(t0, std::false_type{}, idx_a, key_a, bk_a, idx_b, key_b, bk_b);
                else window(t0, std::false_type{}, idx_b, key_b, bk_b, idx_a, key_a, bk_a);
            }
        }
        if (npend) flush_parked();
        if (lane == 0) jq.nlate[(uint64_t)g * kJoinSlices + s] = DBG ? 0u : nlate;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < kJoinSub) {
            // groups whose first item was never reached have no items: they start (and end) at the end of the stream
            const uint32_t st = DBG ? 0u : (lane < jn ? sG[wave][lane] : cursor);
            const uint32_t en = DBG ? 0u : (lane + 1 < jn ? sG[wave][lane + 1] : cursor);
            hsub[(uint32_t)lane * kJoinSlices] = st | ((en - st) << 16);
        }
    }
}

// One workgroup per 4 batches (16 tiles, one per wave).  loc + 1 per window in LDS, 16 bits each; lane l of the wave that owns a
// tile reads its 32 windows as one 64-byte row, the 16-byte pieces of a row rotated by l to spread the banks.
__global__ __launch_bounds__(1024, 8) void k_join_merge(JoinQ jq, uint32_t batch0, uint32_t nb, uint32_t n_tiles, TileArrays ta,
                                                     uint32_t *cand, uint32_t cand_cap, uint32_t *g_cand_count,
                                                     uint32_t *complex_list, uint32_t *complex_n) {
    __shared__ __align__(16) uint16_t locat[kJoinMergeBatches * kJoinWin];  // 64 KiB
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t g8 = blockIdx.x, g64 = g8 / (uint32_t)kJoinSub;  // g8: merge group
    const uint32_t tile_lo = (batch0 + g8 * kJoinMergeBatches) * kJoinTiles;
    if (jq.gflag[g64]) {  // degenerate group: exact generic probe for these tiles
        if (tid < kJoinMergeBatches * kJoinTiles && tile_lo + tid < n_tiles) complex_list[atomicAdd(complex_n, 1u)] = tile_lo + tid;
        return;
    }
    // the 16 hit streams this wave scatters (slices wave, wave + 16, ...): start/count first, so that the loads overlap the clear
    const uint32_t hs_mine = lane < 16 ? jq.hsub[(uint64_t)g8 * kJoinSlices + (uint32_t)wave + 16u * (uint32_t)lane] : 0u;
    const uint32_t ng64 = jq.bpad / 64u;
    // side lists of the 64-batch group (hits found in second-chance buckets / the overflow list, a few per slice, unordered):
    // 4 threads per slice
    const uint32_t late_n = jq.nlate[(uint64_t)g64 * kJoinSlices + ((uint32_t)tid >> 2)];
    const uint2 *late_l = jq.late + ((uint64_t)((uint32_t)tid >> 2) * ng64 + g64) * kJoinLateCap;
    const uint32_t late_mine = g8 % (uint32_t)kJoinSub;  // (batch & 63) / kJoinMergeBatches of my batches
#pragma unroll
    for (int i = 0; i < (int)(sizeof(locat) / 16 / 1024); i++) reinterpret_cast<uint4 *>(locat)[tid + 1024 * i] = make_uint4(0, 0, 0, 0);
    uint2 lrec[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t j = ((uint32_t)tid & 3u) + 4u * (uint32_t)i;
        lrec[i] = j < late_n ? late_l[j] : make_uint2(0u, 0u);  // loc + 1 is never 0
    }
    __syncthreads();
    auto put = [&](uint32_t rec) {
        const uint32_t w = rec & 0x7FFFu;  // window inside the group
        locat[(w & ~31u) | ((((w >> 3) + (w >> 5)) & 3u) << 3) | (w & 7u)] = (uint16_t)(rec >> 16);
    };
    constexpr int MU = 4;  // streams in flight per wave
    for (int i0 = 0; i0 < 16; i0 += MU) {
        uint32_t rec[MU], cnt[MU];
        const uint32_t *h[MU];
#pragma unroll
        for (int u = 0; u < MU; u++) {
            const uint32_t hs = lane_value(hs_mine, i0 + u);
            const uint32_t sl = (uint32_t)wave + 16u * (uint32_t)(i0 + u);
            cnt[u] = hs >> 16;
            h[u] = jq.hits + ((uint64_t)sl * ng64 + g64) * kJoinHitCap + (hs & 0xFFFFu);
            rec[u] = (uint32_t)lane < cnt[u] ? h[u][lane] : 0u;
        }
#pragma unroll
        for (int u = 0; u < MU; u++)
            if (rec[u]) put(rec[u]);
#pragma unroll
        for (int u = 0; u < MU; u++)
            for (uint32_t j = 64u + (uint32_t)lane; j < cnt[u]; j += 64u) put(h[u][j]);  // long streams (rare)
    }
    {  // the side lists: this merge group's records are taken (the first 16 of each list were loaded up front)
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (lrec[i].y && (lrec[i].x >> 13) / (uint32_t)kJoinMergeBatches == late_mine) put((lrec[i].x & 0x7FFFu) | (lrec[i].y << 16));
        for (uint32_t j = 16u + ((uint32_t)tid & 3u); j < late_n; j += 4u) {
            const uint2 r = late_l[j];
            if ((r.x >> 13) / (uint32_t)kJoinMergeBatches == late_mine) put((r.x & 0x7FFFu) | (r.y << 16));
        }
    }
    __syncthreads();
    {
        const uint32_t tl = (uint32_t)wave;  // tile inside the group
        const uint32_t tile = tile_lo + tl;
        if (tile >= n_tiles) return;
        uint32_t raw = 0, chg = 0, first_loc = GS_NONE, prev = GS_NONE;
        const uint4 *row = reinterpret_cast<const uint4 *>(locat + tl * kTileWindows + lane * 32);
#pragma unroll
        for (int pc = 0; pc < 4; pc++) {
            const uint4 v4 = row[(pc + lane) & 3];
            const uint32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int bb = pc * 8 + c;
                const uint32_t x = (v[c >> 1] >> (16 * (c & 1))) & 0xFFFFu;
                if (x) {
                    const uint32_t loc = x - 1u;
                    raw |= 1u << bb;
                    if (prev == GS_NONE) first_loc = loc;
                    else if (loc != prev) chg |= 1u << bb;
                    prev = loc;
                }
            }
        }
        tile_summary(raw, chg, first_loc, prev, tile, n_tiles, ta, cand, cand_cap, g_cand_count);
    }
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
// k_emit: one thread per tile
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_emit(TableView tab, CompactView ct, MzView mz, int compact, ReadsView rd, TileArrays ta,
                                              const uint32_t *cand, uint32_t cand_cap, gs_hit *hits, gs_hit *hits_host,
                                              uint64_t hit_cap, const Agg *emit_total) {
    // short lists (<= kMirrorRows) also go straight to pinned host memory, row by row: no copy after the pass; long ones
    // are fetched in one bulk copy instead (scattered 12-byte writes across PCIe are slow)
    const bool mirror = emit_total->sum <= kMirrorRows;
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= rd.n_tiles) return;
    const uint32_t nc = ta.n_cand[tile];
    if (!nc) return;
    const uint32_t ne = ta.n_emit[tile];
    const uint32_t supp = nc - ne;
    const uint32_t off = ta.cand_off[tile];
    if ((uint64_t)off + nc > cand_cap) return;  // overflowed pass; the host re-runs with a larger buffer
    const uint32_t r = rd.tile_read[tile];
    const uint32_t t0 = rd.read_tile0[r];
    const uint32_t tin = tile - t0;
    const uint64_t raw_before = ta.raw_prefix[tile] - ta.raw_prefix[t0];
    const uint64_t emit_before = ta.emit_prefix[tile] - ta.emit_prefix[t0];
    const uint64_t out0 = ta.emit_prefix[tile];
    const uint64_t rwoff = rd.read_word_off[r];
    const int k = tab.k;
    const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
    for (uint32_t j = supp; j < nc; j++) {
        const uint32_t cv = cand[off + j];
        const uint32_t woff = cv & 0xFFFFu, rank = cv >> 16;
        const uint32_t w = tin * kTileWindows + woff;
        const uint64_t suppressed_before = (raw_before + rank) - (emit_before + (j - supp));
        const uint64_t o = out0 + (j - supp);
        if (o >= hit_cap) return;
        // payload: re-probe this one window
        const uint64_t wi = rwoff + (w >> 5);
        const uint64_t a = rd.words[wi], b2 = rd.words[wi + 1];
        const uint64_t c = canon_at(a, b2, ~revpairs(a), ~revpairs(b2), (int)(w & 31u), k, kmask);
        uint32_t payload = 0, loc_unused = 0;
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
        hrec.read = r;
        hrec.pos = (uint32_t)(w - suppressed_before);
        hrec.payload = payload;
        hits[o] = hrec;
        if (mirror) hits_host[o] = hrec;
    }
}

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
    // small-table fast path (exact compact table resident in the XCD-partitioned L2s)
    int compact = 0;  // 0 generic only, 1 compact-l2, 2 compact-hbm
    uint4 *cbuckets = nullptr;
    uint32_t *cpayload = nullptr;  // payload per compact slot
    int c_qb = 0, c_rembits = 0, c_locbits = 0, c_has_flag = 0, c_w64 = 0;
    uint64_t c_overflow = 0;
    // LDS join (k = 20): the compact table's slices are probed from LDS; keys that overflowed both buckets, in
    // hashed form (slice << 32 | item), sorted
    // minimiser-addressed table (gs_mz.h): large tables
    int mz = 0;
    uint4 *mz_regions = nullptr;
    uint32_t mz_n_regions = 0;
    int mz_m = 0, mz_locbits = 0;
    uint32_t *mz_pay = nullptr;  // payload per slot
    uint64_t mz_overflow = 0, mz_dup = 0;  // keys in the generic side table; keys stored under both strands' picks
    int join = 0;
    uint32_t join_chunk = kJoinChunkMax;
    uint64_t *j_ovf_code = nullptr;
    uint32_t *j_ovf_loc = nullptr;
    uint32_t *j_ovf_start = nullptr;  // [kJoinSlices + 1] first overflow key of every slice
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
    PartRec *part_recs = nullptr;      // [n_tiles * kNPart], fast path only
    uint32_t *complex_list = nullptr;  // [n_tiles]
    JoinQ jq{};                        // LDS join queues for min(batches, kJoinChunk) batches
    PassState ps;
};

namespace {

TableView view(const gs_table *t) { return TableView{t->slots, t->cap - 1, t->shift, t->k}; }
MzView mz_view(const gs_table *t) {
    return MzView{t->mz_regions, t->mz_n_regions, t->k, t->mz_m, t->mz_locbits, t->mz_pay};
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

// batches per pass of the join kernels over one set of queues (GAVISUNK_JOIN_CHUNK, a test knob read once when the
// table is created: small chunks exercise the chunk loop)
uint32_t join_chunk_from_env() {
    const char *e = getenv("GAVISUNK_JOIN_CHUNK");
    const long v = e ? atol(e) : 0;
    return v >= 64 && v <= (long)kJoinChunkMax ? (uint32_t)v / 64u * 64u : kJoinChunkMax;
}

int launch_phase1(gs_table *t, gs_reads *r) {
    HIP_TRY(hipSetDevice(t->device));
    if (t->compact == 1 && t->join && r->n_tiles && !r->jq.item) {
        int rc;
        const uint64_t nb = std::min<uint64_t>(t->join_chunk, (r->n_tiles + kJoinTiles - 1) / kJoinTiles);
        const uint64_t bpad = (nb + 63) / 64 * 64;
        r->jq.bpad = (uint32_t)bpad;
        if ((rc = dev_alloc(&r->jq.item, nb * kJoinWin)) || (rc = dev_alloc(&r->jq.woff, bpad * kJoinWin)) ||
            (rc = dev_alloc(&r->jq.seg, nb * kJoinSlices)) || (rc = dev_alloc(&r->jq.segT, bpad * kJoinSlices)) ||
            (rc = dev_alloc(&r->jq.hits, bpad / 64 * kJoinSlices * kJoinHitCap)) ||
            (rc = dev_alloc(&r->jq.hsub, bpad / kJoinMergeBatches * kJoinSlices)) || (rc = dev_alloc(&r->jq.gflag, bpad / 64)) ||
            (rc = dev_alloc(&r->jq.late, bpad / 64 * kJoinSlices * kJoinLateCap)) || (rc = dev_alloc(&r->jq.nlate, bpad / 64 * kJoinSlices)))
            return rc;
        if (!r->complex_list && (rc = dev_alloc(&r->complex_list, r->n_tiles))) return rc;
    }
    if (t->compact == 1 && !t->join && r->n_tiles && !r->part_recs) {
        int rc;
        if ((rc = dev_alloc(&r->part_recs, (uint64_t)r->n_tiles * kNPart))) return rc;
        if (!r->complex_list && (rc = dev_alloc(&r->complex_list, r->n_tiles))) return rc;
    }
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
        } else if (t->compact == 1 && t->join) {
            const uint32_t n_batches = (r->n_tiles + kJoinTiles - 1) / kJoinTiles;
#ifdef GS_TUNING  // experiments only (tools/, libgavisunk_tuning.so): the shipped library has no such switches
            const char *env_reps = getenv("GAVISUNK_JOIN_REPS");  // workgroups per table slice
            const uint32_t reps = env_reps ? std::max(1, atoi(env_reps)) : 1u;
            const char *env_jt = getenv("GAVISUNK_JOIN_THREADS");  // occupancy experiment: probe workgroup size (multiple of 64)
            const uint32_t jthreads = env_jt ? std::min(1024, std::max(512, atoi(env_jt) / 64 * 64)) : 1024u;
            const char *env_jp = getenv("GAVISUNK_JOIN_PIPE");  // item loads in flight per lane
            const int jpipe = env_jp ? atoi(env_jp) : 4;
#else
            const uint32_t reps = 1u, jthreads = 1024u;
#endif
            const uint32_t chunk = std::min<uint32_t>(t->join_chunk, r->jq.bpad);  // what the queues were sized for
            for (uint32_t b0 = 0; b0 < n_batches; b0 += chunk) {
                const uint32_t nb = std::min(chunk, n_batches - b0);
                const uint32_t ng64 = (nb + 63) / 64;
                hipLaunchKernelGGL((k_join_bin<20>), dim3(nb), dim3(512), 0, t->stream, cv, (const uint64_t *)r->words,
                                   (const TileDesc *)r->tile_desc, r->n_tiles, b0, r->jq);
                hipLaunchKernelGGL(k_join_segT, dim3(ng64, kJoinSlices / 64), dim3(256), 0, t->stream, r->jq, nb);
#define GS_LAUNCH_JOIN_PROBE(PIPE, HF) GS_LAUNCH_JOIN_PROBE_DBG(PIPE, HF, 0)
#define GS_LAUNCH_JOIN_PROBE_DBG(PIPE, HF, DBG)                                                                                \
    hipLaunchKernelGGL((k_join_probe<20, PIPE, HF, DBG>), dim3(kJoinSlices * reps), dim3(jthreads), 0, t->stream, cv, r->jq, nb, reps, \
                       (const uint64_t *)t->j_ovf_code, (const uint32_t *)t->j_ovf_loc, (const uint32_t *)t->j_ovf_start)
                if (!t->c_has_flag) {
                    GS_LAUNCH_JOIN_PROBE(4, false);
                } else {
#ifdef GS_TUNING
                    const char *jdbg = getenv("GAVISUNK_DEBUG_MODE");  // phases switched off for timing: RESULTS ARE WRONG
                    const int jmode = jdbg ? atoi(jdbg) : 0;
                    if (jmode == 11)
                        GS_LAUNCH_JOIN_PROBE_DBG(4, true, 1);
                    else if (jmode == 12)
                        GS_LAUNCH_JOIN_PROBE_DBG(4, true, 2);
                    else if (jmode == 13)
                        GS_LAUNCH_JOIN_PROBE_DBG(4, true, 3);
                    else if (jpipe == 3)
                        GS_LAUNCH_JOIN_PROBE(3, true);
                    else if (jpipe == 5)
                        GS_LAUNCH_JOIN_PROBE(5, true);
                    else
#endif
                        GS_LAUNCH_JOIN_PROBE(4, true);  // 4 steps per window, two windows in flight: measured best (118 VGPRs, no spills)
                }
#undef GS_LAUNCH_JOIN_PROBE
#undef GS_LAUNCH_JOIN_PROBE_DBG
                hipLaunchKernelGGL(k_join_merge, dim3((nb + kJoinMergeBatches - 1) / kJoinMergeBatches), dim3(1024), 0, t->stream,
                                   r->jq, b0, nb, r->n_tiles, r->ta, r->cand, r->cand_cap, r->ps.d_cand_count, r->complex_list,
                                   r->ps.d_complex_n);
            }
            // tiles of degenerate groups (none on ordinary reads): generic exact probe, grid-stride over the list
            hipLaunchKernelGGL(k_probe<true>, dim3(std::min<uint32_t>(blocks, 1024u)), dim3(256), 0, t->stream, view(t), cv,
                               view(r), r->ta, r->cand, r->cand_cap, r->ps.d_cand_count, r->ps.d_err_pos,
                               (const uint32_t *)r->complex_list, (const uint32_t *)r->ps.d_complex_n);
        } else if (t->compact == 1) {
            constexpr int nt = 4;
            const uint32_t pblocks = ((r->n_tiles + nt - 1) / nt + 3) / 4 * kNPart;
#define GS_LAUNCH_PART(KT, NT, PIPE, DBG)                                                                           \
    hipLaunchKernelGGL((k_probe_part<KT, NT, PIPE, DBG>), dim3(pblocks), dim3(256), 0, t->stream, cv, view(t),      \
                       (const uint64_t *)r->words, (const TileDesc *)r->tile_desc, r->n_tiles, r->part_recs)
            if (t->k != 20) {
                GS_LAUNCH_PART(0, 4, 4, 0);
            } else {
#ifdef GS_TUNING
                const char *dbg = getenv("GAVISUNK_DEBUG_MODE");  // phases switched off for timing: RESULTS ARE WRONG
                const int mode = dbg ? atoi(dbg) : 0;
                const char *env_pipe = getenv("GAVISUNK_PART_PIPE");
                const int pipe = env_pipe ? atoi(env_pipe) : 2;
                if (mode == 1)
                    GS_LAUNCH_PART(20, 4, 4, 1);
                else if (mode == 2)
                    GS_LAUNCH_PART(20, 4, 4, 2);
                else if (mode == 3)
                    GS_LAUNCH_PART(20, 4, 2, 3);
                else if (mode == 4)
                    GS_LAUNCH_PART(20, 4, 2, 4);
                else if (mode == 5)
                    GS_LAUNCH_PART(20, 4, 2, 1);
                else if (mode == 6)
                    GS_LAUNCH_PART(20, 4, 2, 6);
                else if (pipe == 8)
                    GS_LAUNCH_PART(20, 4, 8, 0);
                else if (pipe == 2)
                    GS_LAUNCH_PART(20, 4, 2, 0);
                else
#endif
                    GS_LAUNCH_PART(20, 4, 4, 0);
            }
#undef GS_LAUNCH_PART
            hipLaunchKernelGGL(k_merge_part, dim3((r->n_tiles + 15) / 16), dim3(256), 0, t->stream, (const PartRec *)r->part_recs,
                               r->n_tiles, r->ta, r->cand, r->cand_cap, r->ps.d_cand_count, r->complex_list, r->ps.d_complex_n);
            // tiles whose partition summaries cannot be merged exactly: generic probe, grid-stride over the list
            hipLaunchKernelGGL(k_probe<true>, dim3(std::min<uint32_t>(blocks, 1024u)), dim3(256), 0, t->stream, view(t), cv,
                               view(r), r->ta, r->cand, r->cand_cap, r->ps.d_cand_count, r->ps.d_err_pos,
                               (const uint32_t *)r->complex_list, (const uint32_t *)r->ps.d_complex_n);
        } else if (t->compact == 2) {
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
    if (after)
        hipLaunchKernelGGL(k_chain_carry, dim3(1), dim3(1), 0, t->stream, (const Agg *)after->ps.d_totals, r->ps.d_carry_in);
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
//   1 = "compact-l2":  32-bit slots, a 1/16 slice fits the L2 of an XCD next to the read stream (<= 2 MiB) -> k_probe_part
//   2 = "compact-hbm": 32-bit slots, larger; one 16-byte bucket load per window                              -> k_probe<true>
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
        const bool l2 = (16ull << qb) <= (2ull << 20) && locbits <= 18 && !(env && !strcmp(env, "compact-hbm"));
        return l2 ? 1 : 2;
    }
    // 64-bit slots with the natural bucket count
    qb = qb_nat;
    rembits = kbits - qb;
    if (!valid_qb(qb) || rembits + 1 + locbits > 62) return 0;
    has_flag = 1;
    return 3;
}

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
    // slot fits 64 bits): since round 2 its kernel is the fastest on every table size -- cache-resident tables included (config
    // 2: 188 Gbase/s against 175 of the LDS join).  Otherwise, and when forced ("compact-lds": what small tables got by default
    // before), an exact compact copy probed from LDS / the L2s, or one bucket load per window from HBM ("compact-hbm").  The
    // generic 16-byte-slot table holds what those could not take, or everything.
    const char *env = getenv("GAVISUNK_TABLE");
    if (env && !strcmp(env, "auto")) env = nullptr;
    const bool force_mz = env && (!strcmp(env, "minimizer") || !strcmp(env, "minimizer-exact"));
    const bool old_auto = env && !strcmp(env, "compact-lds");  // the compact layouts as they were chosen before
    if ((force_mz || !env) && mz_choose(k, n, maxloc, any_missing, t->mz_m, t->mz_n_regions, t->mz_locbits)) {
        t->mz = force_mz && !strcmp(env, "minimizer-exact") ? 2 : 1;
        t->compact = 0;
    } else {
        const char *cenv = force_mz || old_auto ? nullptr : env;
        t->compact = choose_compact(k, n, cap_full * sizeof(Slot), maxloc, any_missing, cenv, t->c_qb, t->c_rembits, t->c_locbits, t->c_has_flag);
        // a small-table layout forced on a table it does not apply to: the default layout
        if ((t->compact == 2 || t->compact == 3) && (old_auto || (env && !strcmp(env, "compact-l2"))) &&
            mz_choose(k, n, maxloc, any_missing, t->mz_m, t->mz_n_regions, t->mz_locbits)) {
            t->mz = 1;
            t->compact = 0;
        }
    }
    uint64_t n_generic = n;
    // A minimiser length whose regions cannot hold the keys (more than an eighth of them overflow: clustered minimisers, a
    // repeat-rich assembly) is no reason to fail: the next length is tried (shorter runs, emptier regions), and when none
    // does, the layouts below take the table -- only a FORCED layout reports the misfit.
    while (t->mz) {
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
#if GS_MZ_RD
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
#else
        {  // keys sorted by region, stable (rocPRIM radix sort on the 32 region bits; a key's index rides along)
            DevBufs st;
            uint64_t *reg0 = nullptr, *reg1 = nullptr;
            uint32_t *idx0 = nullptr, *idx1 = nullptr;
            if ((rc = st.get(&reg0, n)) || (rc = st.get(&reg1, n)) || (rc = st.get(&idx0, n)) || (rc = st.get(&idx1, n))) return rc;
            const unsigned nb = (unsigned)((n + 255) / 256);
            hipLaunchKernelGGL(k_mz_regions, dim3(nb), dim3(256), 0, t->stream, dk, n, k, t->mz_m, t->mz_n_regions, reg0);
            hipLaunchKernelGGL(k_iota, dim3(nb), dim3(256), 0, t->stream, idx0, n);
            if ((rc = sort_pairs_u64_u32(reg0, reg1, idx0, idx1, n, 32, t->stream))) return rc;
            hipLaunchKernelGGL(k_mz_place, dim3(nb), dim3(256), 0, t->stream, reinterpret_cast<unsigned long long *>(t->mz_regions),
                               t->mz_pay, k, t->mz_locbits, dk, dl, dp, (const uint64_t *)reg1, (const uint32_t *)idx1, n, dov, dlist, ocap);
            HIP_TRY(hipStreamSynchronize(t->stream));
        }
#endif
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
        // LDS join: k = 20 makes a slice (partition x top 4 bucket bits) exactly a 32-bit item space
        t->join = t->compact == 1 && k == 20 && t->c_qb >= 5 && t->c_qb - 4 <= 13 && t->c_locbits <= 15 &&
                  !(env && !strcmp(env, "compact-l2"));
        if (t->join) {
            t->join_chunk = join_chunk_from_env();
            // the overflow keys in hashed form (slice << 32 | item), sorted: the probe kernel searches them per slice
            std::vector<uint64_t> ok(ov);
            std::vector<uint32_t> ol(ov);
            if (ov) {
                uint64_t *d_ok = nullptr;
                uint32_t *d_ol = nullptr;
                if ((rc = tmp.get(&d_ok, ov)) || (rc = tmp.get(&d_ol, ov))) return rc;
                hipLaunchKernelGGL(k_gather_list, dim3((unsigned)((ov + 255) / 256)), dim3(256), 0, t->stream, dk, dl, dlist, ov, d_ok, d_ol);
                HIP_TRY(hipMemcpyAsync(ok.data(), d_ok, ov * sizeof(uint64_t), hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipMemcpyAsync(ol.data(), d_ol, ov * sizeof(uint32_t), hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipStreamSynchronize(t->stream));
            }
            std::vector<std::pair<uint64_t, uint32_t>> codes(ov);
            const int kbits = 2 * k - 4, sb = t->c_qb - 4;
            for (uint64_t i = 0; i < ov; i++) {
                const uint64_t c = ok[i];
                const BucketRem br = hmix2((c >> 4) & ((1ULL << kbits) - 1ULL), kbits, t->c_qb);
                const uint64_t slice = ((uint64_t)part_of(c, k) << 4) | (br.bucket >> sb);
                const uint64_t item = ((uint64_t)(br.bucket & ((1u << sb) - 1u)) << t->c_rembits) | br.rem;
                codes[i] = {(slice << 32) | item, ol[i]};
            }
            std::sort(codes.begin(), codes.end());
            std::vector<uint64_t> hc(ov);
            std::vector<uint32_t> hl(ov), hs(kJoinSlices + 1, 0);
            for (uint64_t i = 0; i < ov; i++) {
                hc[i] = codes[i].first;
                hl[i] = codes[i].second;
                hs[(codes[i].first >> 32) + 1]++;
            }
            for (int i = 0; i < kJoinSlices; i++) hs[i + 1] += hs[i];  // (a slice with more than kJoinOvfMax keys searches them in HBM)
            if ((rc = dev_alloc(&t->j_ovf_code, ov)) || (rc = dev_alloc(&t->j_ovf_loc, ov)) ||
                (rc = dev_alloc(&t->j_ovf_start, kJoinSlices + 1)))
                return rc;
            if (ov) {
                HIP_TRY(hipMemcpy(t->j_ovf_code, hc.data(), ov * sizeof(uint64_t), hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(t->j_ovf_loc, hl.data(), ov * sizeof(uint32_t), hipMemcpyHostToDevice));
            }
            HIP_TRY(hipMemcpy(t->j_ovf_start, hs.data(), hs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
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
    static thread_local char buf[256];
    if (!t) return "";
    if (t->mz)
        snprintf(buf, sizeof buf, "minimizer-hbm: %u regions of 64 B (8 fingerprint bytes + 7 x u64 [%d-bit %s | %d-bit loc]), mod-minimiser of length %d (w = %d, t = %d), %llu overflow keys in a generic 16 B-slot side table",
                 t->mz_n_regions, mz_raw(t->k, t->mz_locbits) ? 2 * t->k : 2 * t->k - 3, mz_raw(t->k, t->mz_locbits) ? "key" : "tag", t->mz_locbits,
                 t->mz_m, t->k - t->mz_m + 1, mz_t_of(t->k, t->mz_m), (unsigned long long)t->mz_overflow);
    else if (t->compact)
        snprintf(buf, sizeof buf, "%s: 16 partitions x 2^%d buckets x 4 x %s (%d-bit remainder, %d-bit loc%s), %llu overflow keys in a generic 16 B-slot side table",
                 t->compact == 1 ? (t->join ? "compact-lds" : "compact-l2") : "compact-hbm", t->c_qb, t->c_w64 ? "u64" : "u32", t->c_rembits, t->c_locbits,
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
    (void)hipFree(t->j_ovf_code);
    (void)hipFree(t->j_ovf_loc);
    (void)hipFree(t->j_ovf_start);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    if (t->copy_stream) (void)hipStreamDestroy(t->copy_stream);
    delete t;
}

uint64_t gs_table_bytes(const gs_table *t) {
    if (!t) return 0;
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

// every device array of a batch, for up to cap_words packed words, cap_reads reads, cap_tiles tiles
static int reads_alloc(int device, int k, uint64_t cap_words, uint32_t cap_reads, uint64_t cap_tiles, uint64_t cand_rows, gs_reads **out) {
    HIP_TRY(hipSetDevice(device));
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
    if ((rc = ensure_out_buffers(r, (uint32_t)std::min<uint64_t>(cand_rows, 0xFFFFFFFFull)))) return rc;
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
static int reads_from_ascii(int device, const uint8_t *seq, bool seq_on_device, const uint64_t *off, uint32_t n_reads, int k,
                            gs_reads **out, const char *who) {
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
    uint64_t *d_boff = nullptr;
    const uint64_t nbytes = off[n_reads];
    if (!seq_on_device && (rc = dev_alloc(&d_seq, nbytes))) return rc;
    if ((rc = dev_alloc(&d_boff, (uint64_t)n_reads + 1))) {
        (void)hipFree(d_seq);
        return rc;
    }
    hipError_t e = hipSuccess;
    if (!seq_on_device && nbytes) e = hipMemcpy(d_seq, seq, nbytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_boff, off, ((uint64_t)n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, 0, seq_on_device ? seq : d_seq, d_boff,
                           guard->read_word_off, n_reads, n_words, guard->words);
        e = hipDeviceSynchronize();
    }
    (void)hipFree(d_seq);
    (void)hipFree(d_boff);
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

void gs_reads_destroy(gs_reads *r) {
    if (!r) return;
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
    (void)hipFree(r->part_recs);
    (void)hipFree(r->jq.item);
    (void)hipFree(r->jq.woff);
    (void)hipFree(r->jq.seg);
    (void)hipFree(r->jq.segT);
    (void)hipFree(r->jq.hits);
    (void)hipFree(r->jq.hsub);
    (void)hipFree(r->jq.gflag);
    (void)hipFree(r->jq.late);
    (void)hipFree(r->jq.nlate);
    (void)hipFree(r->complex_list);
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

int gs_tag_phase1(gs_table *t, gs_reads *r, uint32_t *has_raw, uint32_t *last_loc) {
    int rc;
    if ((rc = check_pair(t, r, "gs_tag_phase1"))) return rc;
    if ((rc = launch_phase1(t, r))) return rc;
    // summary = scan A with no carry-in; cheap, and phase 2 redoes it with the true carry
    r->ps.last_carry_in = GS_NONE;
    HIP_TRY(hipMemcpyAsync(r->ps.d_carry_in, &r->ps.last_carry_in, sizeof(uint32_t), hipMemcpyHostToDevice, t->stream));
    if ((rc = launch_scan(t, r, 0))) return rc;
    Agg tot;
    HIP_TRY(hipMemcpyAsync(&tot, r->ps.d_totals, sizeof tot, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(wait_stream(t->stream));
    if (has_raw) *has_raw = tot.has & 1u;
    if (last_loc) *last_loc = (tot.has & 1u) ? tot.loc : GS_NONE;
    return GS_OK;
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

// Parallel gzip inflation for the read files rule SUNK_annot is given (`.fq.gz`, workflow/rules/tagONT.smk:17,36): the one part of
// ingest that stays on the host, moved off the single zlib thread.
//
// DEFLATE has no index, so a second thread cannot know where a block starts nor what the 32 KiB before it hold.  Both are dealt with
// the way pugz / rapidgzip do it:
//   find    every worker but the first scans its piece of the compressed file bit by bit for a dynamic-Huffman block header that
//           parses (complete code-length, literal/length and distance codes) and whose whole block decodes;
//   decode  from there into 16-bit symbols: a byte, or -- for a back-reference that reaches into the unknown 32 KiB window before
//           the worker's start -- a marker 256 + (position in that window);
//   chain   worker i stops at the first block boundary that IS a later worker's start (exactly: same bit): that start is then a
//           proven boundary.  A start that an earlier worker walks past without landing on it was a false positive: its worker's
//           output is dropped and the earlier worker keeps going.  Only outputs reachable from the stream's true start are used;
//   resolve the last 32 KiB of every used piece, in order (each needs only the window before it), then all markers in parallel
//           while the symbols are narrowed to bytes; CRC-32 and ISIZE of every gzip member are checked (pieces in parallel,
//           combined), so a corrupt file fails like it does under zlib.
// Anything unusual (a decode error on the proven path, a bad trailer, a failed check) fails the read: gs_gz_read returns GS_E_IO
// (GS_E_NOMEM when memory runs out) where zlib would report a damaged file; the command line maps it to the reference's "cannot
// open the file" exit (kmerpos_annot3.py).  There is no second reader behind this one.
#include <zlib.h>

#include <algorithm>
#include <system_error>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <chrono>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "gs_common.h"

using namespace gs;

namespace {

constexpr uint32_t kWindow = 32768;
constexpr uint16_t kMarker = 256;  // symbol 256 + j = byte j of the unknown window

// ---- bit reader (LSB first), bounds-checked: reading past the end yields zeros and sets `over` -------------------------------
struct Bits {
    const uint8_t *p = nullptr;
    size_t n = 0;      // bytes
    size_t byte = 0;   // next byte to load
    uint64_t buf = 0;
    int cnt = 0;
    bool over = false;
    void seek(uint64_t bitpos) {
        byte = (size_t)(bitpos >> 3);
        buf = 0;
        cnt = 0;
        over = false;
        refill();
        const int sh = (int)(bitpos & 7);
        buf >>= sh;
        cnt -= sh;
    }
    inline void refill() {
        if (byte + 8 <= n) {  // (the common case: one unaligned load)
            uint64_t w;
            memcpy(&w, p + byte, 8);
            const int take = (63 - cnt) >> 3;  // whole bytes that fit
            if (take < 8) w &= (take == 0 ? 0ULL : (~0ULL >> (64 - 8 * take)));
            buf |= w << cnt;
            byte += (size_t)take;
            cnt += 8 * take;
            return;
        }
        while (cnt <= 56) {
            uint64_t b = 0;
            if (byte < n) b = p[byte];
            else if (byte > n + 8) over = true;
            byte++;
            buf |= b << cnt;
            cnt += 8;
        }
    }
    inline uint32_t peek(int k) const { return (uint32_t)(buf & ((1ULL << k) - 1ULL)); }
    inline void drop(int k) {
        buf >>= k;
        cnt -= k;
    }
    inline uint32_t take(int k) {
        if (cnt < k) refill();
        const uint32_t v = peek(k);
        drop(k);
        return v;
    }
    uint64_t bitpos() const { return (uint64_t)byte * 8 - (uint64_t)cnt; }
    bool past_end() const { return bitpos() > (uint64_t)n * 8; }
};

// ---- Huffman decoding tables: a primary table of 2^P entries, longer codes through second-level tables ------------------------
// entry: bits 0..4 code length (bits to drop; 0 = invalid), bit 5 = "points to a subtable" (then bits 8..11 = its index bits),
// bits 16..31 symbol or subtable offset
struct Huff {
    std::vector<uint32_t> t;
    int primary = 0;
    // -> false when the lengths are no complete prefix code (or over-subscribed).  `allow_single`: a code with exactly one symbol
    // of length 1 is accepted (DEFLATE allows that for the distance code) as is one with no symbol at all
    bool build(const uint8_t *len, int n, int P, bool allow_single) {
        int count[16] = {0};
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        int used = 0, maxlen = 0;
        for (int l = 1; l <= 15; l++)
            if (count[l]) {
                used += count[l];
                maxlen = l;
            }
        primary = P;
        if (used == 0) {
            if (!allow_single) return false;
            t.assign((size_t)1 << P, 0u);
            return true;
        }
        long left = 1;
        for (int l = 1; l <= 15; l++) {
            left <<= 1;
            left -= count[l];
            if (left < 0) return false;  // over-subscribed
        }
        if (left > 0 && !(allow_single && used == 1 && maxlen == 1)) return false;  // incomplete
        int next[16];
        next[1] = 0;
        {
            int code = 0;
            for (int l = 1; l <= 15; l++) {
                code = (code + count[l - 1]) << 1;
                next[l] = code;
            }
        }
        // subtable sizes: for every primary prefix of a long code, the longest code under it decides
        size_t total = (size_t)1 << P;
        std::vector<uint8_t> subbits;
        if (maxlen > P) subbits.assign((size_t)1 << P, 0);
        std::vector<uint16_t> codes((size_t)n);
        for (int i = 0; i < n; i++)
            if (len[i]) {
                const int l = len[i];
                uint32_t c = (uint32_t)next[l]++;
                uint32_t r = 0;  // bit-reversed: DEFLATE packs Huffman codes MSB first into an LSB-first stream
                for (int b = 0; b < l; b++) r |= ((c >> b) & 1u) << (l - 1 - b);
                codes[(size_t)i] = (uint16_t)r;
                if (l > P) {
                    const uint32_t pre = r & (((uint32_t)1 << P) - 1u);
                    subbits[pre] = std::max<uint8_t>(subbits[pre], (uint8_t)(l - P));
                }
            }
        std::vector<uint32_t> suboff;
        if (maxlen > P) {
            suboff.assign((size_t)1 << P, 0);
            for (size_t pre = 0; pre < ((size_t)1 << P); pre++)
                if (subbits[pre]) {
                    suboff[pre] = (uint32_t)total;
                    total += (size_t)1 << subbits[pre];
                }
        }
        t.assign(total, 0u);
        for (int i = 0; i < n; i++) {
            const int l = len[i];
            if (!l) continue;
            const uint32_t r = codes[(size_t)i];
            if (l <= P) {
                const uint32_t e = ((uint32_t)i << 16) | (uint32_t)l;
                for (uint32_t x = r; x < ((uint32_t)1 << P); x += (uint32_t)1 << l) t[x] = e;
            } else {
                const uint32_t pre = r & (((uint32_t)1 << P) - 1u);
                const int sb = subbits[pre];
                t[pre] = (suboff[pre] << 16) | ((uint32_t)sb << 8) | 32u | (uint32_t)P;
                const uint32_t hi = r >> P;
                const uint32_t e = ((uint32_t)i << 16) | (uint32_t)(l - P);
                for (uint32_t x = hi; x < ((uint32_t)1 << sb); x += (uint32_t)1 << (l - P)) t[suboff[pre] + x] = e;
            }
        }
        if (used == 1 && maxlen == 1 && allow_single) {  // the one-symbol distance code: the other bit value stays invalid
        }
        return true;
    }
    // -> symbol, or -1 (invalid code)
    inline int decode(Bits &b) const {
        if (b.cnt < 32) b.refill();
        uint32_t e = t[b.peek(primary)];
        if (e & 32u) {
            b.drop(primary);
            const int sb = (int)((e >> 8) & 15u);
            e = t[(e >> 16) + b.peek(sb)];
        }
        const int l = (int)(e & 31u);
        if (!l) return -1;
        b.drop(l);
        return (int)(e >> 16);
    }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct BlockCodes {
    Huff lit, dist;
    // up to three LITERALS per lookup (sequence text: A C G T take 2 .. 3 bits each, so the primary table's 11 bits usually hold
    // three of them): entry = three 16-bit symbols | bits to drop << 48 | count << 56; count 0 = take the one-symbol path.
    // Empty when the block's literal codes are too long for it to pay (build_multi).
    std::vector<uint64_t> multi;
    // the two tables as the fast loop reads them: the symbol already turned into what it means --
    //   bits 0..4 code length, bit 5 "subtable" (bits 8..11 its index bits, bits 16.. its offset), else bits 6..7 kind (0 literal,
    //   1 length / distance, 2 end of block), bits 8..11 number of extra bits, bits 16..31 literal / base length / base distance
    // -- so that a match costs two dependent table loads instead of four (kLenBase / kLenExtra / kDistBase / kDistExtra)
    std::vector<uint32_t> flit, fdist;
};

// A prefix of a stream's next P bits decodes the same whatever follows it, so the second and third literal of an entry can be
// looked up in the same table with the bits that are left, zero-padded: a code longer than those bits cannot come back as a
// shorter one (the code is prefix-free).
void build_multi(BlockCodes &bc) {
    bc.multi.clear();
    const int P = bc.lit.primary;
    if (bc.lit.t.size() < ((size_t)1 << P)) return;
    const uint32_t *lt = bc.lit.t.data();
    // worth it when two literals fit the table's bits more often than not: at least two literal codes of <= 4 bits
    int short_lits = 0;
    for (uint32_t i = 0; i < 16u && short_lits < 2; i++) {
        const uint32_t e = lt[i];
        if (!(e & 32u) && (e & 31u) && (e & 31u) <= 4u && (e >> 16) < 256u && (i >> (e & 31u)) == 0u) short_lits++;
    }
    if (short_lits < 2) return;
    bc.multi.resize((size_t)1 << P);
    uint64_t *mt = bc.multi.data();
    for (uint32_t i = 0; i < ((uint32_t)1 << P); i++) {
        uint64_t m = 0;
        const uint32_t e1 = lt[i];
        const int l1 = (int)(e1 & 31u);
        if (!(e1 & 32u) && l1 && (e1 >> 16) < 256u) {
            uint32_t rest = i >> l1;
            int avail = P - l1;
            const uint32_t e2 = lt[rest];
            const int l2 = (int)(e2 & 31u);
            if (!(e2 & 32u) && l2 && l2 <= avail && (e2 >> 16) < 256u) {
                uint64_t syms = (uint64_t)(e1 >> 16) | ((uint64_t)(e2 >> 16) << 16);
                int cnt = 2, tot = l1 + l2;
                rest >>= l2;
                avail -= l2;
                const uint32_t e3 = lt[rest];
                const int l3 = (int)(e3 & 31u);
                if (!(e3 & 32u) && l3 && l3 <= avail && (e3 >> 16) < 256u) {
                    syms |= (uint64_t)(e3 >> 16) << 32;
                    cnt = 3;
                    tot += l3;
                }
                m = syms | ((uint64_t)tot << 48) | ((uint64_t)cnt << 56);
            }
        }
        mt[i] = m;
    }
}

void make_fast(BlockCodes &bc) {
    const size_t P1 = (size_t)1 << bc.lit.primary, P2 = (size_t)1 << bc.dist.primary;
    bc.flit.resize(bc.lit.t.size());
    for (size_t i = 0; i < bc.lit.t.size(); i++) {
        const uint32_t e = bc.lit.t[i];
        uint32_t f = 0;
        if (i < P1 && (e & 32u)) {
            f = e;  // pointer to a subtable: as it is
        } else if (e & 31u) {
            const uint32_t s = e >> 16, l = e & 31u;
            if (s < 256u) f = (s << 16) | l;
            else if (s == 256u) f = (2u << 6) | l;
            else if (s < 286u) f = ((uint32_t)kLenBase[s - 257] << 16) | ((uint32_t)kLenExtra[s - 257] << 8) | (1u << 6) | l;
            // (286, 287: no valid length symbol -> 0 = invalid)
        }
        bc.flit[i] = f;
    }
    bc.fdist.resize(bc.dist.t.size());
    for (size_t i = 0; i < bc.dist.t.size(); i++) {
        const uint32_t e = bc.dist.t[i];
        uint32_t f = 0;
        if (i < P2 && (e & 32u)) {
            f = e;
        } else if (e & 31u) {
            const uint32_t s = e >> 16, l = e & 31u;
            if (s < 30u) f = ((uint32_t)kDistBase[s] << 16) | ((uint32_t)kDistExtra[s] << 8) | (1u << 6) | l;
        }
        bc.fdist[i] = f;
    }
}

// The symbols of a Huffman block with the reader's state in registers: one branch-free refill per symbol (after it at least 56
// bits are at hand: a literal/length code, its extra bits, a distance code and its extra bits are at most 48), no capacity test
// per symbol (the caller guarantees room for `budget` symbols and 8 readable bytes at `in_safe`), matches copied four symbols
// at a time.  -> 1 end of block, 0 stop (input or output margin reached: the careful loop goes on), -1 invalid data.
inline int fast_symbols(const BlockCodes &bc, Bits &b, uint16_t *&o, uint16_t *o_stop) {
    const uint8_t *base = b.p, *in = base + b.byte;
    const uint8_t *in_safe = b.n >= 16 ? base + b.n - 16 : base;
    uint64_t buf = b.buf;
    int cnt = b.cnt;
    const uint32_t *lt = bc.flit.data(), *dt = bc.fdist.data();
    const int lp = bc.lit.primary, dp = bc.dist.primary;
    const uint32_t lmask = (1u << lp) - 1u, dmask = (1u << dp) - 1u;
    uint16_t *w = o;
    int rc = 0;
    const uint64_t *mt = bc.multi.empty() ? nullptr : bc.multi.data();
    while (in < in_safe && w < o_stop) {
        uint64_t x;
        memcpy(&x, in, 8);
        buf |= x << cnt;
        in += (63 - cnt) >> 3;
        cnt |= 56;
        if (mt) {
            // two or three literals per lookup, up to four lookups from the >= 56 bits at hand (11 bits each at most); the store
            // writes four symbols, the fourth is overwritten by what comes next (the output margin holds it)
            uint64_t m = mt[(uint32_t)buf & lmask];
            if (m >> 56) {
                int k = 0;
                do {
                    memcpy(w, &m, 8);
                    w += m >> 56;
                    const int l = (int)((m >> 48) & 255u);
                    buf >>= l;
                    cnt -= l;
                    m = mt[(uint32_t)buf & lmask];
                } while ((m >> 56) && ++k < 4);
                continue;
            }
        }
        uint32_t e = lt[(uint32_t)buf & lmask];
        if (e & 32u) {
            buf >>= lp;
            cnt -= lp;
            e = lt[(e >> 16) + ((uint32_t)buf & ((1u << ((e >> 8) & 15u)) - 1u))];
        }
        const int l = (int)(e & 31u);
        if (!l) {
            rc = -1;
            break;
        }
        const uint32_t kind = (e >> 6) & 3u;
        if (kind == 0) {
            buf >>= l;
            cnt -= l;
            *w++ = (uint16_t)(e >> 16);
            // up to two more literals from the bits already at hand (three codes are at most 45 of the >= 56 bits)
            uint32_t e2 = lt[(uint32_t)buf & lmask];
            if (!(e2 & (32u | 192u)) && (e2 & 31u)) {
                buf >>= (e2 & 31u);
                cnt -= (int)(e2 & 31u);
                *w++ = (uint16_t)(e2 >> 16);
                e2 = lt[(uint32_t)buf & lmask];
                if (!(e2 & (32u | 192u)) && (e2 & 31u)) {
                    buf >>= (e2 & 31u);
                    cnt -= (int)(e2 & 31u);
                    *w++ = (uint16_t)(e2 >> 16);
                }
            }
            continue;
        }
        if (kind == 2) {
            buf >>= l;
            cnt -= l;
            rc = 1;
            break;
        }
        // a match: base length + extra bits, then the distance the same way (at most 15 + 5 + 15 + 13 = 48 of the >= 56 bits)
        const int xl = (int)((e >> 8) & 15u);
        const uint32_t len = (e >> 16) + ((uint32_t)(buf >> l) & ((1u << xl) - 1u));
        buf >>= l + xl;
        cnt -= l + xl;
        uint32_t d = dt[(uint32_t)buf & dmask];
        if (d & 32u) {
            buf >>= dp;
            cnt -= dp;
            d = dt[(d >> 16) + ((uint32_t)buf & ((1u << ((d >> 8) & 15u)) - 1u))];
        }
        const int dl = (int)(d & 31u);
        if (!dl) {
            rc = -1;
            break;
        }
        const int xd = (int)((d >> 8) & 15u);
        const uint32_t dist = (d >> 16) + ((uint32_t)(buf >> dl) & ((1u << xd) - 1u));
        buf >>= dl + xd;
        cnt -= dl + xd;
        const uint16_t *r = w - dist;  // (behind the 32768 markers a distance never reaches before the buffer)
        if (dist >= 4) {
            for (uint32_t i = 0; i < len; i += 4) memcpy(w + i, r + i, 8);  // (up to 3 symbols of slack past the match: the margin holds them)
        } else if (dist == 3) {
            // a period of three symbols: four of the pattern per store, three of them kept (the same slack)
            const uint64_t v = (uint64_t)r[0] | ((uint64_t)r[1] << 16) | ((uint64_t)r[2] << 32) | ((uint64_t)r[0] << 48);
            for (uint32_t i = 0; i < len; i += 3) memcpy(w + i, &v, 8);
        } else {
            // runs and pairs (the quality line of `seqtk seq -F '#'` is one symbol repeated: half of a read file's text comes
            // through here, 258 symbols to a match): the pattern from a register, four symbols per store -- symbol by symbol
            // every load waited for the store before it
            const uint64_t a = r[0], b = dist == 1 ? a : (uint64_t)r[1];
            const uint64_t v = a | (b << 16) | (a << 32) | (b << 48);
            for (uint32_t i = 0; i < len; i += 4) memcpy(w + i, &v, 8);
        }
        w += len;
    }
    o = w;
    // hand the state back in the careful reader's form (no stray bits above cnt)
    b.byte = (size_t)(in - base);
    b.cnt = cnt;
    b.buf = cnt >= 64 ? buf : (buf & ((1ULL << cnt) - 1ULL));
    return rc;
}

// the header of a dynamic block (after its 3 header bits) -> codes; false = not a valid header
bool read_dynamic_header(Bits &b, BlockCodes &bc) {
    b.refill();
    const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; i++) cl[kClOrder[i]] = (uint8_t)b.take(3);
    Huff clh;
    if (!clh.build(cl, 19, 7, false)) return false;
    uint8_t lens[286 + 30 + 138];
    int i = 0;
    const int total = hlit + hdist;
    while (i < total) {
        const int s = clh.decode(b);
        if (s < 0) return false;
        if (s < 16) {
            lens[i++] = (uint8_t)s;
        } else {
            int rep, v = 0;
            if (s == 16) {
                if (i == 0) return false;
                v = lens[i - 1];
                rep = 3 + (int)b.take(2);
            } else if (s == 17) {
                rep = 3 + (int)b.take(3);
            } else {
                rep = 11 + (int)b.take(7);
            }
            if (i + rep > total) return false;
            while (rep--) lens[i++] = (uint8_t)v;
        }
        if (b.over) return false;
    }
    if (!lens[256]) return false;  // a block without an end-of-block code never ends
    if (!bc.lit.build(lens, hlit, 11, false)) return false;
    if (!bc.dist.build(lens + hlit, hdist, 8, true)) return false;
    make_fast(bc);
    return true;
}

const BlockCodes &fixed_codes() {
    static const BlockCodes bc = [] {
        BlockCodes c;
        uint8_t l[288];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        c.lit.build(l, 288, 11, false);
        uint8_t d[32];
        for (int i = 0; i < 32; i++) d[i] = 5;
        c.dist.build(d, 32, 8, true);  // (32 codes of 5 bits; 30 and 31 never occur in valid data and are refused by the decoder)
        make_fast(c);
        return c;
    }();
    return bc;
}

// uninitialised, reusable storage (a std::vector would zero every byte before the decoder writes it, and give the pages back
// between groups: first-touch page faults from many threads at once are what a sandboxed kernel serialises)
template <typename T>
struct RawVec {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    RawVec() = default;
    RawVec(const RawVec &) = delete;
    RawVec &operator=(const RawVec &) = delete;
    RawVec(RawVec &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    ~RawVec() { free(p); }
    void reserve(size_t want) {
        if (want <= cap) return;
        const size_t bytes = want * sizeof(T);
        if (bytes >= ((size_t)8 << 20)) {  // (2 MiB-aligned, transparent huge pages asked for: far fewer first-touch faults)
            void *q = nullptr;
            if (posix_memalign(&q, (size_t)2 << 20, (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1)) != 0 || !q) throw std::bad_alloc();
            (void)madvise(q, bytes, MADV_HUGEPAGE);
            if (n) memcpy(q, p, n * sizeof(T));
            free(p);
            p = static_cast<T *>(q);
        } else {
            T *q = static_cast<T *>(realloc(p, bytes));
            if (!q) throw std::bad_alloc();
            p = q;
        }
        cap = want;
    }
    void grow_for(size_t extra) {
        if (n + extra > cap) reserve(std::max(cap * 2, n + extra + 4096));
    }
    void resize(size_t m) {
        reserve(m);
        n = m;
    }
    size_t size() const { return n; }
    T *data() { return p; }
    const T *data() const { return p; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

// ---- one worker's output: 16-bit symbols behind a window of 32768 markers -----------------------------------------------------
struct Piece {
    RawVec<uint16_t> sym;        // [0, kWindow): the unknown window (markers); [kWindow, ..): the decoded symbols
    uint64_t start_bit = 0;      // where decoding began (a block boundary)
    uint64_t end_bit = 0;        // where it stopped (a block boundary, or the end of the last member)
    bool found = false, ok = false, reached_eof = false, budget_stop = false;
    uint32_t landed_on = 0;      // index of the later piece whose start this one stopped at (0 = none: ran to the end)
    struct MemberEnd {
        uint64_t out_pos;  // symbols of this piece before the member's end
        uint32_t crc, isize;
    };
    std::vector<MemberEnd> members;
    std::string err;
    Piece() = default;
    Piece(Piece &&) noexcept = default;
    size_t n_out() const { return sym.size() - kWindow; }
};

// gzip member header at byte `pos` -> first byte of the deflate stream, or 0 on a malformed header
size_t skip_gzip_header(const uint8_t *p, size_t n, size_t pos) {
    if (pos + 10 > n || p[pos] != 0x1f || p[pos + 1] != 0x8b || p[pos + 2] != 8) return 0;
    const uint8_t flg = p[pos + 3];
    if (flg & 0xE0) return 0;  // (reserved bits: zero in every gzip file, RFC 1952 2.3.1 -- and one more test of a member start found by search)
    size_t q = pos + 10;
    if (flg & 4) {  // FEXTRA
        if (q + 2 > n) return 0;
        q += 2 + ((size_t)p[q] | ((size_t)p[q + 1] << 8));
    }
    if (flg & 8) {  // FNAME
        while (q < n && p[q]) q++;
        q++;
    }
    if (flg & 16) {  // FCOMMENT
        while (q < n && p[q]) q++;
        q++;
    }
    if (flg & 2) q += 2;  // FHCRC
    return q <= n ? q : 0;
}

bool g_prof = getenv("GAVISUNK_GZ_PROF") != nullptr;
std::atomic<uint64_t> g_prof_cycles{0}, g_prof_blocks{0};

// Decodes blocks from b's position.  `stops`: sorted bit positions of later pieces' starts: decoding ends at the first block
// boundary that equals one of them (-> its index + 1 in *landed) and walks past those it misses.  Also ends at the end of
// the file's last member.  `one_block`: decode exactly one block (the finder's trial).
bool decode_blocks(const uint8_t *data, size_t n, Bits &b, Piece &pc, const std::vector<uint64_t> *stops, size_t first_stop, bool one_block,
                   size_t max_symbols) {
    RawVec<uint16_t> &out = pc.sym;
    size_t si = first_stop;
    BlockCodes dyn;
    bool member_start = false;  // the position is the first block of a member whose header this loop has just read
    for (;;) {
        const uint64_t here = b.bitpos();
        if (stops) {
            while (si < stops->size() && (*stops)[si] < here) si++;  // a start this piece walked past: not a boundary after all
            if (si < stops->size() && (*stops)[si] == here && here != pc.start_bit) {
                pc.landed_on = (uint32_t)(si + 1);
                pc.end_bit = here;
                return true;
            }
        }
        if (member_start && !one_block && out.size() - kWindow > max_symbols) {
            // a member has ended and the next one's header has been read: as proven a boundary as a block start on the path --
            // and in a file of many small members (bgzip: one final block per 64 KiB member) the only kind there is
            pc.end_bit = here;
            pc.budget_stop = true;
            return true;
        }
        member_start = false;
        b.refill();
        const uint32_t bfinal = b.take(1), btype = b.take(2);
        if (btype == 3) {
            pc.err = "invalid block type";
            return false;
        }
        if (btype == 0) {  // stored: to the byte boundary, LEN, ~LEN, bytes
            const uint64_t bp = b.bitpos();
            size_t q = (size_t)((bp + 7) >> 3);
            if (q + 4 > n) {
                pc.err = "truncated stored block";
                return false;
            }
            const uint32_t len = (uint32_t)data[q] | ((uint32_t)data[q + 1] << 8), nlen = (uint32_t)data[q + 2] | ((uint32_t)data[q + 3] << 8);
            if ((len ^ 0xFFFFu) != nlen || q + 4 + len > n) {
                pc.err = "bad stored block";
                return false;
            }
            const size_t o = out.size();
            out.grow_for(len);
            out.n = o + len;
            for (uint32_t i = 0; i < len; i++) out[o + i] = data[q + 4 + i];
            b.seek((uint64_t)(q + 4 + len) * 8);
        } else {
            const BlockCodes *bc;
            if (btype == 1) {
                bc = &fixed_codes();
            } else {
                const uint64_t tsc0 = g_prof ? __builtin_ia32_rdtsc() : 0;
                if (!read_dynamic_header(b, dyn)) {
                    pc.err = "bad dynamic block header";
                    return false;
                }
                if (!one_block) build_multi(dyn);  // (the finder's trial blocks do without)
                else dyn.multi.clear();
                if (g_prof) {
                    g_prof_cycles.fetch_add(__builtin_ia32_rdtsc() - tsc0, std::memory_order_relaxed);
                    g_prof_blocks.fetch_add(1, std::memory_order_relaxed);
                }
                bc = &dyn;
            }
            for (;;) {
                if (b.over) {  // (past the end the reader yields zeros, which may well be a literal's code: a cut file would never end)
                    pc.err = "truncated stream";
                    return false;
                }
                if (out.n + 300 > out.cap) out.grow_for(300);
                if (out.cap - out.n > 1024 && b.byte + 32 < b.n && !bc->dist.t.empty()) {
                    // the fast loop, while there is room for a few hundred symbols and input to spare
                    uint16_t *o = out.p + out.n;
                    const int rc = fast_symbols(*bc, b, o, out.p + out.cap - 300);
                    out.n = (size_t)(o - out.p);
                    if (rc < 0) {
                        pc.err = "invalid code";
                        return false;
                    }
                    if (rc == 1) break;
                    if (out.n + 300 > out.cap) out.grow_for(out.cap);  // (the output margin stopped it: double the buffer)
                    if (b.byte + 32 < b.n) continue;
                }
                int s = bc->lit.decode(b);
                if (s < 256) {
                    if (s < 0) {
                        pc.err = "invalid literal/length code";
                        return false;
                    }
                    out.p[out.n++] = (uint16_t)s;
                    continue;
                }
                if (s == 256) break;
                s -= 257;
                if (s >= 29) {
                    pc.err = "invalid length symbol";
                    return false;
                }
                if (b.cnt < 32) b.refill();
                const uint32_t len = kLenBase[s] + b.take(kLenExtra[s]);
                const int ds = bc->dist.decode(b);
                if (ds < 0 || ds >= 30) {
                    pc.err = "invalid distance code";
                    return false;
                }
                if (b.cnt < 32) b.refill();
                const uint32_t dist = kDistBase[ds] + b.take(kDistExtra[ds]);
                const size_t o = out.size();
                if (dist > o) {  // (cannot happen behind the 32768 markers unless the piece's window was cut: a corrupt stream)
                    pc.err = "distance too far back";
                    return false;
                }
                out.n = o + len;  // (room for 300 symbols was made above; a match is at most 258)
                uint16_t *w = out.p + o;
                const uint16_t *r = w - dist;
                for (uint32_t i = 0; i < len; i++) w[i] = r[i];
                if (b.over) {
                    pc.err = "truncated stream";
                    return false;
                }
            }
            if (b.past_end()) {
                pc.err = "truncated stream";
                return false;
            }
        }
        if (one_block) {
            if (out.size() - kWindow > max_symbols) return false;
            pc.end_bit = b.bitpos();
            return true;
        }
        if (!bfinal && out.size() - kWindow > max_symbols) {
            // enough for one piece (a stretch that compresses extremely well, or no later start in sight): any block boundary on
            // the proven path is a good place to end a group
            pc.end_bit = b.bitpos();
            pc.budget_stop = true;
            return true;
        }
        if (bfinal) {  // the member's trailer, then the next member or the end of the file
            size_t q = (size_t)((b.bitpos() + 7) >> 3);
            if (q + 8 > n) {
                pc.err = "truncated gzip trailer";
                return false;
            }
            Piece::MemberEnd me;
            me.out_pos = out.size() - kWindow;
            me.crc = (uint32_t)data[q] | ((uint32_t)data[q + 1] << 8) | ((uint32_t)data[q + 2] << 16) | ((uint32_t)data[q + 3] << 24);
            me.isize = (uint32_t)data[q + 4] | ((uint32_t)data[q + 5] << 8) | ((uint32_t)data[q + 6] << 16) | ((uint32_t)data[q + 7] << 24);
            pc.members.push_back(me);
            q += 8;
            while (q < n && data[q] == 0) q++;  // (zero padding after the last member is tolerated, as gzip does)
            if (q >= n) {
                pc.end_bit = (uint64_t)n * 8;
                pc.reached_eof = true;
                return true;
            }
            const size_t d0 = skip_gzip_header(data, n, q);
            if (!d0) {
                pc.err = "trailing garbage after a gzip member";
                return false;
            }
            b.seek((uint64_t)d0 * 8);
            member_start = true;
        }
    }
}

// the first bit position in [from, to) where a piece can start: a non-final dynamic block that decodes as a whole, or the first
// block of a gzip MEMBER (magic, a well-formed header, a block of any kind that decodes: files of many members -- bgzip writes one
// final block per 64 KiB member -- have no non-final blocks to find).  Either is a candidate only: a start counts when the piece
// before it lands on it.
bool find_block(const uint8_t *data, size_t n, uint64_t from, uint64_t to, uint64_t *found) {
    Bits b;
    b.p = data;
    b.n = n;
    BlockCodes bc;
    Piece trial;
    auto decodes = [&](uint64_t pos) {  // the whole block at `pos` (into a scratch piece: markers and all)
        trial.sym.reserve(kWindow + (1u << 20));
        trial.sym.resize(kWindow);
        memset(trial.sym.data(), 0, kWindow * sizeof(uint16_t));
        trial.members.clear();
        trial.start_bit = pos;
        trial.err.clear();
        b.seek(pos);
        if (!decode_blocks(data, n, b, trial, nullptr, 0, true, (size_t)1 << 24)) return false;
        return trial.n_out() >= 64;  // (a real block of a big file is not this small; a false positive usually is)
    };
    for (uint64_t pos = from; pos < to; pos++) {
        const size_t by = (size_t)(pos >> 3);
        if (by + 4 > n) return false;
        if (!(pos & 7) && data[by] == 0x1f && data[by + 1] == 0x8b && data[by + 2] == 8) {  // a member?
            const size_t d0 = skip_gzip_header(data, n, by);
            if (d0 && (uint64_t)d0 * 8 < to && decodes((uint64_t)d0 * 8)) {
                *found = (uint64_t)d0 * 8;
                return true;
            }
        }
        // quick reject on the first 13 bits: BFINAL = 0, BTYPE = 2 (bits: 0, then 0 1), HLIT <= 29, HDIST <= 29
        uint32_t w;
        memcpy(&w, data + by, 4);
        w >>= (pos & 7);
        if ((w & 7u) != 4u) continue;            // bit0 = BFINAL = 0, bits 1..2 = BTYPE = 2 -> value 0b100
        if (((w >> 3) & 31u) > 29u) continue;    // HLIT
        if (((w >> 8) & 31u) > 29u) continue;    // HDIST
        b.seek(pos + 3);
        if (!read_dynamic_header(b, bc)) continue;
        if (!decodes(pos)) continue;
        *found = pos;
        return true;
    }
    return false;
}

// CRC-32 of p[0, n) continued from `c` (0 = a fresh one)
uint32_t crc32_more(uint32_t c, const uint8_t *p, size_t n) {
    static uint32_t (*fast)(uint32_t, const void *, size_t) = [] {
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        return h ? (uint32_t(*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32") : nullptr;
    }();
    if (fast) return fast(c, p, n);
    while (n) {
        const uInt m = (uInt)std::min<size_t>(n, 1u << 30);
        c = (uint32_t)crc32(c, p, m);
        p += m;
        n -= m;
    }
    return c;
}

}  // namespace

namespace {
// The text buffers handed to the caller come back through gs_gz_free_buffer: a few of them are kept for the next read instead
// of going back to the allocator (a 1 GiB block returned to the system is a quarter of a million page faults when the next one
// is written, taken by many threads at once).
struct BufPool {
    std::mutex m;
    struct E {
        uint8_t *p;
        size_t cap;
    };
    std::vector<E> idle;
    std::vector<E> live;
    uint8_t *get(size_t want, size_t *cap) {
        {
            std::lock_guard<std::mutex> lk(m);
            size_t best = idle.size();
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].cap >= want && (best == idle.size() || idle[i].cap < idle[best].cap)) best = i;
            if (best < idle.size()) {
                E e = idle[best];
                idle.erase(idle.begin() + (long)best);
                live.push_back(e);
                *cap = e.cap;
                return e.p;
            }
        }
        // large buffers on 2 MiB boundaries with transparent huge pages asked for: 512 x fewer first-touch faults
        const size_t sz = std::max<size_t>(want, 64);
        void *q = nullptr;
        if (sz >= ((size_t)8 << 20)) {
            if (posix_memalign(&q, (size_t)2 << 20, (sz + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1)) != 0) q = nullptr;
            if (q) (void)madvise(q, sz, MADV_HUGEPAGE);
        } else {
            q = malloc(sz);
        }
        if (!q) throw std::bad_alloc();
        uint8_t *p = static_cast<uint8_t *>(q);
        std::lock_guard<std::mutex> lk(m);
        live.push_back(E{p, sz});
        *cap = sz;
        return p;
    }
    void put(uint8_t *p) {
        if (!p) return;
        std::lock_guard<std::mutex> lk(m);
        for (size_t i = 0; i < live.size(); i++)
            if (live[i].p == p) {
                E e = live[i];
                live.erase(live.begin() + (long)i);
                if (idle.size() < 6 && e.cap >= ((size_t)16 << 20)) idle.push_back(e);
                else free(e.p);
                return;
            }
        free(p);  // (not one of ours)
    }
};
BufPool &pool() {
    static BufPool *p = new BufPool;  // (never destroyed: buffers may be returned during interpreter shutdown)
    return *p;
}
struct OutBuf {
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    ~OutBuf() { pool().put(p); }
    void reserve(size_t want) {
        if (want <= cap) return;
        size_t c = 0;
        uint8_t *q = pool().get(want, &c);
        if (n) memcpy(q, p, n);
        pool().put(p);
        p = q;
        cap = c;
    }
    void grow_for(size_t extra) {
        if (n + extra > cap) reserve(std::max(cap + cap / 2, n + extra + 4096));
    }
    size_t size() const { return n; }
    uint8_t *data() { return p; }
};
}  // namespace

struct gs_gzfile {
    int fd = -1;
    const uint8_t *data = nullptr;
    size_t n = 0;
    int threads = 1;
    uint64_t next_bit = 0;          // where the next group starts (a proven block boundary), or the end
    bool at_end = false, failed = false;
    std::vector<uint8_t> window;    // the 32 KiB before next_bit (resolved bytes)
    // the member that is open at next_bit: crc / size so far
    uint32_t run_crc = 0;
    uint64_t run_size = 0;
    OutBuf out;                     // the text being made for the caller
    std::vector<Piece> pool;        // the workers' symbol buffers, reused from group to group
    std::string mode, err;
    uint64_t pieces_used = 0, pieces_dropped = 0;
    uint64_t bytes_in = 0, bytes_out = 0;  // compressed bytes consumed / text produced so far (the expansion ratio)
    uint64_t newlines = 0;                 // '\n' bytes among the NEW bytes of the stretch made last (counted while they are written)
};

namespace {
// `fn` on up to `want` threads beside the calling one (fn pulls its work from a shared counter, so any number of threads, the
// calling thread alone included, finishes it).  A thread that cannot be created -- std::system_error under a pids / ulimit cap --
// is simply not there; the ones that were are always joined.
template <typename F>
void run_on_threads(int want, F &fn) {
    std::vector<std::thread> th;
    try {
        for (int t = 0; t < want; t++) th.emplace_back(fn);
    } catch (const std::system_error &) {
    }
    fn();
    for (auto &t : th) t.join();
}
}  // namespace

namespace {

// one group: up to `want` pieces of `piece_bytes` compressed bytes each, decoded in parallel, appended to z->out
bool inflate_group(gs_gzfile *z, size_t piece_bytes, size_t want_pieces) {
    const uint8_t *data = z->data;
    const size_t n = z->n;
    const uint64_t start = z->next_bit;
    const size_t first_byte = (size_t)(start >> 3);
    const size_t left_pieces = std::max<size_t>((n - first_byte + piece_bytes - 1) / piece_bytes, 1);
    const bool last_group = left_pieces <= want_pieces;
    // (a group that is not the file's last gets one more START behind its last piece: where that piece stops, and the next group begins)
    size_t np = last_group ? left_pieces : want_pieces + 1;
    if (z->pool.size() < np) z->pool.resize(np);
    std::vector<Piece> &pcs = z->pool;
    for (size_t i = 0; i < np; i++) {
        Piece &pc = pcs[i];
        pc.found = pc.ok = pc.reached_eof = pc.budget_stop = false;
        pc.landed_on = 0;
        pc.start_bit = pc.end_bit = 0;
        pc.members.clear();
        pc.err.clear();
        pc.sym.n = 0;
    }
    // (an exception must not leave a worker thread: it is reported by the calling thread after the join)
    std::atomic<bool> oom{false};
    auto guarded = [&oom](auto &fn) {
        return [&oom, &fn]() {
            try {
                fn();
            } catch (...) {
                oom.store(true);
            }
        };
    };
    const bool dbg = getenv("GAVISUNK_GZ_DEBUG") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    // ---- find
    pcs[0].found = true;
    pcs[0].start_bit = start;
    {
        std::atomic<size_t> nexti{1};
        auto work = [&]() {
            for (;;) {
                const size_t i = nexti.fetch_add(1);
                if (i >= np) return;
                const uint64_t from = (uint64_t)(first_byte + i * piece_bytes) * 8, to = std::min<uint64_t>(from + (uint64_t)piece_bytes * 8, (uint64_t)n * 8);
                uint64_t f = 0;
                if (from < to && find_block(data, n, from, to, &f)) {
                    pcs[i].found = true;
                    pcs[i].start_bit = f;
                }
            }
        };
        auto gw = guarded(work);
        run_on_threads((int)std::min<size_t>((size_t)std::max(z->threads - 1, 0), np > 1 ? np - 2 : 0), gw);
    }
    if (oom.load()) throw std::bad_alloc();
    if (dbg) fprintf(stderr, "[gz] find %.3f s\n", now() - t0), t0 = now();
    std::vector<uint64_t> stops;      // starts of the found pieces after the first, ascending
    std::vector<uint32_t> stop_piece;
    for (size_t i = 1; i < np; i++)
        if (pcs[i].found) {
            stops.push_back(pcs[i].start_bit);
            stop_piece.push_back((uint32_t)i);
        }
    // ---- decode
    {
        std::atomic<size_t> nexti{0};
        auto work = [&]() {
            for (;;) {
                const size_t i = nexti.fetch_add(1);
                if (i >= np) return;
                Piece &pc = pcs[i];
                if (!pc.found) continue;
                if (!last_group && i + 1 == np) continue;  // (the extra start: a stop for the piece before it, decoded by the next group)
                pc.sym.reserve(kWindow + piece_bytes * 8);
                pc.sym.resize(kWindow);
                for (uint32_t j = 0; j < kWindow; j++) pc.sym[j] = (uint16_t)(kMarker + j);
                Bits b;
                b.p = data;
                b.n = n;
                b.seek(pc.start_bit);
                const size_t fs = (size_t)(std::upper_bound(stops.begin(), stops.end(), pc.start_bit) - stops.begin());
                // a piece that is no later piece's predecessor must not run to the end of the file unless this is the last group
                pc.ok = decode_blocks(data, n, b, pc, &stops, fs, false, std::max<size_t>(piece_bytes * 32, (size_t)1 << 26));
            }
        };
        auto gw = guarded(work);
        run_on_threads((int)std::min<size_t>((size_t)std::max(z->threads - 1, 0), np ? np - 1 : 0), gw);
    }
    if (oom.load()) throw std::bad_alloc();
    if (dbg) fprintf(stderr, "[gz] decode %.3f s\n", now() - t0), t0 = now();
    if (g_prof) fprintf(stderr, "[gz] block headers + tables: %llu blocks, %.1f kcycles each so far\n", (unsigned long long)g_prof_blocks.load(),
                        g_prof_blocks.load() ? (double)g_prof_cycles.load() / (double)g_prof_blocks.load() / 1e3 : 0.0);
    // ---- chain: from piece 0 along the landings; in a group that is not the file's last, the chain must end ON a later piece's
    // start (the next group begins there); pieces after the last landing are decoded again by the next group
    std::vector<size_t> chain;
    size_t i = 0;
    for (;;) {
        Piece &pc = pcs[i];
        if (!pc.ok) {
            z->err = pc.err.empty() ? "decode failed" : pc.err;
            return false;
        }
        chain.push_back(i);
        if (pc.landed_on) {
            i = stop_piece[pc.landed_on - 1];
            if (!last_group && i + 1 == np) break;  // the next group starts at the extra start
            continue;
        }
        if (!pc.reached_eof && !pc.budget_stop) {
            z->err = "a piece ended without reaching a block boundary";
            return false;
        }
        break;
    }
    z->pieces_used += chain.size();
    z->pieces_dropped += np - chain.size();
    // ---- resolve the windows in order, then narrow everything in parallel
    std::vector<std::vector<uint8_t>> win(chain.size() + 1);
    win[0] = z->window;
    win[0].resize(kWindow, 0);
    std::vector<size_t> out_off(chain.size() + 1, 0);
    for (size_t c = 0; c < chain.size(); c++) {
        const Piece &pc = pcs[chain[c]];
        const size_t m = pc.n_out();
        out_off[c + 1] = out_off[c] + m;
        std::vector<uint8_t> &w = win[c + 1];
        w.resize(kWindow);
        const std::vector<uint8_t> &pw = win[c];
        for (uint32_t j = 0; j < kWindow; j++) {  // the last 32 KiB before this piece's end: from its own symbols or the window before it
            const size_t idx = pc.sym.size() - kWindow + j;  // (index into [markers | symbols])
            const uint16_t s = pc.sym[idx];
            uint8_t v;
            if (idx < kWindow) v = pw[idx];  // still inside the window before the piece (a piece shorter than 32 KiB)
            else v = s < kMarker ? (uint8_t)s : pw[s - kMarker];
            w[j] = v;
        }
    }
    std::atomic<uint64_t> newlines{0};
    const size_t base = z->out.size();
    z->out.grow_for(out_off[chain.size()]);
    z->out.n = base + out_off[chain.size()];
    // per piece: the CRC-32 of every stretch between member ends (a piece holds the tail of one member, whole members, the head
    // of the next), taken tile by tile right behind the narrowing while the bytes are in the cache
    struct SegCrc {
        size_t len;
        uint32_t crc;
    };
    std::vector<std::vector<SegCrc>> segs(chain.size());
    {
        std::atomic<size_t> nextc{0};
        auto work = [&]() {
            for (;;) {
                const size_t c = nextc.fetch_add(1);
                if (c >= chain.size()) return;
                const Piece &pc = pcs[chain[c]];
                const uint16_t *s = pc.sym.data() + kWindow;
                const size_t m = pc.n_out();
                uint8_t *o = z->out.data() + base + out_off[c];
                const uint8_t *w = win[c].data();
                // a symbol -> its byte by ONE lookup: [0, 256) the byte itself, behind it the window before the piece.  Where the
                // chain of copies is never cut (FASTA, reads shorter than the window: every base is a copy of an earlier one, back
                // to the unknown window) nearly every symbol of a piece is a marker, and the test "byte or marker" per symbol
                // cost more than the lookup
                std::vector<uint8_t> lut((size_t)kMarker + kWindow);
                for (uint32_t j = 0; j < kMarker; j++) lut[j] = (uint8_t)j;
                memcpy(lut.data() + kMarker, w, kWindow);
                const uint8_t *const lu = lut.data();
                uint64_t nl = 0;
                std::vector<SegCrc> &sg = segs[c];
                size_t seg_lo = 0;
                for (size_t mi = 0; mi <= pc.members.size(); mi++) {
                    const size_t seg_hi = mi < pc.members.size() ? (size_t)pc.members[mi].out_pos : m;
                    uint32_t crc = 0;
                    for (size_t ta = seg_lo; ta < seg_hi;) {
                        const size_t t1 = std::min(seg_hi, ta + ((size_t)1 << 17));
                        size_t k = ta;
#if defined(__SSE2__)
                        // sixteen symbols at a time while none of them is a marker (all but the first stretch of a piece): pack to
                        // bytes, count the newlines by compare + mask
                        const __m128i hi = _mm_set1_epi16((short)0xFF00), nlv = _mm_set1_epi8('\n');
                        for (; k + 16 <= t1; k += 16) {
                            const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + k));
                            const __m128i b2 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + k + 8));
                            __m128i bytes;
                            if (_mm_movemask_epi8(_mm_cmpeq_epi16(_mm_and_si128(_mm_or_si128(a, b2), hi), _mm_setzero_si128())) != 0xFFFF) {
                                for (size_t j = k; j < k + 16; j++) o[j] = lu[s[j]];  // a marker among them: sixteen lookups, no test
                                bytes = _mm_loadu_si128(reinterpret_cast<const __m128i *>(o + k));
                            } else {
                                bytes = _mm_packus_epi16(a, b2);
                                _mm_storeu_si128(reinterpret_cast<__m128i *>(o + k), bytes);
                            }
                            nl += (uint64_t)__builtin_popcount((unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(bytes, nlv)));
                        }
#endif
                        for (; k < t1; k++) {
                            const uint8_t byte = lu[s[k]];
                            o[k] = byte;
                            nl += byte == '\n';
                        }
                        crc = crc32_more(crc, o + ta, t1 - ta);
                        ta = t1;
                    }
                    sg.push_back(SegCrc{seg_hi - seg_lo, crc});
                    seg_lo = seg_hi;
                }
                newlines.fetch_add(nl, std::memory_order_relaxed);
            }
        };
        auto gw = guarded(work);
        run_on_threads((int)std::min<size_t>((size_t)std::max(z->threads - 1, 0), chain.size() ? chain.size() - 1 : 0), gw);
    }
    if (oom.load()) throw std::bad_alloc();
    z->newlines += newlines.load();
    if (dbg) fprintf(stderr, "[gz] resolve + narrow + crc %.3f s\n", now() - t0), t0 = now();
    // ---- members that ended inside this group: CRC-32 and ISIZE, combined from the stretches' values behind what earlier groups
    // saw of the open member
    for (size_t c = 0; c < chain.size(); c++) {
        const Piece &pc = pcs[chain[c]];
        for (size_t mi = 0; mi < segs[c].size(); mi++) {
            const SegCrc &sg = segs[c][mi];
            if (sg.len) {
                z->run_crc = z->run_size ? (uint32_t)crc32_combine(z->run_crc, sg.crc, (z_off_t)sg.len) : sg.crc;
                z->run_size += sg.len;
            }
            if (mi < pc.members.size()) {  // a member ends behind this stretch
                const Piece::MemberEnd &me = pc.members[mi];
                const uint32_t crc = z->run_size ? z->run_crc : 0u;
                if (crc != me.crc || (uint32_t)z->run_size != me.isize) {
                    z->err = "gzip member fails its CRC-32 / size check";
                    return false;
                }
                z->run_crc = 0;
                z->run_size = 0;
            }
        }
    }
    if (dbg) fprintf(stderr, "[gz] crc %.3f s\n", now() - t0), t0 = now();
    const Piece &lastp = pcs[chain.back()];
    z->window = win[chain.size()];
    z->next_bit = lastp.end_bit;
    if (lastp.reached_eof) z->at_end = true;
    if (z->next_bit <= start && !z->at_end) {
        z->err = "no progress";
        return false;
    }
    return true;
}

}  // namespace

extern "C" {

uint64_t gs_text_record_cut(const uint8_t *t, uint64_t n);

// Opens a gzip file for parallel inflation with `threads` workers (0: the machine's hardware threads, at most 32).
// GS_E_IO: cannot be opened / mapped, or does not start with a gzip member.
int gs_gz_open(const char *path, int threads, gs_gzfile **out) {
    if (!path || !out) {
        set_error("gs_gz_open: null argument");
        return GS_E_ARG;
    }
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        set_error("cannot open the file %s", path);
        return GS_E_IO;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 18) {
        close(fd);
        set_error("%s: not a gzip file", path);
        return GS_E_IO;
    }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
        close(fd);
        set_error("%s: mmap failed", path);
        return GS_E_IO;
    }
    const uint8_t *p = static_cast<const uint8_t *>(m);
    const size_t d0 = skip_gzip_header(p, (size_t)st.st_size, 0);
    if (!d0) {
        munmap(m, (size_t)st.st_size);
        close(fd);
        set_error("%s: not a gzip file", path);
        return GS_E_IO;
    }
    gs_gzfile *z = new gs_gzfile;
    z->fd = fd;
    z->data = p;
    z->n = (size_t)st.st_size;
    if (threads <= 0) threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    if (const char *e = getenv("GAVISUNK_GZ_THREADS")) threads = std::max(1, atoi(e));
    z->threads = threads;
    z->next_bit = (uint64_t)d0 * 8;
    z->window.assign(kWindow, 0);
    z->mode = "parallel deflate, " + std::to_string(threads) + " threads";
    *out = z;
    return GS_OK;
}

// The next stretch of inflated bytes: `prefix` (prefix_len bytes the caller carried over: the unfinished record at the end of the
// stretch before), then at least `want` new bytes unless the file ends first.  The buffer is handed over: *data belongs to the
// caller (gs_gz_free_buffer).  *n = prefix_len and nothing new means the end of the file.
// GS_E_IO: the stream is damaged, or nothing in it looks like a block start -- inflate it with zlib instead.
int gs_gz_read(gs_gzfile *z, uint64_t want, const uint8_t *prefix, uint64_t prefix_len, uint8_t **data, uint64_t *n) {
    if (!z || !data || !n || (prefix_len && !prefix)) {
        set_error("gs_gz_read: bad argument");
        return GS_E_ARG;
    }
    *data = nullptr;
    *n = 0;
    if (z->failed) {
        set_error("gs_gz_read: %s", z->err.c_str());
        return GS_E_IO;
    }
    try {
        // (virtual space is free until it is touched: room for what this call will plausibly produce, grown when that is short)
        const uint64_t left = (uint64_t)z->n - (z->next_bit >> 3);
        z->out.reserve((size_t)(prefix_len + std::min<uint64_t>(want + ((uint64_t)64 << 20), left * 8 + ((uint64_t)1 << 20))));
        z->out.n = 0;
        z->newlines = 0;
        if (prefix_len) {
            memcpy(z->out.p, prefix, (size_t)prefix_len);
            z->out.n = (size_t)prefix_len;
        }
        size_t piece = (size_t)2 << 20;
        if (const char *e = getenv("GAVISUNK_GZ_PIECE_KB")) piece = (size_t)std::max(16, atoi(e)) << 10;  // (test knob: small groups)
        while (!z->at_end && z->out.size() - prefix_len < want) {
            // enough pieces for every thread, a few more so that short pieces do not leave threads idle -- fewer when the caller
            // asks for little (the first blocks of a file: the sooner they are out, the sooner the copy to the device and the
            // parser start beside the inflation of the rest), judged by the expansion seen so far
            const double ratio = z->bytes_in > ((uint64_t)1 << 20) ? std::max(1.0, (double)z->bytes_out / (double)z->bytes_in) : 4.0;
            const uint64_t missing = want - (z->out.size() - prefix_len);
            size_t pieces = (size_t)((double)missing / ((double)piece * ratio)) + 1;
            pieces = std::max<size_t>(std::min<size_t>(pieces, (size_t)z->threads * 2), std::max(1, z->threads / 2));
            const uint64_t bit0 = z->next_bit;
            const size_t out0 = z->out.size();
            if (!inflate_group(z, piece, pieces)) {
                z->failed = true;
                set_error("gs_gz_read: %s", z->err.c_str());
                return GS_E_IO;
            }
            z->bytes_in += (z->next_bit - bit0) >> 3;
            z->bytes_out += z->out.size() - out0;
        }
    } catch (const std::bad_alloc &) {
        z->failed = true;
        z->err = "out of memory";
        set_error("gs_gz_read: out of memory");
        return GS_E_NOMEM;
    } catch (const std::exception &e) {  // (nothing may cross the C boundary)
        z->failed = true;
        z->err = e.what();
        set_error("gs_gz_read: %s", e.what());
        return GS_E_IO;
    }
    *data = z->out.p;
    *n = z->out.n;
    z->out.p = nullptr;  // handed over
    z->out.n = z->out.cap = 0;
    return GS_OK;
}
void gs_gz_free_buffer(uint8_t *p) { pool().put(p); }
// '\n' bytes among the new bytes (the prefix not counted) of the stretch gs_gz_read returned last
uint64_t gs_gz_newlines(const gs_gzfile *z) { return z ? z->newlines : 0; }
// gs_text_record_cut for a caller that already knows how many '\n' the text holds (the reader counts them while it writes)
uint64_t gs_text_record_cut_counted(const uint8_t *t, uint64_t n, uint64_t n_newlines) {
    if (!t || !n) return 0;
    if (t[0] == '>') return gs_text_record_cut(t, n);  // (a backward search from the end: cheap)
    if (n_newlines < 4) return 0;
    uint64_t cut = n;
    for (uint64_t k = 0; k < n_newlines % 4 + 1; k++) {
        const void *q = cut ? memrchr(t, '\n', (size_t)cut) : nullptr;
        if (!q) return 0;
        cut = (uint64_t)(static_cast<const uint8_t *>(q) - t);
    }
    return cut + 1;
}
int gs_gz_at_end(const gs_gzfile *z) { return z && z->at_end ? 1 : 0; }

// Where a block of FASTA / FASTQ text can be cut so that it holds whole records (what the on-device record parser takes,
// gs_reads_from_text): FASTA (first byte '>'): before the last line that starts with '>'; FASTQ (the regular four-line form):
// after the last complete group of four lines.  0 = the text holds no complete record followed by the start of another.
uint64_t gs_text_record_cut(const uint8_t *t, uint64_t n) {
    if (!t || !n) return 0;
    if (t[0] == '>') {
        for (uint64_t i = n; i-- > 1;)
            if (t[i] == '>' && t[i - 1] == '\n') return i;
        return 0;
    }
    uint64_t nl = 0;
    for (const uint8_t *p = t, *e = t + n; (p = static_cast<const uint8_t *>(memchr(p, '\n', (size_t)(e - p)))) != nullptr; p++) nl++;
    if (nl < 4) return 0;
    uint64_t cut = n;
    for (uint64_t k = 0; k < nl % 4 + 1; k++) {  // back to the newline that ends the last whole four-line record
        const void *q = cut ? memrchr(t, '\n', (size_t)cut) : nullptr;
        if (!q) return 0;
        cut = (uint64_t)(static_cast<const uint8_t *>(q) - t);
    }
    return cut + 1;
}

const char *gs_gz_mode(const gs_gzfile *z) { return z ? z->mode.c_str() : ""; }
uint64_t gs_gz_pieces(const gs_gzfile *z, uint64_t *dropped) {
    if (dropped) *dropped = z ? z->pieces_dropped : 0;
    return z ? z->pieces_used : 0;
}

void gs_gz_close(gs_gzfile *z) {
    if (!z) return;
    if (z->data) munmap((void *)z->data, z->n);
    if (z->fd >= 0) close(z->fd);
    delete z;
}

}  // extern "C"

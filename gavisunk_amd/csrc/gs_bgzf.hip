// DEFLATE on the GPU for BGZF-shaped `.fq.gz`: the read files rule SUNK_annot is given (workflow/rules/tagONT.smk:17,36) come out of
// `seqtk seq -F '#' | rustybam fastq-split`, whose `.gz` writer is bgzip's format -- a gzip file of thousands of MEMBERS, each at
// most 64 KiB of text in (usually) one final block, its compressed size in the header's `BC` extra field and its inflated size in
// the trailer.  Everything a parallel decoder wants is therefore known before a single Huffman code is read: where every member's
// deflate data starts, and where its text goes.  So
//   * the host only walks the headers of a chunk of the file (pread into pinned memory, 18 bytes looked at per member) and hands the
//     compressed bytes to the device: 3 GB cross PCIe instead of the 20 GB of text they hold;
//   * k_bgzf_inflate: ONE LANE PER MEMBER.  A lane builds its block's Huffman tables in its own columns of the wave's LDS (lit/len:
//     8-bit direct table, distances: 7-bit; longer codes by canonical search with its counts in LDS too) and decodes straight to the
//     member's place in the text: literals gathered 8 to a store, short matches loaded when decoded and stored a turn later, runs
//     and short periods from a register pattern (see the kernel for the loop's two parts);
//   * k_bgzf_crc checks every member's CRC-32 (and the decoder its ISIZE and that the block ends where the member does): a damaged
//     file fails as it does under zlib;
//   * the text block -- cut where a record ends, the rest carried to the front of the next one -- goes to the record parser of
//     gs_ingest.hip where it lies: it never exists on the host.
// Files that are not BGZF (no `BC` field in the first member) are refused with GS_E_IRREGULAR at gs_bgzf_open, a member outside
// the container later in the file with the same status by gs_bgzf_next: the caller reads such files with the host reader
// (csrc/gs_inflate.cpp), from the start.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "gs_common.h"

using gs::set_error;

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GS_E_NODEVICE;                                                             \
        }                                                                                     \
    } while (0)

namespace {

typedef uint64_t u64u __attribute__((aligned(1)));
typedef uint32_t u32u __attribute__((aligned(1)));

struct Member {
    uint32_t coff;   // deflate data: offset in the chunk
    uint32_t clen;
    uint32_t ooff;   // text: offset in the block
    uint32_t isize;
    uint32_t crc;
};
enum : uint32_t { kOk = 0, kBadCode = 1, kBadDist = 2, kOverrun = 3, kBadStored = 4, kBadLens = 5, kBadSize = 6, kBadCrc = 7, kBadType = 8 };

#ifndef GS_BGZF_LL_BITS
#define GS_BGZF_LL_BITS 8
#endif
#ifndef GS_BGZF_D_BITS
#define GS_BGZF_D_BITS 7
#endif
#ifndef GS_BGZF_LANES
#define GS_BGZF_LANES 32
#endif
#ifndef GS_BGZF_WAVES_EU
#define GS_BGZF_WAVES_EU 2
#endif
#ifndef GS_BGZF_OCC  // (the CPU stand-in of tools/hip_cpu has no such attribute)
#define GS_BGZF_OCC __attribute__((amdgpu_waves_per_eu(GS_BGZF_WAVES_EU)))
#endif
constexpr int kLLBits = GS_BGZF_LL_BITS, kDBits = GS_BGZF_D_BITS;
// Members per wavefront.  A lane's tables are 832 bytes of LDS, so a CU holds 192 members whatever this is.  Measured on blocks of
// 2.3 - 3.1 GB (level 1): 64 and 32 members to a wave take 28 ms whatever the block's size -- one wave's own chain of instructions
// and waits (rocprofv3 --pmc: ~440 instructions and 3.4 L2 requests per symbol, 57 % of the wave-cycles waiting); 16 to a wave take
// the same for blocks of up to 36 000 members and 45 - 50 ms for 48 000: three waves to a SIMD with a quarter of the lanes live are
// bound by VALU issue.  32: the smaller workgroup (26 KiB of LDS) finds room beside other kernels sooner than one of 52 KiB.
constexpr int kLanes = GS_BGZF_LANES;
constexpr uint16_t kLong = 0xFFFFu, kInvalid = 0xFFFEu;
constexpr int kScratchBytes = 1408;  // per lane: code lengths (320) + two canonical sets (count / first code / first index / running: 4 x 16 x 2 B each) + sorted symbols (288 + 32) x 2 B

struct Canon {  // a lane's row of global scratch, for codes longer than the direct tables' index
    uint16_t *count, *first_code, *first_idx, *next, *sorted;
};

typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));

// The member's bits: `bc` valid bits in `bb` (what lies above them in `bb` is the stream's next bits too, or zero), `ip` = the next
// byte that is not counted in `bc`.
struct BitIn {
    const uint8_t *ip;
    uint64_t bb;
    uint32_t bc;
    __device__ __forceinline__ void refill() {  // to at least 56 bits
        bb |= *reinterpret_cast<const u64u *>(ip) << bc;
        const uint32_t nb = (63u - bc) >> 3;
        ip += nb;
        bc += nb * 8u;
    }
    __device__ __forceinline__ uint32_t take(uint32_t n) {
        const uint32_t v = (uint32_t)bb & ((1u << n) - 1u);
        bb >>= n;
        bc -= n;
        return v;
    }
};

// What the decoder of codes LONGER than a direct table's index needs, in the lane's LDS columns: the first code of length
// tbits + 1 and the number of codes of every length from there to 15 (`cn_lds`, 16 - tbits entries); the symbols of those codes in
// canonical order either in LDS too (`sym_lds`: distances and code-length codes, at most 30) or in the lane's scratch row from
// `*long_base` on (literals / lengths).  (With these in global memory a long code cost a dozen dependent loads, and with 64 members
// to a wavefront three loop turns in four had a lane on that path: 6 us per symbol.)
constexpr int kSymLds = 30;  // (a distance alphabet has 30 symbols, the code-length alphabet 19; with 30 a workgroup's LDS is 26 x 512 bytes)

// table entry: symbol | length << 9
__device__ __forceinline__ uint32_t build_tables(const uint8_t *lens, int nsym, uint16_t *tab /* [1 << tbits][64], this lane's column */, int tbits,
                                                 const Canon &cn, uint16_t *cn_lds, uint8_t *sym_lds, uint32_t *long_base) {
    for (int l = 0; l < 16; l++) cn.count[l] = 0;
    for (int s = 0; s < nsym; s++) cn.count[lens[s]]++;
    cn.count[0] = 0;
    uint32_t code = 0, idx = 0;
    int left = 1;  // Kraft: over-subscribed sets are refused (an incomplete one is legal: a single distance code)
    for (int l = 1; l < 16; l++) {
        left <<= 1;
        left -= (int)cn.count[l];
        if (left < 0) return kBadLens;
        code = (code + cn.count[l - 1]) << 1;
        cn.first_code[l] = (uint16_t)code;
        cn.next[l] = (uint16_t)code;
        cn.first_idx[l] = (uint16_t)idx;
        idx += cn.count[l];
        if (l == tbits + 1) {
            cn_lds[0] = (uint16_t)code;
            *long_base = cn.first_idx[l];
        }
        if (l > tbits) cn_lds[(l - tbits) * kLanes] = cn.count[l];
    }
    for (int j = 0; j < (1 << tbits); j++) tab[j * kLanes] = kInvalid;
    uint16_t offs[16];
    for (int l = 0; l < 16; l++) offs[l] = cn.first_idx[l];
    for (int s = 0; s < nsym; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t c = cn.next[l]++;
        const uint32_t at = offs[l]++;
        cn.sorted[at] = (uint16_t)s;
        const uint32_t r = __brev(c) >> (32 - l);
        if (l <= tbits) {
            for (uint32_t j = r; j < (1u << tbits); j += (1u << l)) tab[j * kLanes] = (uint16_t)(s | (l << 9));
        } else {
            tab[(r & ((1u << tbits) - 1u)) * kLanes] = kLong;
            if (sym_lds) sym_lds[(at - *long_base) * kLanes] = (uint8_t)s;  // (nsym <= kSymLds for these tables)
        }
    }
    return kOk;
}
// a code longer than the direct table's index: the canonical search from length TBITS + 1 on (the codes of one length are
// consecutive, and no shorter code is a prefix: the direct table said so), every length looked at without a branch
template <int TBITS, bool SYM_LDS>
__device__ __forceinline__ int canon_decode(BitIn &b, const uint16_t *cn_lds, const uint8_t *sym_lds, const uint16_t *sorted_long) {
    constexpr int L0 = TBITS + 1;
    uint32_t cnt[16 - L0];
    uint32_t first = cn_lds[0];
#pragma unroll
    for (int i = 0; i < 16 - L0; i++) cnt[i] = cn_lds[(1 + i) * kLanes];
    const uint32_t bits = (uint32_t)b.bb;
    uint32_t code = __brev(bits) >> (32 - L0), idx = 0, hit_len = 0, hit_at = 0;
#pragma unroll
    for (int l = L0; l < 16; l++) {
        const uint32_t d = code - first, c = cnt[l - L0];
        if (d < c && !hit_len) {
            hit_len = (uint32_t)l;
            hit_at = idx + d;
        }
        idx += c;
        first = (first + c) << 1;
        code = (code << 1) | ((bits >> l) & 1u);
    }
    if (!hit_len) return -1;
    b.bb >>= hit_len;
    b.bc -= hit_len;
    return SYM_LDS ? (int)sym_lds[hit_at * kLanes] : (int)sorted_long[hit_at];
}

#ifdef GS_BGZF_PROF  // (experiments: what a lane's time goes to; never in the product library)
__device__ unsigned long long g_prof[8];
#define PROF_ADD(i, v) prof[i] += (v)
#define PROF_T() wall_clock64()
#else
#define PROF_ADD(i, v) ((void)0)
#define PROF_T() 0ull
#endif

// LDS of one workgroup of k_bgzf_inflate (its five arrays)
constexpr size_t kInflateLdsBytes = (size_t)kLanes * (((size_t)(1 << kLLBits) + (1 << kDBits) + (16 - kLLBits) + (16 - kDBits)) * 2 + kSymLds);

#ifndef GS_BGZF_GROUP
#define GS_BGZF_GROUP 2
#endif
constexpr int kGroup = GS_BGZF_GROUP;  // symbols per fast turn (2, 3 and 4 measure the same: 28 - 31 ms per 3.1 GB block; 2 is the least code)

#ifndef GS_BGZF_FAST_TURNS
#define GS_BGZF_FAST_TURNS 8
#endif

// How the loop is laid out, and why.  The members of a wavefront are decoded in lock step, so a loop turn costs the wave every
// instruction ANY of its lanes needs: with one loop that handles whatever symbol comes (long codes, runs, overlapping copies, block
// headers), three turns in four had some lane on a rare path and every turn cost ~1000 instructions (3.8 us) -- for lanes that
// mostly wanted 100.  So the loop has two parts:
//   * FAST turns (GS_BGZF_FAST_TURNS in a row), up to GS_BGZF_GROUP symbols each: a symbol is taken only if all of it is common -- a
//     literal, or a match whose codes are in the direct tables, at most 32 bytes long, not overlapping its own output, not at the
//     very end of the text.  Anything else leaves the lane's state untouched and the lane sits out the remaining fast turns; a
//     match behind the turn's first symbol that is longer than 16 bytes, or reads what this turn has yet to store, only ends the
//     turn for the lane (it is the next turn's first symbol);
//   * one GENERAL step: the next thing of the lane's stream whatever it is (block header and tables, stored block, any symbol).
// A fast turn waits for memory ONCE, at its top: the 32 input bytes it refills from (`P`) and the source bytes of the matches of the
// turn before were asked for a turn earlier; what they produce is stored then, in stream order (`settle`), and no instruction of the
// turn's symbol steps needs memory.
__global__ __launch_bounds__(64) GS_BGZF_OCC void k_bgzf_inflate(const uint8_t *comp, const Member *mem, uint32_t n_members, uint8_t *text, uint8_t *scratch,
                                                     uint32_t *status) {
    __shared__ uint16_t s_ll[1 << kLLBits][kLanes];
    __shared__ uint16_t s_dt[1 << kDBits][kLanes];
    __shared__ uint16_t s_cnl[16 - kLLBits][kLanes], s_cnd[16 - kDBits][kLanes];
    __shared__ uint8_t s_dsym[kSymLds][kLanes];
    const uint32_t lane = threadIdx.x;
    const uint32_t mi = blockIdx.x * kLanes + lane;
    if (lane >= (uint32_t)kLanes || mi >= n_members) return;
    const Member m = mem[mi];
    uint16_t *ll = &s_ll[0][lane], *dt = &s_dt[0][lane], *cnl = &s_cnl[0][lane], *cnd = &s_cnd[0][lane];
    uint8_t *dsym = &s_dsym[0][lane];
    uint32_t long_ll = 0, long_d = 0;
    uint8_t *row = scratch + (size_t)mi * kScratchBytes;
    uint8_t *lens = row;
    uint16_t *w = reinterpret_cast<uint16_t *>(row + 320);
    const Canon cl{w, w + 16, w + 32, w + 48, w + 128}, cd{w + 64, w + 80, w + 96, w + 112, w + 128 + 288};
    const uint16_t *ll_long = cl.sorted;
    BitIn b{comp + m.coff, 0, 0};
    const uint8_t *const iend = comp + m.coff + m.clen;
    uint8_t *op = text + m.ooff;
    uint8_t *const obase = op, *const oend = op + m.isize;
    uint64_t wbuf = 0;
    uint32_t wn = 0, err = kOk;
    // Stores of a fast turn are made LATE, in stream order, at the top of the next turn (or before any other store of the lane):
    // slot j = what the turn's j-th symbol left behind -- eight bytes of gathered literals at `lit_at` and / or a match's source
    // bytes (16; slot 0: 32 if `wide0`), asked for when the match was decoded, for `mat_at`.  Every store of the lane comes after
    // them, so what a store writes behind its own bytes is overwritten by the next one.
    constexpr uint32_t kNoAt = 0xFFFFFFFFu;
    uint32_t lit_at[kGroup], mat_at[kGroup];
    uint64_t lit_buf[kGroup];
    u32x4u mat_a[kGroup], mat_b0 = {0u, 0u, 0u, 0u};
    bool wide0 = false;
#pragma unroll
    for (int j = 0; j < kGroup; j++) {
        lit_at[j] = kNoAt;
        mat_at[j] = kNoAt;
        lit_buf[j] = 0;
        mat_a[j] = mat_b0;
    }
    auto settle = [&]() {
#pragma unroll
        for (int j = 0; j < kGroup; j++) {
            if (lit_at[j] != kNoAt) {
                *reinterpret_cast<u64u *>(obase + lit_at[j]) = lit_buf[j];
                lit_at[j] = kNoAt;
            }
            if (mat_at[j] != kNoAt) {
                *reinterpret_cast<u32x4u *>(obase + mat_at[j]) = mat_a[j];
                if (j == 0 && wide0) *reinterpret_cast<u32x4u *>(obase + mat_at[j] + 16) = mat_b0;
                mat_at[j] = kNoAt;
            }
        }
    };
    auto flush = [&]() {  // the gathered literals out (the bytes behind them in the store are overwritten by what follows)
        if (!wn) return;
        settle();
        if (op + 8 <= oend) {
            *reinterpret_cast<u64u *>(op) = wbuf;
        } else {
            for (uint32_t i = 0; i < wn && op + i < oend; i++) op[i] = (uint8_t)(wbuf >> (8 * i));
        }
        op += wn;
        wbuf = 0;
        wn = 0;
    };
#ifdef GS_BGZF_PROF
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = PROF_T();
#endif
    bool done = false, in_block = false, last = false;
    uint64_t P[4];  // the 32 bytes at b.ip, asked for at the end of the turn (or general step) before
    auto load_pre = [&]() {
        const u32x4u lo = *reinterpret_cast<const u32x4u *>(b.ip), hi = *reinterpret_cast<const u32x4u *>(b.ip + 16);
        P[0] = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
        P[1] = (uint64_t)lo.z | ((uint64_t)lo.w << 32);
        P[2] = (uint64_t)hi.x | ((uint64_t)hi.y << 32);
        P[3] = (uint64_t)hi.z | ((uint64_t)hi.w << 32);
    };
    load_pre();
    // to at least 56 bits from the bytes of P at offset `poff` (<= 21)
    auto refill_from_pre = [&](uint32_t &poff) {
        const uint32_t kk = poff >> 3, sh = (poff & 7u) * 8u;
        const uint64_t lo = kk == 0u ? P[0] : kk == 1u ? P[1] : P[2], hi = kk == 0u ? P[1] : kk == 1u ? P[2] : P[3];
        const uint64_t w8 = sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
        b.bb |= w8 << b.bc;
        const uint32_t nb = (63u - b.bc) >> 3;
        poff += nb;
        b.ip += nb;
        b.bc += nb * 8u;
    };
    while (!done) {
        // ---- fast turns: up to kGroup symbols each, one wait for memory at the top ----
        const unsigned long long t_fast = PROF_T();
        (void)t_fast;
        if (in_block) {
            bool stay = true;
            for (int turn = 0; turn < GS_BGZF_FAST_TURNS && stay; turn++) {
                PROF_ADD(5, 1);
                settle();
                const uint32_t turn_lo = (uint32_t)(op - obase);  // what lies behind it is not in memory before the next turn
                uint32_t poff = 0;                                // bytes of P this turn has taken
                bool more = true;
#pragma unroll
                for (int j = 0; j < kGroup; j++) {
                    if (!more) continue;
                    more = false;
                    {  // at least 56 bits (a symbol takes at most 48)
                        uint64_t w8;
                        if (j == 0) {
                            w8 = P[0];
                        } else {
                            const uint32_t kk = poff >> 3, sh = (poff & 7u) * 8u;
                            const uint64_t lo = kk == 0u ? P[0] : kk == 1u ? P[1] : P[2], hi = kk == 0u ? P[1] : kk == 1u ? P[2] : P[3];
                            w8 = sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
                        }
                        b.bb |= w8 << b.bc;
                        const uint32_t nb = (63u - b.bc) >> 3;
                        poff += nb;
                        b.ip += nb;
                        b.bc += nb * 8u;
                    }
                    // (leave: the general step takes this symbol -- or reports what is wrong with it)
                    if (b.ip > iend + 8) {
                        stay = false;
                        continue;
                    }
                    const uint32_t e = ll[((uint32_t)b.bb & ((1u << kLLBits) - 1u)) * kLanes];
                    if (e >= kInvalid) {
                        stay = false;
                        continue;
                    }
                    const uint32_t l1 = e >> 9, sym = e & 0x1FFu;
                    if (sym < 256u) {
                        if (wn == 7u) {
                            if (op + 8 > oend) {
                                stay = false;
                                continue;
                            }
                            const uint64_t full = wbuf | ((uint64_t)sym << 56);
                            if (j == 0) {
                                *reinterpret_cast<u64u *>(op) = full;  // (nothing of this turn is pending yet)
                            } else {
                                lit_at[j] = (uint32_t)(op - obase);
                                lit_buf[j] = full;
                            }
                            op += 8;
                            wbuf = 0;
                            wn = 0;
                        } else {
                            wbuf |= (uint64_t)sym << (8u * wn);
                            wn++;
                        }
                        b.bb >>= l1;
                        b.bc -= l1;
                        PROF_ADD(3, 1);
                        more = true;
                        continue;
                    }
                    const uint32_t c = sym - 257u;
                    if (c > 28u) {  // (the end of the block, or a code that is not one)
                        stay = false;
                        continue;
                    }
                    uint64_t t = b.bb >> l1;
                    uint32_t used = l1, len;
                    if (c < 8u) {
                        len = 3u + c;
                    } else if (c == 28u) {
                        len = 258u;
                    } else {
                        const uint32_t eb = (c >> 2) - 1u;
                        len = ((4u + (c & 3u)) << eb) + 3u + ((uint32_t)t & ((1u << eb) - 1u));
                        t >>= eb;
                        used += eb;
                    }
                    const uint32_t e2 = dt[((uint32_t)t & ((1u << kDBits) - 1u)) * kLanes];
                    if (e2 >= kInvalid) {
                        stay = false;
                        continue;
                    }
                    const uint32_t l2 = e2 >> 9, ds = e2 & 0x1FFu;
                    if (ds > 29u) {
                        stay = false;
                        continue;
                    }
                    t >>= l2;
                    used += l2;
                    uint32_t dist;
                    if (ds < 4u) {
                        dist = ds + 1u;
                    } else {
                        const uint32_t eb = (ds >> 1) - 1u;
                        dist = ((2u + (ds & 1u)) << eb) + 1u + ((uint32_t)t & ((1u << eb) - 1u));
                        t >>= eb;
                        used += eb;
                    }
                    uint8_t *const at = op + wn;  // (where the match goes once the gathered literals are out)
                    const uint32_t at_off = (uint32_t)(at - obase);
                    if (len > 32u || len > dist || dist > at_off || at + 32 > oend) {
                        stay = false;
                        continue;
                    }
                    if (j > 0 && (len > 16u || at_off - dist + len > turn_lo)) continue;  // (first symbol of the next turn: more room, nothing pending)
                    b.bb = t;
                    b.bc -= used;
                    if (wn) {
                        if (j == 0) {
                            *reinterpret_cast<u64u *>(op) = wbuf;
                        } else {
                            lit_at[j] = (uint32_t)(op - obase);
                            lit_buf[j] = wbuf;
                        }
                        wbuf = 0;
                        wn = 0;
                    }
                    const uint8_t *src = at - dist;
                    mat_a[j] = *reinterpret_cast<const u32x4u *>(src);
                    if (j == 0) {
                        wide0 = len > 16u;
                        if (wide0) mat_b0 = *reinterpret_cast<const u32x4u *>(src + 16);
                    }
                    mat_at[j] = at_off;
                    op = at + len;
                    PROF_ADD(4, 1);
                    more = true;
                }
                load_pre();
            }
        }
        // ---- the general step ----
        PROF_ADD(6, PROF_T() - t_fast);
        PROF_ADD(2, 1);
        uint32_t gpoff = 0;  // bytes of P this step has taken (P: asked for at the end of the last turn or step)
        refill_from_pre(gpoff);
        if (b.ip > iend + 8) {  // (a damaged member cannot keep a lane busy beyond its own data)
            err = kOverrun;
            break;
        }
        if (!in_block) {
            if (last) {
                done = true;
                break;
            }
            const unsigned long long t_hdr = PROF_T();
            (void)t_hdr;
            last = b.take(1) != 0;
            const uint32_t btype = b.take(2);
            if (btype == 0) {  // stored: to the next byte, LEN / NLEN, the bytes
                flush();
                settle();
                b.take(b.bc & 7u);
                b.refill();
                const uint32_t len = b.take(16), nlen = b.take(16);
                if ((len ^ nlen) != 0xFFFFu) {
                    err = kBadStored;
                    break;
                }
                const uint8_t *src = b.ip - (b.bc >> 3);  // (bytes the bit buffer holds go back to the input)
                b.bb = 0;
                b.bc = 0;
                if (src + len > iend || op + len > oend) {
                    err = kOverrun;
                    break;
                }
                for (uint32_t i = 0; i < len; i++) op[i] = src[i];
                op += len;
                b.ip = src + len;
                load_pre();
                continue;
            }
            if (btype == 3) {
                err = kBadType;
                break;
            }
            int nll = 288, nd = 30;
            if (btype == 1) {
                for (int s = 0; s < 144; s++) lens[s] = 8;
                for (int s = 144; s < 256; s++) lens[s] = 9;
                for (int s = 256; s < 280; s++) lens[s] = 7;
                for (int s = 280; s < 288; s++) lens[s] = 8;
                for (int s = 0; s < 30; s++) lens[288 + s] = 5;
            } else {
                b.refill();
                nll = (int)b.take(5) + 257;
                nd = (int)b.take(5) + 1;
                const int ncl = (int)b.take(4) + 4;
                if (nll > 286 || nd > 30) {
                    err = kBadLens;
                    break;
                }
                const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t *cll = lens + 288;  // (the 19 code-length code lengths: in the distance lengths' place for a moment)
                for (int i = 0; i < 19; i++) cll[i] = 0;
                for (int i = 0; i < ncl; i++) {
                    b.refill();
                    cll[order[i]] = (uint8_t)b.take(3);
                }
                if ((err = build_tables(cll, 19, dt, kDBits, cd, cnd, dsym, &long_d)) != kOk) break;
                int i = 0;
                uint8_t tmp[320];
                while (i < nll + nd) {
                    b.refill();
                    if (b.ip > iend + 8) {
                        err = kOverrun;
                        break;
                    }
                    const uint16_t e = dt[((uint32_t)b.bb & ((1u << kDBits) - 1u)) * kLanes];
                    int sym;
                    if (e < kInvalid) {
                        b.take(e >> 9);
                        sym = e & 0x1FF;
                    } else {
                        sym = e == kLong ? canon_decode<kDBits, true>(b, cnd, dsym, nullptr) : -1;
                    }
                    if (sym < 0) {
                        err = kBadCode;
                        break;
                    }
                    if (sym < 16) {
                        tmp[i++] = (uint8_t)sym;
                    } else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (!i) {
                                err = kBadLens;
                                break;
                            }
                            val = tmp[i - 1];
                            rep = 3 + (int)b.take(2);
                        } else if (sym == 17) {
                            rep = 3 + (int)b.take(3);
                        } else {
                            rep = 11 + (int)b.take(7);
                        }
                        if (i + rep > nll + nd) {
                            err = kBadLens;
                            break;
                        }
                        while (rep--) tmp[i++] = (uint8_t)val;
                    }
                }
                if (err != kOk) break;
                if (!tmp[256]) {  // (no end-of-block code)
                    err = kBadLens;
                    break;
                }
                for (int s = 0; s < nll; s++) lens[s] = tmp[s];
                for (int s = nll; s < 288; s++) lens[s] = 0;
                for (int s = 0; s < nd; s++) lens[288 + s] = tmp[nll + s];
                for (int s = nd; s < 30; s++) lens[288 + s] = 0;
                nll = 288;
                nd = 30;
            }
            if ((err = build_tables(lens, nll, ll, kLLBits, cl, cnl, nullptr, &long_ll)) != kOk) break;
            if ((err = build_tables(lens + 288, nd, dt, kDBits, cd, cnd, dsym, &long_d)) != kOk) break;
            ll_long = cl.sorted + long_ll;
            in_block = true;
            PROF_ADD(1, PROF_T() - t_hdr);
            load_pre();
            continue;
        }
        // one symbol, whatever it is
        int sym;
        {
            const uint16_t e = ll[((uint32_t)b.bb & ((1u << kLLBits) - 1u)) * kLanes];
            if (e < kInvalid) {
                b.take(e >> 9);
                sym = e & 0x1FF;
            } else if (e == kLong) {
                sym = canon_decode<kLLBits, false>(b, cnl, nullptr, ll_long);
            } else {
                sym = -1;
            }
        }
        if (sym < 0) {
            err = kBadCode;
            break;
        }
        if (sym < 256) {
            PROF_ADD(3, 1);
            wbuf |= (uint64_t)sym << (8 * wn);
            if (++wn == 8) {
                if (op + 8 > oend) {
                    err = kOverrun;
                    break;
                }
                settle();
                *reinterpret_cast<u64u *>(op) = wbuf;
                op += 8;
                wbuf = 0;
                wn = 0;
            }
        } else if (sym == 256) {
            in_block = false;
        } else {
            const uint32_t c = (uint32_t)sym - 257u;
            if (c > 28u) {
                err = kBadCode;
                break;
            }
            uint32_t len;
            if (c < 8u) len = 3u + c;
            else if (c == 28u) len = 258u;
            else {
                const uint32_t eb = (c >> 2) - 1u;
                len = ((4u + (c & 3u)) << eb) + 3u + b.take(eb);
            }
            refill_from_pre(gpoff);
            int ds;
            {
                const uint16_t e = dt[((uint32_t)b.bb & ((1u << kDBits) - 1u)) * kLanes];
                if (e < kInvalid) {
                    b.take(e >> 9);
                    ds = e & 0x1FF;
                } else if (e == kLong) {
                    ds = canon_decode<kDBits, true>(b, cnd, dsym, nullptr);
                } else {
                    ds = -1;
                }
            }
            if (ds < 0 || ds > 29) {
                err = kBadCode;
                break;
            }
            uint32_t dist;
            if (ds < 4) dist = (uint32_t)ds + 1u;
            else {
                const uint32_t eb = ((uint32_t)ds >> 1) - 1u;
                dist = ((2u + ((uint32_t)ds & 1u)) << eb) + 1u + b.take(eb);
            }
            if (len <= 32u && len <= dist && dist <= (uint32_t)(op + wn - obase) && op + wn + 32 <= oend) {
                // (here because of a long code, not because of the copy: copied late like a fast turn's first match)
                settle();
                if (wn) {
                    *reinterpret_cast<u64u *>(op) = wbuf;
                    op += wn;
                    wbuf = 0;
                    wn = 0;
                }
                const uint8_t *src = op - dist;
                mat_a[0] = *reinterpret_cast<const u32x4u *>(src);
                wide0 = len > 16u;
                if (wide0) mat_b0 = *reinterpret_cast<const u32x4u *>(src + 16);
                mat_at[0] = (uint32_t)(op - obase);
                op += len;
                PROF_ADD(4, 1);
                load_pre();
                continue;
            }
            flush();
            settle();
            if (dist > (uint32_t)(op - obase)) {
                err = kBadDist;
                break;
            }
            if (op + len > oend) {
                err = kOverrun;
                break;
            }
            const uint8_t *src = op - dist;
            uint8_t *const mend = op + len;
            PROF_ADD(4, 1);
            if (dist >= 16u) {
                if (dist >= 64u) {  // (four loads before their stores: one wait for 64 bytes)
                    while (op + 64 <= mend) {
                        const u32x4u c0 = *reinterpret_cast<const u32x4u *>(src), c1 = *reinterpret_cast<const u32x4u *>(src + 16);
                        const u32x4u c2 = *reinterpret_cast<const u32x4u *>(src + 32), c3 = *reinterpret_cast<const u32x4u *>(src + 48);
                        *reinterpret_cast<u32x4u *>(op) = c0;
                        *reinterpret_cast<u32x4u *>(op + 16) = c1;
                        *reinterpret_cast<u32x4u *>(op + 32) = c2;
                        *reinterpret_cast<u32x4u *>(op + 48) = c3;
                        op += 64;
                        src += 64;
                    }
                }
                while (op + 16 <= mend) {
                    *reinterpret_cast<u32x4u *>(op) = *reinterpret_cast<const u32x4u *>(src);
                    op += 16;
                    src += 16;
                }
                if (op < mend) {
                    if (op + 16 <= oend) {
                        *reinterpret_cast<u32x4u *>(op) = *reinterpret_cast<const u32x4u *>(src);  // (what lies behind the match is written next)
                    } else {
                        for (; op < mend; op++, src++) *op = *src;
                    }
                    op = mend;
                }
            } else if (dist >= 8u) {
                while (op + 8 <= mend) {
                    *reinterpret_cast<u64u *>(op) = *reinterpret_cast<const u64u *>(src);
                    op += 8;
                    src += 8;
                }
                if (op < mend) {
                    if (op + 8 <= oend) {
                        *reinterpret_cast<u64u *>(op) = *reinterpret_cast<const u64u *>(src);
                    } else {
                        for (; op < mend; op++, src++) *op = *src;
                    }
                    op = mend;
                }
            } else {
                // a period of 1 .. 7 bytes (the quality line of `seqtk seq -F '#'`, homopolymers, short tandem repeats): the period
                // from its own bytes, repeated to fill a register, stored in steps of the largest multiple of the period that fits
                // -- no load waits for a store of this match
                uint64_t v;
                if (src + 8 <= oend) {  // the period's bytes by one load (what it brings behind them is masked away)
                    v = *reinterpret_cast<const u64u *>(src) & (~0ULL >> (64u - 8u * dist));
                } else {
                    v = 0;
                    for (uint32_t i = 0; i < dist; i++) v |= (uint64_t)src[i] << (8u * i);
                }
                for (uint32_t sft = dist; sft < 8u; sft <<= 1) v |= v << (8u * sft);
                if ((8u % dist) == 0u) {  // periods 1, 2, 4: sixteen bytes to a store (a quality line's 258-byte run: 17 stores)
                    const uint32_t lo32 = (uint32_t)v, hi32 = (uint32_t)(v >> 32);
                    const u32x4u v16 = {lo32, hi32, lo32, hi32};
                    while (op + 16 <= oend && op < mend) {
                        *reinterpret_cast<u32x4u *>(op) = v16;
                        op += 16;
                    }
                }
                const uint32_t step = (8u / dist) * dist;
                while (op + 8 <= oend && op < mend) {
                    *reinterpret_cast<u64u *>(op) = v;
                    op += step;
                }
                for (uint32_t i = 0; op < mend; op++, i++) *op = (uint8_t)(v >> (8u * (i % dist)));  // (the last bytes of the text)
                op = mend;
            }
        }
        load_pre();
    }
    settle();
    if (err == kOk) {
        flush();
        if (op != oend) err = kBadSize;                                     // ISIZE
        else if (b.ip - (b.bc >> 3) > iend) err = kOverrun;                 // the last block ended behind the member's data
    }
    if (err != kOk) atomicCAS(status, 0u, (mi << 4) | err);
#ifdef GS_BGZF_PROF
    prof[0] = PROF_T() - t_begin;
    for (int i = 0; i < 8; i++) atomicAdd(&g_prof[i], prof[i]);
#endif
}

// CRC-32 of every member's text against its trailer (a lane per member; eight bytes per load, the table in LDS)
__global__ __launch_bounds__(64) void k_bgzf_crc(const uint8_t *text, const Member *mem, uint32_t n_members, uint32_t *status) {
    __shared__ uint32_t tab[256];
    for (uint32_t i = threadIdx.x; i < 256u; i += 64u) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        tab[i] = c;
    }
    __syncthreads();
    const uint32_t mi = blockIdx.x * 64u + threadIdx.x;
    if (mi >= n_members) return;
    const Member m = mem[mi];
    const uint8_t *p = text + m.ooff, *e = p + m.isize;
    uint32_t c = 0xFFFFFFFFu;
    for (; p + 8 <= e; p += 8) {
        uint64_t v = *reinterpret_cast<const u64u *>(p);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            c = tab[(c ^ (uint32_t)v) & 0xFFu] ^ (c >> 8);
            v >>= 8;
        }
    }
    for (; p < e; p++) c = tab[(c ^ *p) & 0xFFu] ^ (c >> 8);
    if ((c ^ 0xFFFFFFFFu) != m.crc) atomicCAS(status, 0u, (mi << 4) | kBadCrc);
}

#ifndef GS_BGZF_NL_BLOCKS
#define GS_BGZF_NL_BLOCKS 2048  // (a grid-stride kernel: the CPU stand-in of tools/hip_cpu runs it with a handful of blocks)
#endif
__global__ void k_count_newlines(const uint8_t *t, uint64_t n, unsigned long long *out) {
    uint32_t mine = 0;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (uint64_t)gridDim.x * blockDim.x * 8) {
        if (i + 8 <= n) {
            uint64_t v = *reinterpret_cast<const u64u *>(t + i) ^ 0x0A0A0A0A0A0A0A0AULL;
            // bytes that are zero: (v - 0x01..) & ~v & 0x80..   (exact per byte only without borrows: count them one by one instead)
#pragma unroll
            for (int k = 0; k < 8; k++) mine += ((v >> (8 * k)) & 0xFFu) == 0u ? 1u : 0u;
        } else {
            for (uint64_t j = i; j < n; j++) mine += t[j] == '\n' ? 1u : 0u;
        }
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor((int)mine, o, 64);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(out, (unsigned long long)mine);
}

__global__ void k_gather_names(const uint8_t *text, const uint64_t *name_off, const uint32_t *name_len, const uint64_t *blob_off, uint32_t n, uint8_t *blob) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint8_t *s = text + name_off[r];
    uint8_t *d = blob + blob_off[r];
    for (uint32_t i = 0; i < name_len[r]; i++) d[i] = s[i];
}

const char *err_text(uint32_t e) {
    switch (e) {
        case kBadCode: return "invalid Huffman code";
        case kBadDist: return "a match reaches before the member's start";
        case kOverrun: return "the data runs past the member's end";
        case kBadStored: return "a stored block's length check fails";
        case kBadLens: return "invalid code lengths";
        case kBadSize: return "a member's text is not as long as its trailer says";
        case kBadCrc: return "a member fails its CRC-32 check";
        default: return "invalid block type";
    }
}

constexpr size_t kTailPinned = (size_t)1 << 20;
constexpr size_t kChunk = (size_t)64 << 20;  // compressed bytes read, indexed and handed over at a time (at most)

// members that lie completely in buf[0, n) and whose text fits below text_end: their descriptors (offsets into buf / into the text
// from `text_at`), bytes of buf used
int walk_members(const uint8_t *buf, size_t n, bool at_eof, uint64_t text_at, uint64_t text_end, std::vector<Member> &out, size_t *used,
                 uint64_t *text_bytes) {
    size_t p = 0;
    uint64_t t = text_at;
    while (p < n) {
        if (p + 18 > n) {
            if (at_eof) {
                set_error("truncated gzip member header");
                return GS_E_IO;
            }
            break;
        }
        if (buf[p] != 0x1f || buf[p + 1] != 0x8b || buf[p + 2] != 8 || !(buf[p + 3] & 4)) {
            set_error("not a BGZF member (no gzip magic / no extra field) at byte %zu of a chunk", p);
            return GS_E_IRREGULAR;
        }
        if (buf[p + 3] & ~4u) {  // (name, comment, header CRC: bgzip writes none of them)
            set_error("a BGZF member with header fields bgzip does not write");
            return GS_E_IRREGULAR;
        }
        const size_t xlen = (size_t)buf[p + 10] | ((size_t)buf[p + 11] << 8);
        if (p + 12 + xlen > n) {
            if (at_eof) {
                set_error("truncated gzip member header");
                return GS_E_IO;
            }
            break;
        }
        size_t bsize = 0;
        for (size_t q = p + 12; q + 4 <= p + 12 + xlen;) {
            const size_t slen = (size_t)buf[q + 2] | ((size_t)buf[q + 3] << 8);
            if (buf[q] == 'B' && buf[q + 1] == 'C' && slen == 2 && q + 6 <= p + 12 + xlen) bsize = ((size_t)buf[q + 4] | ((size_t)buf[q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        if (!bsize || bsize < 12 + xlen + 8) {
            set_error("a gzip member without a BGZF block size");
            return GS_E_IRREGULAR;
        }
        if (p + bsize > n) {
            if (at_eof) {
                set_error("truncated BGZF member");
                return GS_E_IO;
            }
            break;
        }
        Member m;
        m.coff = (uint32_t)(p + 12 + xlen);
        m.clen = (uint32_t)(bsize - 12 - xlen - 8);
        const uint8_t *tr = buf + p + bsize - 8;
        m.crc = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
        m.isize = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
        if (m.isize > (1u << 16)) {
            set_error("a BGZF member of more than 64 KiB");
            return GS_E_IRREGULAR;
        }
        if (t + m.isize > text_end) break;  // (the block is full: this member opens the next one)
        m.ooff = (uint32_t)t;
        t += m.isize;
        // (bgzip's empty end-of-file member -- one empty fixed block, CRC 0 -- carries nothing.  Any other member that CLAIMS to be
        // empty is decoded like the rest: a damaged ISIZE field over a member full of text must fail as it does under zlib, not
        // drop the member's text silently)
        const bool eof_marker = !m.isize && m.clen == 2 && buf[m.coff] == 0x03 && buf[m.coff + 1] == 0x00 && m.crc == 0;
        if (!eof_marker) out.push_back(m);
        p += bsize;
    }
    *used = p;
    *text_bytes = t - text_at;
    return GS_OK;
}

}  // namespace

struct gs_bgzf {
    size_t chunk = kChunk;  // (GAVISUNK_BGZF_CHUNK_KB: a test knob -- small chunks, many blocks)
    int device = 0;
    int fd = -1;
    uint64_t file_size = 0, file_at = 0;       // file_at: behind the last member handed to a slot
    uint8_t *h_chunk[2] = {nullptr, nullptr};  // pinned: one is read into while the other's copy to the device is in flight
    hipEvent_t h_free[2] = {nullptr, nullptr};
    int hk = 0;
    // the compressed bytes of one text block as they lie in the file, its members, how much text they make: two of them, so that the
    // NEXT block's bytes are read (page cache -> pinned -> device, on the copy stream) while this block's members are being inflated
    struct Slot {
        uint8_t *d_comp = nullptr;
        size_t comp_cap = 0, comp_at = 0;
        std::vector<Member> members;
        uint64_t text = 0;       // bytes of text of the members (their ooff counts from 0: the carried bytes go in front at launch)
        bool filled = false, last = false;
        hipEvent_t copied = nullptr;
    } slot[2];
    int cur = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    Member *d_mem = nullptr;
    uint8_t *d_scratch = nullptr;
    size_t mem_cap = 0;
    unsigned readers = 4;  // threads that read a chunk of the file (GAVISUNK_BGZF_READERS)
    size_t members_in_flight = 49152;  // what one launch of k_bgzf_inflate keeps resident: CUs x (LDS of a CU / LDS of a lane)
    uint32_t *d_status = nullptr;
    unsigned long long *d_count = nullptr;
    hipStream_t st = nullptr, st_copy = nullptr;
    uint8_t *d_rem = nullptr;     // the bytes behind the last block's last whole record
    size_t rem_n = 0, rem_cap = 0;
    std::vector<uint8_t> tail;
    uint8_t *h_tail = nullptr;  // pinned, kTailPinned bytes
    uint64_t bytes_in = 0, bytes_out = 0;
    bool read_done = false;       // every member of the file is in a slot (or has been inflated)
    bool at_end = false;          // ... and inflated
};

namespace {
// members worth about `want` bytes of text (at most `room`) from the file's current position into slot `sl`: chunk by chunk into pinned
// memory (headers walked there) and on to the device on the copy stream
int fill_slot(gs_bgzf *z, gs_bgzf::Slot &sl, uint64_t want, uint64_t room, bool trace) {
    sl.members.clear();
    sl.comp_at = 0;
    sl.text = 0;
    sl.filled = false;
    sl.last = false;
    if (z->read_done) return GS_OK;
    const uint64_t left0 = z->file_size - z->file_at;
    const size_t need = (size_t)std::min<uint64_t>(left0, std::max<uint64_t>(want / 2, (uint64_t)128 << 20)) + z->chunk + 64;
    if (need > sl.comp_cap) {
        HIP_TRY(hipStreamSynchronize(z->st_copy));
        HIP_TRY(hipStreamSynchronize(z->st));
        (void)hipFree(sl.d_comp);
        sl.d_comp = nullptr;
        sl.comp_cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sl.d_comp), need));
        sl.comp_cap = need;
    }
    std::vector<Member> part;
    size_t per_chunk = 0;  // members of the last chunk: about what the next one holds
    const bool timing = getenv("GAVISUNK_TIMING") != nullptr;
    double ms_wait = 0, ms_read = 0, ms_walk = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };
    const auto t_fill = now();
    while (!z->read_done && sl.text < want && sl.comp_at + z->chunk + 64 <= sl.comp_cap) {
        // (a launch with a few members more than the device keeps resident runs twice as long as one without: 46 ms against 28)
        if (!sl.members.empty() && sl.members.size() + per_chunk > z->members_in_flight && sl.members.size() * 4 >= z->members_in_flight * 3) break;
        const int k = z->hk;
        const uint64_t left = z->file_size - z->file_at;
        const size_t take = (size_t)std::min<uint64_t>(z->chunk, left);
        auto t0 = now();
        HIP_TRY(hipEventSynchronize(z->h_free[k]));  // (this pinned buffer's last copy has left it)
        ms_wait += since(t0);
        t0 = now();
        if (trace) fprintf(stderr, "[bgzf] chunk at %llu: reading %zu bytes\n", (unsigned long long)z->file_at, take);
        {  // four threads read a quarter each (one thread copies out of the page cache at 5-6 GB/s: it would be the whole run)
            constexpr unsigned kMaxReaders = 16;
            bool ok[kMaxReaders];
            for (unsigned t = 0; t < kMaxReaders; t++) ok[t] = true;
            uint8_t *dst = z->h_chunk[k];
            const int fd = z->fd;
            const uint64_t at = z->file_at;
            const unsigned T = take >= ((size_t)4 << 20) ? z->readers : 1u;
            gs_parallel_for(T, [&](unsigned t) {
                size_t lo = take * t / T;
                const size_t hi = take * (t + 1) / T;
                while (lo < hi) {
                    const ssize_t r = pread(fd, dst + lo, hi - lo, (off_t)(at + lo));
                    if (r <= 0) {
                        ok[t] = false;
                        return;
                    }
                    lo += (size_t)r;
                }
            });
            bool all_ok = true;
            for (unsigned t = 0; t < T; t++) all_ok = all_ok && ok[t];
            if (!all_ok) {
                set_error("gs_bgzf_next: read failed");
                return GS_E_IO;
            }
        }
        memset(z->h_chunk[k] + take, 0, 64);
        const bool eof = take == left;
        part.clear();
        size_t used = 0;
        uint64_t text_bytes = 0;
        ms_read += since(t0);
        t0 = now();
        int rc = walk_members(z->h_chunk[k], take, eof, sl.text, room, part, &used, &text_bytes);
        ms_walk += since(t0);
        if (rc != GS_OK) return rc;
        if (!used) {
            if (sl.text) break;  // (the block is full)
            set_error("gs_bgzf_next: a member larger than a chunk");
            return GS_E_IRREGULAR;
        }
        HIP_TRY(hipMemcpyAsync(sl.d_comp + sl.comp_at, z->h_chunk[k], used + 64, hipMemcpyHostToDevice, z->st_copy));
        HIP_TRY(hipEventRecord(z->h_free[k], z->st_copy));
        per_chunk = part.size();
        for (Member &m : part) {
            m.coff += (uint32_t)sl.comp_at;
            sl.members.push_back(m);
        }
        sl.comp_at += used;
        sl.text += text_bytes;
        z->file_at += used;
        z->bytes_in += used;
        if (z->file_at >= z->file_size) z->read_done = true;
        z->hk ^= 1;
    }
    HIP_TRY(hipEventRecord(sl.copied, z->st_copy));
    sl.filled = true;
    sl.last = z->read_done;
    if (timing && sl.comp_at)
        fprintf(stderr, "[gs_bgzf fill] %.1f MB of the file in %.1f ms: %.1f reading (%u threads), %.1f walking member headers, %.1f waiting for a pinned buffer's copy\n",
                sl.comp_at / 1e6, since(t_fill), ms_read, z->readers, ms_walk, ms_wait);
    return GS_OK;
}
}  // namespace

extern "C" {

void gs_bgzf_close(gs_bgzf *z) {
    if (!z) return;
    (void)hipSetDevice(z->device);
    if (z->st) (void)hipStreamSynchronize(z->st);
    if (z->st_copy) (void)hipStreamSynchronize(z->st_copy);
    if (z->fd >= 0) close(z->fd);
    if (z->h_tail) (void)hipHostFree(z->h_tail);
    for (int k = 0; k < 2; k++) {
        if (z->h_chunk[k]) (void)hipHostFree(z->h_chunk[k]);
        if (z->h_free[k]) (void)hipEventDestroy(z->h_free[k]);
        (void)hipFree(z->slot[k].d_comp);
        if (z->slot[k].copied) (void)hipEventDestroy(z->slot[k].copied);
    }
    if (z->ev0) (void)hipEventDestroy(z->ev0);
    if (z->ev1) (void)hipEventDestroy(z->ev1);
    (void)hipFree(z->d_mem);
    (void)hipFree(z->d_scratch);
    (void)hipFree(z->d_status);
    (void)hipFree(z->d_count);
    (void)hipFree(z->d_rem);
    if (z->st) (void)hipStreamDestroy(z->st);
    if (z->st_copy) (void)hipStreamDestroy(z->st_copy);
    delete z;
}

// GS_E_IRREGULAR: the file is not BGZF (read it with gs_gz_*); GS_E_IO: it cannot be opened
int gs_bgzf_open(const char *path, int device, gs_bgzf **out) {
    if (!path || !out) {
        set_error("gs_bgzf_open: null argument");
        return GS_E_ARG;
    }
    *out = nullptr;
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_bgzf_open: HIP device %d not available -- there is no CPU fallback", device);
        return GS_E_NODEVICE;
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        set_error("cannot open the file %s", path);
        return GS_E_IO;
    }
    struct stat sb;
    uint8_t head[18];
    if (fstat(fd, &sb) != 0 || pread(fd, head, sizeof head, 0) != (ssize_t)sizeof head || head[0] != 0x1f || head[1] != 0x8b || head[2] != 8 || head[3] != 4 ||
        head[12] != 'B' || head[13] != 'C') {  // (bgzip writes the BC field first)
        close(fd);
        set_error("%s is not a BGZF file", path);
        return GS_E_IRREGULAR;
    }
    HIP_TRY(hipSetDevice(device));
    auto *z = new gs_bgzf();
    z->device = device;
    z->fd = fd;
    z->file_size = (uint64_t)sb.st_size;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            z->members_in_flight = (size_t)prop.multiProcessorCount * (size_t)(prop.maxSharedMemoryPerMultiProcessor / kInflateLdsBytes) * (size_t)kLanes;
    }
    if (const char *rk = getenv("GAVISUNK_BGZF_READERS")) z->readers = (unsigned)std::min(16, std::max(1, atoi(rk)));
    if (const char *ck = getenv("GAVISUNK_BGZF_CHUNK_KB")) z->chunk = std::min(kChunk, (size_t)std::max(128, atoi(ck)) << 10);
    hipError_t e = hipStreamCreateWithFlags(&z->st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&z->st_copy, hipStreamNonBlocking);
    for (int k = 0; k < 2 && e == hipSuccess; k++) {
        e = hipHostMalloc(reinterpret_cast<void **>(&z->h_chunk[k]), z->chunk + 64, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&z->h_free[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&z->slot[k].copied, hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&z->h_tail), kTailPinned, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreate(&z->ev0);
    if (e == hipSuccess) e = hipEventCreate(&z->ev1);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&z->d_status), sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&z->d_count), sizeof(unsigned long long));
    if (e != hipSuccess) {
        gs_bgzf_close(z);
        set_error("gs_bgzf_open: %s", hipGetErrorString(e));
        return GS_E_NOMEM;
    }
    *out = z;
    return GS_OK;
}

int gs_bgzf_at_end(const gs_bgzf *z) { return z ? ((z->at_end && !z->rem_n) ? 1 : 0) : 1; }

// The next text block: members worth about `want` bytes of text inflated on the device behind what the last block left over, cut
// where the last whole record ends (FASTQ: four lines per record; FASTA: before the last '>' line).  *stage = NULL at the end of
// the file, or when not one whole record fit (ask again, with a larger `want`).  GS_E_IO: a damaged file (what zlib reports as such).
int gs_bgzf_next(gs_bgzf *z, uint64_t want, gs_textstage **stage) {
    if (!z || !stage) {
        set_error("gs_bgzf_next: null argument");
        return GS_E_ARG;
    }
    *stage = nullptr;
    if (z->at_end && !z->rem_n) return GS_OK;
    HIP_TRY(hipSetDevice(z->device));
    want = std::min<uint64_t>(std::max<uint64_t>(want, 1u << 20), 0xC0000000ull);
    const bool trace = getenv("GAVISUNK_BGZF_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[bgzf] next: want %llu, carried %zu, file at %llu of %llu\n", (unsigned long long)want, z->rem_n, (unsigned long long)z->file_at, (unsigned long long)z->file_size);
    // the members of a block make at most `room` bytes of text: what is wanted and one chunk's worth beyond it (a chunk is taken whole)
    const uint64_t room = std::min<uint64_t>(want + (uint64_t)z->chunk * 8, 0xF0000000ull);
    gs_bgzf::Slot &sl = z->slot[z->cur];
    if (!sl.filled && !z->at_end) {
        int rc = fill_slot(z, sl, want, room, trace);
        if (rc != GS_OK) return rc;
    }
    const uint64_t block_text = z->at_end ? 0 : sl.text;
    if ((uint64_t)z->rem_n + block_text > 0xFFFFFF00ull) {
        set_error("gs_bgzf_next: a record of more than %llu bytes of text", (unsigned long long)z->rem_n);
        return GS_E_IRREGULAR;
    }
    size_t cap = 0;
    uint8_t *d_text = gs::stage_buffer(z->device, (size_t)(z->rem_n + block_text) + 64, &cap);
    if (!d_text) return GS_E_NOMEM;
    struct Release {
        int dev;
        uint8_t *p;
        size_t cap;
        ~Release() {
            if (p) gs::stage_buffer_release(dev, p, cap);
        }
    } rel{z->device, d_text, cap};
    uint64_t n = 0;
    if (z->rem_n) {
        HIP_TRY(hipMemcpyAsync(d_text, z->d_rem, z->rem_n, hipMemcpyDeviceToDevice, z->st));
        n = z->rem_n;
        z->rem_n = 0;
    }
    const uint32_t nm = z->at_end ? 0u : (uint32_t)sl.members.size();
    if (trace) fprintf(stderr, "[bgzf] %u members, %zu compressed bytes, %llu bytes of text behind %llu carried\n", nm, sl.comp_at, (unsigned long long)block_text, (unsigned long long)n);
    bool launched = false;
    if (nm) {
        if (nm > z->mem_cap) {
            HIP_TRY(hipStreamSynchronize(z->st));
            (void)hipFree(z->d_mem);
            (void)hipFree(z->d_scratch);
            z->d_mem = nullptr;
            z->d_scratch = nullptr;
            z->mem_cap = std::max<size_t>((size_t)nm + nm / 4 + 1024, 60000);
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&z->d_mem), z->mem_cap * sizeof(Member)));
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&z->d_scratch), z->mem_cap * (size_t)kScratchBytes));
        }
        HIP_TRY(hipStreamWaitEvent(z->st, sl.copied, 0));  // (the slot's compressed bytes have arrived)
        HIP_TRY(hipMemcpyAsync(z->d_mem, sl.members.data(), (size_t)nm * sizeof(Member), hipMemcpyHostToDevice, z->st));
        HIP_TRY(hipMemsetAsync(z->d_status, 0, sizeof(uint32_t), z->st));
        HIP_TRY(hipEventRecord(z->ev0, z->st));
        hipLaunchKernelGGL(k_bgzf_inflate, dim3((nm + kLanes - 1) / kLanes), dim3(64), 0, z->st, sl.d_comp, z->d_mem, nm, d_text + n, z->d_scratch, z->d_status);
        hipLaunchKernelGGL(k_bgzf_crc, dim3((nm + 63) / 64), dim3(64), 0, z->st, d_text + n, z->d_mem, nm, z->d_status);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(z->ev1, z->st));
        launched = true;
    }
    if (!z->at_end) {
        n += block_text;
        if (sl.last) z->at_end = true;
    }
    // while this block's members are being inflated: the next block's compressed bytes
    gs_bgzf::Slot &nx = z->slot[z->cur ^ 1];
    if (!z->read_done) {
        int rc = fill_slot(z, nx, want, room, trace);
        if (rc != GS_OK) {
            (void)hipStreamSynchronize(z->st);
            return rc;
        }
    }
    if (launched) {
        uint32_t status = 0;
        HIP_TRY(hipMemcpyAsync(&status, z->d_status, sizeof status, hipMemcpyDeviceToHost, z->st));
        HIP_TRY(hipStreamSynchronize(z->st));
        if (getenv("GAVISUNK_TIMING")) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, z->ev0, z->ev1);
            fprintf(stderr, "[gs_bgzf_next] %u members, %.1f MB -> %.1f MB: inflate + crc kernels %.2f ms\n", nm, sl.comp_at / 1e6, block_text / 1e6, ms);
#ifdef GS_BGZF_PROF
            unsigned long long pr[8], zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            (void)hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_prof), sizeof pr);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof zero);
            fprintf(stderr, "[prof] per member: %.0f ticks (10 ns) in all, %.0f in headers + tables, %.0f in fast turns; %.0f general steps, %.0f fast turns, %.0f literals + %.0f matches\n",
                    (double)pr[0] / nm, (double)pr[1] / nm, (double)pr[6] / nm, (double)pr[2] / nm, (double)pr[5] / nm, (double)pr[3] / nm, (double)pr[4] / nm);
#endif
        }
        if (status) {
            set_error("gs_bgzf_next: %s (member %u of the block that ends at byte %llu)", err_text(status & 15u), status >> 4, (unsigned long long)z->file_at);
            return GS_E_IO;
        }
    }
    sl.filled = false;
    z->cur ^= 1;
    if (!n) return GS_OK;
    // ---- the cut: where the last whole record ends ----
    uint8_t first = 0;
    HIP_TRY(hipMemcpyAsync(&first, d_text, 1, hipMemcpyDeviceToHost, z->st));
    unsigned long long nl = 0;
    HIP_TRY(hipMemsetAsync(z->d_count, 0, sizeof(unsigned long long), z->st));
    hipLaunchKernelGGL(k_count_newlines, dim3(GS_BGZF_NL_BLOCKS), dim3(256), 0, z->st, d_text, n, z->d_count);
    HIP_TRY(hipMemcpyAsync(&nl, z->d_count, sizeof nl, hipMemcpyDeviceToHost, z->st));
    HIP_TRY(hipStreamSynchronize(z->st));
    uint64_t cut = n;
    if (!z->at_end) {
        // (the end of the text: 1 MiB through a pinned buffer holds a dozen records of long reads; more only if it does not)
        size_t tail_n = (size_t)std::min<uint64_t>(n, (uint64_t)kTailPinned);
        for (;;) {
            uint8_t *dst = z->h_tail;
            if (tail_n > kTailPinned || !dst) {
                z->tail.resize(tail_n);
                dst = z->tail.data();
            }
            HIP_TRY(hipMemcpyAsync(dst, d_text + (n - tail_n), tail_n, hipMemcpyDeviceToHost, z->st));
            HIP_TRY(hipStreamSynchronize(z->st));
            const uint8_t *t = dst;
            int64_t at = -1;  // index in the tail of the byte BEHIND the last whole record
            if (first == '>') {  // FASTA: the last record may be incomplete, whatever it ends with: cut in front of its '>' line
                for (size_t i = tail_n; i-- > 1;)
                    if (t[i] == '>' && t[i - 1] == '\n') {
                        at = (int64_t)i;
                        break;
                    }
            } else {  // four lines per record: behind the newline whose number is the largest multiple of four
                uint64_t skip = nl & 3u;
                for (size_t i = tail_n; i-- > 0;)
                    if (t[i] == '\n') {
                        if (!skip) {
                            at = (int64_t)i + 1;
                            break;
                        }
                        skip--;
                    }
                if (nl < 4) at = -2;  // (not one whole record yet)
            }
            if (at >= 0) {
                cut = (n - tail_n) + (uint64_t)at;
                break;
            }
            if (tail_n == n || at == -2) {
                cut = 0;
                break;
            }
            tail_n = (size_t)std::min<uint64_t>(n, (uint64_t)tail_n * 16);
        }
    }
    if (cut < n) {
        const size_t r = (size_t)(n - cut);
        if (r > z->rem_cap) {
            (void)hipFree(z->d_rem);
            z->d_rem = nullptr;
            z->rem_cap = r + r / 2 + (1u << 20);
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&z->d_rem), z->rem_cap));
        }
        HIP_TRY(hipMemcpyAsync(z->d_rem, d_text + cut, r, hipMemcpyDeviceToDevice, z->st));
        HIP_TRY(hipStreamSynchronize(z->st));
        z->rem_n = r;
    }
    z->bytes_out += cut;
    if (!cut) {
        if (z->at_end) z->rem_n = 0;
        // (a record longer than a block: the caller asks again, with more room)
        return GS_OK;
    }
    *stage = gs::stage_adopt(z->device, d_text, (uint32_t)cut, cap, first);
    rel.p = nullptr;
    return GS_OK;
}

// the names of a parsed block whose text exists only on the device: gathered there, copied as one blob (blob_off[n + 1] on the host)
int gs_textstage_names(const gs_textstage *stage, const uint64_t *name_off, const uint32_t *name_len, uint32_t n, uint8_t *blob, uint64_t *blob_off) {
    if (!stage || (n && (!name_off || !name_len || !blob_off))) {
        set_error("gs_textstage_names: null argument");
        return GS_E_ARG;
    }
    blob_off[0] = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (name_off[i] > stage->n || name_len[i] > stage->n - name_off[i]) {  // (the kernel below trusts them)
            set_error("gs_textstage_names: name %u lies outside the staged text (%u bytes)", i, stage->n);
            return GS_E_ARG;
        }
        blob_off[i + 1] = blob_off[i] + name_len[i];
    }
    if (!n || !blob_off[n]) return GS_OK;
    if (!blob) {
        set_error("gs_textstage_names: null argument");
        return GS_E_ARG;
    }
    HIP_TRY(hipSetDevice(stage->device));
    // one device buffer per calling thread, kept between calls (a hipMalloc / hipFree pair per block would synchronise the device
    // four times -- each time behind the reader thread's inflate kernel)
    struct Scratch {
        uint8_t *p = nullptr;
        size_t cap = 0;
        int device = -1;
    };
    static thread_local Scratch sc;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_bo = up((size_t)n * 8), o_nl = o_bo + up(((size_t)n + 1) * 8), o_blob = o_nl + up((size_t)n * 4), need = o_blob + up(blob_off[n]);
    hipError_t e = hipSuccess;
    if (sc.device != stage->device || sc.cap < need) {
        if (sc.p) {
            (void)hipSetDevice(sc.device);
            (void)hipFree(sc.p);
            (void)hipSetDevice(stage->device);
        }
        sc = Scratch();
        e = hipMalloc(reinterpret_cast<void **>(&sc.p), need + need / 2);
        if (e == hipSuccess) {
            sc.cap = need + need / 2;
            sc.device = stage->device;
        } else {
            sc.p = nullptr;
        }
    }
    uint64_t *d_no = reinterpret_cast<uint64_t *>(sc.p), *d_bo = reinterpret_cast<uint64_t *>(sc.p + o_bo);
    uint32_t *d_nl = reinterpret_cast<uint32_t *>(sc.p + o_nl);
    uint8_t *d_blob = sc.p + o_blob;
    if (e == hipSuccess) e = hipMemcpy(d_no, name_off, (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_bo, blob_off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_nl, name_len, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_gather_names, dim3((n + 255) / 256), dim3(256), 0, 0, stage->d_text, d_no, d_nl, d_bo, n, d_blob);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(blob, d_blob, blob_off[n], hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        set_error("gs_textstage_names: %s", hipGetErrorString(e));
        return GS_E_NODEVICE;
    }
    return GS_OK;
}

// (tests, tools) the staged text on the host
int gs_textstage_copy(const gs_textstage *stage, uint8_t *out, uint64_t cap, uint64_t *n) {
    if (!stage || !n) {
        set_error("gs_textstage_copy: null argument");
        return GS_E_ARG;
    }
    *n = stage->n;
    if (!out) return GS_OK;
    if (cap < stage->n) {
        set_error("gs_textstage_copy: %u bytes do not fit in %llu", stage->n, (unsigned long long)cap);
        return GS_E_CAPACITY;
    }
    HIP_TRY(hipSetDevice(stage->device));
    if (stage->n) HIP_TRY(hipMemcpy(out, stage->d_text, stage->n, hipMemcpyDeviceToHost));
    return GS_OK;
}

uint64_t gs_bgzf_bytes_in(const gs_bgzf *z) { return z ? z->bytes_in : 0; }

}  // extern "C"

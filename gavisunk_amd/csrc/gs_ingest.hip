// On-device FASTA / FASTQ record parsing: the bytes of a block of whole records go to the GPU as
// they are; line starts, record boundaries, names and sequence lengths are found there, the bases are gathered (newlines and
// headers dropped) and 2-bit packed by the same kernels as every other read batch.  The host keeps only the text (for the
// names) and gets back four small arrays.
//
// What is parsed is the REGULAR subset of what the reference's reader accepts (`readfq`, kmerpos_annot3.nim:85; the host
// parser gs_fastx_* in gs_host.cpp restates its rules and is the fallback):
//   FASTA  first byte '>'; a line that starts with '>' is a header, every other line is sequence (may be empty) and starts
//          with neither '@' nor '+' (such a line would end the record in readfq)
//   FASTQ  first byte '@'; records of exactly four lines: '@' header, non-empty sequence line that starts with none of
//          '>', '@', '+', a '+' line, a quality line at least as long as the sequence; only empty lines after the last one
// '\r' before '\n' is dropped; name = header up to the first white space.  Anything else returns GS_E_IRREGULAR and the caller
// reads the file with gs_fastx_*: the same bytes give the same reads either way (tests/test_ingest_gpu.py compares them).
#include <hip/hip_runtime.h>

#include <cstring>
#include <chrono>
#include <cstdlib>
#include <algorithm>
#include <mutex>
#include <vector>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/functional.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

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

struct gs_textindex {
    std::vector<uint64_t> name_off;  // into the caller's text
    std::vector<uint32_t> name_len, len;
};

namespace {

// The parser's temporaries come out of ONE device arena that is kept from call to call (a file is parsed block after block, the
// blocks alike): fifteen hipMalloc / hipFree per block synchronise the device fifteen times, each waiting for whatever another
// thread has in flight (the command line stages the next block meanwhile).  What does not fit is allocated on the side and
// makes the arena larger for the next call.  One parse at a time holds the arena (a lock for the call's duration).
struct ParseArena {
    std::mutex m;
    int device = -1;
    uint8_t *base = nullptr;
    size_t cap = 0, want = 0;
};
ParseArena &parse_arena() {
    static ParseArena *a = new ParseArena;
    return *a;
}
struct DevMem {  // bump allocation from the arena; side allocations are freed when it goes out of scope
    std::vector<void *> side;
    std::unique_lock<std::mutex> lk;
    ParseArena *ar = nullptr;
    size_t used = 0, asked = 0;
    explicit DevMem(int device) : lk(parse_arena().m), ar(&parse_arena()) {
        if (ar->base && (ar->device != device || ar->want > ar->cap)) {  // (idle here: nobody else holds it)
            (void)hipFree(ar->base);
            ar->base = nullptr;
            ar->cap = 0;
        }
        if (!ar->base && ar->want) {
            void *q = nullptr;
            if (hipMalloc(&q, ar->want + ar->want / 8) == hipSuccess) {
                ar->base = static_cast<uint8_t *>(q);
                ar->cap = ar->want + ar->want / 8;
                ar->device = device;
            }
        }
    }
    ~DevMem() {
        for (void *q : side) (void)hipFree(q);
        ar->want = std::max(ar->want, asked);
    }
    template <typename T>
    hipError_t get(T **out, size_t n) {
        const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
        asked += bytes;
        if (ar->base && used + bytes <= ar->cap) {
            *out = reinterpret_cast<T *>(ar->base + used);
            used += bytes;
            return hipSuccess;
        }
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, bytes);
        if (e == hipSuccess) side.push_back(q);
        *out = static_cast<T *>(q);
        return e;
    }
};

// Line starts = position 0 and every position after a '\n' (if inside the text), in order.  Two passes over the text, 16 bytes per
// lane and load: the newlines of each 16 KB tile are counted, the counts are scanned, and the second pass writes the positions to
// their places (a wave-level prefix over the lanes' counts, the waves' totals through LDS).  The text is addressed from the 16-byte
// boundary at or below its first byte ("virtual" offsets, `head` = the bytes before the text), so that every load is aligned.
constexpr uint32_t kLsTile = 16384, kLsSub = 4096;  // bytes per workgroup, bytes per round of its 256 lanes

// bit 7 of every byte of `x` that is '\n'
__device__ inline uint32_t nl_bytes(uint32_t x) {
    const uint32_t y = x ^ 0x0A0A0A0Au;
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
}

// the newline bytes of the 16 bytes at virtual offset `v`, as four words of bit-7 flags; only those at [lo, hi) count
__device__ inline uint4 nl_of_chunk(const uint8_t *__restrict__ vbase, uint64_t v, uint64_t lo, uint64_t hi) {
    uint4 m = make_uint4(0u, 0u, 0u, 0u);
    if (v + 16u <= lo || v >= hi) return m;
    const uint4 w = *reinterpret_cast<const uint4 *>(vbase + v);
    m = make_uint4(nl_bytes(w.x), nl_bytes(w.y), nl_bytes(w.z), nl_bytes(w.w));
    if (v < lo || v + 16u > hi) {  // a chunk at either end of the text: drop the flags of the bytes outside
        uint32_t *mm = reinterpret_cast<uint32_t *>(&m);
        for (uint32_t b = 0; b < 16u; b++)
            if (v + b < lo || v + b >= hi) mm[b >> 2] &= ~(0x80u << ((b & 3u) * 8u));
    }
    return m;
}

__global__ __launch_bounds__(256) void k_nl_count(const uint8_t *__restrict__ vbase, uint64_t lo, uint64_t hi, uint32_t *__restrict__ tile_count) {
    __shared__ uint32_t s_w[4];
    uint32_t c = 0;
    const uint64_t t0 = (uint64_t)blockIdx.x * kLsTile;
    for (uint32_t j = 0; j < kLsTile / kLsSub; j++) {
        const uint4 m = nl_of_chunk(vbase, t0 + j * kLsSub + threadIdx.x * 16u, lo, hi);
        c += __popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w);
    }
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void k_nl_write(const uint8_t *__restrict__ vbase, uint64_t lo, uint64_t hi, uint32_t head,
                                                  const uint32_t *__restrict__ tile_base, uint32_t *__restrict__ ls) {
    __shared__ uint32_t s_w[4];
    if (blockIdx.x == 0 && threadIdx.x == 0) ls[0] = 0u;
    uint32_t base = 1u + tile_base[blockIdx.x];
    const uint64_t t0 = (uint64_t)blockIdx.x * kLsTile;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t j = 0; j < kLsTile / kLsSub; j++) {
        const uint64_t v = t0 + j * kLsSub + threadIdx.x * 16u;
        const uint4 m = nl_of_chunk(vbase, v, lo, hi);
        const uint32_t c = __popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w);
        uint32_t inc = c;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o);
            if ((int)lane >= o) inc += up;
        }
        __syncthreads();  // (the totals of the round before have been read)
        if (lane == 63u) s_w[wave] = inc;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < 4u; w++) {
            if (w < wave) before += s_w[w];
            all += s_w[w];
        }
        if (c) {
            uint32_t at = base + before + inc - c;
            const uint32_t mm[4] = {m.x, m.y, m.z, m.w};
            for (uint32_t q = 0; q < 4u; q++)
                for (uint32_t f = mm[q]; f; f &= f - 1u) ls[at++] = (uint32_t)(v - head) + q * 4u + ((uint32_t)__ffs(f) >> 3);  // the byte AFTER the newline
        }
        base += all;
    }
}

// per line: its length without '\n' / '\r', its first byte (0 for an empty line); the index of the last non-empty line
__global__ void k_line_info(const uint8_t *__restrict__ text, uint32_t n, const uint32_t *__restrict__ ls, uint32_t n_lines,
                            uint32_t *__restrict__ llen, uint8_t *__restrict__ c0, uint32_t *last_nonempty_plus1) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_lines) return;
    const uint32_t a = ls[j];
    uint32_t b = j + 1 < n_lines ? ls[j + 1] - 1u : (text[n - 1] == '\n' ? n - 1u : n);
    if (b > a && text[b - 1] == '\r') b--;
    llen[j] = b - a;
    c0[j] = b > a ? text[a] : 0;
    if (b > a) atomicMax(last_nonempty_plus1, j + 1u);
}

// FASTA: header flags, sequence bytes per line, validity
__global__ void k_fasta_lines(const uint32_t *__restrict__ llen, const uint8_t *__restrict__ c0, uint32_t n_lines,
                              uint8_t *__restrict__ hdr, uint32_t *__restrict__ seqbytes, uint32_t *bad) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_lines) return;
    const uint8_t c = c0[j];
    const bool h = c == '>';
    hdr[j] = h ? 1 : 0;
    seqbytes[j] = h ? 0u : llen[j];
    if ((j == 0 && !h) || c == '@' || c == '+') atomicOr(bad, 1u);
}

// FASTQ: the same per line from the line's place in its record; lines past the last record are empty by construction
__global__ void k_fastq_lines(const uint32_t *__restrict__ llen, const uint8_t *__restrict__ c0, uint32_t n_lines, uint32_t n_rec_lines,
                              uint8_t *__restrict__ hdr, uint32_t *__restrict__ seqbytes, uint32_t *bad) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_lines) return;
    uint8_t h = 0;
    uint32_t sb = 0;
    if (j < n_rec_lines) {
        const uint8_t c = c0[j];
        switch (j & 3u) {
            case 0:
                h = 1;
                if (c != '@') atomicOr(bad, 1u);
                break;
            case 1:
                sb = llen[j];
                if (sb == 0u || c == '>' || c == '@' || c == '+') atomicOr(bad, 1u);
                break;
            case 2:
                if (c != '+') atomicOr(bad, 1u);
                break;
            default:
                if (llen[j] < llen[j - 2]) atomicOr(bad, 1u);
                break;
        }
    }
    hdr[j] = h;
    seqbytes[j] = sb;
}

// per record: where its name is in the text, and where its bases start in the gathered stream
__global__ void k_records(const uint8_t *__restrict__ text, const uint32_t *__restrict__ ls, const uint32_t *__restrict__ llen,
                          const uint32_t *__restrict__ base_off, const uint32_t *__restrict__ hdr_line, uint32_t n_rec,
                          uint32_t *__restrict__ name_off, uint32_t *__restrict__ name_len, uint32_t *__restrict__ rec_off) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    const uint32_t j = hdr_line[r], a = ls[j] + 1u, e = ls[j] + llen[j];
    uint32_t q = a;
    while (q < e) {  // isspace of the C locale
        const uint8_t c = text[q];
        if (c == ' ' || (c >= '\t' && c <= '\r')) break;
        q++;
    }
    name_off[r] = a;
    name_len[r] = q - a;
    rec_off[r] = base_off[j];
}

// the bases of every sequence line to their place in the gathered stream: one wavefront per line
__global__ __launch_bounds__(256) void k_gather_lines(const uint8_t *__restrict__ text, const uint32_t *__restrict__ ls,
                                                      const uint32_t *__restrict__ seqbytes, const uint32_t *__restrict__ base_off,
                                                      uint32_t n_lines, uint8_t *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t j = blockIdx.x * 4u + (threadIdx.x >> 6); j < n_lines; j += gridDim.x * 4u) {
        const uint32_t nb = seqbytes[j];
        if (!nb) continue;
        const uint8_t *src = text + ls[j];
        uint8_t *dst = out + base_off[j];
        for (uint32_t i = lane; i < nb; i += 64u) dst[i] = src[i];
    }
}

}  // namespace

extern "C" {

namespace {
// Device buffers of staged text blocks are kept for the next block instead of being freed: hipFree / hipMalloc synchronise the
// whole device, and in the command-line pipeline one thread stages block i + 1 while another parses and tags block i -- each
// would wait for the other's work at every allocation.  (A block is released only after its parse has finished: the caller's
// contract, and gs_reads_from_staged_text ends with a device synchronisation.)
struct StageCache {
    std::mutex m;
    struct E {
        int device;
        uint8_t *p;
        size_t cap;
    };
    std::vector<E> idle;
    uint8_t *get(int device, size_t want, size_t *cap) {
        std::lock_guard<std::mutex> lk(m);
        size_t best = idle.size();
        for (size_t i = 0; i < idle.size(); i++)
            if (idle[i].device == device && idle[i].cap >= want && (best == idle.size() || idle[i].cap < idle[best].cap)) best = i;
        if (best == idle.size()) return nullptr;
        E e = idle[best];
        idle.erase(idle.begin() + (long)best);
        *cap = e.cap;
        return e.p;
    }
    void put(int device, uint8_t *p, size_t cap) {
        {
            std::lock_guard<std::mutex> lk(m);
            if (idle.size() < 3 && cap >= ((size_t)32 << 20)) {
                idle.push_back(E{device, p, cap});
                return;
            }
        }
        (void)hipSetDevice(device);
        (void)hipFree(p);
    }
};
StageCache &stage_cache() {
    static StageCache *c = new StageCache;
    return *c;
}
}  // namespace

}  // extern "C"
// (for gs_bgzf.hip, whose text blocks are MADE on the device: a buffer from the same cache, and a stage handle around it)
uint8_t *gs::stage_buffer(int device, size_t want, size_t *cap) {
    uint8_t *p = stage_cache().get(device, want, cap);
    if (p) return p;
    *cap = std::max<size_t>(want + want / 8, 1);
    if (hipSetDevice(device) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&p), *cap) != hipSuccess) {
        (void)hipGetLastError();
        set_error("out of device memory (%zu bytes of text)", want);
        return nullptr;
    }
    return p;
}
void gs::stage_buffer_release(int device, uint8_t *p, size_t cap) { stage_cache().put(device, p, cap); }
gs_textstage *gs::stage_adopt(int device, uint8_t *d_text, uint32_t n, size_t cap, uint8_t first) {
    auto *st = new gs_textstage();
    st->device = device;
    st->d_text = d_text;
    st->n = n;
    st->cap = cap;
    st->first = first;
    return st;
}
extern "C" {

int gs_text_stage(int device, const uint8_t *text, uint64_t n_bytes, gs_textstage **out) {
    if (!out || (!text && n_bytes)) {
        set_error("gs_text_stage: bad argument");
        return GS_E_ARG;
    }
    *out = nullptr;
    if (n_bytes >= 0xFFFFFF00ull) {
        set_error("gs_text_stage: a block of %llu bytes (cut the text at record boundaries into blocks below 4 GiB)", (unsigned long long)n_bytes);
        return GS_E_ARG;
    }
    if (gs_device_count() <= device || device < 0) {
        set_error("gs_text_stage: HIP device %d not available -- there is no CPU fallback", device);
        return GS_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    auto *st = new gs_textstage();
    st->device = device;
    st->n = (uint32_t)n_bytes;
    st->first = n_bytes ? text[0] : 0;
    const size_t n = st->n;
    if (const char *sk = getenv("GAVISUNK_TEST_STAGE_SKEW")) st->skew = (uint32_t)atoi(sk) & 15u;
    // (+ 16: the line-start kernels read the text in aligned 16-byte chunks, the last of which may end behind the text's last byte)
    st->d_text = stage_cache().get(device, n + st->skew + 16, &st->cap);
    if (!st->d_text) {
        st->cap = n + n / 8 + st->skew + 16;  // (a little room: the next block of a file is about this long)
        if (hipMalloc(reinterpret_cast<void **>(&st->d_text), st->cap) != hipSuccess) {
            delete st;
            set_error("gs_text_stage: out of device memory (%zu bytes)", n);
            return GS_E_NOMEM;
        }
    }
    st->d_text += st->skew;
    // the text through two small pinned buffers on a stream of its own (a pageable copy of a few hundred MB makes the runtime
    // pin the whole range first); callable from a second thread while the first one loads the SUNK table
    constexpr size_t kStage = 8u << 20;
    struct Stage {
        uint8_t *h[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        hipStream_t s = nullptr;
        // (never torn down: a thread's exit may come after the HIP runtime's; the process's end frees them)
    };
    // (one set per staging thread, made on its first block: creating and freeing pinned buffers and streams synchronises the device)
    static thread_local Stage sg;
    auto fail = [&](hipError_t e) {
        set_error("gs_text_stage: %s", hipGetErrorString(e));
        (void)hipFree(st->d_text - st->skew);
        delete st;
        return GS_E_NODEVICE;
    };
    hipError_t e = hipSuccess;
    if (!sg.s) {
        Stage ns;
        e = hipStreamCreateWithFlags(&ns.s, hipStreamNonBlocking);
        for (int i = 0; i < 2 && e == hipSuccess; i++) {
            e = hipHostMalloc(reinterpret_cast<void **>(&ns.h[i]), kStage, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ns.ev[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) return fail(e);
        sg = ns;
    }
    int b = 0;
    for (size_t o = 0; o < n; o += kStage, b ^= 1) {
        const size_t len = std::min(kStage, n - o);
        if (o >= 2 * kStage && (e = hipEventSynchronize(sg.ev[b])) != hipSuccess) return fail(e);  // this buffer's last copy has left it
        // (a call ends with a stream synchronisation: at o < 2 kStage both buffers are free)
        memcpy(sg.h[b], text + o, len);
        if ((e = hipMemcpyAsync(st->d_text + o, sg.h[b], len, hipMemcpyHostToDevice, sg.s)) != hipSuccess) return fail(e);
        if ((e = hipEventRecord(sg.ev[b], sg.s)) != hipSuccess) return fail(e);
    }
    if ((e = hipStreamSynchronize(sg.s)) != hipSuccess) return fail(e);
    *out = st;
    return GS_OK;
}

void gs_text_stage_free(gs_textstage *st) {
    if (!st) return;
    if (st->d_text) stage_cache().put(st->device, st->d_text - st->skew, st->cap);
    delete st;
}

int gs_reads_from_staged_text(const gs_textstage *stage, int k, gs_reads **out, gs_textindex **index) {
    if (!stage || !out || !index || k < 1 || k > 32) {
        set_error("gs_reads_from_staged_text: bad argument");
        return GS_E_ARG;
    }
    *out = nullptr;
    *index = nullptr;
    const int device = stage->device;
    const uint32_t n = stage->n;
    if (n == 0 || (stage->first != '>' && stage->first != '@')) {
        set_error("gs_reads_from_text: the text does not start with '>' or '@'");
        return GS_E_IRREGULAR;
    }
    HIP_TRY(hipSetDevice(device));
    const bool timing = getenv("GAVISUNK_INGEST_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipDeviceSynchronize();
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[ingest] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
        t_last = t;
    };
    const bool fastq = stage->first == '@';
    DevMem mem(device);
    const uint8_t *d_text = stage->d_text;
    uint8_t *d_c0, *d_hdr, *d_seq;
    uint32_t *d_ls, *d_llen, *d_seqbytes, *d_base, *d_scalars, *d_hdr_line, *d_name_off, *d_name_len, *d_rec_off, *d_tile;
    unsigned long long *d_count;
    // line starts (see k_nl_count): a newline at the text's last byte starts no line
    const uint32_t head = (uint32_t)(reinterpret_cast<uintptr_t>(stage->d_text) & 15u);
    const uint8_t *vbase = stage->d_text - head;
    const uint64_t lo = head, hi = (uint64_t)head + n - 1u;
    const uint32_t n_tiles = (uint32_t)(((uint64_t)head + n + kLsTile - 1u) / kLsTile);
    HIP_TRY(mem.get(&d_tile, (size_t)n_tiles + 1));
    HIP_TRY(mem.get(&d_count, 2));
    HIP_TRY(mem.get(&d_scalars, 4));
    lap("mallocs");
    HIP_TRY(hipMemset(d_scalars, 0, 16));
    HIP_TRY(hipMemset(d_count, 0, 16));
    HIP_TRY(hipMemset(d_tile + n_tiles, 0, 4));
    hipLaunchKernelGGL(k_nl_count, dim3(n_tiles), dim3(256), 0, 0, vbase, lo, hi, d_tile);
    size_t tb = 0;
    void *d_tmp = nullptr;
    rocprim::counting_iterator<uint32_t> iota(0u);
    HIP_TRY(rocprim::exclusive_scan(nullptr, tb, d_tile, d_tile, 0u, (size_t)n_tiles + 1, rocprim::plus<uint32_t>()));
    HIP_TRY(mem.get(reinterpret_cast<uint8_t **>(&d_tmp), tb));
    HIP_TRY(rocprim::exclusive_scan(d_tmp, tb, d_tile, d_tile, 0u, (size_t)n_tiles + 1, rocprim::plus<uint32_t>()));
    uint32_t n_newlines = 0;
    HIP_TRY(hipMemcpy(&n_newlines, d_tile + n_tiles, 4, hipMemcpyDeviceToHost));
    const uint32_t n_lines = n_newlines + 1u;
    HIP_TRY(mem.get(&d_ls, (size_t)n_lines + 1));
    hipLaunchKernelGGL(k_nl_write, dim3(n_tiles), dim3(256), 0, 0, vbase, lo, hi, head, d_tile, d_ls);
    lap("line starts");
    HIP_TRY(mem.get(&d_llen, n_lines));
    HIP_TRY(mem.get(&d_c0, n_lines));
    HIP_TRY(mem.get(&d_hdr, n_lines));
    HIP_TRY(mem.get(&d_seqbytes, (size_t)n_lines + 1));
    HIP_TRY(mem.get(&d_base, (size_t)n_lines + 1));
    const dim3 lgrid((n_lines + 255) / 256), lblock(256);
    hipLaunchKernelGGL(k_line_info, lgrid, lblock, 0, 0, d_text, n, d_ls, n_lines, d_llen, d_c0, d_scalars + 0);
    uint32_t n_rec_lines = 0;
    if (fastq) {
        HIP_TRY(hipMemcpy(&n_rec_lines, d_scalars + 0, 4, hipMemcpyDeviceToHost));
        if (n_rec_lines % 4u) {
            set_error("gs_reads_from_text: FASTQ text whose last record is not four lines");
            return GS_E_IRREGULAR;
        }
        hipLaunchKernelGGL(k_fastq_lines, lgrid, lblock, 0, 0, d_llen, d_c0, n_lines, n_rec_lines, d_hdr, d_seqbytes, d_scalars + 1);
    } else {
        hipLaunchKernelGGL(k_fasta_lines, lgrid, lblock, 0, 0, d_llen, d_c0, n_lines, d_hdr, d_seqbytes, d_scalars + 1);
    }
    HIP_TRY(hipMemset(d_seqbytes + n_lines, 0, 4));
    uint32_t bad = 0;
    HIP_TRY(hipMemcpy(&bad, d_scalars + 1, 4, hipMemcpyDeviceToHost));
    if (bad) {
        set_error("gs_reads_from_text: not a regular %s text (see gs_ingest.hip): read it with gs_fastx_*", fastq ? "four-line FASTQ" : "FASTA");
        return GS_E_IRREGULAR;
    }
    // where each line's bases go in the gathered stream (exclusive prefix over n_lines + 1 values: the last one is the total)
    size_t tb2 = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tb2, d_seqbytes, d_base, 0u, (size_t)n_lines + 1, rocprim::plus<uint32_t>()));
    void *d_tmp2 = nullptr;
    HIP_TRY(mem.get(reinterpret_cast<uint8_t **>(&d_tmp2), tb2));
    HIP_TRY(rocprim::exclusive_scan(d_tmp2, tb2, d_seqbytes, d_base, 0u, (size_t)n_lines + 1, rocprim::plus<uint32_t>()));
    uint32_t total = 0;
    HIP_TRY(hipMemcpy(&total, d_base + n_lines, 4, hipMemcpyDeviceToHost));
    lap("lines, scan");
    // records = header lines
    HIP_TRY(mem.get(&d_hdr_line, n_lines));
    size_t tb3 = 0;
    HIP_TRY(rocprim::select(nullptr, tb3, iota, d_hdr, d_hdr_line, d_count + 1, (size_t)n_lines));
    void *d_tmp3 = nullptr;
    HIP_TRY(mem.get(reinterpret_cast<uint8_t **>(&d_tmp3), tb3));
    HIP_TRY(rocprim::select(d_tmp3, tb3, iota, d_hdr, d_hdr_line, d_count + 1, (size_t)n_lines));
    unsigned long long n_rec64 = 0;
    HIP_TRY(hipMemcpy(&n_rec64, d_count + 1, 8, hipMemcpyDeviceToHost));
    const uint32_t n_rec = (uint32_t)n_rec64;
    HIP_TRY(mem.get(&d_name_off, n_rec));
    HIP_TRY(mem.get(&d_name_len, n_rec));
    HIP_TRY(mem.get(&d_rec_off, n_rec));
    hipLaunchKernelGGL(k_records, dim3((n_rec + 255) / 256), dim3(256), 0, 0, d_text, d_ls, d_llen, d_base, d_hdr_line, n_rec, d_name_off,
                       d_name_len, d_rec_off);
    HIP_TRY(mem.get(&d_seq, (size_t)total + 64));
    hipLaunchKernelGGL(k_gather_lines, dim3(std::min<uint32_t>((n_lines + 3) / 4, 1u << 16)), dim3(256), 0, 0, d_text, d_ls, d_seqbytes, d_base,
                       n_lines, d_seq);
    HIP_TRY(hipGetLastError());
    lap("records, gather");
    auto *ix = new gs_textindex();
    std::vector<uint32_t> h32(n_rec);
    ix->name_len.resize(n_rec);
    ix->name_off.resize(n_rec);
    ix->len.resize(n_rec);
    HIP_TRY(hipMemcpy(h32.data(), d_name_off, (size_t)n_rec * 4, hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < n_rec; r++) ix->name_off[r] = h32[r];
    HIP_TRY(hipMemcpy(ix->name_len.data(), d_name_len, (size_t)n_rec * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h32.data(), d_rec_off, (size_t)n_rec * 4, hipMemcpyDeviceToHost));
    std::vector<uint64_t> off((size_t)n_rec + 1);
    for (uint32_t r = 0; r < n_rec; r++) off[r] = h32[r];
    off[n_rec] = total;
    for (uint32_t r = 0; r < n_rec; r++) ix->len[r] = (uint32_t)(off[r + 1] - off[r]);
    uint64_t *d_boff;
    HIP_TRY(mem.get(&d_boff, (size_t)n_rec + 1));
    const int rc = gs::reads_from_parsed_text(device, d_seq, off.data(), n_rec, k, d_boff, out);  // (waits for the null stream: every kernel above ran there)
    lap("index copies, pack");
    if (rc != GS_OK) {
        delete ix;
        return rc;
    }
    *index = ix;
    return GS_OK;
}

int gs_reads_from_text(int device, const uint8_t *text, uint64_t n_bytes, int k, gs_reads **out, gs_textindex **index) {
    if (!out || !index || (!text && n_bytes) || k < 1 || k > 32) {
        set_error("gs_reads_from_text: bad argument");
        return GS_E_ARG;
    }
    *out = nullptr;
    *index = nullptr;
    if (n_bytes == 0 || (text[0] != '>' && text[0] != '@')) {
        set_error("gs_reads_from_text: the text does not start with '>' or '@'");
        return GS_E_IRREGULAR;
    }
    gs_textstage *st = nullptr;
    int rc = gs_text_stage(device, text, n_bytes, &st);
    if (rc != GS_OK) return rc;
    rc = gs_reads_from_staged_text(st, k, out, index);
    gs_text_stage_free(st);
    return rc;
}

void gs_textindex_free(gs_textindex *ix) { delete ix; }
uint32_t gs_textindex_n_reads(const gs_textindex *ix) { return ix ? (uint32_t)ix->len.size() : 0u; }
const uint64_t *gs_textindex_name_off(const gs_textindex *ix) { return ix ? ix->name_off.data() : nullptr; }
const uint32_t *gs_textindex_name_len(const gs_textindex *ix) { return ix ? ix->name_len.data() : nullptr; }
const uint32_t *gs_textindex_len(const gs_textindex *ix) { return ix ? ix->len.data() : nullptr; }

}  // extern "C"

// Internal declarations shared by the host (gs_host.cpp) and device (gs_device.hip) halves of
// libgavisunk.so.  The public contract is include/gavisunk.h.
#pragma once
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <system_error>
#include <thread>
#include <cstdlib>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "gavisunk.h"

namespace gs {

void set_error(const char *fmt, ...);

// nim-kmer 0.2.6 `encode` byte table (A/a 0, C/c 1, G/g 2, T/t/U/u 3, everything else 0).
struct EncodeTable {
    uint8_t v[256];
    constexpr EncodeTable() : v{} {
        v['C'] = v['c'] = 1;
        v['G'] = v['g'] = 2;
        v['T'] = v['t'] = v['U'] = v['u'] = 3;
    }
};
inline constexpr EncodeTable kEncode{};

inline uint8_t enc(uint8_t c) { return kEncode.v[c]; }

inline uint64_t encode_kmer(const char *s, size_t n) {
    uint64_t c = 0;
    for (size_t i = 0; i < n; i++) c = (c << 2) | kEncode.v[(uint8_t)s[i]];
    return c;
}

// device-wide radix sort of (u64 key, u32 value) pairs by the low `end_bit` key bits (rocPRIM; gs_sunkdef.hip).
// All pointers are device memory; `stream` is a hipStream_t.
int sort_pairs_u64_u32(const uint64_t *k_in, uint64_t *k_out, const uint32_t *v_in, uint32_t *v_out, uint64_t n, int end_bit,
                       void *stream);

// radix sort of 64-bit keys on all 64 bits (rocPRIM; gs_sunkdef.hip)
int sort_keys_u64(const uint64_t *k_in, uint64_t *k_out, uint64_t n, void *stream);
// stable selection: out[] = in[i] for the i with flag[i] != 0, in order; *n_out = how many (rocPRIM select; gs_sunkdef.hip)
int select_flagged_u32(const uint32_t *in, const uint8_t *flag, uint32_t *out, unsigned long long *d_count, uint64_t n, uint64_t *n_out,
                       void *stream);

// the join of a parsed <kmers.db> with its <kmer.loc> on the device (gs_sunkdef.hip): distinct db codes (sorted), per code the dense
// (contig, group) id and the row of the LAST loc row that carries it, GS_MISSING where none does.  hipMalloc'ed outputs.
int db_join_device(int device, const gs_sunkdb *db, uint64_t **d_keys, uint32_t **d_loc, uint32_t **d_payload, uint64_t *n_keys);

// the pair test / clean-up / largest component kernels over device arrays (gs_pair.hip); rows of a read = [d_off[r], d_off[r+1])
int pair_component_device(int device, uint32_t n_reads, uint64_t n_rows, const uint64_t *d_off, const int64_t *d_pos, const int64_t *d_start,
                          const int64_t *d_id, uint32_t *d_outn, int64_t *d_out);
// contig-wide union-find over device arrays (gs_pair.hip)
int union_regions_device(int device, uint64_t n_rows, const uint32_t *d_row_read, const uint32_t *d_row_vertex, uint32_t n_vertices,
                         const int64_t *d_vertex_id, uint32_t *d_size, int64_t *d_min, int64_t *d_max);

// (gs_device.hip, for the on-device record parser of gs_ingest.hip) gs_reads_from_device_ascii for bases that kernels on the
// NULL stream wrote: waits for that stream only, takes its n_reads + 1 words of scratch from the caller
int reads_from_parsed_text(int device, const void *d_seq, const uint64_t *off, uint32_t n_reads, int k, uint64_t *d_boff_scratch,
                           gs_reads **out);

// a FASTA / FASTQ text block in device memory (gs_ingest.hip: copied there by gs_text_stage; gs_bgzf.hip: inflated there)
}  // namespace gs
struct gs_textstage {
    int device = 0;
    uint8_t *d_text = nullptr;
    uint32_t n = 0;
    uint8_t first = 0;
    size_t cap = 0;
    uint32_t skew = 0;  // bytes of the buffer in front of the text (GAVISUNK_TEST_STAGE_SKEW: the parser takes text at any alignment)
};
namespace gs {
uint8_t *stage_buffer(int device, size_t want, size_t *cap);          // from the cache of staged-text buffers, or fresh
void stage_buffer_release(int device, uint8_t *p, size_t cap);
gs_textstage *stage_adopt(int device, uint8_t *d_text, uint32_t n, size_t cap, uint8_t first);

constexpr int kTileWindows = 2048;  // windows per tile = 64 lanes x 32 windows (one 64-bit word per lane)
constexpr int kTileWords = 64;

}  // namespace gs

void gs_advise_huge(void *p, size_t bytes);  // madvise(MADV_HUGEPAGE) (gs_host.cpp)

// fn(t) for t = 0 .. T - 1, one thread per t beside the calling one (which takes t = 0).  A thread that cannot be started --
// std::system_error under a pids / ulimit cap -- has its t run by the caller instead; the ones that were started are always joined
// (an exception out of std::thread's constructor with joinable threads in the vector would end in std::terminate, or cross the C
// boundary into ctypes).
template <typename F>
void gs_parallel_for(unsigned T, F fn) {
    std::vector<std::thread> pool;
    std::vector<unsigned> mine(1, 0u);
    for (unsigned t = 1; t < T; t++) {
        try {
            pool.emplace_back(fn, t);
        } catch (const std::system_error &) {
            mine.push_back(t);
        }
    }
    for (unsigned t : mine) fn(t);
    for (auto &th : pool) th.join();
}

// std::vector for the row arrays of a parsed table (1e8 .. 1e9 entries each): resize() leaves the new elements uninitialised
// (they are written right after, by several threads: no single-threaded zero fill, first touch in parallel), and allocations of
// 4 MiB and more are 2 MiB-aligned and marked for transparent huge pages (a quarter of the page faults of 1e9 rows x 32 bytes)
template <typename T>
struct HugeAlloc {
    using value_type = T;
    HugeAlloc() = default;
    template <typename U>
    HugeAlloc(const HugeAlloc<U> &) {}
    T *allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes >= ((size_t)4 << 20)) {
            const size_t a = (size_t)2 << 20;
            void *p = aligned_alloc(a, (bytes + a - 1) / a * a);
            if (!p) throw std::bad_alloc();
            gs_advise_huge(p, (bytes + a - 1) / a * a);
            return static_cast<T *>(p);
        }
        void *p = malloc(bytes ? bytes : 1);
        if (!p) throw std::bad_alloc();
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t) { free(p); }
    template <typename U>
    void construct(U *p) {
        ::new ((void *)p) U;  // default-initialisation: nothing for the integer types this is used with
    }
    template <typename U, typename... A>
    void construct(U *p, A &&...a) {
        ::new ((void *)p) U(std::forward<A>(a)...);
    }
    template <typename U>
    bool operator==(const HugeAlloc<U> &) const { return true; }
    template <typename U>
    bool operator!=(const HugeAlloc<U> &) const { return false; }
};
template <typename T>
using HugeVec = std::vector<T, HugeAlloc<T>>;

struct gs_sunkdb {
    int k = 0;
    // what the parsers leave: every line of <kmers.db> (file order, duplicates included) and the k-mer of every loc row.  The join
    // of the two -- distinct db codes, each with the LAST loc row that carries it (kmerpos_annot3.nim:68) -- is made where it is
    // needed: on the device for the table (gs_table_create_from_db: sorts + a merge), on the host when a caller asks for the
    // arrays below (gs_sunkdb_keys / _key_row / _key_loc: the first call pays for it)
    HugeVec<uint64_t> db_codes;
    HugeVec<uint64_t> row_code;
    std::atomic<bool> joined{false};  // (set under join_mu when the host join is complete; read without it by the accessors)
    std::mutex join_mu;
    std::vector<uint64_t> keys;     // distinct db codes, first-seen order
    std::vector<uint32_t> key_row;  // winning loc row or GS_MISSING
    std::vector<uint32_t> key_loc;  // dense (contig, group) id or GS_MISSING
    HugeVec<uint32_t> row_contig;
    HugeVec<int64_t> row_start;
    HugeVec<int64_t> row_group;
    HugeVec<uint32_t> row_loc;
    std::vector<std::string> contigs;
};

struct gs_batch {
    std::vector<uint64_t> words;
    std::vector<uint64_t> word_off;  // n_reads + 1
    std::vector<uint32_t> len;
    std::vector<char> names;         // NUL-terminated, back to back
    std::vector<uint64_t> name_off;
    uint64_t bases = 0;
};

// the tagger's output file (gs_host.cpp: rows formatted on the host; gs_format.hip: rows formatted on the device)
struct gs_rowwriter {
    FILE *f = nullptr;
    std::vector<char> buf;
    uint64_t rows = 0;
    void *pinned[4] = {nullptr, nullptr, nullptr, nullptr};  // staging of device-formatted text (allocated on first use, gs_format.hip)
    bool flush() {
        if (buf.empty()) return true;
        const bool ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
        buf.clear();
        return ok;
    }
};
void rowwriter_release_pinned(gs_rowwriter *w);  // gs_format.hip

struct gs_sunkpos {
    std::vector<uint32_t> row_seg, row_contig;
    std::vector<int64_t> row_pos, row_start, row_group;
    std::vector<uint64_t> row_text_off;  // n_rows + 1 offsets into `text`
    std::vector<char> text;              // rows back to back, '\0' after each
    std::vector<std::string> seg_names, contigs;
    // the distinct names (built on demand, gs_sunkpos_n_names): first segment of every name, name id per segment / per row,
    // rank of every name in sorted order; the distinct (contig, ID) pairs (gs_sunkpos_row_keys)
    std::vector<uint32_t> names, name_of_seg, row_name, name_rank, key_contig;
    std::vector<int64_t> key_id;
};

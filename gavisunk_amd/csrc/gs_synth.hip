// libgssynth.so -- synthetic genome-scale inputs generated ON the GPU.  BENCH / TEST SUPPORT, not part of the drop-in
// boundary (include/gavisunk.h) and never loaded by the product path.
//
// BASELINE.json configs[2] ("config 3") is a CHM13-scale haplotype with 10^6 ONT reads; real data do
// not exist offline, so the workload is a size- and density-matched surrogate.  At 3.1 Gb / 4e10 read bases nothing may
// touch the host, and torch cannot share a process with libgavisunk.so on the GPU (its wheel bundles a second HIP
// runtime; two runtimes cannot both acquire the device), hence these few kernels.  Everything is a pure function of
// (seed, index): counter-based hashing, no generator state.
//
//   assembly   uniform ACGT; the sequence is cut into slots of `family_len` bases and a slot is, with probability
//              `repeat_fraction`, an exact copy of one of `n_families` repeat families (none of its k-mers is singly unique)
//   reads      error model of gavisunk_amd/synth.py `mutate`: per source base 2 % substitution (uniform over ACGT), 2 %
//              deletion, 2 % insertion (the base doubled); reverse strand = reverse complement of the source interval
//   select     rows of a device code array that occur in a (small, sorted) query list: how the bench finds the SUNKs that
//              can matter for the reads its CPU oracle walks
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "gs_synth.h"

namespace {

thread_local char g_err[256];
#define SY_TRY(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            snprintf(g_err, sizeof g_err, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -4;                                                                                    \
        }                                                                                                 \
    } while (0)

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

__device__ __forceinline__ uint8_t ascii_of(uint32_t c) { return (uint8_t)("ACGT"[c & 3u]); }
__device__ __forceinline__ uint8_t complement(uint8_t b) { return b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : 'A'; }

// base i of the (ancestral) assembly: a pure function of (seed, i)
// break_period > 0: unique sequence is interrupted every break_period bases by a 44-base interspersed element (one of 32, the
// same bases wherever it lands): its 25 windows are no SUNKs, a gap longer than k, so the run of SUNKs -- the location group
// (workflow/rules/defineSUNKs.smk:124-126: one group per merged interval) -- ends there.  0: groups as long as the unique slots.
constexpr uint32_t kBreakLen = 44;
__device__ __forceinline__ uint32_t ancestor_base(uint64_t i, uint64_t seed, uint32_t family_len, uint32_t n_families, uint32_t repeat_thr,
                                                  uint32_t break_period = 0) {
    const uint64_t slot = i / family_len;
    const uint64_t hs = mix64(seed ^ (slot * 0xD1B54A32D192ED03ULL) ^ 0xABCDEF);
    uint64_t h;
    if ((uint32_t)hs < repeat_thr) {  // a repeat slot: base (i mod family_len) of family (hs >> 32) mod n_families
        const uint64_t fam = (hs >> 32) % n_families;
        h = mix64(seed ^ 0x5EED0FFA ^ ((fam << 32) | (i % family_len)));
    } else if (break_period && i % break_period < kBreakLen) {
        const uint64_t elem = mix64(seed ^ 0xB4EA ^ (i / break_period)) & 31u;
        h = mix64(seed ^ 0xE1E3 ^ ((elem << 32) | (i % break_period)));
    } else {
        h = mix64(seed ^ (i * 0x9E3779B97F4A7C15ULL));
    }
    return (uint32_t)(h >> 13) & 3u;
}

__global__ void k_genome(uint8_t *out, uint64_t n, uint64_t seed, uint32_t family_len, uint32_t n_families, uint32_t repeat_thr,
                         uint32_t break_period) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = ascii_of(ancestor_base(i, seed, family_len, n_families, repeat_thr, break_period));
}

// BASELINE.json configs[3] (cfg 4): haplotype `hap` (0 / 1) of a diploid assembly -- both descend from
// the ancestor above; a site is a SNP with probability snp (hap 1 then carries another base than the ancestor, hap 0 keeps
// it: the haplotypes DIFFER at that rate), and a slot of spec_len bases is, per haplotype and independently, replaced by
// sequence that exists nowhere else with probability spec ("5 % hap-specific sequence"; same length in both haplotypes).
__global__ void k_haplotype(uint8_t *out, uint64_t n, uint64_t seed, uint32_t family_len, uint32_t n_families, uint32_t repeat_thr,
                            uint32_t hap, uint32_t snp_thr, uint32_t spec_thr, uint32_t spec_len) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t hs = mix64(seed ^ ((i / spec_len) * 0xA24BAED4963EE407ULL) ^ (0x51ED27 + hap));
        uint32_t b;
        if ((uint32_t)hs < spec_thr) {
            b = (uint32_t)(mix64(seed ^ 0x4A9C0DE ^ ((uint64_t)(hap + 1) << 60) ^ (i * 0x9E3779B97F4A7C15ULL)) >> 13) & 3u;
        } else {
            b = ancestor_base(i, seed, family_len, n_families, repeat_thr);
            if (hap) {
                const uint64_t h2 = mix64(seed ^ 0x5A9 ^ (i * 0xC2B2AE3D27D4EB4FULL));
                if ((uint32_t)h2 < snp_thr) b = (b + 1u + (uint32_t)((h2 >> 40) % 3u)) & 3u;
            }
        }
        out[i] = ascii_of(b);
    }
}

struct ReadSrc {
    uint64_t start;  // first base of the source interval in the assembly
    uint32_t len;
    uint32_t rev;
};

// what happens to source base j of read r: bits 1:0 = copies written (0, 1, 2), bits 3:2 = substituted base or 4 = keep
__device__ __forceinline__ uint32_t fate(uint64_t seed, uint32_t r, uint32_t j, uint32_t t_sub, uint32_t t_del, uint32_t t_ins) {
    const uint64_t h = mix64(seed ^ (((uint64_t)r << 32) | j) * 0xC2B2AE3D27D4EB4FULL);
    const uint32_t u = (uint32_t)h;
    if (u < t_sub) return 1u | (((uint32_t)(h >> 40) & 3u) << 2);
    if (u < t_del) return 0u | (4u << 2);
    if (u < t_ins) return 2u | (4u << 2);
    return 1u | (4u << 2);
}

// one wavefront per read; pass 0 counts the output bases, pass 1 writes them
template <int PASS>
__global__ __launch_bounds__(256) void k_reads(const uint8_t *__restrict__ genome, const ReadSrc *__restrict__ src, uint32_t n_reads,
                                               uint64_t seed, uint32_t t_sub, uint32_t t_del, uint32_t t_ins, uint64_t *out_len,
                                               const uint64_t *__restrict__ out_off, uint8_t *out) {
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= n_reads) return;
    const ReadSrc s = src[r];
    const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    uint64_t w = PASS ? out_off[r] : 0;  // wave-uniform write cursor / count
    for (uint32_t j0 = 0; j0 < s.len; j0 += 64u) {
        const uint32_t j = j0 + (uint32_t)lane;
        const bool act = j < s.len;
        const uint32_t f = act ? fate(seed, r, j, t_sub, t_del, t_ins) : (4u << 2);
        const uint32_t rep = act ? (f & 3u) : 0u;
        const unsigned long long m1 = __ballot(rep >= 1), m2 = __ballot(rep == 2);
        if (PASS) {
            if (rep) {
                uint8_t b = s.rev ? complement(genome[s.start + (s.len - 1u - j)]) : genome[s.start + j];
                if ((f >> 2) < 4u) b = ascii_of(f >> 2);
                const uint64_t o = w + (uint64_t)__popcll(m1 & lt) + (uint64_t)__popcll(m2 & lt);
                out[o] = b;
                if (rep == 2) out[o + 1] = b;
            }
        }
        w += (uint64_t)__popcll(m1) + (uint64_t)__popcll(m2);
    }
    if (!PASS && lane == 0) out_len[r] = w;
}

__global__ void k_select(const uint64_t *codes, uint64_t n, const uint64_t *q, uint64_t nq, uint32_t *rows, unsigned long long *count,
                         uint64_t cap) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t c = codes[i];
        uint64_t lo = 0, hi = nq;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (q[mid] < c) lo = mid + 1;
            else hi = mid;
        }
        if (lo < nq && q[lo] == c) {
            const unsigned long long o = atomicAdd(count, 1ULL);
            if (o < cap) rows[o] = (uint32_t)i;
        }
    }
}

template <typename T>
__global__ void k_gather(const T *src, const uint32_t *rows, uint64_t n, T *dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[rows[i]];
}

}  // namespace

extern "C" {

const char *gs_synth_last_error(void) { return g_err; }

void gs_synth_free(int device, void *d) {
    (void)hipSetDevice(device);
    (void)hipFree(d);
}

int gs_synth_genome(int device, uint64_t seed, uint64_t n_bases, double repeat_fraction, uint32_t n_families, uint32_t family_len,
                    uint32_t break_period, void **d_ascii_out) {
    if (!d_ascii_out || !n_bases || !n_families || !family_len || repeat_fraction < 0 || repeat_fraction > 1) {
        snprintf(g_err, sizeof g_err, "gs_synth_genome: bad argument");
        return -1;
    }
    SY_TRY(hipSetDevice(device));
    uint8_t *d = nullptr;
    SY_TRY(hipMalloc((void **)&d, n_bases + 64));
    const uint32_t thr = (uint32_t)(repeat_fraction * 4294967295.0);
    hipLaunchKernelGGL(k_genome, dim3(8192), dim3(256), 0, 0, d, n_bases, seed, family_len, n_families, thr, break_period);
    SY_TRY(hipGetLastError());
    SY_TRY(hipDeviceSynchronize());
    *d_ascii_out = d;
    return 0;
}

int gs_synth_diploid(int device, uint64_t seed, uint64_t n_bases_per_hap, double repeat_fraction, uint32_t n_families,
                     uint32_t family_len, double snp, double spec, uint32_t spec_len, void **d_ascii_out) {
    if (!d_ascii_out || !n_bases_per_hap || !n_families || !family_len || !spec_len || repeat_fraction < 0 || repeat_fraction > 1 ||
        snp < 0 || snp > 1 || spec < 0 || spec > 1) {
        snprintf(g_err, sizeof g_err, "gs_synth_diploid: bad argument");
        return -1;
    }
    SY_TRY(hipSetDevice(device));
    uint8_t *d = nullptr;
    SY_TRY(hipMalloc((void **)&d, 2 * n_bases_per_hap + 64));
    const uint32_t thr = (uint32_t)(repeat_fraction * 4294967295.0);
    for (uint32_t hap = 0; hap < 2; hap++)
        hipLaunchKernelGGL(k_haplotype, dim3(8192), dim3(256), 0, 0, d + (uint64_t)hap * n_bases_per_hap, n_bases_per_hap, seed,
                           family_len, n_families, thr, hap, (uint32_t)(snp * 4294967295.0), (uint32_t)(spec * 4294967295.0), spec_len);
    SY_TRY(hipGetLastError());
    SY_TRY(hipDeviceSynchronize());
    *d_ascii_out = d;
    return 0;
}

int gs_synth_reads(int device, uint64_t seed, const void *d_genome_ascii, uint32_t n_reads, const uint64_t *src_start,
                   const uint32_t *src_len, const uint8_t *rev, double sub, double dele, double ins, void **d_ascii_out,
                   uint64_t *off_out) {
    if (!d_genome_ascii || !d_ascii_out || !off_out || (n_reads && (!src_start || !src_len || !rev))) {
        snprintf(g_err, sizeof g_err, "gs_synth_reads: bad argument");
        return -1;
    }
    SY_TRY(hipSetDevice(device));
    std::vector<ReadSrc> hs(n_reads);
    for (uint32_t i = 0; i < n_reads; i++) hs[i] = ReadSrc{src_start[i], src_len[i], rev[i] ? 1u : 0u};
    ReadSrc *d_src = nullptr;
    uint64_t *d_len = nullptr, *d_off = nullptr;
    SY_TRY(hipMalloc((void **)&d_src, (size_t)(n_reads + 1) * sizeof(ReadSrc)));
    SY_TRY(hipMalloc((void **)&d_len, (size_t)(n_reads + 1) * sizeof(uint64_t)));
    SY_TRY(hipMalloc((void **)&d_off, (size_t)(n_reads + 1) * sizeof(uint64_t)));
    SY_TRY(hipMemcpy(d_src, hs.data(), (size_t)n_reads * sizeof(ReadSrc), hipMemcpyHostToDevice));
    const uint32_t t_sub = (uint32_t)(sub * 4294967295.0), t_del = (uint32_t)((sub + dele) * 4294967295.0),
                   t_ins = (uint32_t)((sub + dele + ins) * 4294967295.0);
    const unsigned blocks = (n_reads + 3) / 4;
    if (n_reads)
        hipLaunchKernelGGL((k_reads<0>), dim3(blocks), dim3(256), 0, 0, (const uint8_t *)d_genome_ascii, d_src, n_reads, seed, t_sub,
                           t_del, t_ins, d_len, (const uint64_t *)nullptr, (uint8_t *)nullptr);
    std::vector<uint64_t> len(n_reads);
    SY_TRY(hipMemcpy(len.data(), d_len, (size_t)n_reads * sizeof(uint64_t), hipMemcpyDeviceToHost));
    off_out[0] = 0;
    for (uint32_t i = 0; i < n_reads; i++) off_out[i + 1] = off_out[i] + len[i];
    SY_TRY(hipMemcpy(d_off, off_out, (size_t)(n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    uint8_t *d_out = nullptr;
    SY_TRY(hipMalloc((void **)&d_out, off_out[n_reads] + 64));
    if (n_reads)
        hipLaunchKernelGGL((k_reads<1>), dim3(blocks), dim3(256), 0, 0, (const uint8_t *)d_genome_ascii, d_src, n_reads, seed, t_sub,
                           t_del, t_ins, (uint64_t *)nullptr, (const uint64_t *)d_off, d_out);
    SY_TRY(hipGetLastError());
    SY_TRY(hipDeviceSynchronize());
    (void)hipFree(d_src);
    (void)hipFree(d_len);
    (void)hipFree(d_off);
    *d_ascii_out = d_out;
    return 0;
}

int gs_synth_select_rows(int device, const void *d_codes, uint64_t n, const uint64_t *sorted_queries, uint64_t nq, uint32_t *rows_out,
                         uint64_t cap, uint64_t *n_rows) {
    if (!n_rows || (n && !d_codes) || (nq && !sorted_queries) || (cap && !rows_out)) {
        snprintf(g_err, sizeof g_err, "gs_synth_select_rows: bad argument");
        return -1;
    }
    SY_TRY(hipSetDevice(device));
    uint64_t *d_q = nullptr;
    uint32_t *d_rows = nullptr;
    unsigned long long *d_cnt = nullptr;
    SY_TRY(hipMalloc((void **)&d_q, (nq + 1) * sizeof(uint64_t)));
    SY_TRY(hipMalloc((void **)&d_rows, (cap + 1) * sizeof(uint32_t)));
    SY_TRY(hipMalloc((void **)&d_cnt, sizeof(unsigned long long)));
    SY_TRY(hipMemcpy(d_q, sorted_queries, nq * sizeof(uint64_t), hipMemcpyHostToDevice));
    SY_TRY(hipMemset(d_cnt, 0, sizeof(unsigned long long)));
    if (n) hipLaunchKernelGGL(k_select, dim3(8192), dim3(256), 0, 0, (const uint64_t *)d_codes, n, d_q, nq, d_rows, d_cnt, cap);
    unsigned long long cnt = 0;
    SY_TRY(hipMemcpy(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost));
    *n_rows = cnt;
    if (cnt <= cap && cnt) SY_TRY(hipMemcpy(rows_out, d_rows, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost));
    (void)hipFree(d_q);
    (void)hipFree(d_rows);
    (void)hipFree(d_cnt);
    if (cnt > cap) {
        snprintf(g_err, sizeof g_err, "gs_synth_select_rows: %llu rows do not fit cap %llu", cnt, (unsigned long long)cap);
        return -3;
    }
    return 0;
}

int gs_synth_gather(int device, const void *d_src, int elem_bytes, const uint32_t *rows, uint64_t n, void *out) {
    if (n && (!d_src || !rows || !out || (elem_bytes != 4 && elem_bytes != 8))) {
        snprintf(g_err, sizeof g_err, "gs_synth_gather: bad argument");
        return -1;
    }
    if (!n) return 0;
    SY_TRY(hipSetDevice(device));
    uint32_t *d_rows = nullptr;
    void *d_dst = nullptr;
    SY_TRY(hipMalloc((void **)&d_rows, n * sizeof(uint32_t)));
    SY_TRY(hipMalloc(&d_dst, n * (size_t)elem_bytes));
    SY_TRY(hipMemcpy(d_rows, rows, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (elem_bytes == 8)
        hipLaunchKernelGGL((k_gather<uint64_t>), dim3(blocks), dim3(256), 0, 0, (const uint64_t *)d_src, d_rows, n, (uint64_t *)d_dst);
    else
        hipLaunchKernelGGL((k_gather<uint32_t>), dim3(blocks), dim3(256), 0, 0, (const uint32_t *)d_src, d_rows, n, (uint32_t *)d_dst);
    SY_TRY(hipMemcpy(out, d_dst, n * (size_t)elem_bytes, hipMemcpyDeviceToHost));
    (void)hipFree(d_rows);
    (void)hipFree(d_dst);
    return 0;
}

}  // extern "C"

// Microbenchmark (design input for genome-scale tables, round 2): lane = window.  Consecutive windows of a read that share
// a minimiser sit in ADJACENT LANES of one wave-load; if their table buckets share a 128-byte line the texture path
// coalesces them into one request.  What does a run of L adjacent lanes per line buy on a table far larger than the caches?
//   mode 0: one 16-byte load per window, runs of exactly L lanes per 128-byte region (piece = hash(window) & 7)
//   mode 1: the same with 64-byte regions (piece & 3)
//   mode 2: two 16-byte loads per window (a 32-byte piece), 128-byte regions
//   mode 3: one 8-byte load per window, 128-byte regions
//   mode 4: run lengths drawn per run from {1..2L-1} (mean L), one 16-byte load per window
// U wave-steps are in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x *= 0x9E3779B97F4A7C15ULL; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; return x ^ (x >> 32); }

template <int MODE, int U>
__global__ __launch_bounds__(256) void k_run(const uint4 *__restrict__ tab, uint64_t n_lines, int L, int rounds, uint32_t *out) {
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        uint4 v[U];
        uint4 v2[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t g = ((wave * rounds + r) * U + u) * 64 + lane;  // window index
            uint64_t run;
            if (MODE == 4) {
                // variable runs: cut the window axis into cells of 2L windows holding runs of length a and 2L - a
                const uint64_t cell = g / (2 * L);
                const int a = 1 + (int)(mix(cell * 77 + 5) % (uint64_t)(2 * L - 1));
                run = cell * 2 + ((int)(g % (2 * L)) >= a ? 1 : 0);
            } else {
                run = g / L;
            }
            const uint64_t h = mix(run + 1), hw = mix(g ^ 0x5555);
            if (MODE == 1) {
                const uint64_t reg = h % (n_lines * 2);
                v[u] = tab[reg * 4 + (hw & 3)];
            } else if (MODE == 2) {
                const uint64_t reg = h % n_lines;
                v[u] = tab[reg * 8 + (hw & 3) * 2];
                v2[u] = tab[reg * 8 + (hw & 3) * 2 + 1];
            } else if (MODE == 3) {
                const uint64_t reg = h % n_lines;
                const uint2 x = reinterpret_cast<const uint2 *>(tab)[reg * 16 + (hw & 15)];
                v[u] = make_uint4(x.x, x.y, 0, 0);
            } else {
                const uint64_t reg = h % n_lines;
                v[u] = tab[reg * 8 + (hw & 7)];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            acc += v[u].x ^ v[u].w;
            if (MODE == 2) acc += v2[u].y;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// mode 5: the shape of k_probe_mz -- per step of 64 windows the R = 64 / L distinct regions are fetched whole (8 lanes x 16 B
// each) by LDS-DMA, NI wave-instructions per step (runs past R repeat the last one), two steps in flight behind a counted vmcnt,
// then every lane reads one 16-byte piece of its run's region from LDS.  LDSB bytes of LDS per wave set the occupancy.
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p; }
template <int NI, int LDSB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_dma(const uint4 *__restrict__ tab, uint64_t n_lines, int L, int rounds, uint32_t *out) {
    __shared__ uint4 cache[WAVES][LDSB / 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t gw = (uint64_t)blockIdx.x * WAVES + wave;
    const uint32_t R = (64 + L - 1) / L;
    uint32_t acc = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#define STAGE(r, slot)                                                                                                         \
    _Pragma("unroll") for (int it = 0; it < NI; it++) {                                                                        \
        const uint32_t q = min((uint32_t)(it * 8 + (lane >> 3)), R - 1u);                                                      \
        const uint64_t reg = mix((gw * rounds + (r)) * 64 + q + 1) % n_lines;                                                  \
        __builtin_amdgcn_global_load_lds(tab + reg * 8 + (lane & 7), &cache[wave][(slot) * NI * 64 + it * 64], 16, 0, 0);     \
    }
#else
#define STAGE(r, slot)
#endif
    STAGE(0, 0);
    STAGE(1, 1);
    for (int r = 0; r < rounds; r += 2) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
            const uint32_t myrun = (uint32_t)lane / (uint32_t)L;
            const uint32_t a = lds_addr(&cache[wave][u * NI * 64]) + myrun * 128u + ((uint32_t)mix(gw * 77 + r * 64 + lane) & 7u) * 16u;
            uint4 v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v.x ^ v.w;
            STAGE(r + u + 2, u);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) out[0] = acc;
}

template <int NI, int LDSB, int WAVES>
int run_dma(const uint4 *tab, uint64_t n_lines, int L, uint32_t *out, int waves_per_cu) {
    const int rounds = 256, blocks = 256 * waves_per_cu / WAVES * 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_dma<NI, LDSB, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, tab, n_lines, L, rounds, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    const double win = (double)blocks * WAVES * 64 * rounds;
    printf("LDS-DMA whole regions: NI=%d, %5.1f KiB LDS/wave (%2d waves/CU)  L=%2d : %7.1f G windows/s  %6.1f G lines/s  (%.3f ms)\n", NI, LDSB / 1024.0,
           waves_per_cu, L, win / ms / 1e6, win / L / ms / 1e6, ms);
    fflush(stdout);
    return 0;
}

template <int MODE, int U>
int run(const uint4 *tab, uint64_t n_lines, int L, uint32_t *out, const char *what) {
    const int rounds = 64, blocks = 256 * 8;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_run<MODE, U>), dim3(blocks), dim3(256), 0, 0, tab, n_lines, L, rounds, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    const double win = (double)blocks * 256 * rounds * U;
    printf("%-44s L=%2d U=%d : %7.1f G windows/s  %6.1f G lines/s  (%.3f ms)\n", what, L, U, win / ms / 1e6, win / L / ms / 1e6, ms);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv) {
    const size_t gib = argc > 1 ? (size_t)atol(argv[1]) : 16;
    const size_t bytes = gib << 30;
    uint4 *tab; uint32_t *out;
    CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(tab, 1, bytes));
    const uint64_t n_lines = bytes / 128;
    printf("table %zu GiB, lane = window, runs of L adjacent lanes share a region\n", gib);
    if (argc > 2) {  // only the LDS-DMA form
        for (int L : {2, 3, 4, 6, 8}) if (run_dma<4, 13312, 2>(tab, n_lines, L, out, 12)) return 1;
        for (int L : {3, 4}) if (run_dma<4, 8192, 2>(tab, n_lines, L, out, 18)) return 1;
        for (int L : {3, 4, 8}) if (run_dma<2, 4096, 4>(tab, n_lines, L, out, 32)) return 1;
        for (int L : {3, 4}) if (run_dma<3, 13312, 2>(tab, n_lines, L, out, 12)) return 1;
        return 0;
    }
    const int Ls[] = {1, 2, 3, 4, 6, 8, 16};
    for (int L : Ls) if (run<0, 4>(tab, n_lines, L, out, "16 B/window, 128-B regions")) return 1;
    for (int L : Ls) if (run<0, 8>(tab, n_lines, L, out, "16 B/window, 128-B regions")) return 1;
    for (int L : {1, 2, 3, 4}) if (run<1, 8>(tab, n_lines, L, out, "16 B/window, 64-B regions")) return 1;
    for (int L : {1, 2, 3, 4, 6}) if (run<2, 4>(tab, n_lines, L, out, "2 x 16 B/window, 128-B regions")) return 1;
    for (int L : {1, 2, 3, 4, 6}) if (run<3, 8>(tab, n_lines, L, out, "8 B/window, 128-B regions")) return 1;
    for (int L : {2, 3, 4, 6}) if (run<4, 8>(tab, n_lines, L, out, "16 B/window, variable runs (mean L)")) return 1;
    return 0;
}

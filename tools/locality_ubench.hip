// Microbenchmark (design input): what does locality between consecutive probes of a lane buy on a table far larger
// than the caches?  Each lane issues 8 independent 16-byte loads per round; groups of REP consecutive loads go to the
// same 128-byte region (different 16-byte pieces), as consecutive windows that share a minimiser would.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x *= 0x9E3779B97F4A7C15ULL; return x ^ (x >> 31); }

template <int REP>
__global__ __launch_bounds__(256) void k_rep(const uint4 *__restrict__ tab, uint64_t region_mask, int rounds, uint32_t *out) {
    uint64_t s = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x632BE59BD9B4E019ULL + 1;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j % REP == 0) s = mix(s + j + 1);
            v[j] = tab[(((s >> 20) & region_mask) << 3) | (uint64_t)((j + (int)(s & 7)) & 7)];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc += v[j].x;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int REP>
int run(const uint4 *tab, uint64_t regions, uint32_t *out) {
    const int rounds = 256, blocks = 256 * 8;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_rep<REP>), dim3(blocks), dim3(256), 0, 0, tab, regions - 1, rounds, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("%d consecutive probes per 128-B region: %7.1f G probes/s (%.3f ms)\n", REP, (double)blocks * 256 * rounds * 8 / ms / 1e6, ms);
    return 0;
}

int main() {
    const size_t bytes = 4096ull << 20;
    uint4 *tab; uint32_t *out;
    CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(tab, 1, bytes));
    const uint64_t regions = bytes / 128;
    if (run<1>(tab, regions, out) || run<2>(tab, regions, out) || run<4>(tab, regions, out) || run<8>(tab, regions, out)) return 1;
    return 0;
}

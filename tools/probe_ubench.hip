// Microbenchmark (design input, not product code): random-probe throughput of MI355X by table
// size and access width.  Each lane issues `UNROLL` independent loads per round at hashed
// addresses, like k_probe does.  Prints G probes/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x *= 0x9E3779B97F4A7C15ULL; return x ^ (x >> 31); }

template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void k_rand(const T *__restrict__ tab, uint64_t mask, int rounds, uint32_t *out) {
    uint64_t s = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x632BE59BD9B4E019ULL + 1;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        T v[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; j++) { s = mix(s + j + 1); v[j] = tab[(s >> 20) & mask]; }
#pragma unroll
        for (int j = 0; j < UNROLL; j++) acc += ((const uint32_t *)&v[j])[0];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void k_lds(const uint32_t *__restrict__ src, int words, int rounds, uint32_t *out) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    uint64_t s = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x632BE59BD9B4E019ULL + 1;
    uint32_t acc = 0;
    const uint32_t mask = words - 1;
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int j = 0; j < UNROLL; j++) { s = mix(s + j + 1); acc += lds[(s >> 20) & mask]; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename T>
int run(const char *name, size_t bytes, int blocks_per_cu) {
    const size_t n = bytes / sizeof(T);
    T *tab; uint32_t *out;
    CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(tab, 1, bytes));
    const int rounds = 256, blocks = 256 * blocks_per_cu;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_rand<T, 8>), dim3(blocks), dim3(256), 0, 0, tab, (uint64_t)n - 1, rounds, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    }
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double probes = (double)blocks * 256 * rounds * 8;
    printf("%-6s table %8.1f MiB  blocks/CU %d : %7.1f G probes/s  (%.3f ms)\n", name, bytes / 1048576.0, blocks_per_cu, probes / ms / 1e6, ms);
    CK(hipFree(tab)); CK(hipFree(out));
    return 0;
}

int main() {
    const size_t sizes[] = {1u << 20, 2u << 20, 4u << 20, 8u << 20, 16u << 20, 32u << 20, 64u << 20, 128u << 20, 256u << 20, 1024u << 20, 4096ull << 20};
    for (size_t s : sizes) { if (run<uint32_t>("u32", s, 8)) return 1; }
    for (size_t s : sizes) { if (run<uint2>("u64", s, 8)) return 1; }
    for (size_t s : sizes) { if (run<uint4>("u128", s, 8)) return 1; }
    for (int bpc : {2, 4, 16}) { if (run<uint4>("u128", 128u << 20, bpc)) return 1; if (run<uint32_t>("u32", 2u << 20, bpc)) return 1; }
    // LDS
    uint32_t *src, *out; CK(hipMalloc(&src, 128 << 10)); CK(hipMalloc(&out, 4)); CK(hipMemset(src, 1, 128 << 10));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int kb : {16, 64, 128}) {
        const int words = kb * 256, rounds = 1024, blocks = 256 * (160 / kb > 8 ? 8 : 160 / kb);
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL((k_lds<8>), dim3(blocks), dim3(256), kb * 1024, 0, src, words, rounds, out);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        }
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("LDS %3d KiB/block, %d blocks: %7.1f G probes/s (%.3f ms)\n", kb, blocks, (double)blocks * 256 * rounds * 8 / ms / 1e6, ms);
    }
    return 0;
}

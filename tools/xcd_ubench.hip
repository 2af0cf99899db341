// Microbenchmark (design input): (a) is block b placed on XCD b % 8?  (b) random-probe rate when
// block b only touches the (b % 8)-th eighth of a table (XCD-partitioned L2 residency).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x *= 0x9E3779B97F4A7C15ULL; return x ^ (x >> 31); }

__global__ void k_xcc(uint32_t *out) {
    if (threadIdx.x == 0) {
        uint32_t x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = x & 0xF;
    }
}

template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void k_part(const T *__restrict__ tab, uint64_t part_mask, uint64_t part_elems, int nparts,
                                              int rounds, uint32_t *out) {
    const T *base = tab + (uint64_t)(blockIdx.x % nparts) * part_elems;
    uint64_t s = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x632BE59BD9B4E019ULL + 1;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        T v[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; j++) { s = mix(s + j + 1); v[j] = base[(s >> 20) & part_mask]; }
#pragma unroll
        for (int j = 0; j < UNROLL; j++) acc += ((const uint32_t *)&v[j])[0];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    uint32_t *d; CK(hipMalloc(&d, 4096 * 4));
    hipLaunchKernelGGL(k_xcc, dim3(4096), dim3(64), 0, 0, d);
    std::vector<uint32_t> h(4096); CK(hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost));
    int match = 0; for (int i = 0; i < 4096; i++) match += (h[i] == (uint32_t)((i + (h[0] - 0 + 8)) % 8));
    printf("XCC of blocks 0..15:"); for (int i = 0; i < 16; i++) printf(" %u", h[i]);
    printf("\nblocks with xcc == (b + xcc0) %% 8: %d / 4096\n", match);
    uint32_t *out; CK(hipMalloc(&out, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (size_t mb : {8, 16, 24, 32, 64}) {
        for (int nparts : {1, 8}) {
            const size_t bytes = mb << 20; uint4 *tab; CK(hipMalloc(&tab, bytes)); CK(hipMemset(tab, 1, bytes));
            const uint64_t part_elems = bytes / 16 / nparts;
            uint64_t pm = 1; while (pm * 2 <= part_elems) pm *= 2;  // power-of-two window inside the part
            const int rounds = 256, blocks = 2048;
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL((k_part<uint4, 8>), dim3(blocks), dim3(256), 0, 0, tab, pm - 1, part_elems, nparts, rounds, out);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            }
            printf("table %3zu MiB, %d partition(s) (%.1f MiB touched per XCD): %7.1f G probes/s\n", mb, nparts,
                   (double)pm * 16 / 1048576.0, (double)blocks * 256 * rounds * 8 / ms / 1e6);
            CK(hipFree(tab));
        }
    }
    return 0;
}

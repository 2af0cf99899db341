// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes this engine uses
// (the guide: FETCH_SIZE reads 1/2 of a wide coalesced stream; other widths are uncalibrated).
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; every kernel moves a known byte count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x *= 0x9E3779B97F4A7C15ULL; return x ^ (x >> 31); }

__global__ void calib_read8_1GiB(const uint64_t *p, uint64_t n, uint64_t *out) {  // 8 B per lane, coalesced
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc += __builtin_nontemporal_load(p + i);
    if (acc == 0x1234) out[0] = acc;
}
__global__ void calib_read16_1GiB(const uint4 *p, uint64_t n, uint64_t *out) {  // 16 B per lane, coalesced
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc += p[i].x;
    if (acc == 0x1234) out[0] = acc;
}
__global__ void calib_rand16_64Mprobes(const uint4 *p, uint64_t mask, int per_thread, uint64_t *out) {  // random 16 B probes, 1 GiB table
    uint64_t s = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x632BE59BD9B4E019ULL + 1, acc = 0;
    for (int i = 0; i < per_thread; i++) { s = mix(s + i); acc += p[(s >> 20) & mask].x; }
    if (acc == 0x1234) out[0] = acc;
}
__global__ void calib_write16_1GiB(uint4 *p, uint64_t n) {  // 16 B per lane stores
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4);
}
__global__ void calib_write48_scattered_48MB(uint4 *p, uint64_t nrec) {  // 48-byte records, 3 x 16 B by 3 lanes of a wave (PartRec shape)
    const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w < nrec && lane < 3) p[w * 3 + lane] = make_uint4(1, 2, 3, 4);
}
__global__ void calib_read4_1GiB(const uint32_t *p, uint64_t n, uint64_t *out) {  // 4 B per lane, coalesced
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x1234) out[0] = acc;
}
// the join probe's shape: a wave reads 64 consecutive dwords that start at an arbitrary dword (not line-aligned),
// two runs of 32 per wave-load taken from places 32 KiB apart; 1/4 of the buffer is read (every 4th 128-byte run)
__global__ void calib_read4_segments_256MiB(const uint32_t *p, uint64_t nseg, uint64_t *out) {
    uint64_t acc = 0;
    const int lane = threadIdx.x & 63;
    for (uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w * 2 < nseg; w += ((uint64_t)gridDim.x * blockDim.x) >> 6) {
        const uint64_t seg = w * 2 + (lane >> 5);                // 32-dword segment
        const uint64_t base = seg * 128 + ((seg * 7) & 31);      // in a 128-dword (512 B) cell, at an odd offset
        acc += p[base + (lane & 31)];
    }
    if (acc == 0x1234) out[0] = acc;
}
__global__ void calib_write4_compact_256MiB(uint32_t *p, uint64_t n) {  // 4 B per lane, coalesced stores (hit streams)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
int main() {
    const uint64_t bytes = 1ull << 30;
    void *buf; uint64_t *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 8)); CK(hipMemset(buf, 1, bytes));
    hipLaunchKernelGGL(calib_read8_1GiB, dim3(4096), dim3(256), 0, 0, (const uint64_t *)buf, bytes / 8, out);
    hipLaunchKernelGGL(calib_read16_1GiB, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, out);
    hipLaunchKernelGGL(calib_rand16_64Mprobes, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16 - 1, 64, out);
    hipLaunchKernelGGL(calib_write16_1GiB, dim3(4096), dim3(256), 0, 0, (uint4 *)buf, bytes / 16);
    hipLaunchKernelGGL(calib_write48_scattered_48MB, dim3(1 << 20 >> 2), dim3(256), 0, 0, (uint4 *)buf, (uint64_t)1 << 20);
    hipLaunchKernelGGL(calib_read4_1GiB, dim3(4096), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 4, out);
    hipLaunchKernelGGL(calib_read4_segments_256MiB, dim3(4096), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 512 - 1, out);
    hipLaunchKernelGGL(calib_write4_compact_256MiB, dim3(4096), dim3(256), 0, 0, (uint32_t *)buf, bytes / 16);
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}

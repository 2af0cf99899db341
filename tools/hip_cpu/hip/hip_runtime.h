// CPU stand-in for the slice of the HIP runtime that csrc/gs_bgzf.hip and csrc/gs_ingest.hip use -- TEST INFRASTRUCTURE
// (tools/bgzf_cpu_asan.sh, tools/ingest_cpu_asan.sh), never part of the product library.  The GPU pool refuses sanitizer runs, and an out-of-bounds access of a kernel on a damaged `.gz` is at
// best silent there and at worst a fault that resets the node: with this header in front of the real one the SAME source file --
// its host code and its kernels, unchanged -- compiles as plain C++ (clang++: the kernels use ext_vector_type) and runs under
// AddressSanitizer + UBSan on the host:
//   * device and pinned memory are malloc'ed blocks of exactly the sizes the code asks for (a read or a store one byte past a
//     buffer is ASan's to report), copies are memcpy, streams and events do nothing (everything is synchronous);
//   * a kernel launch runs the grid's blocks one after the other, the threads of a block as REAL threads: __syncthreads is a
//     barrier of the block's threads that have not returned yet, __shfl_xor an exchange through a per-block array between two
//     barriers, `__shared__` a static array (one block runs at a time), atomics are the host's.
// Nothing here knows what the kernels compute.
#pragma once

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __constant__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define HIP_SYMBOL(x) (&(x))

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hip_cpu_idx {
    unsigned x = 0, y = 0, z = 0;
};
inline thread_local hip_cpu_idx threadIdx, blockIdx, blockDim, gridDim;

// ---- the runtime API ------------------------------------------------------------------------------------------------
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0, hipErrorOutOfMemory = 2;
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
constexpr unsigned hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2;
struct hipDeviceProp_t {
    int multiProcessorCount = 256;
    size_t maxSharedMemoryPerMultiProcessor = 160 << 10;
};

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "out of memory (CPU stand-in)"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    *p = hipDeviceProp_t();
    return hipSuccess;
}
inline hipError_t hipMalloc(void **p, size_t n) {
    *p = malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
inline hipError_t hipFree(void *p) {
    free(p);
    return hipSuccess;
}
inline hipError_t hipHostFree(void *p) { return hipFree(p); }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
    memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
    memmove(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
    memset(d, v, n);
    return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n) { return hipMemsetAsync(d, v, n, nullptr); }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMemcpyFromSymbol(void *d, const void *sym, size_t n) {
    memcpy(d, sym, n);
    return hipSuccess;
}
inline hipError_t hipMemcpyToSymbol(void *sym, const void *s, size_t n) {
    memcpy(sym, s, n);
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
    *s = nullptr;
    return hipSuccess;
}
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) {
    *e = nullptr;
    return hipSuccess;
}
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
    *ms = 0.f;
    return hipSuccess;
}

// ---- a block's threads -----------------------------------------------------------------------------------------------------
namespace hip_cpu {

struct Block {
    std::mutex m;
    std::condition_variable cv;
    unsigned threads = 0;  // of the block
    unsigned active = 0;   // that have not returned from the kernel yet (the ones __syncthreads waits for)
    unsigned waiting = 0, gen = 0;
    unsigned done = 0, done_gen = 0;  // the barrier between two blocks
    std::vector<uint64_t> xch;        // __shfl_xor's values
    void release() {
        waiting = 0;
        gen++;
        cv.notify_all();
    }
    void sync() {
        std::unique_lock<std::mutex> lk(m);
        if (++waiting == active) {
            release();
            return;
        }
        const unsigned g = gen;
        cv.wait(lk, [&] { return gen != g; });
    }
    void leave() {  // the thread returned from the kernel
        std::unique_lock<std::mutex> lk(m);
        active--;
        if (waiting && waiting == active) release();
    }
    void next_block() {  // every thread of the block, after leave(): the next block starts with all of them
        std::unique_lock<std::mutex> lk(m);
        if (++done == threads) {
            done = 0;
            active = threads;
            done_gen++;
            cv.notify_all();
            return;
        }
        const unsigned g = done_gen;
        cv.wait(lk, [&] { return done_gen != g; });
    }
};
inline thread_local Block *cur = nullptr;

template <typename K, typename... A>
void launch(K kernel, dim3 grid, dim3 block, A... args) {
    Block b;
    b.threads = b.active = block.x * block.y * block.z;
    b.xch.assign(b.threads, 0);
    auto body = [&](unsigned t) {
        cur = &b;
        blockDim.x = block.x, blockDim.y = block.y, blockDim.z = block.z;
        gridDim.x = grid.x, gridDim.y = grid.y, gridDim.z = grid.z;
        threadIdx.x = t % block.x, threadIdx.y = (t / block.x) % block.y, threadIdx.z = t / (block.x * block.y);
        for (unsigned bz = 0; bz < grid.z; bz++)
            for (unsigned by = 0; by < grid.y; by++)
                for (unsigned bx = 0; bx < grid.x; bx++) {
                    blockIdx.x = bx, blockIdx.y = by, blockIdx.z = bz;
                    kernel(args...);
                    b.leave();
                    b.next_block();
                }
        cur = nullptr;
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < b.threads; t++) pool.emplace_back(body, t);
    body(0);
    for (auto &th : pool) th.join();
}

}  // namespace hip_cpu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hip_cpu::launch(kernel, grid, block, __VA_ARGS__)

inline void __syncthreads() { hip_cpu::cur->sync(); }
// (every thread of the block that has not returned takes part: the kernels here shuffle before any thread leaves)
template <typename T, typename F>
inline T hip_cpu_exchange(T v, F partner_of) {
    static_assert(sizeof(T) <= 8, "32- and 64-bit values");
    hip_cpu::Block &b = *hip_cpu::cur;
    const unsigned me = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    b.xch[me] = bits;
    b.sync();
    const unsigned partner = partner_of(me);
    T r = v;
    if (partner < b.threads) memcpy(&r, &b.xch[partner], sizeof(T));
    b.sync();
    return r;
}
template <typename T>
inline T __shfl_xor(T v, int lane_mask, int width = 64) {
    return hip_cpu_exchange(v, [=](unsigned me) { return (me & ~(unsigned)(width - 1)) | ((me ^ (unsigned)lane_mask) & (unsigned)(width - 1)); });
}
template <typename T>
inline T __shfl_up(T v, unsigned delta, int width = 64) {  // (a lane below `delta` keeps its own value)
    return hip_cpu_exchange(v, [=](unsigned me) { return (me & (unsigned)(width - 1)) >= delta ? me - delta : me; });
}
inline int __popc(uint32_t v) { return __builtin_popcount(v); }
inline int __ffs(uint32_t v) { return __builtin_ffs((int)v); }
struct alignas(16) uint4 {
    uint32_t x, y, z, w;
};
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
inline uint32_t __brev(uint32_t v) { return __builtin_bitreverse32(v); }
inline unsigned long long wall_clock64() { return 0; }
template <typename T>
inline T atomicAdd(T *p, T v) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
template <typename T>
inline T atomicOr(T *p, T v) {
    return __atomic_fetch_or(p, v, __ATOMIC_RELAXED);
}
template <typename T>
inline T atomicMax(T *p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return old;
}
inline uint32_t atomicCAS(uint32_t *p, uint32_t expect, uint32_t v) {
    __atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return expect;
}

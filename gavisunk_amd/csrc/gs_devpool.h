// Device memory and the stream of the second half (gs_chain.hip, gs_diag.hip, gs_pair.hip).
//
// Every hipMalloc / hipFree synchronises the whole device, and the second half used to make ~150 of them per call: with the
// tagger's next pass running on its own streams beside it, each one would stall both.  So the temporaries of the chain come
// from a caching pool (blocks go back to the pool, not to the driver; gs_device_pool_trim hands them back), and every kernel,
// scan, sort and copy of the chain runs on ONE non-blocking stream per device: a block is reused only by work queued behind
// the work that used it before, so handing it out again needs no wait.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <map>
#include <mutex>
#include <vector>

#include "gs_common.h"

namespace gs {

hipStream_t chain_stream(int device);         // the second half's stream on this device (created on first use; gs_chain.hip)
void *pool_alloc(int device, size_t bytes);   // nullptr (and set_error) when the device has no room
void pool_free(int device, void *p);          // back to the pool
void pool_trim(int device);                   // cached blocks back to the driver (synchronises)

// allocations of one call: released together when the call ends
struct PoolScope {
    int device;
    std::vector<void *> ptrs;
    explicit PoolScope(int dev) : device(dev) {}
    PoolScope(const PoolScope &) = delete;
    PoolScope &operator=(const PoolScope &) = delete;
    ~PoolScope() {
        for (void *p : ptrs)
            if (p) pool_free(device, p);
    }
    template <typename T>
    int get(T **d, uint64_t n) {
        *d = static_cast<T *>(pool_alloc(device, std::max<uint64_t>(n, 1) * sizeof(T)));
        if (!*d) return GS_E_NOMEM;
        ptrs.push_back(*d);
        return GS_OK;
    }
    template <typename T>
    int put(T **d, const T *h, uint64_t n) {  // host array -> device, behind the chain stream's earlier work; the host array is free on return
        int rc = get(d, n);
        if (rc) return rc;
        if (h && n) {
            hipError_t e = hipMemcpyAsync(*d, h, n * sizeof(T), hipMemcpyHostToDevice, chain_stream(device));
            if (e == hipSuccess) e = hipStreamSynchronize(chain_stream(device));
            if (e != hipSuccess) {
                set_error("copy to the device failed: %s", hipGetErrorString(e));
                return GS_E_NODEVICE;
            }
        }
        return GS_OK;
    }
    void release(void *p) {  // early
        for (auto &q : ptrs)
            if (q == p && p) {
                pool_free(device, p);
                q = nullptr;
            }
    }
};

}  // namespace gs

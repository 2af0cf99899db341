// CPU stand-in (see ../../hip/hip_runtime.h): rocprim::exclusive_scan with its two-call protocol, in order, in == out allowed.
#pragma once
#include <hip/hip_runtime.h>
namespace rocprim {
template <typename In, typename Out, typename Init, typename Op>
inline hipError_t exclusive_scan(void *temp, size_t &bytes, In in, Out out, Init init, size_t n, Op op, hipStream_t = nullptr, bool = false) {
    if (!temp) {
        bytes = 64;
        return hipSuccess;
    }
    auto acc = init;
    for (size_t i = 0; i < n; i++) {
        const auto v = in[i];
        out[i] = acc;
        acc = op(acc, v);
    }
    return hipSuccess;
}
}  // namespace rocprim

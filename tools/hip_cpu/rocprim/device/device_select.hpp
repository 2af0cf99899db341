// CPU stand-in (see ../../hip/hip_runtime.h): rocprim::select by flags.
#pragma once
#include <hip/hip_runtime.h>
namespace rocprim {
template <typename In, typename Flags, typename Out, typename Count>
inline hipError_t select(void *temp, size_t &bytes, In in, Flags flags, Out out, Count count, size_t n, hipStream_t = nullptr, bool = false) {
    if (!temp) {
        bytes = 64;
        return hipSuccess;
    }
    size_t k = 0;
    for (size_t i = 0; i < n; i++)
        if (flags[i]) out[k++] = in[i];
    *count = k;
    return hipSuccess;
}
}  // namespace rocprim

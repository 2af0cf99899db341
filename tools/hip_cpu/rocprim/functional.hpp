#pragma once
namespace rocprim {
template <typename T>
struct plus {
    T operator()(const T &a, const T &b) const { return a + b; }
};
}  // namespace rocprim

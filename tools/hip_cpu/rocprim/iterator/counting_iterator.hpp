#pragma once
#include <cstddef>
namespace rocprim {
template <typename T>
struct counting_iterator {
    T first;
    explicit counting_iterator(T f) : first(f) {}
    T operator[](size_t i) const { return first + (T)i; }
};
}  // namespace rocprim

// TEST INFRASTRUCTURE ONLY (see tests/emu/hip/hip_runtime.h): rocprim's device scans, sequentially.
#pragma once
#include <cstddef>

namespace rocprim {
template <typename T> struct plus { T operator()(const T &a, const T &b) const { return a + b; } };
template <typename T, typename Op>
hipError_t inclusive_scan(void *temp, size_t &bytes, const T *in, T *out, size_t n, Op op, hipStream_t) {
    if (!temp) { bytes = 64; return hipSuccess; }
    T acc{};
    for (size_t i = 0; i < n; i++) { acc = i ? op(acc, in[i]) : in[i]; out[i] = acc; }
    return hipSuccess;
}
template <typename T, typename Op>
hipError_t exclusive_scan(void *temp, size_t &bytes, const T *in, T *out, T init, size_t n, Op op, hipStream_t) {
    if (!temp) { bytes = 64; return hipSuccess; }
    T acc = init;
    for (size_t i = 0; i < n; i++) { const T v = in[i]; out[i] = acc; acc = op(acc, v); }
    return hipSuccess;
}
}  // namespace rocprim

// TEST INFRASTRUCTURE ONLY (see tests/emu/hip/hip_runtime.h): rocprim::radix_sort_pairs as a stable sort on the key bits.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>

namespace rocprim {
template <typename K, typename V>
hipError_t radix_sort_pairs(void *temp, size_t &bytes, const K *kin, K *kout, const V *vin, V *vout, size_t n, unsigned begin_bit,
                            unsigned end_bit, hipStream_t) {
    if (!temp) { bytes = 64; return hipSuccess; }
    const K mask = end_bit >= 8 * sizeof(K) ? ~K(0) : ((K(1) << end_bit) - 1);
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), size_t(0));
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ((kin[a] & mask) >> begin_bit) < ((kin[b] & mask) >> begin_bit); });
    std::vector<K> ks(n);
    std::vector<V> vs(n);
    for (size_t i = 0; i < n; i++) { ks[i] = kin[idx[i]]; vs[i] = vin[idx[i]]; }
    std::copy(ks.begin(), ks.end(), kout);
    std::copy(vs.begin(), vs.end(), vout);
    return hipSuccess;
}
}  // namespace rocprim

namespace rocprim {
template <typename K>
hipError_t radix_sort_keys(void *temp, size_t &bytes, const K *kin, K *kout, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t) {
    if (!temp) { bytes = 64; return hipSuccess; }
    const K mask = end_bit >= 8 * sizeof(K) ? ~K(0) : ((K(1) << end_bit) - 1);
    std::vector<K> ks(kin, kin + n);
    std::stable_sort(ks.begin(), ks.end(), [&](K a, K b) { return ((a & mask) >> begin_bit) < ((b & mask) >> begin_bit); });
    std::copy(ks.begin(), ks.end(), kout);
    return hipSuccess;
}
}  // namespace rocprim

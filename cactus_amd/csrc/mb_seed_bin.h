// mb_seed_bin.h -- the seed hits of a strand grouped by diagonal WITHOUT a device-wide sort (gfx950, wave64): the keys are dealt into bins
// by the top bits of their (scrambled) diagonal, and a work-group per bin orders its bin in LDS.  Included by mb_kernels.hip inside
// namespace mb after mb_seed_dense.h (and, with MB_EMU defined, by tests/emu/emu_seed_dense.cpp).
//
// What the ungapped stage wants (mb_ungapped_ux.h, mb_runs.h): the hits of a diagonal side by side and in q order.  rocprim's radix sort
// gave that with four passes over the keys (16 B per key and pass, plus the histogram, plus k_keys_unhash: ~ 96 B per key) and was a
// fifth of the kernel time of a human x mouse chunk pair.  The keys need no ORDER between diagonals, only grouping -- and the scrambled
// diagonal (d * C mod 2^B, mb_seed_dense.h) spreads real homology evenly over its top bits:
//   k_bin_count    histogram of the top `nbits` bits of the scrambled diagonal (LDS histogram per work-group, one flush).  nbits follows
//                  from the number of keys, which the device knows before the host does: the kernel reads it where k_seed_hits left it
//   k_bin_scan     one work-group: the bins' places (exclusive scan), their write cursors, the largest bin and the number of bins beyond
//                  the small sorter's room -- the plan the host reads back with the strand's hit count (no extra synchronisation)
//   k_bin_scatter  a work-group per chunk of 32 768 keys: LDS histogram of the chunk, one returning atomic per non-empty bin to reserve
//                  the chunk's stretch of each bin, then the keys go to their bins (order inside a bin: whatever the atomics gave)
//   k_bin_sort     a work-group per bin: the bin's keys into LDS by BUCKET = the next 11 / 12 bits of the scrambled diagonal (count, scan,
//                  place), then every key finds its rank among the keys of its bucket ((diagonal, q) ascending; a bucket holds a key or
//                  two, a diagonal of real homology some hundreds) and is written to that place with its diagonal unscrambled.
//                  Two instantiations: bins of up to 4 096 keys (48 KB of LDS, three work-groups per CU) and of up to 16 384 (144 KB).
// The result is the SAME array rocprim's stable sort by the scrambled diagonal followed by k_keys_unhash gives (bins, buckets and ranks
// are all ascending in (scrambled diagonal, q), and a (diagonal, q) pair occurs once), so everything behind it is untouched and the two
// paths are compared key by key in the tests.  24 B of reads + 16 B of writes per key instead of ~ 96.
// A strand whose largest bin does not fit the large sorter (one diagonal with > ~ 13 000 hits: a self alignment; or diagonals left
// unscrambled) goes through rocprim as before: the host sees the plan before it queues either.
#pragma once

constexpr int kBinBitsMax = 13;                // at most 8 192 bins (the LDS histograms of k_bin_count / k_bin_scatter: 32 KB)
constexpr int kBinMeanDefault = 2800;          // bins are as few as keep the mean bin at or below this many keys (Poisson spread stays within 4 096)
constexpr int kBinCapSmall = 4096, kBinCapBig = 16384;
constexpr int kBinChunk = 32768;               // keys per work-group of k_bin_scatter
constexpr int kBinPlanWords = 8;               // u32 words at the head of a strand's bin state
// bin state of a strand (u32 words): [0] nbits [1] largest bin [2] bins beyond kBinCapSmall [3] keys (low word) [4] 1 if the keys did not
// fit their buffer (nothing was counted); [8, 8 + 8192) counts; then 8193 starts; then 8192 cursors
constexpr int kBinStateWords = kBinPlanWords + (1 << kBinBitsMax) + (1 << kBinBitsMax) + 8 + (1 << kBinBitsMax);
__host__ __device__ __forceinline__ uint32_t *bin_counts(uint32_t *state) { return state + kBinPlanWords; }
__host__ __device__ __forceinline__ uint32_t *bin_starts(uint32_t *state) { return state + kBinPlanWords + (1 << kBinBitsMax); }
__host__ __device__ __forceinline__ uint32_t *bin_cursors(uint32_t *state) { return state + kBinPlanWords + 2 * (1 << kBinBitsMax) + 8; }

__host__ __device__ __forceinline__ int bin_bits(const unsigned long long n, const int diag_bits, const int mean) {
    int b = 0;
    while (b < kBinBitsMax && b < diag_bits && (n >> b) > (unsigned long long)mean) b++;
    return b;
}
__device__ __forceinline__ uint32_t bin_of(const unsigned long long key, const int diag_bits, const int nbits) {
    return nbits ? (uint32_t)(key >> 32) >> (diag_bits - nbits) : 0u;
}

__global__ __launch_bounds__(1024) void k_bin_count(const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ n_ptr,
                                                     const unsigned long long cap, const int diag_bits, const int mean, uint32_t *__restrict__ state) {
    __shared__ uint32_t hist[1 << kBinBitsMax];
    const unsigned long long n = *n_ptr;
    if (n > cap || n >= (1ull << 31)) {                                 // (the keys were not all written: the host searches again with more room)
        if (blockIdx.x == 0 && threadIdx.x == 0) state[4] = 1u;
        return;
    }
    const int nbits = bin_bits(n, diag_bits, mean), nb = 1 << nbits;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) hist[b] = 0u;
    __syncthreads();
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        atomicAdd(&hist[bin_of(keys[i], diag_bits, nbits)], 1u);
    __syncthreads();
    uint32_t *counts = bin_counts(state);
    for (int b = threadIdx.x; b < nb; b += blockDim.x) { const uint32_t c = hist[b]; if (c) atomicAdd(&counts[b], c); }
}

// one work-group of 1024: eight bins per work-item
__global__ __launch_bounds__(1024) void k_bin_scan(const unsigned long long *__restrict__ n_ptr, const unsigned long long cap, const int diag_bits, const int mean,
                                                    uint32_t *__restrict__ state) {
    constexpr int kPer = (1 << kBinBitsMax) / 1024;
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t s_max, s_big;
    const unsigned long long n = *n_ptr;
    if (n > cap || n >= (1ull << 31)) return;
    const int nbits = bin_bits(n, diag_bits, mean), nb = 1 << nbits;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { s_max = 0u; s_big = 0u; }
    const uint32_t *counts = bin_counts(state);
    uint32_t *starts = bin_starts(state), *cursors = bin_cursors(state);
    uint32_t c[kPer], sum = 0, mx = 0, big = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int b = tid * kPer + k;
        c[k] = b < nb ? counts[b] : 0u;
        sum += c[k]; mx = c[k] > mx ? c[k] : mx; big += c[k] > (uint32_t)kBinCapSmall ? 1u : 0u;
    }
    const uint32_t incl = (uint32_t)dpp_scan_add((int)sum);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) before += k < wv ? wsum[k] : 0u;
    uint32_t at = before + incl - sum;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int b = tid * kPer + k;
        if (b < nb) { starts[b] = at; cursors[b] = at; }
        at += c[k];
    }
    if (mx) atomicMax(&s_max, mx);
    if (big) atomicAdd(&s_big, big);
    __syncthreads();
    if (tid == 0) { state[0] = (uint32_t)nbits; state[1] = s_max; state[2] = s_big; state[3] = (uint32_t)n; starts[nb] = (uint32_t)n; }
}

__global__ __launch_bounds__(1024) void k_bin_scatter(const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out, const int64_t n,
                                                       const int diag_bits, const int nbits, uint32_t *__restrict__ state) {
    __shared__ uint32_t base[1 << kBinBitsMax];
    const int nb = 1 << nbits;
    const int64_t i0 = (int64_t)blockIdx.x * kBinChunk, i1 = i0 + kBinChunk < n ? i0 + kBinChunk : n;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) base[b] = 0u;
    __syncthreads();
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) atomicAdd(&base[bin_of(in[i], diag_bits, nbits)], 1u);
    __syncthreads();
    uint32_t *cursors = bin_cursors(state);
    for (int b = threadIdx.x; b < nb; b += blockDim.x) { const uint32_t c = base[b]; if (c) base[b] = atomicAdd(&cursors[b], c); }
    __syncthreads();
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const unsigned long long key = in[i];                              // (the chunk is 256 KB: this read comes from the L2)
        out[atomicAdd(&base[bin_of(key, diag_bits, nbits)], 1u)] = key;
    }
}

// A bin in LDS.  Buckets: the kBk bits below the bin bits of the scrambled diagonal (all of the rest of it, if that is fewer).
template <int kCap, int kBk, int kThreads>
__global__ __launch_bounds__(kThreads) void k_bin_sort(const unsigned long long *in, unsigned long long *out, const uint32_t *__restrict__ state,
                                                        const int diag_bits, const int nbits, const uint32_t hinv, const uint32_t hmask) {
    constexpr int kPer = kCap / kThreads, kNbk = 1 << kBk, kCPer = kNbk / kThreads;
    static_assert(kCap % kThreads == 0 && kNbk % kThreads == 0, "whole keys and counters per work-item");
    __shared__ unsigned long long sk[kCap];
    __shared__ uint32_t cur[kNbk];
    __shared__ uint32_t wsum[kThreads / 64];
    const uint32_t *starts = state + kBinPlanWords + (1 << kBinBitsMax);
    const uint32_t lo = starts[blockIdx.x], m = starts[blockIdx.x + 1] - lo;
    if (m == 0u || m > (uint32_t)kCap || (kCap > kBinCapSmall && m <= (uint32_t)kBinCapSmall)) return;      // (empty, or the other instantiation's, or nobody's: the host did not come here then)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = diag_bits - nbits, bshift = r > kBk ? r - kBk : 0;
    const uint32_t rmask = r >= 32 ? 0xFFFFFFFFu : ((1u << r) - 1u);
    auto bucket = [&](const unsigned long long key) { return (((uint32_t)(key >> 32)) & rmask) >> bshift; };
#pragma unroll
    for (int k = 0; k < kCPer; k++) cur[tid * kCPer + k] = 0u;
    unsigned long long key[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const uint32_t i = (uint32_t)(tid + k * kThreads);
        key[k] = i < m ? in[(size_t)lo + i] : ~0ull;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; k++) if ((uint32_t)(tid + k * kThreads) < m) atomicAdd(&cur[bucket(key[k])], 1u);
    __syncthreads();
    {   // counts -> places: kCPer counters per work-item
        uint32_t c[kCPer], sum = 0;
#pragma unroll
        for (int k = 0; k < kCPer; k++) { c[k] = cur[tid * kCPer + k]; sum += c[k]; }
        const uint32_t incl = (uint32_t)dpp_scan_add((int)sum);
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t at = incl - sum;
#pragma unroll
        for (int k = 0; k < kThreads / 64; k++) at += k < wv ? wsum[k] : 0u;
#pragma unroll
        for (int k = 0; k < kCPer; k++) { cur[tid * kCPer + k] = at; at += c[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; k++) if ((uint32_t)(tid + k * kThreads) < m) sk[atomicAdd(&cur[bucket(key[k])], 1u)] = key[k];
    __syncthreads();
    // cur[b] is now the END of bucket b (its beginning: the end of the bucket before it)
#pragma unroll 1
    for (uint32_t i = (uint32_t)tid; i < m; i += (uint32_t)kThreads) {
        const unsigned long long mine = sk[i];
        const uint32_t b = bucket(mine), b0 = b ? cur[b - 1] : 0u, b1 = cur[b];
        uint32_t rank = 0;
        for (uint32_t j = b0; j < b1; j++) rank += sk[j] < mine ? 1u : 0u;
        out[(size_t)lo + b0 + rank] = ((unsigned long long)((((uint32_t)(mine >> 32)) * hinv) & hmask) << 32) | (uint32_t)mine;
    }
}

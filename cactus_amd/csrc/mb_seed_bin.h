// mb_seed_bin.h -- the seed hits of a strand grouped by diagonal WITHOUT a device-wide sort (gfx950, wave64): the keys are dealt into bins
// by the top bits of their (scrambled) diagonal, and a work-group per bin orders its bin in LDS.  Included by mb_kernels.hip inside
// namespace mb after mb_seed_dense.h (and, with MB_EMU defined, by tests/emu/emu_seed_dense.cpp).
//
// What the ungapped stage wants (mb_ungapped_ux.h, mb_runs.h): the hits of a diagonal side by side and in q order.  rocprim's radix sort
// gave that with four passes over the keys (16 B per key and pass, plus the histogram, plus k_keys_unhash: ~ 96 B per key) and was a
// fifth of the kernel time of a human x mouse chunk pair.  The keys need no ORDER between diagonals, only grouping -- and the scrambled
// diagonal (d * C mod 2^B, mb_seed_dense.h) spreads real homology evenly over its top bits:
//   k_bin_count    a work-group per chunk of 16 384 keys: LDS histogram of the top `nbits` bits of the scrambled diagonal, written as the
//                  chunk's row of a (chunks x bins) matrix.  nbits follows from the number of keys, which the device knows before the
//                  host does: the kernel reads it where k_seed_hits left it
//   k_bin_scan     a work-item per bin runs down its column (the row entries become "keys of this bin in earlier chunks"); the last
//                  work-group to finish scans the bins' totals: their places, the largest bin and the number of bins beyond the small
//                  sorter's room -- the plan the host reads back with the strand's hit count (no extra synchronisation)
//   k_bin_scatter_staged   a work-group per chunk: the chunk dealt into its bins IN LDS (histogram, scan, place), then written out bin by bin
//                  to the places its matrix row gives -- runs of 64-128 B per bin instead of one 8-B store per key and cache line.  (No atomic
//                  on device memory anywhere.  k_bin_scatter: the same without the staging, for more than 2 048 bins.)
//   k_bin_sort     a work-group per bin: the bin's keys into LDS by BUCKET = the next 11 / 12 bits of the scrambled diagonal (count, scan,
//                  place).  A bucket holds a key or two: every key finds its rank among the keys of its bucket ((diagonal, q) ascending)
//                  and is written to that place with its diagonal unscrambled.  A bucket with more than 32 keys is a diagonal of real
//                  homology (hundreds of hits, thousands between near-identical sequences): those are sorted in place by a bitonic
//                  network first -- one wave per run of up to 2 048 keys, the whole work-group for the longer ones -- because ranking
//                  by counting is quadratic in the run.
//                  Two instantiations: bins of up to 4 096 keys (41 KB of LDS, three work-groups per CU) and of up to 16 384 (146 KB).
// The result is the SAME array rocprim's stable sort by the scrambled diagonal followed by k_keys_unhash gives (bins, buckets and ranks
// are all ascending in (scrambled diagonal, q), and a (diagonal, q) pair occurs once), so everything behind it is untouched and the two
// paths are compared key by key in the tests.  24 B read + 16 B written per key (PMC: 43 B with the matrix and the scatter's partial lines) instead of ~ 96;
// standing alone 192 us per strand of 11.5 million keys against rocprim's 289 (MI355X, profiles/README.md).
// A strand whose largest bin does not fit the large sorter (one diagonal with more than ~ 8 000 hits on top of a full bin: a self alignment; or
// diagonals left unscrambled), or of more than 2^26 keys, goes through rocprim as before: the host sees the plan before it queues either.
#pragma once

constexpr int kBinBitsMax = 13;                // at most 8 192 bins (the LDS histograms of k_bin_count / k_bin_scatter: 32 KB)
constexpr int kBinMeanDefault = 11000;         // bins are as few as keep the mean bin at or below this many keys: the large sorter's, with room for the diagonals of real homology
constexpr int kBinCapSmall = 4096, kBinCapBig = 16384;
constexpr int kBinChunk = 16384;               // keys per work-group of k_bin_count / k_bin_scatter: what the scatter stages in LDS
constexpr unsigned long long kBinKeysMax = 1ull << 26;     // more keys than this: no plan (the matrix would outgrow its use; the mean bin the large sorter)
constexpr int kBinRunShort = 32;               // a bucket of up to this many keys is ranked by counting
constexpr int kBinRunWave = 2048;              // a longer one is sorted by one wave, beyond this size by the work-group
constexpr int kBinPlanWords = 8;               // u32 words at the head of a strand's bin state
// bin state of a strand (u32 words): [0] nbits [1] largest bin [2] bins beyond kBinCapSmall [3] keys [4] 1 if there is no plan (the keys did
// not fit their buffer, or are too many) [5] work-groups of k_bin_scan that are done [6] non-empty bins within kBinCapSmall; [8, 8 + 8192) keys per bin; then 8193 places
constexpr int kBinStateWords = kBinPlanWords + (1 << kBinBitsMax) + (1 << kBinBitsMax) + 8;
__host__ __device__ __forceinline__ uint32_t *bin_counts(uint32_t *state) { return state + kBinPlanWords; }
__host__ __device__ __forceinline__ uint32_t *bin_starts(uint32_t *state) { return state + kBinPlanWords + (1 << kBinBitsMax); }

__host__ __device__ __forceinline__ int bin_bits(const unsigned long long n, const int diag_bits, const int mean) {
    int b = 0;
    while (b < kBinBitsMax && b < diag_bits && (n >> b) > (unsigned long long)mean) b++;
    return b;
}
// u32 words of the (chunks x bins) matrix for up to `cap` keys
__host__ __device__ __forceinline__ unsigned long long bin_matrix_words(unsigned long long cap, const int diag_bits, const int mean) {
    if (cap > kBinKeysMax) cap = kBinKeysMax;
    return ((cap + kBinChunk - 1) / kBinChunk) << bin_bits(cap, diag_bits, mean);
}
__device__ __forceinline__ uint32_t bin_of(const unsigned long long key, const int diag_bits, const int nbits) {
    return nbits ? (uint32_t)(key >> 32) >> (diag_bits - nbits) : 0u;
}
__device__ __forceinline__ bool bin_no_plan(const unsigned long long n, const unsigned long long cap) { return n > cap || n > kBinKeysMax; }

#ifdef MB_EMU
#define MB_WAVE_SYNC() emu_wave_sync()
#else
// LDS traffic of ONE wave: its DS instructions execute in the order they were issued; the fence keeps the compiler from moving them
#define MB_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

// grid: one work-group (of 256: four waves find room on a CU that others share; sixteen wait for a free one) per chunk the key buffer could hold
// (those behind the last key leave at once)
__global__ __launch_bounds__(256) void k_bin_count(const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ n_ptr,
                                                     const unsigned long long cap, const int diag_bits, const int mean, uint32_t *__restrict__ state,
                                                     uint32_t *__restrict__ matrix) {
    __shared__ uint32_t hist[1 << kBinBitsMax];
    const unsigned long long n = *n_ptr;
    if (bin_no_plan(n, cap)) {                                         // (the keys were not all written: the host searches again with more room -- or they are too many)
        if (blockIdx.x == 0 && threadIdx.x == 0) state[4] = 1u;
        return;
    }
    const unsigned long long i0 = (unsigned long long)blockIdx.x * kBinChunk, i1 = i0 + kBinChunk < n ? i0 + kBinChunk : n;
    if (i0 >= n) return;
    const int nbits = bin_bits(n, diag_bits, mean), nb = 1 << nbits;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) hist[b] = 0u;
    __syncthreads();
    for (unsigned long long i = i0 + threadIdx.x; i < i1; i += blockDim.x) atomicAdd(&hist[bin_of(keys[i], diag_bits, nbits)], 1u);
    __syncthreads();
    uint32_t *row = matrix + ((size_t)blockIdx.x << nbits);
    for (int b = threadIdx.x; b < nb; b += blockDim.x) row[b] = hist[b];
}

// grid: 8192 / 256 work-groups of 256 -- a work-item per bin; the last work-group to finish makes the plan
__global__ __launch_bounds__(256) void k_bin_scan(const unsigned long long *__restrict__ n_ptr, const unsigned long long cap, const int diag_bits, const int mean,
                                                   uint32_t *state, uint32_t *__restrict__ matrix) {
    constexpr int kPer = (1 << kBinBitsMax) / 256;
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_max, s_big, s_small, s_ticket;
    const unsigned long long n = *n_ptr;
    if (bin_no_plan(n, cap)) return;
    const int nbits = bin_bits(n, diag_bits, mean), nb = 1 << nbits;
    const uint32_t n_chunks = (uint32_t)((n + kBinChunk - 1) / kBinChunk);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t *counts = bin_counts(state), *starts = bin_starts(state);
    {
        const int b = (int)(blockIdx.x * blockDim.x) + tid;
        if (b < nb) {
            uint32_t run = 0, c = 0;
            for (; c + 32 <= n_chunks; c += 32) {                        // (the loads of a round do not wait for each other: 700 chunks are 22 round trips)
                uint32_t t[32];
#pragma unroll
                for (int k = 0; k < 32; k++) t[k] = matrix[((size_t)(c + k) << nbits) + b];
#pragma unroll
                for (int k = 0; k < 32; k++) { matrix[((size_t)(c + k) << nbits) + b] = run; run += t[k]; }
            }
            for (; c < n_chunks; c++) { const uint32_t t = matrix[((size_t)c << nbits) + b]; matrix[((size_t)c << nbits) + b] = run; run += t; }
            counts[b] = run;
        }
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) s_ticket = atomicAdd(&state[5], 1u);
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;
    __threadfence();
    if (tid == 0) { s_max = 0u; s_big = 0u; s_small = 0u; }
    uint32_t c[kPer], sum = 0, mx = 0, big = 0, small = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int b = tid * kPer + k;
        c[k] = b < nb ? __atomic_load_n(&counts[b], __ATOMIC_RELAXED) : 0u;
        sum += c[k]; mx = c[k] > mx ? c[k] : mx; big += c[k] > (uint32_t)kBinCapSmall ? 1u : 0u; small += c[k] > 0u && c[k] <= (uint32_t)kBinCapSmall ? 1u : 0u;
    }
    const uint32_t incl = (uint32_t)dpp_scan_add((int)sum);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t at = incl - sum;
#pragma unroll
    for (int k = 0; k < 4; k++) at += k < wv ? wsum[k] : 0u;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int b = tid * kPer + k;
        if (b < nb) starts[b] = at;
        at += c[k];
    }
    if (mx) atomicMax(&s_max, mx);
    if (big) atomicAdd(&s_big, big);
    if (small) atomicAdd(&s_small, small);
    __syncthreads();
    if (tid == 0) { state[0] = (uint32_t)nbits; state[1] = s_max; state[2] = s_big; state[3] = (uint32_t)n; state[6] = s_small; starts[nb] = (uint32_t)n; }
}

// grid: a work-group per chunk of the n keys; at most 2 048 bins.  The chunk is dealt into its bins in LDS first and leaves it bin by bin:
// a wave's store touches the few cache lines its keys' runs lie in (16 K keys into 1 024 bins: runs of 128 B) instead of 64 lines, one per
// key -- stores of single keys to 4 096 bins made the direct scatter below 273 us per 11.5 million keys where a radix pass takes 70.
constexpr int kBinStagedBits = 11;
__global__ __launch_bounds__(1024) void k_bin_scatter_staged(const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out, const int64_t n,
                                                              const int diag_bits, const int nbits, const uint32_t *__restrict__ state, const uint32_t *__restrict__ matrix) {
    constexpr int kPer = kBinChunk / 1024, kNb = 1 << kBinStagedBits, kCPer = kNb / 1024;
    __shared__ unsigned long long tile[kBinChunk];
    __shared__ uint32_t cursor[kNb];
    __shared__ uint32_t delta[kNb];
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * kBinChunk;
    const uint32_t m = (uint32_t)(i0 + kBinChunk < n ? kBinChunk : n - i0);
    const uint32_t *row = matrix + ((size_t)blockIdx.x << nbits), *starts = state + kBinPlanWords + (1 << kBinBitsMax);
#pragma unroll
    for (int k = 0; k < kCPer; k++) cursor[tid * kCPer + k] = 0u;
    unsigned long long key[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const uint32_t i = (uint32_t)(tid + k * 1024);
        key[k] = i < m ? in[i0 + i] : 0ull;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; k++) if ((uint32_t)(tid + k * 1024) < m) atomicAdd(&cursor[bin_of(key[k], diag_bits, nbits)], 1u);
    __syncthreads();
    {
        uint32_t c[kCPer], sum = 0;
#pragma unroll
        for (int k = 0; k < kCPer; k++) { c[k] = cursor[tid * kCPer + k]; sum += c[k]; }
        const uint32_t incl = (uint32_t)dpp_scan_add((int)sum);
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t at = incl - sum;
#pragma unroll
        for (int k = 0; k < 16; k++) at += k < wv ? wsum[k] : 0u;
#pragma unroll
        for (int k = 0; k < kCPer; k++) {
            const int b = tid * kCPer + k;
            cursor[b] = at;
            delta[b] = b < (1 << nbits) ? starts[b] + row[b] - at : 0u;      // (place in the bin of the chunk's first key of bin b, minus its place in the tile)
            at += c[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; k++) if ((uint32_t)(tid + k * 1024) < m) tile[atomicAdd(&cursor[bin_of(key[k], diag_bits, nbits)], 1u)] = key[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const uint32_t i = (uint32_t)(tid + k * 1024);
        if (i < m) { const unsigned long long kk = tile[i]; out[delta[bin_of(kk, diag_bits, nbits)] + i] = kk; }
    }
}

// the same without the staging (more than 2 048 bins: a chunk holds a key or two of each)
__global__ __launch_bounds__(1024) void k_bin_scatter(const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out, const int64_t n,
                                                       const int diag_bits, const int nbits, const uint32_t *__restrict__ state, const uint32_t *__restrict__ matrix) {
    __shared__ uint32_t base[1 << kBinBitsMax];
    const int nb = 1 << nbits;
    const int64_t i0 = (int64_t)blockIdx.x * kBinChunk, i1 = i0 + kBinChunk < n ? i0 + kBinChunk : n;
    const uint32_t *row = matrix + ((size_t)blockIdx.x << nbits), *starts = state + kBinPlanWords + (1 << kBinBitsMax);
    for (int b = threadIdx.x; b < nb; b += blockDim.x) base[b] = starts[b] + row[b];
    __syncthreads();
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const unsigned long long key = in[i];
        out[atomicAdd(&base[bin_of(key, diag_bits, nbits)], 1u)] = key;
    }
}

// s[0 .. L) ascending, by the nt work-items t = 0 .. nt - 1 that call this together (one wave, or the work-group): a bitonic network whose
// comparisons all point the same way (the first step of a merge compares with the mirror position), so positions beyond L count as +inf
// and are simply left out.  sync(): between two steps.
template <typename Sync>
__device__ __forceinline__ void bin_bitonic(unsigned long long *s, const uint32_t L, const uint32_t t, const uint32_t nt, Sync sync) {
    uint32_t P = 2;
    while (P < L) P <<= 1;
    auto ce = [&](const uint32_t i, const uint32_t l) {
        if (l < L) { const unsigned long long a = s[i], b = s[l]; if (a > b) { s[i] = b; s[l] = a; } }
    };
    for (uint32_t k = 2; k <= P; k <<= 1) {
        const uint32_t half = k >> 1;
        for (uint32_t x = t; x < (P >> 1); x += nt) { const uint32_t blk = (x / half) * k, o = x & (half - 1u); ce(blk + o, blk + k - 1u - o); }
        sync();
        for (uint32_t j = k >> 2; j > 0u; j >>= 1) {
            for (uint32_t x = t; x < (P >> 1); x += nt) { const uint32_t i = ((x & ~(j - 1u)) << 1) | (x & (j - 1u)); ce(i, i + j); }
            sync();
        }
    }
}

// A bin in LDS.  Buckets: the kBk bits below the bin bits of the scrambled diagonal (all of the rest of it, if that is fewer).
template <int kCap, int kBk, int kThreads>
__global__ __launch_bounds__(kThreads) void k_bin_sort(const unsigned long long *in, unsigned long long *out, const uint32_t *__restrict__ state,
                                                        const int diag_bits, const int nbits, const uint32_t hinv, const uint32_t hmask) {
    constexpr int kPer = kCap / kThreads, kNbk = 1 << kBk, kCPer = kNbk / kThreads, kMaxRuns = kCap / (kBinRunShort + 1) + 1;
    static_assert(kCap % kThreads == 0 && kNbk % kThreads == 0, "whole keys and counters per work-item");
    __shared__ unsigned long long sk[kCap];
    __shared__ uint32_t cur[kNbk];
    __shared__ uint32_t wsum[kThreads / 64];
    __shared__ uint32_t runs[kMaxRuns];
    __shared__ uint32_t n_runs;
    const uint32_t *starts = state + kBinPlanWords + (1 << kBinBitsMax);
    const uint32_t lo = starts[blockIdx.x], m = starts[blockIdx.x + 1] - lo;
    if (m == 0u || m > (uint32_t)kCap || (kCap > kBinCapSmall && m <= (uint32_t)kBinCapSmall)) return;      // (empty, or the other instantiation's, or nobody's: the host did not come here then)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = diag_bits - nbits, bshift = r > kBk ? r - kBk : 0;
    const uint32_t rmask = r >= 32 ? 0xFFFFFFFFu : ((1u << r) - 1u);
    auto bucket = [&](const unsigned long long key) { return (((uint32_t)(key >> 32)) & rmask) >> bshift; };
#pragma unroll
    for (int k = 0; k < kCPer; k++) cur[tid * kCPer + k] = 0u;
    if (tid == 0) n_runs = 0u;
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
        for (int k = 0; k < kCPer; k++) {
            cur[tid * kCPer + k] = at; at += c[k];
            if (c[k] > (uint32_t)kBinRunShort) runs[atomicAdd(&n_runs, 1u)] = (uint32_t)(tid * kCPer + k);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; k++) if ((uint32_t)(tid + k * kThreads) < m) sk[atomicAdd(&cur[bucket(key[k])], 1u)] = key[k];
    __syncthreads();
    // cur[b] is now the END of bucket b (its beginning: the end of the bucket before it).  The long runs first, in place:
    const uint32_t nr = n_runs;
    if (nr) {
        for (uint32_t x = (uint32_t)wv; x < nr; x += (uint32_t)(kThreads / 64)) {                       // (wave-uniform: a wave takes a run at a time)
            const uint32_t b = runs[x], b0 = b ? cur[b - 1] : 0u, L = cur[b] - b0;
            if (L <= (uint32_t)kBinRunWave) bin_bitonic(sk + b0, L, (uint32_t)lane, 64u, [] { MB_WAVE_SYNC(); });
        }
        __syncthreads();
        for (uint32_t x = 0; x < nr; x++) {                                                              // (the same for every work-item)
            const uint32_t b = runs[x], b0 = b ? cur[b - 1] : 0u, L = cur[b] - b0;
            if (L > (uint32_t)kBinRunWave) bin_bitonic(sk + b0, L, (uint32_t)tid, (uint32_t)kThreads, [] { __syncthreads(); });
        }
    }
#pragma unroll 1
    for (uint32_t i = (uint32_t)tid; i < m; i += (uint32_t)kThreads) {
        const unsigned long long mine = sk[i];
        const uint32_t b = bucket(mine), b0 = b ? cur[b - 1] : 0u, b1 = cur[b];
        uint32_t at = i;                                                 // (a long run is in order already)
        if (b1 - b0 <= (uint32_t)kBinRunShort) {
            at = b0;
            for (uint32_t j = b0; j < b1; j++) at += sk[j] < mine ? 1u : 0u;
        }
        out[(size_t)lo + at] = ((unsigned long long)((((uint32_t)(mine >> 32)) * hinv) & hmask) << 32) | (uint32_t)mine;
    }
}

// mb_seed_index.h -- the dense seed position table of one large target (CSR over the 2^24 seed words + occupancy bitmap: index words, the
// three-launch exclusive scan, scatter) and the seed search of a strand that needs several q batches (count, scan, fill) or is asked for
// without q order (k_seed_search) (gfx950, wave64).  Included by mb_kernels.hip inside namespace mb after the wave helpers (dpp_scan_add)
// and mb_seedword.h -- and, with MB_EMU defined, by the host-side emulation under tests/emu (emu_seed_dense.cpp).
#pragma once

// ------------------------------------------------------------------------------------------------
// seed index
__global__ void k_index_words(const uint8_t *__restrict__ codes, int64_t n, int step, int64_t first, uint32_t *__restrict__ words,
                              int64_t n_slots, uint32_t *__restrict__ counts) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    int64_t p = first + s * step;
    uint32_t w = 0xFFFFFFFFu;
    if (p + kSeedSpan <= n) {
        uint32_t ww;
        if (window_word(codes, p, ww)) { w = dense_bucket(ww); atomicAdd(&counts[w], 1u); }
    }
    words[s] = w;
}

void launch_index_words(const uint8_t *codes, int64_t n, int step, int64_t first, uint32_t *words, int64_t n_slots, uint32_t *counts,
                        hipStream_t s) {
    if (n_slots <= 0) return;
    hipLaunchKernelGGL(k_index_words, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, s, codes, n, step, first, words,
                       n_slots, counts);
}

__global__ void k_index_scatter(const uint32_t *__restrict__ words, int64_t n_slots, int step, int64_t first,
                                const uint32_t *__restrict__ offsets, uint32_t *__restrict__ cursor,
                                uint32_t *__restrict__ positions) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    uint32_t w = words[s];
    if (w == 0xFFFFFFFFu) return;
    uint32_t k = atomicAdd(&cursor[w], 1u);
    positions[offsets[w] + k] = (uint32_t)(first + s * step);
}

void launch_index_scatter(const uint32_t *words, int64_t n_slots, int step, int64_t first, const uint32_t *offsets, uint32_t *cursor,
                          uint32_t *positions, hipStream_t s) {
    if (n_slots <= 0) return;
    hipLaunchKernelGGL(k_index_scatter, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, s, words, n_slots, step, first,
                       offsets, cursor, positions);
}

// the scatter's cursors (= the bucket counts again) back to zero for the next build: by the indexed words when they are few,
// else with a memset of the 64 MiB
__global__ void k_index_clear(const uint32_t *__restrict__ words, int64_t n_slots, uint32_t *__restrict__ cursor) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const uint32_t w = words[s];
    if (w != 0xFFFFFFFFu) cursor[w] = 0u;
}

void launch_index_clear(const uint32_t *words, int64_t n_slots, uint32_t *cursor, hipStream_t s) {
    if (n_slots <= 0) return;
    if (n_slots > (2 << 20)) { (void)hipMemsetAsync(cursor, 0, ((size_t)kBuckets + 1) * 4, s); return; }
    hipLaunchKernelGGL(k_index_clear, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, s, words, n_slots, cursor);
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of u32 (three launches: block totals, scan of totals, apply)
constexpr int kScanBlock = 256;
constexpr int kScanPerThread = 8;
constexpr int kScanTile = kScanBlock * kScanPerThread;    // 2048

__global__ void k_block_sums(const uint32_t *__restrict__ in, int64_t n, unsigned long long *__restrict__ bsum) {
    __shared__ unsigned long long red[kScanBlock / 64];
    int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPerThread;
    // (thread sums in 64 bits; the wave total from three 20-bit slices summed with DPP scans: no LDS round trips)
    unsigned long long v = 0;
    if (base + kScanPerThread <= n) {
        const uint4 a = *(const uint4 *)(in + base), b = *(const uint4 *)(in + base + 4);      // (base is a multiple of 8: 16-byte aligned)
        v = (unsigned long long)a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    } else {
#pragma unroll
        for (int k = 0; k < kScanPerThread; k++) if (base + k < n) v += in[base + k];
    }
    {
        const int s0 = dpp_scan_add((int)(v & 0xFFFFFu)), s1 = dpp_scan_add((int)((v >> 20) & 0xFFFFFu)), s2 = dpp_scan_add((int)(v >> 40));
        v = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(s0, 63) + ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(s1, 63) << 20) +
            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(s2, 63) << 40);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int k = 0; k < kScanBlock / 64; k++) t += red[k];
        bsum[blockIdx.x] = t;
    }
}

__global__ void k_scan_bsums(unsigned long long *bsum, int64_t nb) {
    // single block of 1024 threads; serial over chunks with a running carry
    __shared__ unsigned long long tmp[1024];
    __shared__ unsigned long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        int64_t i = base + threadIdx.x;
        unsigned long long v = (i < nb) ? bsum[i] : 0;
        tmp[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            unsigned long long add = (threadIdx.x >= (unsigned)o) ? tmp[threadIdx.x - o] : 0;
            __syncthreads();
            tmp[threadIdx.x] += add;
            __syncthreads();
        }
        unsigned long long incl = tmp[threadIdx.x], carry = carry_s;
        if (i < nb) bsum[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[nb] = carry_s;
}

// INDEX: the scan of the seed table's bucket counts.  The same pass leaves the counts zeroed (they are the scatter's cursors next,
// and the next build's histogram after that: no 64 MiB memset between) and writes the occupancy bitmap of the buckets (bit b of
// word w = bucket 32 w + b holds at least one position; 2 MiB, so it stays in L2 while the 64 MiB offset table does not:
// k_seed_search asks it first).
template <bool INDEX>
__global__ void k_scan_apply(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int64_t n,
                             const unsigned long long *__restrict__ bsum, uint32_t *__restrict__ zero_in, uint32_t *__restrict__ occ,
                             int64_t n_occ) {
    __shared__ uint32_t wsum[kScanBlock / 64];
    int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPerThread;
    uint32_t v[kScanPerThread];
    uint32_t tsum = 0;
    if (base + kScanPerThread <= n) {
        const uint4 a = *(const uint4 *)(in + base), b = *(const uint4 *)(in + base + 4);      // (base is a multiple of 8: 16-byte aligned)
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < kScanPerThread; k++) v[k] = (base + k < n) ? in[base + k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < kScanPerThread; k++) tsum += v[k];
    // inclusive scan of tsum across the wave (DPP: no LDS round trips)
    const int lane = threadIdx.x & 63;
    const uint32_t incl = (uint32_t)dpp_scan_add((int)tsum);
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < (int)(threadIdx.x >> 6); k++) woff += wsum[k];
    uint32_t run = (uint32_t)bsum[blockIdx.x] + woff + incl - tsum;
    if (base + kScanPerThread <= n) {
        uint4 a, b;
        a.x = run; a.y = a.x + v[0]; a.z = a.y + v[1]; a.w = a.z + v[2]; b.x = a.w + v[3]; b.y = b.x + v[4]; b.z = b.y + v[5]; b.w = b.z + v[6];
        *(uint4 *)(out + base) = a; *(uint4 *)(out + base + 4) = b;
    } else {
#pragma unroll
        for (int k = 0; k < kScanPerThread; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
    }
    if (INDEX) {
        static_assert(kScanPerThread == 8, "four threads make a bitmap word");
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < kScanPerThread; k++) {
            m |= (v[k] != 0u ? 1u : 0u) << k;
            if (v[k] != 0u) zero_in[base + k] = 0u;                       // (sparse: most buckets of a small target are empty already)
        }
        m <<= 8 * (lane & 3);
        m |= __shfl_xor(m, 1);
        m |= __shfl_xor(m, 2);
        if ((lane & 3) == 0 && base < n_occ) occ[base >> 5] = m;          // (n_occ = kBuckets: a multiple of 32; the total slot behind it is not a bucket)
    }
}

void launch_block_sums(const uint32_t *in, int64_t n, unsigned long long *block_sums, hipStream_t s) {
    if (n <= 0) return;
    int64_t nb = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_block_sums, dim3((unsigned)nb), dim3(kScanBlock), 0, s, in, n, block_sums);
}

void launch_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, unsigned long long *block_sums, hipStream_t s) {
    if (n <= 0) return;
    int64_t nb = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_block_sums, dim3((unsigned)nb), dim3(kScanBlock), 0, s, in, n, block_sums);
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(1024), 0, s, block_sums, nb);
    hipLaunchKernelGGL(k_scan_apply<false>, dim3((unsigned)nb), dim3(kScanBlock), 0, s, in, out, n, block_sums, (uint32_t *)nullptr, (uint32_t *)nullptr, (int64_t)0);
}

// exclusive scan of the kBuckets + 1 bucket counts of the seed table -> offsets; the counts come out zeroed and the occupancy
// bitmap of the buckets is written (k_scan_apply<true>)
void launch_scan_index(uint32_t *counts, uint32_t *offsets, unsigned long long *block_sums, uint32_t *occ, hipStream_t s) {
    const int64_t n = (int64_t)kBuckets + 1, nb = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_block_sums, dim3((unsigned)nb), dim3(kScanBlock), 0, s, counts, n, block_sums);
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(1024), 0, s, block_sums, nb);
    hipLaunchKernelGGL(k_scan_apply<true>, dim3((unsigned)nb), dim3(kScanBlock), 0, s, counts, offsets, n, block_sums, counts, occ, (int64_t)kBuckets);
}

// ------------------------------------------------------------------------------------------------
// seed search
// (count and fill ask the 2 MiB occupancy bitmap before the 64 MiB offset table, like k_seed_search: most lookups of a sparse
//  table end in L2)
__global__ void k_seed_count(const uint8_t *__restrict__ qcodes, int64_t qn, const uint32_t *__restrict__ offsets,
                             const uint32_t *__restrict__ occ, int nvar, uint32_t *__restrict__ qcnt) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= qn) return;
    uint32_t cnt = 0, w;
    if (q + kSeedSpan <= qn && window_word(qcodes, q, w)) {
        const uint32_t b = dense_bucket(w);
        for (int v = 0; v < nvar; v++) {
            uint32_t wv = dense_variant(b, v);
            if ((occ[wv >> 5] >> (wv & 31u)) & 1u) cnt += offsets[wv + 1] - offsets[wv];
        }
    }
    qcnt[q] = cnt;
}

void launch_seed_count(const uint8_t *qcodes, int64_t qn, const uint32_t *offsets, const uint32_t *occ, int transitions, uint32_t *qcnt,
                       hipStream_t s) {
    if (qn <= 0) return;
    hipLaunchKernelGGL(k_seed_count, dim3((unsigned)((qn + 255) / 256)), dim3(256), 0, s, qcodes, qn, offsets, occ,
                       transitions ? 1 + kSeedWeight : 1, qcnt);
}

__global__ void k_seed_fill(const uint8_t *__restrict__ qcodes, int64_t q0, int64_t q1, int64_t qn, int64_t qtot,
                            const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ occ, const uint32_t *__restrict__ positions, int nvar,
                            const uint32_t *__restrict__ hit_off, unsigned long long *__restrict__ keys, const uint32_t hmul, const uint32_t hmask) {
    int64_t q = q0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= q1) return;
    uint32_t w;
    if (!(q + kSeedSpan <= qn && window_word(qcodes, q, w))) return;
    uint32_t o = hit_off[q - q0];
    unsigned long long q_end = (unsigned long long)(q + kSeedSpan);
    const uint32_t bkt = dense_bucket(w);
    for (int v = 0; v < nvar; v++) {
        uint32_t wv = dense_variant(bkt, v);
        if (!((occ[wv >> 5] >> (wv & 31u)) & 1u)) continue;
        uint32_t b0 = offsets[wv], b1 = offsets[wv + 1];
        for (uint32_t k = b0; k < b1; k++) {
            // diagonal d = t_end - q_end = p - q ; stored biased by qtot so it is non-negative, scrambled for the sort (mb_seed_dense.h)
            const uint32_t dq = (uint32_t)((int64_t)positions[k] - q + qtot);
            keys[o++] = ((unsigned long long)((dq * hmul) & hmask) << 32) | q_end;
        }
    }
}

void launch_seed_fill(const uint8_t *qcodes, int64_t q0, int64_t q1, int64_t qtot, const uint32_t *offsets, const uint32_t *occ,
                      const uint32_t *positions, int transitions, const uint32_t *hit_off, unsigned long long *keys,
                      hipStream_t s, uint32_t hmul, uint32_t hmask) {
    if (q1 <= q0) return;
    hipLaunchKernelGGL(k_seed_fill, dim3((unsigned)((q1 - q0 + 255) / 256)), dim3(256), 0, s, qcodes, q0, q1, qtot, qtot,
                       offsets, occ, positions, transitions ? 1 + kSeedWeight : 1, hit_off, keys, hmul, hmask);
}

// seed search in one pass: count, reserve and fill.  Every block counts the hits of its 256 query positions (the bucket
// bounds of the 13 word variants stay in registers), takes its share of the key buffer with ONE atomicAdd and writes the
// keys.  The key order in the buffer depends on the order the blocks get there, but a (diagonal, q_end) key occurs at
// most once and the keys are radix-sorted next, so the result does not.  total[0] receives the number of hits even when
// they did not fit (cap): the host then falls back to the two-pass path (k_seed_count, scan, k_seed_fill).
__global__ __launch_bounds__(256) void k_seed_search(const uint8_t *__restrict__ qcodes, int64_t qn, int64_t qtot,
                                                     const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ occ,
                                                     const uint32_t *__restrict__ positions, int nvar,
                                                     unsigned long long *__restrict__ keys, unsigned long long cap,
                                                     unsigned long long *__restrict__ total) {
    __shared__ unsigned wave_sum[4];
    __shared__ unsigned long long block_base;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t b0[1 + kSeedWeight], b1[1 + kSeedWeight];
    unsigned cnt = 0;
    uint32_t w;
    const bool valid = q < qn && q + kSeedSpan <= qn && window_word(qcodes, q, w);
#pragma unroll
    for (int v = 0; v < 1 + kSeedWeight; v++) {
        b0[v] = b1[v] = 0;
        if (valid && v < nvar) {
            const uint32_t wvv = dense_variant(dense_bucket(w), v);
            if ((occ[wvv >> 5] >> (wvv & 31u)) & 1u) { b0[v] = offsets[wvv]; b1[v] = offsets[wvv + 1]; cnt += b1[v] - b0[v]; }
        }
    }
    const unsigned incl = (unsigned)dpp_scan_add((int)cnt);
    if (lane == 63) wave_sum[wv] = incl;
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const unsigned ws = wave_sum[k]; all += ws; if (k < wv) before += ws; }
    if (threadIdx.x == 0) block_base = all ? atomicAdd(total, (unsigned long long)all) : 0ull;
    __syncthreads();
    unsigned long long o = block_base + before + (incl - cnt);
    if (block_base + all > cap) return;                                  // does not fit: the host reruns the strand in two passes
    const unsigned long long q_end = (unsigned long long)(q + kSeedSpan);
#pragma unroll
    for (int v = 0; v < 1 + kSeedWeight; v++)
        for (uint32_t k = b0[v]; k < b1[v]; k++) {
            // diagonal d = t_end - q_end = p - q ; stored biased by qtot so it is non-negative
            const unsigned long long dq = (unsigned long long)((int64_t)positions[k] - q + qtot);
            keys[o++] = (dq << 32) | q_end;
        }
}

void launch_seed_search(const uint8_t *qcodes, int64_t qtot, const uint32_t *offsets, const uint32_t *occ, const uint32_t *positions, int transitions,
                        unsigned long long *keys, unsigned long long cap, unsigned long long *total, hipStream_t s) {
    if (qtot <= 0) return;
    hipLaunchKernelGGL(k_seed_search, dim3((unsigned)((qtot + 255) / 256)), dim3(256), 0, s, qcodes, qtot, qtot, offsets, occ, positions,
                       transitions ? 1 + kSeedWeight : 1, keys, cap, total);
}

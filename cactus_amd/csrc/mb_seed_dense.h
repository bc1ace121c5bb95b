// mb_seed_dense.h -- seed stage of a large chunk pair (gfx950, wave64): packed sequences, the q-ordered one-pass seed search with
// LDS-staged key output, and the diagonal scrambling that keeps the key sort balanced.  Included by mb_kernels.hip inside namespace mb
// after mb_seedword.h.
//
//   k_pack2bit_mask     a strand as 2 bits per base + 1 mask bit per base (0.375 B/base): what the index build and the seed
//                       search read instead of 19 code bytes per window
//   k_index_words_packed  the target's seed words from the packed form (replaces k_index_words on this path)
//   k_seed_search_ord   seed search of a strand in ONE pass with the keys in q order: tiles of 1024 query positions are taken in order
//                       (one ticket per tile), a tile's share of the key buffer comes from a decoupled look-back over the tiles before
//                       it, and its keys are put together in LDS and leave in whole cache lines
//   k_keys_unhash       the keys' diagonals back from their scrambled form after the sort
//
// Why the keys are q ordered: the ungapped kernels want the hits of a diagonal together and in q order.  Keys that come out in q order
// only have to be sorted (stably) by diagonal: 4 radix passes on a 30 Mb x 30 Mb pair instead of 8.
// Why the diagonal is scrambled: the hits of real homology lie on a few thousand neighbouring diagonals (a 30 Mb x 30 Mb pair at 1.3 %
// divergence: 5 of its 8 million hits per strand within +-10^4 of the main diagonal), so every pass above the lowest byte of the
// diagonal sends a third of the keys to one radix bin -- the onesweep passes took 6 ms instead of 0.3.  The sort only has to GROUP the
// diagonals, not order them: the key holds (d * C) mod 2^B (C odd, B = bits of the diagonal space), neighbouring diagonals land
// all over the bins, and k_keys_unhash multiplies by C^-1 afterwards.
#pragma once

// ---- packed strands ------------------------------------------------------------------------------------------------------------
// p2: base i = bits 63 - 2 (i & 31) .. 62 - 2 (i & 31) of word i >> 5 (first base most significant: a window's care bases come out in the
// order of the seed word).  pm: bit 63 - (i & 63) of word i >> 6 is set when base i cannot be part of a seed window (N, soft-masked,
// separator, beyond the end).  Both arrays hold two words more than the bases need (a window reads word w and w + 1).
// (packed_words2 / packed_wordsm: mb_common.h)

__global__ __launch_bounds__(256) void k_pack2bit_mask(const uint8_t *__restrict__ codes, const int64_t n, unsigned long long *__restrict__ p2,
                                                        unsigned long long *__restrict__ pm, const int64_t n_words_m) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // mask word = 64 bases = two 2-bit words
    if (w >= n_words_m) return;
    const int64_t i0 = w * 64;
    unsigned long long a = 0, b = 0, m = 0;
    if (i0 + 64 <= n) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned long long lo, hi;
            __builtin_memcpy(&lo, codes + i0 + 16 * j, 8); __builtin_memcpy(&hi, codes + i0 + 16 * j + 8, 8);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const unsigned c = (unsigned)((k < 8 ? lo >> (8 * k) : hi >> (8 * (k - 8))) & 0xFFu);
                const int i = 16 * j + k;
                const unsigned long long two = (unsigned long long)(c & 3u);
                if (i < 32) a |= two << (62 - 2 * i); else b |= two << (62 - 2 * (i - 32));
                m |= (unsigned long long)((c & 0xFCu) != 0u) << (63 - i);
            }
        }
    } else {
        for (int i = 0; i < 64; i++) {
            const unsigned c = i0 + i < n ? codes[i0 + i] : 0xFFu;
            const unsigned long long two = (unsigned long long)(c & 3u);
            if (i < 32) a |= two << (62 - 2 * i); else b |= two << (62 - 2 * (i - 32));
            m |= (unsigned long long)((c & 0xFCu) != 0u) << (63 - i);
        }
    }
    p2[2 * w] = a; p2[2 * w + 1] = b; pm[w] = m;
}

// the seed word of the window at p (12 of 19, first care base most significant) and whether all 19 bases may be seeded
__device__ __forceinline__ bool packed_window_word(const unsigned long long *__restrict__ p2, const unsigned long long *__restrict__ pm, const int64_t p, uint32_t &word) {
    const int64_t w2 = p >> 5, wm = p >> 6;
    const unsigned s2 = (unsigned)(p & 31) * 2u, sm = (unsigned)(p & 63);
    const unsigned long long a0 = p2[w2], a1 = p2[w2 + 1], m0 = pm[wm], m1 = pm[wm + 1];
    const unsigned long long x = s2 ? (a0 << s2) | (a1 >> (64u - s2)) : a0;
    const unsigned long long y = sm ? (m0 << sm) | (m1 >> (64u - sm)) : m0;
    const unsigned long long v = x >> 26;                              // 38 bits: base k of the window at bits 37 - 2 k, 36 - 2 k
    // care offsets 0 1 2 | 4 | 7 8 | 11 | 13 | 15 16 17 18 of 1110100110010101111
    word = (uint32_t)(((v >> 32) & 0x3Full) << 18 | ((v >> 28) & 0x3ull) << 16 | ((v >> 20) & 0xFull) << 12 | ((v >> 14) & 0x3ull) << 10 | ((v >> 10) & 0x3ull) << 8 | (v & 0xFFull));
    return (y >> 45) == 0ull;
}

__global__ __launch_bounds__(256) void k_index_words_packed(const unsigned long long *__restrict__ p2, const unsigned long long *__restrict__ pm, const int64_t n,
                                                             const int step, const int64_t first, uint32_t *__restrict__ words, const int64_t n_slots,
                                                             uint32_t *__restrict__ counts) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    const int64_t p = first + s * step;
    uint32_t w = 0xFFFFFFFFu;
    if (p + kSeedSpan <= n) {
        uint32_t ww;
        if (packed_window_word(p2, pm, p, ww)) { w = dense_bucket(ww); atomicAdd(&counts[w], 1u); }
    }
    words[s] = w;
}

// ---- q-ordered one-pass seed search ---------------------------------------------------------------------------------------------
constexpr int kOrdThreadsMin = 512;           // threads per block: 1024 with one word variant, 512 with thirteen (74 VGPRs: three blocks of 8 waves per CU instead of one of 16);
                                              // a tile = threads x R query positions (R consecutive ones per thread)
constexpr int kOrdStage = 6144;               // keys a tile puts together in LDS (48 KiB); a tile with more writes them one by one
constexpr unsigned long long kOrdFlagA = 1ull << 62, kOrdFlagP = 2ull << 62, kOrdValue = (1ull << 62) - 1ull;

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// state: [0] ticket, [1] total hits of the strand (written by the last tile), [2 + t] look-back word of tile t -- all zero before the launch.
// NV word variants per position (1 with --notransition, else 13); R positions per thread (4 with one variant: a quarter of the tiles --
// tickets, block scans, look-backs -- for the 3 x 10^7 positions of a chunk's strand).
#ifdef MB_ORD_PROF            // (lab build: shader clocks of a tile's phases, summed over the tiles of block 0 into the words behind the look-back words)
#define MB_ORD_T(k) const long long ordt##k = (tid == 0) ? (long long)__builtin_readcyclecounter() : 0
#define MB_ORD_FLUSH() do { if (tid == 0 && blockIdx.x == 0) { unsigned long long *pp = state + 2 + n_tiles; atomicAdd(pp + 0, (unsigned long long)(ordt1 - ordt0)); atomicAdd(pp + 1, (unsigned long long)(ordt2 - ordt1)); \
    atomicAdd(pp + 2, (unsigned long long)(ordt3 - ordt2)); atomicAdd(pp + 3, (unsigned long long)(ordt4 - ordt3)); atomicAdd(pp + 4, 1ull); } } while (0)
#else
#define MB_ORD_T(k) do { } while (0)
#define MB_ORD_FLUSH() do { } while (0)
#endif
template <bool PACKED, int R, int NV, int kOrdThreads>
__global__ __launch_bounds__(kOrdThreads, NV == 1 ? 8 : 6) void k_seed_search_ord(const uint8_t *__restrict__ qcodes, const unsigned long long *__restrict__ p2,
                                                                  const unsigned long long *__restrict__ pm, const int64_t qn, const int64_t qtot,
                                                                  const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ occ,
                                                                  const uint32_t *__restrict__ positions, const uint32_t hmul, const uint32_t hmask,
                                                                  unsigned long long *__restrict__ keys, const unsigned long long cap,
                                                                  unsigned long long *__restrict__ state, const int n_tiles) {
    constexpr int kTile = kOrdThreads * R;
    __shared__ unsigned long long stage[kOrdStage];
    __shared__ unsigned wave_sum[kOrdThreads / 64];
    __shared__ unsigned long long s_excl;
    __shared__ int s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    while (true) {
        if (tid == 0) s_tile = (int)atomicAdd(&state[0], 1ull);          // tiles start in order: the tiles a look-back waits for are running or done
        __syncthreads();
        const int tile = s_tile;
        if (tile >= n_tiles) return;
        MB_ORD_T(0);
        const int64_t q0 = (int64_t)tile * kTile + (int64_t)tid * R;
        uint32_t b0[R * NV], b1[R * NV];
        unsigned cnt = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int64_t q = q0 + r;
            uint32_t w = 0;
            bool valid = q + kSeedSpan <= qn;
            if (valid) valid = PACKED ? packed_window_word(p2, pm, q, w) : window_word(qcodes, q, w);
            const uint32_t bkt = dense_bucket(w);                     // (mb_seedword.h: the 13 variants lie within one stretch of 4096 buckets)
#pragma unroll
            for (int v = 0; v < NV; v++) {
                b0[r * NV + v] = b1[r * NV + v] = 0;
                if (valid) {
                    const uint32_t wvv = dense_variant(bkt, v);
                    if ((occ[wvv >> 5] >> (wvv & 31u)) & 1u) { b0[r * NV + v] = offsets[wvv]; b1[r * NV + v] = offsets[wvv + 1]; cnt += b1[r * NV + v] - b0[r * NV + v]; }
                }
            }
        }
        MB_ORD_T(1);
        const unsigned incl = (unsigned)dpp_scan_add((int)cnt);
        if (lane == 63) wave_sum[wv] = incl;
        __syncthreads();
        MB_ORD_T(2);
        unsigned before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < kOrdThreads / 64; k++) { const unsigned ws = wave_sum[k]; all += ws; before += k < wv ? ws : 0u; }
        if (wv == 0) {
            // decoupled look-back by one wave: the tile's own count is published first; then 64 tiles at a time, nearest first, the counts of
            // the tiles before it are added up until one of them has its inclusive prefix out.  The words carry their values themselves:
            // relaxed device-scope accesses do.
            unsigned long long excl = 0;
            if (lane == 0 && tile > 0) __hip_atomic_store(&state[2 + tile], kOrdFlagA | (unsigned long long)all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int base = tile - 1;
            while (base >= 0) {
                const int j = base - lane;
                unsigned long long v = kOrdFlagP;                       // (before tile 0: an inclusive prefix of 0)
                if (j >= 0) v = __hip_atomic_load(&state[2 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned f = (unsigned)(v >> 62);
                const unsigned long long mp = wballot(f == 2u), mz = wballot(f == 0u);
                const int pl = mp ? (int)__ffsll((long long)mp) - 1 : 63;    // the nearest tile with its prefix out (none: all 64 counts are needed)
                const unsigned long long need = (2ull << pl) - 1ull;       // lanes 0 .. pl
                if (mz & need) { __builtin_amdgcn_s_sleep(2); continue; }   // a tile in between has not published yet
                excl += wave_sum_u64(lane <= pl ? (v & kOrdValue) : 0ull);
                if (mp) break;
                base -= 64;
            }
            if (lane == 0) {
                __hip_atomic_store(&state[2 + tile], kOrdFlagP | (excl + (unsigned long long)all), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tile == n_tiles - 1) state[1] = excl + (unsigned long long)all;
                s_excl = excl;
            }
        }
        __syncthreads();
        MB_ORD_T(3);
        const unsigned long long excl = s_excl;
        if (excl + all <= cap && all) {                                 // (does not fit: the host makes room and searches again; the totals still come out)
            unsigned o = before + (incl - cnt);
            const bool staged = all <= (unsigned)kOrdStage;
            unsigned long long *const out = keys + excl;
            if (staged && NV > 1) {
                // a position's hits are listed in LDS first -- (slot in the table's position array, position's offset in the tile): no memory
                // access in the lanes' serial loops -- and the tile's threads then fetch the target positions of ALL hits side by side and
                // write the keys in order, whole cache lines at a time
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const unsigned long long qrel = (unsigned long long)(unsigned)(tid * R + r) << 32;
#pragma unroll
                    for (int v = 0; v < NV; v++)
                        for (uint32_t k = b0[r * NV + v]; k < b1[r * NV + v]; k++) stage[o++] = qrel | k;
                }
                __syncthreads();
                const int64_t qbase = (int64_t)tile * kTile;
                for (unsigned i = tid; i < all; i += kOrdThreads) {
                    const unsigned long long e = stage[i];
                    const int64_t q = qbase + (int64_t)(e >> 32);
                    const uint32_t dq = (uint32_t)((int64_t)positions[(uint32_t)e] - q + qtot);
                    out[i] = ((unsigned long long)((dq * hmul) & hmask) << 32) | (unsigned long long)(q + kSeedSpan);
                }
            } else {
                // (one word variant: a position has a hit or two, the lanes' loops are short -- the keys are made where the bounds are and only
                //  pass through LDS to leave in whole cache lines)
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int64_t q = q0 + r;
                    const unsigned long long q_end = (unsigned long long)(q + kSeedSpan);
#pragma unroll
                    for (int v = 0; v < NV; v++)
                        for (uint32_t k = b0[r * NV + v]; k < b1[r * NV + v]; k++) {
                            // diagonal d = t_end - q_end = p - q, biased by qtot so that it is not negative, then scrambled (see the head of the file)
                            const uint32_t dq = (uint32_t)((int64_t)positions[k] - q + qtot);
                            const unsigned long long key = ((unsigned long long)((dq * hmul) & hmask) << 32) | q_end;
                            if (staged) stage[o] = key; else out[o] = key;
                            o++;
                        }
                }
                if (staged) {
                    __syncthreads();
                    for (unsigned i = tid; i < all; i += kOrdThreads) out[i] = stage[i];
                }
            }
        }
        __syncthreads();                                                // (stage, s_tile and s_excl are reused by the next tile)
        MB_ORD_T(4);
        MB_ORD_FLUSH();
    }
}

__global__ __launch_bounds__(256) void k_keys_unhash(unsigned long long *__restrict__ keys, const int64_t n, const uint32_t hinv, const uint32_t hmask) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < n) {
        ulonglong2 k = *(const ulonglong2 *)(keys + i);                 // (i is even: 16-byte aligned)
        k.x = ((unsigned long long)(((uint32_t)(k.x >> 32) * hinv) & hmask) << 32) | (uint32_t)k.x;
        k.y = ((unsigned long long)(((uint32_t)(k.y >> 32) * hinv) & hmask) << 32) | (uint32_t)k.y;
        *(ulonglong2 *)(keys + i) = k;
    } else if (i < n) {
        const unsigned long long k = keys[i];
        keys[i] = ((unsigned long long)(((uint32_t)(k >> 32) * hinv) & hmask) << 32) | (uint32_t)k;
    }
}

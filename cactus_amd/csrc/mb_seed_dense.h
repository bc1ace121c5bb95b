// mb_seed_dense.h -- seed stage of a large chunk pair (gfx950, wave64): packed sequences, the q-ordered one-pass seed search (hit lists
// per tile of query positions, keys written in q order by a second kernel), and the diagonal scrambling that keeps the key sort balanced.
// (Behind it the keys are grouped by diagonal through bins + LDS -- mb_seed_bin.h -- or, for a strand that does not fit, by rocprim's radix
// sort.)  Included by mb_kernels.hip inside namespace mb after mb_seedword.h.
//
//   k_pack2bit_mask     a strand as 2 bits per base + 1 mask bit per base (0.375 B/base): what the index build and the seed
//                       search read instead of 19 code bytes per window
//   k_index_words_packed  the target's seed words from the packed form (replaces k_index_words on this path)
//   k_seed_hits / k_seed_keys   seed search of a strand with ONE look-up pass and the keys in q order: a tile of query positions lists its
//                       hits into a stretch of scratch reserved with one atomic add; after a scan of the tiles' counts a block per
//                       tile turns the list into keys at the tile's place in q order, whole cache lines at a time
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

// px (optional): the ungapped extension's form of the strand (mb_ungapped_ux.h) -- per 32 bases a 12-byte record {low, high dword of the 2-bit
// word, one flag per base (bit 31 - k) that is set when the base is an N / IUPAC code or a separator (or lies beyond the end)}: what the
// extension cannot score from two bits; a soft-masked base is NOT flagged (it extends like any other base).
__global__ __launch_bounds__(256) void k_pack2bit_mask(const uint8_t *__restrict__ codes, const int64_t n, unsigned long long *__restrict__ p2,
                                                        unsigned long long *__restrict__ pm, const int64_t n_words_m, uint32_t *__restrict__ px) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // mask word = 64 bases = two 2-bit words
    if (w >= n_words_m) return;
    const int64_t i0 = w * 64;
    unsigned long long a = 0, b = 0, m = 0, sp = 0;
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
                sp |= (unsigned long long)((c & 0x84u) != 0u) << (63 - i);
            }
        }
    } else {
        for (int i = 0; i < 64; i++) {
            const unsigned c = i0 + i < n ? codes[i0 + i] : 0xFFu;
            const unsigned long long two = (unsigned long long)(c & 3u);
            if (i < 32) a |= two << (62 - 2 * i); else b |= two << (62 - 2 * (i - 32));
            m |= (unsigned long long)((c & 0xFCu) != 0u) << (63 - i);
            sp |= (unsigned long long)((c & 0x84u) != 0u) << (63 - i);
        }
    }
    p2[2 * w] = a; p2[2 * w + 1] = b; pm[w] = m;
    if (px) {
        uint32_t *r = px + 6 * w;
        r[0] = (uint32_t)a; r[1] = (uint32_t)(a >> 32); r[2] = (uint32_t)(sp >> 32);
        r[3] = (uint32_t)b; r[4] = (uint32_t)(b >> 32); r[5] = (uint32_t)sp;
    }
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

// ---- q-ordered seed search without a second look-up pass ---------------------------------------------------------------------------
// k_seed_hits: a tile of query positions (1024 threads x 4 positions with one word variant, 512 x 1 with thirteen) looks its words up and
//   lists its hits -- (offset of the position in the tile) << 32 | slot in the table's position array -- in thread order into a stretch
//   of a scratch buffer that it reserves with ONE atomic add: the stretches lie in the order the tiles got there;
// a scan of the tiles' hit counts gives every tile its place in q order;
// k_seed_keys: a block per tile fetches the target positions of the tile's hits side by side and writes the keys to that place,
//   whole cache lines at a time.
// Nothing waits for anything inside a kernel.  (The first version of this round kept q order with a decoupled look-back over the
// tiles: half of a tile's time was the wait for the tiles before it, and with a dozen such kernels of other chunk pairs on the GPU a
// step now and then took seconds -- hundreds of spinning blocks polling the same words.  The scratch costs 8 B written and 8 B read per
// hit; the sort's output buffer serves.)
constexpr int kOrdThreadsMin = 512;           // threads per block: 1024 with one word variant, 512 with thirteen (72 VGPRs: three blocks of 8 waves per CU)

template <bool PACKED, int R, int NV, int kOrdThreads>
__global__ __launch_bounds__(kOrdThreads, NV == 1 ? 8 : 6) void k_seed_hits(const uint8_t *__restrict__ qcodes, const unsigned long long *__restrict__ p2,
                                                                             const unsigned long long *__restrict__ pm, const int64_t qn,
                                                                             const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ occ,
                                                                             unsigned long long *__restrict__ scratch, const unsigned long long cap,
                                                                             unsigned long long *__restrict__ total, unsigned long long *__restrict__ tile_base,
                                                                             uint32_t *__restrict__ tile_cnt, const int n_tiles) {
    constexpr int kTile = kOrdThreads * R;
    __shared__ unsigned wave_sum[kOrdThreads / 64];
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
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
        const unsigned incl = (unsigned)dpp_scan_add((int)cnt);
        if (lane == 63) wave_sum[wv] = incl;
        __syncthreads();
        unsigned before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < kOrdThreads / 64; k++) { const unsigned ws = wave_sum[k]; all += ws; before += k < wv ? ws : 0u; }
        if (tid == 0) {
            const unsigned long long base = all ? atomicAdd(total, (unsigned long long)all) : 0ull;
            tile_base[tile] = base; tile_cnt[tile] = all;
            s_base = base;
        }
        __syncthreads();
        const unsigned long long base = s_base;
        if (all && base + all <= cap) {                                 // (does not fit: the host makes room and searches again; the counts still come out)
            unsigned long long *out = scratch + base + before + (incl - cnt);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const unsigned long long qrel = (unsigned long long)(unsigned)(tid * R + r) << 32;
#pragma unroll
                for (int v = 0; v < NV; v++)
                    for (uint32_t k = b0[r * NV + v]; k < b1[r * NV + v]; k++) *out++ = qrel | k;
            }
        }
        __syncthreads();                                                // (wave_sum and s_base are reused by the block's next tile)
    }
}

// keys of tile t = its listed hits, in order, at tile_off[t] of the key buffer (tile_off: exclusive scan of tile_cnt)
__global__ __launch_bounds__(256) void k_seed_keys(const unsigned long long *__restrict__ scratch, const unsigned long long *__restrict__ tile_base,
                                                   const uint32_t *__restrict__ tile_cnt, const uint32_t *__restrict__ tile_off,
                                                   const uint32_t *__restrict__ positions, unsigned long long *__restrict__ keys, const unsigned long long cap,
                                                   const int64_t qtot, const int tile_positions, const uint32_t hmul, const uint32_t hmask, const int n_tiles) {
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const unsigned n = tile_cnt[tile];
        const unsigned long long base = tile_base[tile], off = tile_off[tile];
        if (!n || base + n > cap || off + n > cap) continue;
        const int64_t qbase = (int64_t)tile * tile_positions;
        for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned long long e = scratch[base + i];
            const int64_t q = qbase + (int64_t)(e >> 32);
            // diagonal d = t_end - q_end = p - q, biased by qtot so that it is not negative, then scrambled (see the head of the file)
            const uint32_t dq = (uint32_t)((int64_t)positions[(uint32_t)e] - q + qtot);
            keys[off + i] = ((unsigned long long)((dq * hmul) & hmask) << 32) | (unsigned long long)(q + kSeedSpan);
        }
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

// mb_seed_batch.h -- the seed stage of ALL chunk pairs of a call in shared launches (gfx950, wave64): a sparse seed position table
// per distinct target and one seed search over every (pair, strand) unit.  Included by mb_kernels.hip inside namespace mb (and, with
// MB_EMU defined, by the host-side emulation under tests/emu).
//
// Why: the chunk pairs of a call are independent jobs (/root/reference/src/cactus/paf/local_alignment.py:395-405) and at evolver sizes
// (0.6 Mb x 0.6 Mb) every one of their ~25 seed-stage kernels is far too small for the GPU; a dense 2^24-bucket table is streamed
// three times per build (64 MiB each) for 6 x 10^5 positions.  Here a target's table is
//     bits   2^24 bits   bucket w holds at least one position                      (2 MiB, stays in L2)
//     dir    2^18 x u32  number of occupied buckets before bucket 64 k             (1 MiB)
//     starts             CSR bounds of the occupied buckets in rank order          (<= 4 B per indexed position)
//     positions          the indexed positions, bucket by bucket (arrival order inside a bucket, as in the dense table)
// so a build touches a few MB, and a look-up that passes the bit test costs one more dependent load (dir) than the dense table.
// All targets / all units of the call share every launch: a block finds its target or unit in a small table by its block index.
#pragma once

// (BatchTarget, kBxWordsPerTarget, kBxDirBlocks, kBsTile: mb_common.h)

__device__ __forceinline__ int bx_target_of_block(const BatchTarget *__restrict__ tg, const int n, const int64_t blk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tg[mid].blk0 <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// words of the indexed positions + occupancy bits of their buckets
__global__ __launch_bounds__(256) void k_bx_words(const BatchTarget *__restrict__ tg, const int n_targets, uint32_t *__restrict__ words,
                                                   unsigned long long *__restrict__ bits) {
    const int t = bx_target_of_block(tg, n_targets, (int64_t)blockIdx.x);
    const BatchTarget g = tg[t];
    const int64_t s = ((int64_t)blockIdx.x - g.blk0) * 256 + threadIdx.x;
    if (s >= g.n_slots) return;
    const int64_t p = g.first + s * g.step;
    uint32_t w = 0xFFFFFFFFu;
    if (p + kSeedSpan <= g.n) {
        uint32_t ww;
        if (window_word(unit_glob(g.codes), p, ww)) {
            w = ww;
            atomicOr(&bits[(size_t)t * kBxWordsPerTarget + (ww >> 6)], 1ull << (ww & 63u));
        }
    }
    words[g.slot0 + s] = w;
}

// occupied buckets per 2048 bitmap words (one block)
__global__ __launch_bounds__(256) void k_bx_popc(const unsigned long long *__restrict__ bits, uint32_t *__restrict__ bsum) {
    __shared__ uint32_t tot;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * 2048 + (size_t)threadIdx.x * 8;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) c += (uint32_t)__popcll(bits[base + k]);
    if (c) atomicAdd(&tot, c);                                                  // (LDS)
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// dir[w] = occupied buckets of the target before bitmap word w
__global__ __launch_bounds__(256) void k_bx_dir(const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ bsum, uint32_t *__restrict__ dir) {
    __shared__ uint32_t sc[256], gs[16];
    __shared__ uint32_t before;
    if (threadIdx.x == 0) before = 0;
    __syncthreads();
    const int b_in_t = (int)(blockIdx.x % kBxDirBlocks);
    // blocks of this target before this one (kBxDirBlocks <= 256)
    if ((int)threadIdx.x < b_in_t) { const uint32_t v = bsum[(size_t)(blockIdx.x - b_in_t) + threadIdx.x]; if (v) atomicAdd(&before, v); }
    const size_t base = (size_t)blockIdx.x * 2048 + (size_t)threadIdx.x * 8;
    uint32_t c[8], tsum = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { c[k] = (uint32_t)__popcll(bits[base + k]); tsum += c[k]; }
    sc[threadIdx.x] = tsum;
    __syncthreads();
    if (threadIdx.x < 16) {                                                     // totals of the 16 groups of 16 threads
        uint32_t g = 0;
        for (int k = 0; k < 16; k++) g += sc[16 * threadIdx.x + k];
        gs[threadIdx.x] = g;
    }
    __syncthreads();
    uint32_t run = before;
    for (int g = 0; g < (int)(threadIdx.x >> 4); g++) run += gs[g];
    for (int k = (int)(threadIdx.x & ~15u); k < (int)threadIdx.x; k++) run += sc[k];
#pragma unroll
    for (int k = 0; k < 8; k++) { dir[base + k] = run; run += c[k]; }
}

// rank of an occupied bucket among the occupied buckets of its target
__device__ __forceinline__ uint32_t bx_rank(const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ dir, const int t, const uint32_t w) {
    const size_t at = (size_t)t * kBxWordsPerTarget + (w >> 6);
    return dir[at] + (uint32_t)__popcll(bits[at] & ((1ull << (w & 63u)) - 1ull));
}

// positions per occupied bucket
__global__ __launch_bounds__(256) void k_bx_count(const BatchTarget *__restrict__ tg, const int n_targets, const uint32_t *__restrict__ words,
                                                   const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ dir, uint32_t *__restrict__ cnt) {
    const int t = bx_target_of_block(tg, n_targets, (int64_t)blockIdx.x);
    const BatchTarget g = tg[t];
    const int64_t s = ((int64_t)blockIdx.x - g.blk0) * 256 + threadIdx.x;
    if (s >= g.n_slots) return;
    const uint32_t w = words[g.slot0 + s];
    if (w == 0xFFFFFFFFu) return;
    atomicAdd(&cnt[g.cbase + bx_rank(bits, dir, t, w)], 1u);
}

__global__ __launch_bounds__(256) void k_bx_scatter(const BatchTarget *__restrict__ tg, const int n_targets, const uint32_t *__restrict__ words,
                                                     const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ dir,
                                                     const uint32_t *__restrict__ starts, uint32_t *__restrict__ cursor, uint32_t *__restrict__ positions) {
    const int t = bx_target_of_block(tg, n_targets, (int64_t)blockIdx.x);
    const BatchTarget g = tg[t];
    const int64_t s = ((int64_t)blockIdx.x - g.blk0) * 256 + threadIdx.x;
    if (s >= g.n_slots) return;
    const uint32_t w = words[g.slot0 + s];
    if (w == 0xFFFFFFFFu) return;
    const int64_t r = g.cbase + bx_rank(bits, dir, t, w);
    const uint32_t k = atomicAdd(&cursor[r], 1u);
    positions[starts[r] + k] = (uint32_t)(g.first + s * g.step);
}

// ---- seed search over all units of the call ------------------------------------------------------------------------------------
// The q space of a launch: unit u owns the slots [qpos0, qpos0 + qtot rounded up to 2048) -- a whole number of scan tiles, so the
// scanned tile totals (launch_scan_u32's block_sums) hold, at tile qpos0 / 2048, the number of hits before the unit.

__device__ __forceinline__ int bs_unit_of_slot(const SeedUnit *__restrict__ units, const int n, const int64_t slot) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (units[mid].qpos0 <= slot) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ void bs_lookup(const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ dir, const uint32_t *__restrict__ starts,
                                          const int t, const int64_t cbase, const uint32_t w, uint32_t &b0, uint32_t &b1) {
    const size_t at = (size_t)t * kBxWordsPerTarget + (w >> 6);
    const unsigned long long m = bits[at];
    b0 = b1 = 0;
    if (!((m >> (w & 63u)) & 1ull)) return;
    const int64_t r = cbase + dir[at] + (uint32_t)__popcll(m & ((1ull << (w & 63u)) - 1ull));
    b0 = starts[r]; b1 = starts[r + 1];
}

// (every variant's bitmap word is requested before the first one is looked at, then every directory entry, then every pair of
//  bucket bounds: three rounds of independent loads per thread instead of thirteen dependent chains)
__device__ __forceinline__ void bs_lookup_all(const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ dir, const uint32_t *__restrict__ starts,
                                              const int t, const int64_t cbase, const uint32_t w, const int nvar, uint32_t (&b0)[1 + kSeedWeight],
                                              uint32_t (&b1)[1 + kSeedWeight]) {
    unsigned long long m[1 + kSeedWeight];
    uint32_t wv[1 + kSeedWeight], d[1 + kSeedWeight];
#pragma unroll
    for (int v = 0; v < 1 + kSeedWeight; v++) {
        wv[v] = variant_word(w, v);
        m[v] = v < nvar ? bits[(size_t)t * kBxWordsPerTarget + (wv[v] >> 6)] : 0ull;
    }
#pragma unroll
    for (int v = 0; v < 1 + kSeedWeight; v++) {
        const bool hit = (m[v] >> (wv[v] & 63u)) & 1ull;
        d[v] = hit ? dir[(size_t)t * kBxWordsPerTarget + (wv[v] >> 6)] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int v = 0; v < 1 + kSeedWeight; v++) {
        b0[v] = b1[v] = 0;
        if (d[v] != 0xFFFFFFFFu) {
            const int64_t r = cbase + d[v] + (uint32_t)__popcll(m[v] & ((1ull << (wv[v] & 63u)) - 1ull));
            b0[v] = starts[r]; b1[v] = starts[r + 1];
        }
    }
}

__global__ __launch_bounds__(256) void k_bs_count(const SeedUnit *__restrict__ units, const int n_units, const BatchTarget *__restrict__ tg,
                                                   const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ dir,
                                                   const uint32_t *__restrict__ starts, const int nvar, uint32_t *__restrict__ qcnt) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int u = bs_unit_of_slot(units, n_units, (int64_t)blockIdx.x * 256);     // (a block lies inside one unit: qpos0 is a multiple of 2048)
    const SeedUnit su = units[u];
    const int64_t q = slot - su.qpos0;
    uint32_t cnt = 0, w;
    if (q + kSeedSpan <= su.qtot && window_word(unit_glob(su.qc), q, w)) {
        uint32_t b0[1 + kSeedWeight], b1[1 + kSeedWeight];
        bs_lookup_all(bits, dir, starts, su.index, tg[su.index].cbase, w, nvar, b0, b1);
#pragma unroll
        for (int v = 0; v < 1 + kSeedWeight; v++) cnt += b1[v] - b0[v];
    }
    qcnt[slot] = cnt;
}

__global__ __launch_bounds__(256) void k_bs_fill(const SeedUnit *__restrict__ units, const int n_units, const BatchTarget *__restrict__ tg,
                                                  const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ dir,
                                                  const uint32_t *__restrict__ starts, const uint32_t *__restrict__ positions, const int nvar,
                                                  const uint32_t *__restrict__ hit_off, unsigned long long *__restrict__ keys) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int u = bs_unit_of_slot(units, n_units, (int64_t)blockIdx.x * 256);
    const SeedUnit su = units[u];
    const int64_t q = slot - su.qpos0;
    uint32_t w;
    if (!(q + kSeedSpan <= su.qtot && window_word(unit_glob(su.qc), q, w))) return;
    uint32_t b0[1 + kSeedWeight], b1[1 + kSeedWeight];
    bs_lookup_all(bits, dir, starts, su.index, tg[su.index].cbase, w, nvar, b0, b1);
    uint32_t o = hit_off[slot];
    const unsigned long long q_end = (unsigned long long)(q + kSeedSpan);
    const int64_t dq0 = (int64_t)su.dbase + su.qtot - q;                          // diagonal of target position 0 against this q
#pragma unroll
    for (int v = 0; v < 1 + kSeedWeight; v++)
        for (uint32_t k = b0[v]; k < b1[v]; k++) keys[o++] = ((unsigned long long)(dq0 + (int64_t)positions[k]) << 32) | q_end;
}

// mb_seedword.h -- the 12-of-19 spaced seed word of a window and its one-transition variants.  Included inside
// namespace mb by mb_kernels.hip (and, with MB_EMU defined, by the host-side emulation under tests/emu).
#pragma once

__device__ __forceinline__ bool window_word(const uint8_t *codes, int64_t p, uint32_t &word) {
    // care offsets of 1110100110010101111
    unsigned bad = 0;
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < kSeedSpan; k++) {
        unsigned c = codes[p + k];
        bad |= c;                                  // any code >= 4 (N, lowercase bit 3, separator) sets bits 2..7
        const bool care = (k == 0 || k == 1 || k == 2 || k == 4 || k == 7 || k == 8 || k == 11 || k == 13 || k == 15 ||
                           k == 16 || k == 17 || k == 18);
        if (care) w = (w << 2) | (c & 3u);
    }
    word = w;
    return (bad & 0xFCu) == 0;
}

__device__ __forceinline__ uint32_t variant_word(uint32_t w, int v) {
    // v = 0 exact ; v = 1..12 transition (xor 2) at care position v-1, first care base most significant
    return v == 0 ? w : (w ^ (2u << (2 * (kSeedWeight - v))));
}


// ---- bucket numbering of the DENSE seed table (one large target; mb_seed_dense.h) ---------------------------------------------------------
// The 13 words a query position looks up differ from one another in the HIGH bit of one base (A <-> G, C <-> T).  Numbering the buckets
// by (low bits of the 12 bases) << 12 | (high bits of the 12 bases) puts the 13 of them into one stretch of 4096 buckets -- 512 B of the
// occupancy bitmap, 10 of the 13 bits in one 128-byte line, 6 in one word -- where the plain word spreads them over 13 cache lines
// (8 MiB apart for the first base).  A look-up is a cache-line transaction whatever it reads; the transactions are what bounds the search.
__device__ __forceinline__ uint32_t dense_bucket(uint32_t w) {
    uint32_t e = w & 0x555555u, o = (w >> 1) & 0x555555u;
    e = (e | (e >> 1)) & 0x333333u; o = (o | (o >> 1)) & 0x333333u;
    e = (e | (e >> 2)) & 0x0F0F0Fu; o = (o | (o >> 2)) & 0x0F0F0Fu;
    e = (e | (e >> 4)) & 0x00FF00FFu; o = (o | (o >> 4)) & 0x00FF00FFu;
    e = (e | (e >> 8)) & 0xFFFu; o = (o | (o >> 8)) & 0xFFFu;
    return (e << 12) | o;
}
// bucket of variant v of the word whose bucket is b: v = 0 exact; v = 1..12 transition at care position v - 1 (first care base most significant)
__device__ __forceinline__ uint32_t dense_variant(uint32_t b, int v) { return v == 0 ? b : (b ^ (1u << (kSeedWeight - v))); }

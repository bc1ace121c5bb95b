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


// mb_xdrop.h -- ungapped x-drop extension, lane-per-direction building blocks: substitution scores on code bytes,
// 8 columns per load, branch-free inside a chunk.  Included by mb_kernels.hip inside namespace mb (and, with MB_EMU defined, by the
// host-side emulation test under tests/emu, which supplies the wave primitives itself).
#pragma once

#ifndef MB_EMU
// wave primitives (the emulation defines its own)
template <int CTRL, int BANK>
__device__ __forceinline__ int wdpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, BANK, false); }
__device__ __forceinline__ unsigned long long wballot(bool p) { return __ballot(p); }
__device__ __forceinline__ uint32_t wperm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ int wsdot4(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }
__device__ __forceinline__ int wreadlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }                      // l: wave-uniform
__device__ __forceinline__ unsigned wsignin(unsigned acc, int d) { return __builtin_amdgcn_alignbit(acc, (unsigned)d, 31); }    // (acc << 1) | sign(d)
#endif

// ------------------------------------------------------------------------------------------------
// substitution score on code bytes: HOXD70, N (code 4) scores -100 against anything (A.2)
__device__ __forceinline__ int sub_score(unsigned a, unsigned b) {
    // branch-free on purpose (the compiler turns an if-chain into exec-mask branches inside the hot loops):
    // d = x ^ y selects match / transition / the two transversion classes, `at` tells A,T from C,G
    const unsigned x = a & 7u, y = b & 7u;
    const unsigned d = (x ^ y) & 3u;
    const unsigned cg = (x ^ (x >> 1)) & 1u;                   // 1 for C,G ; 0 for A,T
    // byte k of the table = score(d = k) + 128 :  match, A-C/G-T (-114), transition (-31), A-T / C-G
    const unsigned lut = cg ? (228u | (14u << 8) | (97u << 16) | (3u << 24)) : (219u | (14u << 8) | (97u << 16) | (5u << 24));
    const int v = (int)((lut >> (d * 8u)) & 0xFFu) - 128;
    return ((x | y) & 4u) ? -100 : v;
}


__device__ __forceinline__ unsigned long long load8(const uint8_t *p) {
    unsigned long long v;
    __builtin_memcpy(&v, p, 8);                       // unaligned global_load_dwordx2
    return v;
}


// One x-drop direction, 8 columns per load, branch-free inside a chunk (every per-lane condition is a select).  The first
// NPRE chunks of both sequences arrive PRELOADED: the extension of a chance hit ends within ~35 columns to the left (the seed
// itself is 19 of them) and ~17 to the right, and a chunk-by-chunk loop would pay one dependent cache-line round trip per
// chunk and sequence -- the whole cost of this stage.  The caller issues every preload of both directions before the first
// column is scored, so a typical hit waits for memory once.
struct XState { int run, best, bpos; bool live; };

// four substitution scores at once: byte m of the result = score(a_m, b_m) + 128 for the code bytes a_m, b_m of a4 / b4.
// HOXD70 is a function of (a ^ b) and of whether a is C/G: one v_perm_b32 over an 8-byte table; N (code bit 2) scores -100.
// (Separator bytes have bit 2 set as well: the caller deals with them before looking at the score.)
__device__ __forceinline__ uint32_t scores4(const uint32_t a4, const uint32_t b4) {
    const uint32_t d = (a4 ^ b4) & 0x03030303u;                       // 0 match, 2 transition, 1 / 3 the two transversion classes
    const uint32_t cg = ((a4 ^ (a4 >> 1)) & 0x01010101u) << 2;        // 4 where a is C or G
    constexpr uint32_t kAT = 219u | (14u << 8) | (97u << 16) | (5u << 24);      // a in {A,T}: 91, -114, -31, -123 (+128)
    constexpr uint32_t kCG = 228u | (14u << 8) | (97u << 16) | (3u << 24);      // a in {C,G}: 100, -114, -31, -125
    const uint32_t s = wperm(kCG, kAT, d | cg);       // selector 0..3 -> kAT, 4..7 -> kCG
    const uint32_t nm = (((a4 | b4) & 0x04040404u) >> 2) * 255u;      // 0xFF where either base is N
    return (s & ~nm) | (0x1c1c1c1cu & nm);                            // -100 + 128
}

template <int DIR, typename CNT>
__device__ __forceinline__ void xdrop_chunk(const unsigned long long a8, const unsigned long long b8, const int c, const int xdrop,
                                            XState &x, CNT &ncols) {
    if (((a8 | b8) & 0x8080808080808080ull) != 0ull) {
        // a contig separator (0xFF) inside the chunk -- the only codes with bit 7: the extension ends there, column by column
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int sh = DIR > 0 ? 8 * m : 8 * (7 - m);
            const unsigned a = (unsigned)(a8 >> sh) & 0xFFu, b = (unsigned)(b8 >> sh) & 0xFFu;
            x.live = x.live & (a != kSep) & (b != kSep);
            x.run = x.live ? x.run + sub_score(a, b) : x.run;
            ncols += x.live ? 1u : 0u;
            const bool upd = x.live & (x.run > x.best);
            x.best = upd ? x.run : x.best;
            x.bpos = upd ? 8 * c + m + 1 : x.bpos;
            x.live = x.live & (upd | (x.run >= x.best - xdrop));
        }
        return;
    }
    const uint32_t s_lo = scores4((uint32_t)a8, (uint32_t)b8), s_hi = scores4((uint32_t)(a8 >> 32), (uint32_t)(b8 >> 32));
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const int k = DIR > 0 ? m : 7 - m;                               // byte of the chunk holding column m
        const int sc = (int)(((k < 4 ? s_lo : s_hi) >> (8 * (k & 3))) & 0xFFu) - 128;
        x.run = x.live ? x.run + sc : x.run;
        ncols += x.live ? 1u : 0u;
        const bool upd = x.live & (x.run > x.best);
        x.best = upd ? x.run : x.best;
        x.bpos = upd ? 8 * c + m + 1 : x.bpos;
        x.live = x.live & (upd | (x.run >= x.best - xdrop));
    }
}

template <int DIR, int NPRE>
__device__ __forceinline__ void xdrop_preload(const uint8_t *__restrict__ tp, const uint8_t *__restrict__ qp,
                                              unsigned long long (&a)[NPRE], unsigned long long (&b)[NPRE]) {
#pragma unroll
    for (int c = 0; c < NPRE; c++) {
        a[c] = DIR > 0 ? load8(tp + 8 * c) : load8(tp - 8 * (c + 1));
        b[c] = DIR > 0 ? load8(qp + 8 * c) : load8(qp - 8 * (c + 1));
    }
}

template <int DIR, int NPRE>
__device__ __forceinline__ void xdrop_dir(const uint8_t *__restrict__ tp, const uint8_t *__restrict__ qp, const int xdrop,
                                          const unsigned long long (&a)[NPRE], const unsigned long long (&b)[NPRE],
                                          int &best_out, int &pos_out, unsigned long long &ncols) {
    XState x{0, 0, 0, true};
#pragma unroll
    for (int c = 0; c < NPRE; c++)
        if (x.live) xdrop_chunk<DIR>(a[c], b[c], c, xdrop, x, ncols);
    for (int c = NPRE; x.live; c++) {
        const unsigned long long a8 = DIR > 0 ? load8(tp + 8 * c) : load8(tp - 8 * (c + 1));
        const unsigned long long b8 = DIR > 0 ? load8(qp + 8 * c) : load8(qp - 8 * (c + 1));
        xdrop_chunk<DIR>(a8, b8, c, xdrop, x, ncols);
    }
    best_out = x.best; pos_out = x.bpos;
}


// mb_ungapped_ux.h -- level-synchronous ungapped extension for dense hit sets (gfx950, wave64).
// Included by mb_kernels.hip inside namespace mb after mb_xdrop.h and mb_ungapped_grp.h (and, with MB_EMU defined, by the host-side
// emulation test under tests/emu).
//
// Why: k_ungapped gives a diagonal run to a lane, and a wave lasts as long as its longest lane -- longest run times longest
// extension.  On a dense chance-hit set (8 Mb x 8 Mb random: 9 hits per diagonal, 53 columns per hit) that is 97 issue slots per
// hit against ~15 of useful work.  A chance hit's two walks almost always end inside the first 64 columns and
// never reach another hit of their diagonal, so the walks do not have to wait for each other:
//   1. k_ux_extend   one HIT per lane.  Level 1 = a fixed 5 + 3 chunks (40 columns to the left, 24 to the right) with no early
//                    exit: every lane of every wave does the same work.  The hits that are not finished (about a third) are
//                    packed at the front of the block through LDS and get a fixed second helping (level 2) from the waves
//                    they fill; what is still running after that (real homology) is written to a list.
//   2. k_ux_tail     the listed hits, eight lanes per hit and 64 columns per step (grp_step of mb_ungapped_grp.h) to the end.
//   Every hit leaves an 8-byte record (length to the right, columns counted) and, if it scores >= K, a candidate HSP.
//   3. The sequential rule (a hit with q_end <= extent is dropped, else extent = q_end + length to the right).  If every hit of
//      a run ends before the next hit of the run begins, nothing is dropped: a finished hit looks at its successor and marks
//      the diagonal "dirty" otherwise.  k_ux_accept, one hit per lane, then takes every hit of the clean short runs (counters,
//      extent[], valid marks on candidates); k_ux_resolve walks the few dirty runs hit by hit, one run per lane.
//   4. k_ux_census   valid candidates get their identical-base census; the others are marked (score = INT_MIN) and the host
//                    skips them.
// A walk depends on the sequences only, never on other hits, so extending a hit that the rule later drops changes nothing
// but the work done: results are the sequential ones.  Runs longer than the long-run threshold stay with k_ungapped_long
// (almost all of their hits are dropped there with one ballot per 64); their diagonals are marked in a bitmap so that this
// pipeline stops working on them after level 1.
#pragma once

// (UxEntry and UxScratch: mb_common.h)

namespace ux {

// Chance hits of the 12-of-19 seed under HOXD70 / x-drop 910 (simulation, 2 x 10^5 hits): the left walk looks at 40 columns on
// average (the seed is 19 of them), P(> 48) = 0.14, P(> 72) = 0.005; the right walk at 21, P(> 32) = 0.09, P(> 56) = 0.003.
constexpr int kL1 = 6, kR1 = 4;               // level 1: chunks to the left / right -- 80 % of the chance hits end here
constexpr int kL2 = 2, kR2 = 2;               // level 2: further chunks per direction -- all but ~2.5 % of the hits are done after it
static_assert(kL1 % 2 == 0 && kR1 % 2 == 0 && kL2 % 2 == 0 && kR2 % 2 == 0, "chunks are loaded two at a time");
constexpr int kBlock = 256;
constexpr unsigned kSlots0 = 12, kSlots1 = 4;  // entry slots owned by wave 0 / wave 1 of a block of k_ux_extend (blk_entries, blk_cnt)
constexpr uint32_t kCand = 0x80000000u;

__device__ __forceinline__ uint32_t plane_bit(const UxScratch &sc, const uint32_t dq) { return (dq * sc.plane_mul) & sc.plane_mask; }
__device__ __forceinline__ bool on_long_diagonal(const UxScratch &sc, const uint32_t dq) { const uint32_t b = plane_bit(sc, dq); return (sc.long_bits[b >> 5] >> (b & 31u)) & 1u; }

// Eight columns of one direction, no early exit and six instructions per column: the running score of every column is a
// v_dot4c prefix of the signed score bytes; "best so far" is a running maximum of KEYS (score << 3 | 7 - column), so that the
// first column of the best score comes with it; column m stops iff (score + xdrop) << 3 | 7 is below the key maximum before it
// (carried best: best << 3 | 7, so an equal score never looks like an improvement).  What a sequential walk would not have
// looked at lies behind the first stop and is masked out afterwards.  Same result as xdrop_chunk on chunks without a contig
// separator; a hit with a separator inside the columns of a level skips that level (k_ux_tail's steps know separators).  Keeping
// the column-by-column separator path out of here also keeps k_ux_extend's 18 chunk instances inside the instruction cache.
template <int DIR, typename CNT>
__device__ __forceinline__ void ux_chunk(const unsigned long long a8, const unsigned long long b8, const int c, const int xdrop,
                                         XState &x, CNT &ncols) {
    // (precondition: no contig separator in the chunk -- the callers hand such hits on untouched)
    // column m = byte m: to the left the bytes come reversed
    const uint32_t a_lo = DIR > 0 ? (uint32_t)a8 : wperm((uint32_t)(a8 >> 32), (uint32_t)a8, 0x04050607u);
    const uint32_t a_hi = DIR > 0 ? (uint32_t)(a8 >> 32) : wperm((uint32_t)(a8 >> 32), (uint32_t)a8, 0x00010203u);
    const uint32_t b_lo = DIR > 0 ? (uint32_t)b8 : wperm((uint32_t)(b8 >> 32), (uint32_t)b8, 0x04050607u);
    const uint32_t b_hi = DIR > 0 ? (uint32_t)(b8 >> 32) : wperm((uint32_t)(b8 >> 32), (uint32_t)b8, 0x00010203u);
    uint32_t s_lo = ugrp::scores4s(a_lo, b_lo), s_hi = ugrp::scores4s(a_hi, b_hi);
    const uint32_t ab_lo = a_lo | b_lo, ab_hi = a_hi | b_hi;
    if (((ab_lo | ab_hi) & 0x04040404u) != 0u) { s_lo = ugrp::scores4_fix_n(s_lo, ab_lo); s_hi = ugrp::scores4_fix_n(s_hi, ab_hi); }   // an N: rare
    int p[8];
    p[0] = wsdot4(s_lo, 0x00000001u, x.run); p[1] = wsdot4(s_lo, 0x00000101u, x.run);
    p[2] = wsdot4(s_lo, 0x00010101u, x.run); p[3] = wsdot4(s_lo, 0x01010101u, x.run);
    p[4] = wsdot4(s_hi, 0x00000001u, p[3]); p[5] = wsdot4(s_hi, 0x00000101u, p[3]);
    p[6] = wsdot4(s_hi, 0x00010101u, p[3]); p[7] = wsdot4(s_hi, 0x01010101u, p[3]);
    const int kin = x.best * 8 + 7;
    const int x8 = xdrop * 8;
    int bk[8];
    unsigned sm = 0;
    int prevk = kin;
    int keys8[8];
#pragma unroll
    for (int m = 0; m < 8; m++) { keys8[m] = p[m] * 8 + (7 - m); bk[m] = max(prevk, keys8[m]); prevk = bk[m]; }
#pragma unroll
    for (int m = 7; m >= 0; m--) sm = wsignin(sm, keys8[m] + x8 + m - (m == 0 ? kin : bk[m - 1]));
    const int fs = sm ? __ffs((int)sm) - 1 : 8;                        // first stopping column (it is looked at, and never a new best)
    const int e = fs - 1;                                             // last column that may hold a new best
    const int b0 = -(e & 1), b1 = -((e >> 1) & 1), b2 = -((e >> 2) & 1);
    auto sel = [](int mask, int one, int zero) { return (one & mask) | (zero & ~mask); };
    int bke = sel(b2, sel(b1, sel(b0, bk[7], bk[6]), sel(b0, bk[5], bk[4])), sel(b1, sel(b0, bk[3], bk[2]), sel(b0, bk[1], bk[0])));
    bke = e < 0 ? kin : bke;
    const bool upd = x.live & (bke > kin);
    x.best = upd ? bke >> 3 : x.best;
    x.bpos = upd ? 8 * c + 8 - (bke & 7) : x.bpos;
    ncols += x.live ? (CNT)min(fs + 1, 8) : (CNT)0;
    x.run = p[7];
    x.live = x.live & (sm == 0u);
}

// 16 bytes at any alignment (one global_load_dwordx4): two chunks of a walk.  k_ux_extend is bound by the cache lines its
// lanes look up in the vector L1 (one line per clock and CU), not by arithmetic: half the loads, half the look-ups.
struct U16 { unsigned long long lo, hi; };
__device__ __forceinline__ U16 load16(const uint8_t *p) { U16 v; __builtin_memcpy(&v, p, 16); return v; }
// chunks 2 j and 2 j + 1 of the left walk (columns 16 j .. 16 j + 15 counted from the seed end) / of the right walk
__device__ __forceinline__ void load_left2(const uint8_t *p_end, const int j, unsigned long long &c_even, unsigned long long &c_odd) {
    const U16 v = load16(p_end - 16 * (j + 1));
    c_even = v.hi; c_odd = v.lo;
}
__device__ __forceinline__ void load_right2(const uint8_t *p_end, const int j, unsigned long long &c_even, unsigned long long &c_odd) {
    const U16 v = load16(p_end + 16 * j);
    c_even = v.lo; c_odd = v.hi;
}

// ---- level 1 from the PACKED strands (round 6) ------------------------------------------------------------------------------------
// The windows of a hit -- 48 columns to the left of the seed end, 32 to the right, in both sequences -- are 160 random bytes: 4.5 cache
// lines of 128 B per hit, which is what bounds this kernel on a chunk pair (~ 400 B fetched per hit at 4.6 TB/s).
// The extension's own packed form of a strand (k_pack2bit_mask writes it beside the seed stage's planes) holds 32 bases in a 12-byte record:
// the 2-bit codes as in the seed stage's plane (first base most significant) and ONE BIT per base that is set for N / IUPAC codes and
// separators -- not for soft-masked bases, which extend like any other.  The 128 bases around a seed end are five records, 60 contiguous
// bytes per sequence (1.5 cache lines instead of 2.25), and a strand is 0.375 B per base instead of 1: a 4.5 Mb chunk pair's strands stay
// in an XCD's L2.  A window that holds an N or a separator, or crosses an end of the set, takes the byte path as before.  The 2-bit fields
// become the code bytes ux_chunk works on through a 256-entry table in LDS (four bases per look-up): everything behind the loads is the
// byte path's own code, so the results are the byte path's.
struct PkWin { unsigned long long a[4], n0, n1; };     // bases [e - 64, e + 64): base e - 64 + i = bits 63 - 2 (i & 31), 62 - 2 (i & 31) of a[i >> 5]; n0 / n1: its bit 63 - (i & 63)
struct PkRec { uint32_t a_lo, a_hi, n; };               // 32 bases: base k = bits 63 - 2 k, 62 - 2 k of (a_hi : a_lo), its N / separator flag bit 31 - k of n
__device__ __forceinline__ unsigned long long pk_funnel(const unsigned long long hi, const unsigned long long lo, const unsigned sh) {
    return sh ? (hi << sh) | (lo >> (64u - sh)) : hi;             // 64 bits from bit `sh` (counted from the top, 0 .. 63) of hi:lo
}
__device__ __forceinline__ PkWin pk_load(const uint32_t *__restrict__ px, const int64_t e) {
    const int64_t b = e - 64;
    PkRec r[5];                                                       // 60 contiguous bytes, dword aligned: three 16-byte loads and one of 12
#ifdef MB_EMU
    __builtin_memcpy(r, px + 3 * (b >> 5), sizeof r);
#else
    typedef uint32_t pk_u4 __attribute__((ext_vector_type(4), aligned(4)));
    typedef uint32_t pk_u3 __attribute__((ext_vector_type(3), aligned(4)));
    const uint32_t *src = px + 3 * (b >> 5);
    const pk_u4 v0 = *(const pk_u4 *)src, v1 = *(const pk_u4 *)(src + 4), v2 = *(const pk_u4 *)(src + 8);
    const pk_u3 v3 = *(const pk_u3 *)(src + 12);
    r[0] = PkRec{v0.x, v0.y, v0.z}; r[1] = PkRec{v0.w, v1.x, v1.y}; r[2] = PkRec{v1.z, v1.w, v2.x}; r[3] = PkRec{v2.y, v2.z, v2.w}; r[4] = PkRec{v3.x, v3.y, v3.z};
#endif
    const unsigned s1 = (unsigned)(b & 31), s2 = 2u * s1;
    unsigned long long w[5];
#pragma unroll
    for (int k = 0; k < 5; k++) w[k] = ((unsigned long long)r[k].a_hi << 32) | r[k].a_lo;
    PkWin o;
#pragma unroll
    for (int k = 0; k < 4; k++) o.a[k] = pk_funnel(w[k], w[k + 1], s2);
    // (s1 <= 31: the third word's share is its top s1 bits)
    o.n0 = ((((unsigned long long)r[0].n << 32) | r[1].n) << s1) | ((unsigned long long)r[2].n >> (32u - s1));
    o.n1 = ((((unsigned long long)r[2].n << 32) | r[3].n) << s1) | ((unsigned long long)r[4].n >> (32u - s1));
    return o;
}
// the eight code bytes (memory order: byte k = the base at position start + k) of a 16-bit field whose first base is most significant
__device__ __forceinline__ unsigned long long pk_bytes(const uint32_t *lut, const unsigned f16) {
    return (unsigned long long)lut[(f16 >> 8) & 0xFFu] | ((unsigned long long)lut[f16 & 0xFFu] << 32);
}
// chunk c of the left walk = bases [e - 8 (c + 1), e - 8 c), of the right walk = bases [e + 8 c, e + 8 c + 8)   (c: compile-time)
template <int C> __device__ __forceinline__ unsigned pk_left(const PkWin &w) { return (unsigned)(w.a[1 - (C >> 2)] >> (16 * (C & 3))) & 0xFFFFu; }
template <int C> __device__ __forceinline__ unsigned pk_right(const PkWin &w) { return (unsigned)(w.a[2 + (C >> 2)] >> (48 - 16 * (C & 3))) & 0xFFFFu; }
__device__ __forceinline__ uint32_t pk_lut_entry(const unsigned h) {      // h = four bases, the first in bits 7..6
    return ((h >> 6) & 3u) | (((h >> 4) & 3u) << 8) | (((h >> 2) & 3u) << 16) | ((h & 3u) << 24);
}

// a finished hit: its record and, if it scores, its candidate HSP (census later, only if the rule keeps the hit)
__device__ __forceinline__ void finish_hit(const uint32_t i, const uint32_t dq, const int unit, const int32_t q_end, const int64_t t_end, const int best_l,
                                           const int bl, const int best_r, const int br, const uint32_t cols, const int K,
                                           const unsigned long long *__restrict__ keys, const int64_t n_hits, const UxScratch &sc,
                                           DevHsp *__restrict__ hsps, const int64_t hsp_cap, UngappedCounters *__restrict__ ctr, const UnitTab &ut) {
    unsigned long long *__restrict__ rec = sc.rec;
    // Does this walk reach the next hit of the diagonal -- or, in a later q batch, does an earlier batch's extent reach the run's
    // first hit?  Then the run needs the sequential rule (k_ux_resolve).
    bool dirty = false;
    if ((int64_t)i + 1 < n_hits) {
        const unsigned long long nk = keys[i + 1];
        dirty = (uint32_t)(nk >> 32) == dq && q_end + br >= (int32_t)(uint32_t)nk;
    }
    if (sc.extent_live && sc.extent && (i == 0 || (uint32_t)(keys[i - 1] >> 32) != dq)) dirty |= sc.extent[extent_slot(ut, dq)] >= q_end;
    if (dirty) { const uint32_t b = plane_bit(sc, dq); atomicOr(&sc.dirty_bits[b >> 5], 1u << (b & 31u)); }
    uint32_t x = cols;
    const int score = best_l + best_r;
    if (score >= K && !on_long_diagonal(sc, dq)) {
        const unsigned long long slot = atomicAdd(&ctr->hsps, 1ull);
        if ((int64_t)slot < hsp_cap) {
            DevHsp hs;
            hs.t_start = (int32_t)(t_end - bl); hs.q_start = q_end - bl; hs.len = bl + br; hs.score = score;
            hs.seed_t_end = (int32_t)t_end; hs.seed_q_end = q_end;
            hs.cnt[0] = (int32_t)cols; hs.cnt[1] = 0; hs.cnt[2] = 0; hs.cnt[3] = 0;       // cnt[1]: set by k_ux_resolve when the rule keeps the hit
            hs.unit = unit;
            hsps[slot] = hs;
            x = kCand | (uint32_t)slot;
        }
    }
    rec[i] = ((unsigned long long)x << 32) | (uint32_t)br;
}

}  // namespace ux

__global__ __launch_bounds__(256) void k_ux_mark_long(const unsigned long long *__restrict__ keys, const unsigned *__restrict__ heads_long,
                                                      const unsigned *__restrict__ n_long, const UxScratch sc) {
    const unsigned h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= *n_long) return;
    const uint32_t b = ux::plane_bit(sc, (uint32_t)(keys[heads_long[h]] >> 32));
    atomicOr(&sc.long_bits[b >> 5], 1u << (b & 31u));
}

template <bool PACKED>
__device__ __forceinline__ void ux_extend_body(const unsigned long long *__restrict__ keys, const int64_t n_hits,
                                               const UnitTab ut,
                                               const int xdrop, const int K, const UxScratch sc, DevHsp *__restrict__ hsps,
                                               const int64_t hsp_cap, UngappedCounters *__restrict__ ctr) {
    using namespace ux;
    __shared__ UxEntry slots[kBlock];
    __shared__ unsigned wave_cnt[kBlock / 64];
    __shared__ uint32_t pk_lut[PACKED ? 256 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * kBlock + tid;
    const bool valid = i < n_hits;
    if (PACKED) {                                                        // (kBlock = 256: an entry per work-item)
        pk_lut[tid & 255] = pk_lut_entry((unsigned)tid & 255u);
        __syncthreads();
    }
    // ---- level 1: every lane the same 5 + 3 chunks
    UxEntry e;
    e.i = (uint32_t)i; e.cl = -1; e.cr = -1; e.run_l = 0; e.best_l = 0; e.bpos_l = 0; e.run_r = 0; e.best_r = 0; e.bpos_r = 0; e.cols = 0;
    bool unfinished = false;
    if (valid) {
        const unsigned long long key = keys[i];
        const uint32_t dq = (uint32_t)(key >> 32);
        const int32_t q_end = (int32_t)(uint32_t)key;
        const UnitRef un = unit_of(ut, dq);
        const uint8_t *tc = un.tc, *qc = un.qc;
        const int64_t t_end = (int64_t)dq - un.qoff + q_end;
        unsigned long long aL[kL1], bL[kL1], aR[kR1], bR[kR1];
        bool from_packed = false;
        if (PACKED) {
            static_assert(kL1 == 6 && kR1 == 4, "the packed window holds 64 bases either side of the seed end");
            // (both windows inside their sets: the packed planes hold nothing before position 0)
            if (t_end >= 64 && t_end + 64 <= sc.t_n && q_end >= 64 && (int64_t)q_end + 64 <= sc.q_n) {
                const PkWin wt = pk_load(sc.t_px, t_end), wq = pk_load(sc.q_px, (int64_t)q_end);
                // an N or a separator among the level's columns [e - 48, e + 32) = bits 16 .. 95 of the window
                const unsigned long long special = ((wt.n0 | wq.n0) & 0x0000FFFFFFFFFFFFull) | ((wt.n1 | wq.n1) & 0xFFFFFFFF00000000ull);
                if (special == 0ull) {
                    aL[0] = pk_bytes(pk_lut, pk_left<0>(wt)); aL[1] = pk_bytes(pk_lut, pk_left<1>(wt)); aL[2] = pk_bytes(pk_lut, pk_left<2>(wt));
                    aL[3] = pk_bytes(pk_lut, pk_left<3>(wt)); aL[4] = pk_bytes(pk_lut, pk_left<4>(wt)); aL[5] = pk_bytes(pk_lut, pk_left<5>(wt));
                    bL[0] = pk_bytes(pk_lut, pk_left<0>(wq)); bL[1] = pk_bytes(pk_lut, pk_left<1>(wq)); bL[2] = pk_bytes(pk_lut, pk_left<2>(wq));
                    bL[3] = pk_bytes(pk_lut, pk_left<3>(wq)); bL[4] = pk_bytes(pk_lut, pk_left<4>(wq)); bL[5] = pk_bytes(pk_lut, pk_left<5>(wq));
                    aR[0] = pk_bytes(pk_lut, pk_right<0>(wt)); aR[1] = pk_bytes(pk_lut, pk_right<1>(wt)); aR[2] = pk_bytes(pk_lut, pk_right<2>(wt)); aR[3] = pk_bytes(pk_lut, pk_right<3>(wt));
                    bR[0] = pk_bytes(pk_lut, pk_right<0>(wq)); bR[1] = pk_bytes(pk_lut, pk_right<1>(wq)); bR[2] = pk_bytes(pk_lut, pk_right<2>(wq)); bR[3] = pk_bytes(pk_lut, pk_right<3>(wq));
                    from_packed = true;
                }
            }
        }
        if (!from_packed) {
#pragma unroll
            for (int j = 0; j < kL1 / 2; j++) { load_left2(tc + t_end, j, aL[2 * j], aL[2 * j + 1]); load_left2(qc + q_end, j, bL[2 * j], bL[2 * j + 1]); }
#pragma unroll
            for (int j = 0; j < kR1 / 2; j++) { load_right2(tc + t_end, j, aR[2 * j], aR[2 * j + 1]); load_right2(qc + q_end, j, bR[2 * j], bR[2 * j + 1]); }
        }
        XState xl{0, 0, 0, true}, xr{0, 0, 0, true};
        uint32_t cols = 0;
        unsigned long long seps = 0;
#pragma unroll
        for (int c = 0; c < kL1; c++) seps |= aL[c] | bL[c];
#pragma unroll
        for (int c = 0; c < kR1; c++) seps |= aR[c] | bR[c];
        const bool clean = (seps & 0x8080808080808080ull) == 0ull;      // no contig separator within the level's columns
        if (clean) {
#pragma unroll
            for (int c = 0; c < kL1; c++) ux_chunk<-1>(aL[c], bL[c], c, xdrop, xl, cols);
#pragma unroll
            for (int c = 0; c < kR1; c++) ux_chunk<+1>(aR[c], bR[c], c, xdrop, xr, cols);
        }
        if (xl.live | xr.live) {
            // (a diagonal of k_ungapped_long: nobody reads this hit's record)
            unfinished = !on_long_diagonal(sc, dq);
            e.cl = xl.live ? (clean ? kL1 : 0) : -1; e.cr = xr.live ? (clean ? kR1 : 0) : -1;
            e.run_l = xl.run; e.best_l = xl.best; e.bpos_l = xl.bpos; e.run_r = xr.run; e.best_r = xr.best; e.bpos_r = xr.bpos; e.cols = cols;
        } else {
            finish_hit((uint32_t)i, dq, un.id, q_end, t_end, xl.best, xl.bpos, xr.best, xr.bpos, cols, K, keys, n_hits, sc, hsps, hsp_cap, ctr, ut);
        }
    }
    // ---- the unfinished hits of the block, packed at the front
    const unsigned long long um = wballot(unfinished);
    if (lane == 0) wave_cnt[wv] = (unsigned)__popcll(um);
    __syncthreads();
    unsigned before = 0, n_unf = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) { const unsigned c = wave_cnt[w]; before += w < wv ? c : 0u; n_unf += c; }
    if (unfinished) slots[before + (unsigned)__popcll(um & ((1ull << lane) - 1ull))] = e;
    __syncthreads();
    if ((unsigned)(tid & ~63) >= n_unf) return;                          // (whole waves leave)
    // ---- level 2: 4 more chunks of each direction that is still running
    const bool mine = (unsigned)tid < n_unf;
    bool spill = false;
    uint32_t dq = 0; int32_t q_end = 0; int64_t t_end = 0;
    XState xl{0, 0, 0, false}, xr{0, 0, 0, false};
    UnitRef un = unit_by_id(ut, 0);
    const uint8_t *tc = un.tc, *qc = un.qc;
    if (mine) {
        e = slots[tid];
        const unsigned long long key = keys[e.i];
        dq = (uint32_t)(key >> 32);
        q_end = (int32_t)(uint32_t)key;
        un = unit_of(ut, dq); tc = un.tc; qc = un.qc;
        t_end = (int64_t)dq - un.qoff + q_end;
        const int cl0 = max(e.cl, 0), cr0 = max(e.cr, 0);
        unsigned long long aL[kL2], bL[kL2], aR[kR2], bR[kR2];
        bool from_packed = false;
        if (PACKED) {
            static_assert(kL2 == 2 && kR2 == 2, "level 2 = the outer 16 columns either side of the packed window");
            // a hit that comes from a complete level 1 (a direction that has ended does not matter): its level-2 columns [e - 64, e - 48) and
            // [e + 32, e + 48) are the ends of the same packed window -- the records level 1 read a moment ago
            if ((e.cl == kL1 || e.cl < 0) && (e.cr == kR1 || e.cr < 0) && t_end >= 64 && t_end + 64 <= sc.t_n && q_end >= 64 && (int64_t)q_end + 64 <= sc.q_n) {
                const PkWin wt = pk_load(sc.t_px, t_end), wq = pk_load(sc.q_px, (int64_t)q_end);
                const unsigned long long special = ((wt.n0 | wq.n0) >> 48) | (((wt.n1 | wq.n1) >> 16) & 0xFFFFull);
                if (special == 0ull) {
                    aL[0] = pk_bytes(pk_lut, pk_left<6>(wt)); aL[1] = pk_bytes(pk_lut, pk_left<7>(wt)); bL[0] = pk_bytes(pk_lut, pk_left<6>(wq)); bL[1] = pk_bytes(pk_lut, pk_left<7>(wq));
                    aR[0] = pk_bytes(pk_lut, pk_right<4>(wt)); aR[1] = pk_bytes(pk_lut, pk_right<5>(wt)); bR[0] = pk_bytes(pk_lut, pk_right<4>(wq)); bR[1] = pk_bytes(pk_lut, pk_right<5>(wq));
                    from_packed = true;
                }
            }
        }
        if (!from_packed) {
        // (cl0 and cr0 are even: level 1 took an even number of chunks)
#pragma unroll
        for (int j = 0; j < kL2 / 2; j++) { load_left2(tc + t_end, cl0 / 2 + j, aL[2 * j], aL[2 * j + 1]); load_left2(qc + q_end, cl0 / 2 + j, bL[2 * j], bL[2 * j + 1]); }
#pragma unroll
        for (int j = 0; j < kR2 / 2; j++) { load_right2(tc + t_end, cr0 / 2 + j, aR[2 * j], aR[2 * j + 1]); load_right2(qc + q_end, cr0 / 2 + j, bR[2 * j], bR[2 * j + 1]); }
        }
        xl = XState{e.run_l, e.best_l, e.bpos_l, e.cl >= 0};
        xr = XState{e.run_r, e.best_r, e.bpos_r, e.cr >= 0};
        unsigned long long seps = 0;
#pragma unroll
        for (int c = 0; c < kL2; c++) seps |= aL[c] | bL[c];
#pragma unroll
        for (int c = 0; c < kR2; c++) seps |= aR[c] | bR[c];
        if ((seps & 0x8080808080808080ull) == 0ull) {
#pragma unroll
            for (int c = 0; c < kL2; c++) ux_chunk<-1>(aL[c], bL[c], cl0 + c, xdrop, xl, e.cols);
#pragma unroll
            for (int c = 0; c < kR2; c++) ux_chunk<+1>(aR[c], bR[c], cr0 + c, xdrop, xr, e.cols);
            e.cl = xl.live ? cl0 + kL2 : -1; e.cr = xr.live ? cr0 + kR2 : -1;
            e.run_l = xl.run; e.best_l = xl.best; e.bpos_l = xl.bpos; e.run_r = xr.run; e.best_r = xr.best; e.bpos_r = xr.bpos;
        }                                                             // (else: a separator ahead -- the entry goes on as it came)
        spill = xl.live | xr.live;
        if (!spill) finish_hit(e.i, dq, un.id, q_end, t_end, xl.best, xl.bpos, xr.best, xr.bpos, e.cols, K, keys, n_hits, sc, hsps, hsp_cap, ctr, ut);
    }
    // ---- still running: to k_ux_tail.  A returning atomic on one address costs ~7 ns whoever issues it, and nearly every block has
    //      a straggler or two: the first two waves of a block own 12 + 4 slots of the entry array (count stored, no atomic); only
    //      what does not fit there goes to the shared list (one atomic per wave).
    const unsigned long long sm = wballot(spill);
    const unsigned my_pos = (unsigned)__popcll(sm & ((1ull << lane) - 1ull)), n_sp = (unsigned)__popcll(sm);
    const unsigned own = wv == 0 ? kSlots0 : wv == 1 ? kSlots1 : 0u;
    if (wv < 2 && lane == 0) sc.blk_cnt[2 * blockIdx.x + wv] = min(n_sp, own);
    if (spill && my_pos < own) sc.blk_entries[(size_t)blockIdx.x * (kSlots0 + kSlots1) + (wv == 0 ? 0u : kSlots0) + my_pos] = e;
    if (n_sp > own) {
        unsigned base_slot = 0;
        if (lane == 0) base_slot = atomicAdd(sc.n_entries, n_sp - own);
        base_slot = (unsigned)wreadlane((int)base_slot, 0);
        if (spill && my_pos >= own) {
            const unsigned at = base_slot + (my_pos - own);
            if (at < sc.entry_cap) sc.entries[at] = e;
            else {
                // the list is full: this lane walks its hit to the end itself (slow and rare; same result)
                while (xl.live) { xdrop_chunk<-1>(load8(tc + t_end - 8 * (e.cl + 1)), load8(qc + q_end - 8 * (e.cl + 1)), e.cl, xdrop, xl, e.cols); e.cl++; }
                while (xr.live) { xdrop_chunk<+1>(load8(tc + t_end + 8 * e.cr), load8(qc + q_end + 8 * e.cr), e.cr, xdrop, xr, e.cols); e.cr++; }
                finish_hit(e.i, dq, un.id, q_end, t_end, xl.best, xl.bpos, xr.best, xr.bpos, e.cols, K, keys, n_hits, sc, hsps, hsp_cap, ctr, ut);
            }
        }
    }
}

__global__ __launch_bounds__(ux::kBlock) void k_ux_extend(const unsigned long long *__restrict__ keys, const int64_t n_hits, const UnitTab ut,
                                                           const int xdrop, const int K, const UxScratch sc, DevHsp *__restrict__ hsps,
                                                           const int64_t hsp_cap, UngappedCounters *__restrict__ ctr) {
    ux_extend_body<false>(keys, n_hits, ut, xdrop, K, sc, hsps, hsp_cap, ctr);
}
// level 1 from the packed strands of the launch's ONE unit (UxScratch::t_px / q_px); everything else as k_ux_extend
__global__ __launch_bounds__(ux::kBlock) void k_ux_extend_pk(const unsigned long long *__restrict__ keys, const int64_t n_hits, const UnitTab ut,
                                                              const int xdrop, const int K, const UxScratch sc, DevHsp *__restrict__ hsps,
                                                              const int64_t hsp_cap, UngappedCounters *__restrict__ ctr) {
    ux_extend_body<true>(keys, n_hits, ut, xdrop, K, sc, hsps, hsp_cap, ctr);
}

// The listed hits to the end: a group of 8 lanes per hit, 64 columns per step (the state machine of k_ungapped_grp with the
// list in place of the runs).
__global__ __launch_bounds__(256) void k_ux_tail(const unsigned long long *__restrict__ keys, const int64_t n_hits, const UnitTab ut,
                                                  const int xdrop, const int K,
                                                  const UxScratch sc, DevHsp *__restrict__ hsps, const int64_t hsp_cap,
                                                  UngappedCounters *__restrict__ ctr) {
    using namespace ugrp;
    const int l8 = threadIdx.x & 7;
    // work: the shared list (one entry each) and then the slot regions of the blocks of k_ux_extend (blk_cnt entries each)
    const unsigned n_list = min(*sc.n_entries, sc.entry_cap), n_regions = n_list + 2 * sc.n_blk;
    const unsigned G = gridDim.x * (blockDim.x >> 3);
    unsigned at = blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3);
    const UxEntry *src = nullptr;                                     // entries of the current region still to do
    unsigned left = 0;
    int phase = 0;                                                    // 0 fetch, 1 left, 2 right, 3 nothing left
    UxEntry e;
    e.i = 0; e.cl = -1; e.cr = -1; e.run_l = 0; e.best_l = 0; e.bpos_l = 0; e.run_r = 0; e.best_r = 0; e.bpos_r = 0; e.cols = 0;
    uint32_t dq = 0; int32_t q_end = 0; int64_t t_end = 0;
    int base = 0, runb = 0, best = 0, bpos = 0;
    UnitRef un = unit_by_id(ut, 0);
    const uint8_t *tc = un.tc, *qc = un.qc;
    while (true) {
        if (phase == 0) {
            while (left == 0 && at < n_regions) {                       // (most block regions are empty)
                if (at < n_list) { src = sc.entries + at; left = 1; }
                else {
                    const unsigned r = at - n_list;
                    left = sc.blk_cnt[r];
                    src = sc.blk_entries + (size_t)(r >> 1) * (ux::kSlots0 + ux::kSlots1) + ((r & 1u) ? ux::kSlots0 : 0u);
                }
                at += G;
            }
            if (left) {
                e = *src;
                src++; left--;
                const unsigned long long key = keys[e.i];
                dq = (uint32_t)(key >> 32);
                q_end = (int32_t)(uint32_t)key;
                un = unit_of(ut, dq); tc = un.tc; qc = un.qc;
                t_end = (int64_t)dq - un.qoff + q_end;
                if (e.cl >= 0) { phase = 1; base = 8 * e.cl; runb = e.run_l; best = e.best_l; bpos = e.bpos_l; }
                else { phase = 2; base = 8 * e.cr; runb = e.run_r; best = e.best_r; bpos = e.bpos_r; }
            } else {
                phase = 3;
            }
        }
        const bool act = phase == 1 || phase == 2;
        if (!wballot(act)) break;                                       // (a group without work is in phase 3: nothing comes back)
        const StepOut so = grp_step(act, phase == 1, tc + t_end, qc + q_end, base, best - runb, xdrop, l8);
        if (act) {
            e.cols += (uint32_t)so.lim;
            if (so.improved) { best = runb + so.gmax; bpos = base + so.col + 1; }
            if (!so.stopped) { base += 64; runb += so.gtot; }
            else if (phase == 1) {
                e.best_l = best; e.bpos_l = bpos;
                if (e.cr >= 0) { phase = 2; base = 8 * e.cr; runb = e.run_r; best = e.best_r; bpos = e.bpos_r; }
                else phase = 4;
            } else {
                e.best_r = best; e.bpos_r = bpos;
                phase = 4;
            }
            if (phase == 4) {
                if (l8 == 0) ux::finish_hit(e.i, dq, un.id, q_end, t_end, e.best_l, e.bpos_l, e.best_r, e.bpos_r, e.cols, K, keys, n_hits, sc, hsps, hsp_cap, ctr, ut);
                phase = 0;
            }
        }
    }
}

constexpr int kDirtyBuf = 2048;

// Every hit of the clean short runs is kept: one hit per lane.  The first hit of a dirty short run puts the run on the list of
// k_ux_resolve.
__global__ __launch_bounds__(256) void k_ux_accept(const unsigned long long *__restrict__ keys, const int64_t n_hits, const UnitTab ut, int32_t *__restrict__ extent,
                                                    const UxScratch sc, DevHsp *__restrict__ hsps, UngappedCounters *__restrict__ ctr) {
    // (every block takes one stretch of the sorted hits with the counters in registers: a pair of same-address atomics per wave of
    //  hits costs more than the rest.  The hits of a unit are one stretch of the sorted keys, so a block sees a unit or two: a
    //  thread's numbers go to their unit when the unit changes, and once per block at the end.)
    __shared__ unsigned dbuf[kDirtyBuf];                             // first hits of dirty runs found by this block, appended to the list at the end
    __shared__ unsigned n_dbuf, dbase, sh_unit1;
    __shared__ unsigned long long sh_kept, sh_cols;
    if (threadIdx.x == 0) { n_dbuf = 0; sh_kept = 0; sh_cols = 0; sh_unit1 = 0; }
    __syncthreads();
    unsigned n_kept = 0;
    unsigned long long n_cols = 0;
    int cur = 0;                                                     // unit of n_kept / n_cols
    // (one unit: the grid strides over the hits, so that the list of dirty runs comes out interleaved -- k_ux_resolve's lanes then
    //  walk runs from all over the diagonal space; several units: a stretch per block)
    const bool strided = ut.n <= 1;
    const int64_t per_blk = (((n_hits + (int64_t)gridDim.x - 1) / (int64_t)gridDim.x) + (int64_t)blockDim.x - 1) / (int64_t)blockDim.x * (int64_t)blockDim.x;
    const int64_t i_end = strided ? n_hits : min(n_hits, ((int64_t)blockIdx.x + 1) * per_blk);
    const int64_t i_step = strided ? (int64_t)gridDim.x * blockDim.x : (int64_t)blockDim.x;
    for (int64_t i = (strided ? (int64_t)blockIdx.x * blockDim.x : (int64_t)blockIdx.x * per_blk) + threadIdx.x; i < i_end; i += i_step) {
        // (everything that does not depend on the bit planes is requested first: one round trip + one dependent one per hit)
        const unsigned long long key = keys[i];
        const unsigned long long rc = sc.rec[i];
        const unsigned long long nk = i + 1 < n_hits ? keys[i + 1] : ~0ull;
        const unsigned long long pk = i > 0 ? keys[i - 1] : ~0ull;
        const uint32_t dq = (uint32_t)(key >> 32);
        const uint32_t pb = ux::plane_bit(sc, dq);
        const bool is_long = (sc.long_bits[pb >> 5] >> (pb & 31u)) & 1u, is_dirty = (sc.dirty_bits[pb >> 5] >> (pb & 31u)) & 1u;
        if (!is_long && !is_dirty) {
            if (ut.n > 1) {
                const int u = unit_index(ut, dq);
                if (u != cur) {
                    if (n_kept | n_cols) { atomicAdd(&ctr[cur].extended, (unsigned long long)n_kept); atomicAdd(&ctr[cur].cols, n_cols); }
                    n_kept = 0; n_cols = 0; cur = u;
                }
            }
            const uint32_t x = (uint32_t)(rc >> 32);
            uint32_t cols = x;
            if (x & ux::kCand) {
                DevHsp *hs = hsps + (x & ~ux::kCand);
                cols = (uint32_t)hs->cnt[0];
                hs->cnt[1] = 1;                                        // kept by the rule
            }
            n_kept++; n_cols += cols;
            // the last hit of the run leaves the diagonal's extent
            if ((uint32_t)(nk >> 32) != dq) extent_put(extent, ut, dq, (int32_t)(uint32_t)key + (int32_t)(uint32_t)rc);
        } else if (is_dirty && !is_long && (uint32_t)(pk >> 32) != dq) {
            const unsigned slot = atomicAdd(&n_dbuf, 1u);                // (LDS)
            if (slot < kDirtyBuf) dbuf[slot] = (unsigned)i;
            else {
                const unsigned at = atomicAdd(sc.n_entries + 1, 1u);
                if (at < sc.dirty_cap) sc.dirty_runs[at] = (unsigned)i;   // (the list holds 1.25 entries per hit: it cannot overflow)
            }
        }
    }
    if (n_kept | n_cols) atomicMax(&sh_unit1, (unsigned)cur + 1u);       // (LDS) the block's last unit takes the block-wide sum
    __syncthreads();
    const unsigned n_mine = min(n_dbuf, (unsigned)kDirtyBuf);
    if (threadIdx.x == 0 && n_mine) dbase = atomicAdd(sc.n_entries + 1, n_mine);
    const int blk_unit = (int)sh_unit1 - 1;
    if ((n_kept | n_cols) && cur != blk_unit) {                          // (a thread whose last hits belong to an earlier unit: few)
        atomicAdd(&ctr[cur].extended, (unsigned long long)n_kept); atomicAdd(&ctr[cur].cols, n_cols);
        n_kept = 0; n_cols = 0;
    }
    __syncthreads();
    for (unsigned j = threadIdx.x; j < n_mine; j += blockDim.x)
        if (dbase + j < sc.dirty_cap) sc.dirty_runs[dbase + j] = dbuf[j];
    // one pair of atomics per block: a pair of same-address atomics per wave would cost more than the rest of the kernel
    // (wave sums first: 24-bit slices of the 64-bit column count, so that 64 of them cannot overflow 32 bits)
    const int s_k = ugrp::grp_sum((int)n_kept), s0 = ugrp::grp_sum((int)(n_cols & 0xFFFFFFu)), s1 = ugrp::grp_sum((int)((n_cols >> 24) & 0xFFFFFFu)),
              s2 = ugrp::grp_sum((int)(n_cols >> 48));
    unsigned long long kept = 0, cols = 0;
#pragma unroll
    for (int g = 0; g < 8; g++) {
        kept += (unsigned)wreadlane(s_k, 8 * g);
        cols += (unsigned long long)(unsigned)wreadlane(s0, 8 * g) + ((unsigned long long)(unsigned)wreadlane(s1, 8 * g) << 24) +
                ((unsigned long long)(unsigned)wreadlane(s2, 8 * g) << 48);
    }
    if ((threadIdx.x & 63) == 0 && kept) { atomicAdd(&sh_kept, kept); atomicAdd(&sh_cols, cols); }
    __syncthreads();
    if (threadIdx.x == 0 && sh_kept) { atomicAdd(&ctr[blk_unit].extended, sh_kept); atomicAdd(&ctr[blk_unit].cols, sh_cols); }
}

// The sequential rule over the records of the dirty short runs, one run per lane.
__global__ __launch_bounds__(256) void k_ux_resolve(const unsigned long long *__restrict__ keys, const int64_t n_hits, const UnitTab ut, int32_t *__restrict__ extent,
                                                     const UxScratch sc, DevHsp *__restrict__ hsps, UngappedCounters *__restrict__ ctr) {
    const unsigned long long *__restrict__ rec = sc.rec;
    const unsigned total = min(sc.n_entries[1], sc.dirty_cap);
    unsigned long long n_ext = 0, n_cols = 0;
    int cur = 0;                                                     // unit of n_ext / n_cols
    for (unsigned r = blockIdx.x * blockDim.x + threadIdx.x; r < total; r += gridDim.x * blockDim.x) {
        int64_t k = sc.dirty_runs[r];
        unsigned long long key = keys[k];
        const uint32_t dq = (uint32_t)(key >> 32);
        if (ut.n > 1) {
            const int u = unit_index(ut, dq);
            if (u != cur) {
                if (n_ext | n_cols) { atomicAdd(&ctr[cur].extended, n_ext); atomicAdd(&ctr[cur].cols, n_cols); }
                n_ext = 0; n_cols = 0; cur = u;
            }
        }
        int32_t ext = extent_get(extent, ut, dq);
        while (true) {
            const int32_t q_end = (int32_t)(uint32_t)key;
            if (q_end > ext) {
                const unsigned long long rc = rec[k];
                const uint32_t x = (uint32_t)(rc >> 32);
                uint32_t cols = x;
                if (x & ux::kCand) {
                    DevHsp *hs = hsps + (x & ~ux::kCand);
                    cols = (uint32_t)hs->cnt[0];
                    hs->cnt[1] = 1;                                    // kept by the rule
                }
                ext = q_end + (int32_t)(uint32_t)rc;
                n_ext++;
                n_cols += cols;
            }
            k++;
            if (k >= n_hits) break;
            key = keys[k];
            if ((uint32_t)(key >> 32) != dq) break;
        }
        extent_put(extent, ut, dq, ext);
    }
    unit_count(ctr, cur, n_ext, n_cols);                             // (one pair of atomics per wave)
}

// Candidates the rule kept get their identical-base census (SURVEY A.5 entropy filter input); the others are marked.
__global__ __launch_bounds__(256) void k_ux_census(const UnitTab ut, DevHsp *__restrict__ hsps,
                                                    const int64_t hsp_cap, const UngappedCounters *__restrict__ ctr) {
    const unsigned long long n = min((unsigned long long)hsp_cap, ctr->hsps);
    for (unsigned long long s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (unsigned long long)gridDim.x * blockDim.x) {
        DevHsp hs = hsps[s];
        if (hs.cnt[1] != 1) { hsps[s].score = -2147483647 - 1; continue; }
        const UnitRef un = unit_by_id(ut, hs.unit);
        const uint8_t *tc = un.tc, *qc = un.qc;
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int kk = 0; kk < hs.len; kk += 8) {
            const unsigned long long a8 = load8(tc + hs.t_start + kk), b8 = load8(qc + hs.q_start + kk);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const unsigned a = (unsigned)(a8 >> (8 * m)) & 7u, b = (unsigned)(b8 >> (8 * m)) & 7u;
                const bool in = kk + m < hs.len;
                c0 += (in & (a == b) & (a == 0u)); c1 += (in & (a == b) & (a == 1u));
                c2 += (in & (a == b) & (a == 2u)); c3 += (in & (a == b) & (a == 3u));
            }
        }
        hs.cnt[0] = c0; hs.cnt[1] = c1; hs.cnt[2] = c2; hs.cnt[3] = c3;
        hsps[s] = hs;
    }
}

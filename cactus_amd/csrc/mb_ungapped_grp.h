// mb_ungapped_grp.h -- k_ungapped_grp: ungapped x-drop extension with EIGHT LANES PER DIAGONAL RUN (gfx950, wave64).
// Included by mb_kernels.hip inside namespace mb (and, with MB_EMU defined, by the host-side emulation test under tests/emu).
//
// Rule restated (SURVEY A.4 / A.5 / A.10 UNGAPPED and SEARCH): the hits of a diagonal are
// taken in q order; a hit with q_end <= extent[d] is skipped; otherwise the seed end is extended to the left and to the right,
//     run += score; if (run > best) { best = run; pos = k } else if (run < best - xdrop) stop      (the stop column is counted)
// a contig separator ends a direction before its column; extent[d] = q_end + pos_right; score = best_left + best_right.
//
// The lane-per-run kernel (k_ungapped) walks that loop one column per ~10 instructions in every lane, every lane gathering
// its own cache lines, and a wave lasts as long as its longest lane: 183 wave-instruction slots per hit on the 8 Mb random
// pair.  Here a run belongs to a GROUP of 8 adjacent lanes (half a DPP row), and a step evaluates 64
// columns of one direction at once, lane l the columns 8 l .. 8 l + 7 of the step:
//   * one 8-byte load per lane and sequence: the group reads 64 contiguous bytes of T and of Q;
//   * 8 signed substitution scores from two v_perm_b32, their prefix sums from eight v_dot4c_i32_i8 (mask 1,1,1,1 ...);
//   * the running score across lanes = an 8-lane exclusive sum scan, "best before this column" = an 8-lane exclusive max
//     scan (DPP row_shr / quad_perm inside the half row), the first stopping column = an 8-lane min (quad_perm butterflies
//     + row_half_mirror);
//   * everything a sequential walk would not have looked at lies behind the first stop and is ignored, so the result
//     (best, position, columns counted) is the sequential one, bit for bit.
// The wave runs a small state machine (fetch the next hit of the run / one step of a direction / finish the hit); groups
// take runs in a strided order from the class lists of k_run_heads, longest class first.
#pragma once


namespace ugrp {

constexpr int kNegG = -(1 << 29);             // below every step-relative score
constexpr int kNone = 255;                    // "no stop in this step"

// DPP controls used on the 8-lane groups
constexpr int kShr1 = 0x111, kShr2 = 0x112, kShr4 = 0x114;       // row_shr:n
constexpr int kQ1032 = 0xB1, kQ2301 = 0x4E;                       // quad_perm [1,0,3,2] / [2,3,0,1]
constexpr int kQ0022 = 0xA0, kQ0111 = 0x54, kQ3333 = 0xFF;       // quad_perm [0,0,2,2] / [0,1,1,1] / [3,3,3,3]
constexpr int kHalfMirror = 0x141;                                // row_half_mirror: lane i <-> 7 - i inside 8 lanes

// all-reduce over the 8 lanes of a group
// (the `old` operand is the identity of the operation: the DPP move then folds into the v_min / v_max itself)
__device__ __forceinline__ int grp_min(int v) {
    constexpr int kId = 2147483647;
    v = min(v, wdpp<kQ1032, 0xf>(kId, v));
    v = min(v, wdpp<kQ2301, 0xf>(kId, v));
    return min(v, wdpp<kHalfMirror, 0xf>(kId, v));
}
__device__ __forceinline__ int grp_max(int v) {
    constexpr int kId = -2147483647 - 1;
    v = max(v, wdpp<kQ1032, 0xf>(kId, v));
    v = max(v, wdpp<kQ2301, 0xf>(kId, v));
    return max(v, wdpp<kHalfMirror, 0xf>(kId, v));
}
__device__ __forceinline__ int grp_sum(int v) {
    v += wdpp<kQ1032, 0xf>(0, v);
    v += wdpp<kQ2301, 0xf>(0, v);
    return v + wdpp<kHalfMirror, 0xf>(0, v);
}

// four signed substitution scores: byte m = score(a_m, b_m) as int8.  HOXD70 is a function of (a ^ b) and of whether a is
// C/G: one v_perm_b32 over an 8-byte table.  N (code bit 2) scores -100 against anything: scores4_fix_n, applied only by the
// waves that hold an N (or a separator) somewhere in the step.
__device__ __forceinline__ uint32_t scores4s(const uint32_t a4, const uint32_t b4) {
    const uint32_t d = (a4 ^ b4) & 0x03030303u;                       // 0 match, 2 transition, 1 / 3 the two transversion classes
    const uint32_t cg = ((a4 ^ (a4 >> 1)) & 0x01010101u) << 2;        // 4 where a is C or G
    constexpr uint32_t kAT = 0x85E18E5Bu;                             // a in {A,T}:  91, -114, -31, -123
    constexpr uint32_t kCG = 0x83E18E64u;                             // a in {C,G}: 100, -114, -31, -125
    return wperm(kCG, kAT, d | cg);                                   // selector 0..3 -> kAT, 4..7 -> kCG
}
__device__ __forceinline__ uint32_t scores4_fix_n(const uint32_t s, const uint32_t ab) {     // ab = a4 | b4
    const uint32_t t = (ab >> 2) & 0x01010101u;
    const uint32_t nm = (t << 8) - t;                                 // 0xFF where either base is N
    return (s & ~nm) | (0x9c9c9c9cu & nm);                            // -100
}

// first column of a 4-column word whose byte has bit 7 (a separator), 4 if none
__device__ __forceinline__ int first_sep4(const uint32_t w) {
    const uint32_t m = w & 0x80808080u;
    return m ? (__ffs((int)m) - 1) >> 3 : 4;
}

// One step of 64 columns of one direction for every group of the wave (all 64 lanes execute it; `act` says whether the lane's
// group takes part).  tp / qp: the seed end in T and Q; base: columns of this direction already walked; bestrel: best so far minus
// the run value at the step's origin.  Lane l8 takes columns 8 l8 .. 8 l8 + 7 of the step.
struct StepOut {
    bool stopped, improved;                   // the direction ends inside this step / a new best was found
    int lim;                                  // columns looked at (counted)
    int gmax, col;                            // improved: new best (relative to the step's origin) and its first column in the step
    int gtot;                                 // !stopped: sum of the 64 scores
};
__device__ __forceinline__ StepOut grp_step(const bool act, const bool left, const uint8_t *__restrict__ tp, const uint8_t *__restrict__ qp,
                                            const int base, const int bestrel, const int xdrop, const int l8) {
    const bool ge1 = l8 >= 1;
    const int m_ge1 = l8 >= 1 ? -1 : 0, m_ge2 = l8 >= 2 ? -1 : 0;
    StepOut o_;
    unsigned long long a8 = ~0ull, b8 = ~0ull;                    // (an idle group sees a separator in column 0)
    if (act) {
        const int64_t o = left ? -(int64_t)(base + 8 * (l8 + 1)) : (int64_t)(base + 8 * l8);
        a8 = load8(tp + o);
        b8 = load8(qp + o);
    }
    // column m of the lane = byte m: to the left the bytes come reversed
    const uint32_t sel_lo = left ? 0x04050607u : 0x03020100u, sel_hi = left ? 0x00010203u : 0x07060504u;
    const uint32_t a_lo = wperm((uint32_t)(a8 >> 32), (uint32_t)a8, sel_lo), a_hi = wperm((uint32_t)(a8 >> 32), (uint32_t)a8, sel_hi);
    const uint32_t b_lo = wperm((uint32_t)(b8 >> 32), (uint32_t)b8, sel_lo), b_hi = wperm((uint32_t)(b8 >> 32), (uint32_t)b8, sel_hi);
    uint32_t s_lo = scores4s(a_lo, b_lo), s_hi = scores4s(a_hi, b_hi);
    int fsep = 8;                                                 // first separator column of the lane, 8 if none
    const uint32_t ab_lo = a_lo | b_lo, ab_hi = a_hi | b_hi;
    if (wballot(((ab_lo | ab_hi) & 0x84848484u) != 0u)) {         // an N or a separator somewhere in the wave's 8 x 64 columns: rare
        s_lo = scores4_fix_n(s_lo, ab_lo); s_hi = scores4_fix_n(s_hi, ab_hi);
        const int fs_lo = first_sep4(ab_lo), fs_hi = first_sep4(ab_hi);
        fsep = fs_lo < 4 ? fs_lo : 4 + fs_hi;
    }
    // prefix sums of the lane's scores (relative to the lane's first column), and the same + xdrop
    int p[8], q[8];
    p[0] = wsdot4(s_lo, 0x00000001u, 0); p[1] = wsdot4(s_lo, 0x00000101u, 0);
    p[2] = wsdot4(s_lo, 0x00010101u, 0); p[3] = wsdot4(s_lo, 0x01010101u, 0);
    p[4] = wsdot4(s_hi, 0x00000001u, p[3]); p[5] = wsdot4(s_hi, 0x00000101u, p[3]);
    p[6] = wsdot4(s_hi, 0x00010101u, p[3]); p[7] = wsdot4(s_hi, 0x01010101u, p[3]);
#pragma unroll
    for (int m = 0; m < 8; m++) q[m] = p[m] + xdrop;
    int lm[8];                                                    // running maximum inside the lane
    lm[0] = p[0];
#pragma unroll
    for (int m = 1; m < 8; m++) lm[m] = max(lm[m - 1], p[m]);
    // exclusive sum of the lane totals over the group: the run value (relative to the step's origin) before the lane's first column
    const int tot = p[7];
    int inc = tot;
    inc += wdpp<kShr1, 0xf>(0, inc) & m_ge1;
    inc += wdpp<kShr2, 0xf>(0, inc) & m_ge2;
    inc += wdpp<kShr4, 0xa>(0, inc);                              // (banks 1 and 3 = lanes 4..7 of either group)
    const int L = inc - tot;
    // exclusive maximum over the lanes before this one of (run value at the lane's best column)
    constexpr int kIdMax = -2147483647 - 1;                        // (identity as `old`: the DPP moves fold into the v_max)
    int mx = L + lm[7];
    mx = max(mx, wdpp<kQ0022, 0xf>(kIdMax, mx));
    mx = max(mx, wdpp<kQ0111, 0xf>(kIdMax, mx));
    { const int t = wdpp<kQ3333, 0xf>(kIdMax, mx); mx = max(mx, wdpp<kShr4, 0xa>(kIdMax, t)); }
    int ex = wdpp<kShr1, 0xf>(kIdMax, mx);
    ex = ge1 ? ex : kIdMax;
        const int gl = max(bestrel, ex) - L;                          // best before the lane's first column, relative to the lane
    // x-drop test of the lane's columns: stop at m iff p[m] + xdrop < (best before m)
    unsigned sm = 0;
#pragma unroll
    for (int m = 7; m >= 0; m--) {
        const int thr = m == 0 ? gl : max(gl, lm[m - 1]);
        const int d = q[m] - thr;
        sm = wsignin(sm, d);
    }
    const int fx = sm ? __ffs((int)sm) - 1 : 8;
    // candidate = 2 * column for a separator (the column is not looked at), 2 * column + 1 for an x-drop stop (it is)
    const int c_sep = fsep < 8 ? 16 * l8 + 2 * fsep : kNone, c_x = fx < 8 ? 16 * l8 + 2 * fx + 1 : kNone;
    const int gc = grp_min(min(c_sep, c_x));
    const bool stopped = gc != kNone;
    const int lim = stopped ? (gc + 1) >> 1 : 64;                 // columns looked at in this step
    const int cb = stopped ? gc >> 1 : 64;                        // columns that may hold a new best (the stop column never does)
    // the lane's best among its eligible columns: lm[e], e = cb - 8 l8 - 1 clamped
    const int e = cb - 8 * l8 - 1;
    const int e7 = min(e, 7);
    const int b0 = -(e7 & 1), b1 = -((e7 >> 1) & 1), b2 = -((e7 >> 2) & 1);
    auto sel = [](int mask, int one, int zero) { return (one & mask) | (zero & ~mask); };
    const int lmE = sel(b2, sel(b1, sel(b0, lm[7], lm[6]), sel(b0, lm[5], lm[4])), sel(b1, sel(b0, lm[3], lm[2]), sel(b0, lm[1], lm[0])));
    const int mE = e < 0 ? kNegG : L + lmE;
    const int gmax = grp_max(mE);
    o_.improved = act && gmax > bestrel;
    // first column that attains it: lm is non-decreasing, so inside the lane it is the number of columns with lm < target
    const int tgt = gmax - L;
    const int m1 = (lm[3] - tgt) >> 31;
    const int x1 = sel(m1, lm[5], lm[1]);
    const int m2 = (x1 - tgt) >> 31;
    const int y1 = sel(m1, sel(m2, lm[6], lm[4]), sel(m2, lm[2], lm[0]));
    const int m3 = (y1 - tgt) >> 31;
    const int idx = (m1 & 4) | (m2 & 2) | (m3 & 1);
    o_.col = grp_min(mE == gmax ? 8 * l8 + idx : kNone);
    o_.gtot = grp_sum(tot);
    o_.stopped = stopped; o_.lim = lim; o_.gmax = gmax;
    return o_;
}

}  // namespace ugrp

template <int kMinWaves>
__global__ __launch_bounds__(256, kMinWaves) void k_ungapped_grp(const unsigned long long *__restrict__ keys, int64_t n_hits,
                                                      const unsigned *__restrict__ heads, const unsigned *__restrict__ n_heads_p,
                                                      const UnitTab ut,
                                                      int32_t *__restrict__ extent, int xdrop, int K, DevHsp *__restrict__ hsps,
                                                      int64_t hsp_cap, UngappedCounters *__restrict__ ctr) {
    using namespace ugrp;
    const int lane = threadIdx.x & 63;
    const int l8 = lane & 7;
    // the short-run lists of k_run_heads, longest class first: run r of the concatenation
    const unsigned n3 = n_heads_p[3], n2 = n_heads_p[2], n1 = n_heads_p[1], n0 = n_heads_p[0];
    const unsigned r32 = n3 + n2, r321 = r32 + n1, total = r321 + n0;
    const uint64_t nh = (uint64_t)n_hits;
    const uint64_t off1 = nh, off2 = nh + nh / 2, off3 = nh + nh / 2 + nh / 4;
    const unsigned G = gridDim.x * (blockDim.x >> 3);                  // groups in the grid
    unsigned r = blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3); // this group's next run

    // ---- group state (identical in the 8 lanes of a group)
    int phase = 0;                                                    // 0 fetch, 1 left, 2 right, 3 no runs left
    bool need_run = true;
    unsigned long long cur = 0, nxt = 0;                              // key of the current hit, of the next one (prefetched)
    int64_t k = 0;
    uint32_t dq = 0;
    int32_t ext = 0, q_end = 0;
    int64_t t_end = 0;
    int base = 0, runb = 0, best = 0, bpos = 0, bestL = 0, bl = 0;    // step origin (columns), run value there, best so far and its length
    bool emit = false;
    int em_len = 0, em_score = 0, em_qend = 0;
    int64_t em_t = 0, em_q = 0, em_tend = 0;
    unsigned long long n_ext = 0, n_cols = 0;
    // the unit of the current run (a group walks runs of several units of a batched launch: its counters go to the unit they belong
    // to whenever the unit changes) and of the HSP waiting to be written
    UnitRef un = unit_by_id(ut, 0);
    const uint8_t *tc = un.tc, *qc = un.qc, *em_tc = un.tc, *em_qc = un.qc;
    int em_unit = 0;

    auto advance = [&]() {                                            // to the next hit of the run, or the run is over
        if ((uint32_t)(nxt >> 32) == dq) {
            k++; cur = nxt;
            nxt = k + 1 < n_hits ? keys[k + 1] : ~0ull;
        } else {
            if (l8 == 0) extent_put(extent, ut, dq, ext);
            need_run = true;
        }
        phase = 0;
    };

    while (true) {
        // ---- fetch: the next hit that is not inside an extended stretch (loads and per-group state only: no cross-lane operation
        //      sits inside a divergent region)
        if (phase == 0) {
            if (need_run) {
                if (r < total) {
                    const uint64_t at = r < n3 ? off3 + r : r < r32 ? off2 + (r - n3) : r < r321 ? off1 + (r - r32) : (uint64_t)(r - r321);
                    k = heads[at];
                    r += G;
                    cur = keys[k];
                    nxt = k + 1 < n_hits ? keys[k + 1] : ~0ull;
                    dq = (uint32_t)(cur >> 32);
                    ext = extent_get(extent, ut, dq);
                    need_run = false;
                    if (ut.n > 1) {
                        const UnitRef nu = unit_of(ut, dq);
                        if (nu.id != un.id) {
                            if (l8 == 0 && (n_ext | n_cols)) { atomicAdd(&ctr[un.id].extended, n_ext); atomicAdd(&ctr[un.id].cols, n_cols); }
                            n_ext = 0; n_cols = 0;
                            un = nu; tc = un.tc; qc = un.qc;
                        }
                    }
                } else {
                    phase = 3;
                }
            }
            if (phase == 0) {
                q_end = (int32_t)(uint32_t)cur;
                if (q_end > ext) {
                    t_end = (int64_t)dq - un.qoff + q_end;
                    phase = 1; base = 0; runb = 0; best = 0; bpos = 0;
                } else {
                    advance();
                }
            }
        }
        const bool act = phase == 1 || phase == 2;
        if (!wballot(act)) {
            if (!wballot(phase == 0)) break;                           // every group of the wave is out of runs
            continue;
        }
        // ---- one step of 64 columns for every active group
        const StepOut so = grp_step(act, phase == 1, tc + t_end, qc + q_end, base, best - runb, xdrop, l8);
        const bool stopped = so.stopped, improved = so.improved, left = phase == 1;
        const int lim = so.lim, gmax = so.gmax, col = so.col, gtot = so.gtot;
        // ---- group state after the step
        if (act) n_cols += (unsigned long long)lim;
        if (improved) { best = runb + gmax; bpos = base + col + 1; }
        bool fin = false;
        if (act) {
            if (!stopped) { base += 64; runb += gtot; }
            else if (left) { bestL = best; bl = bpos; phase = 2; base = 0; runb = 0; best = 0; bpos = 0; }
            else fin = true;
        }
        if (fin) {                                                    // both directions done
            const int br = bpos, score = bestL + best;
            n_ext++;
            ext = q_end + br;
            if (score >= K) {
                emit = true;
                em_t = t_end - bl; em_q = (int64_t)q_end - bl; em_len = bl + br; em_score = score; em_tend = t_end; em_qend = q_end;
                em_tc = tc; em_qc = qc; em_unit = un.id;
            }
            advance();
        }
        // ---- HSPs: identical-base census by the group (64 columns per turn), record written by its first lane
        if (wballot(emit)) {
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            for (int kk = 8 * l8; wballot(emit && kk < em_len); kk += 64) {
                if (emit && kk < em_len) {
                    const unsigned long long x8 = load8(em_tc + em_t + kk), y8 = load8(em_qc + em_q + kk);
#pragma unroll
                    for (int m = 0; m < 8; m++) {
                        const unsigned a = (unsigned)(x8 >> (8 * m)) & 7u, b = (unsigned)(y8 >> (8 * m)) & 7u;
                        const bool in = kk + m < em_len;
                        c0 += (in & (a == b) & (a == 0u)); c1 += (in & (a == b) & (a == 1u));
                        c2 += (in & (a == b) & (a == 2u)); c3 += (in & (a == b) & (a == 3u));
                    }
                }
            }
            c0 = grp_sum(c0); c1 = grp_sum(c1); c2 = grp_sum(c2); c3 = grp_sum(c3);
            if (emit && l8 == 0) {
                const unsigned long long slot = atomicAdd(&ctr->hsps, 1ull);
                if ((int64_t)slot < hsp_cap) {
                    DevHsp hs;
                    hs.t_start = (int32_t)em_t; hs.q_start = (int32_t)em_q; hs.len = em_len; hs.score = em_score;
                    hs.seed_t_end = (int32_t)em_tend; hs.seed_q_end = em_qend;
                    hs.cnt[0] = c0; hs.cnt[1] = c1; hs.cnt[2] = c2; hs.cnt[3] = c3;
                    hs.unit = em_unit;
                    hsps[slot] = hs;
                }
            }
            emit = false;
        }
    }
    // counters: the first lane of a group holds the group's numbers (a group lives for the whole launch: few atomics)
    if (l8 == 0 && (n_ext | n_cols)) { atomicAdd(&ctr[un.id].extended, n_ext); atomicAdd(&ctr[un.id].cols, n_cols); }
}

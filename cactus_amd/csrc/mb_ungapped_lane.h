// mb_ungapped_lane.h -- the run-per-lane ungapped extension kernel k_ungapped (sparse hit sets: the default for the many small pairs of a
// phase) and the wave-per-run kernel k_ungapped_long for busy diagonals (every kernel choice hands those to it) (gfx950, wave64).
// Included by mb_kernels.hip inside namespace mb after mb_runs.h (and, with MB_EMU defined, by the host-side emulation test under
// tests/emu, which supplies the wave primitives).
#pragma once

// One x-drop direction, 8 columns per load: see mb_xdrop.h.

__global__ __launch_bounds__(256) void k_ungapped(const unsigned long long *__restrict__ keys, int64_t n_hits,
                                                  const unsigned *__restrict__ heads, const unsigned *__restrict__ n_heads_p,
                                                  const UnitTab ut,
                                                  int32_t *__restrict__ extent, int xdrop, int K, DevHsp *__restrict__ hsps,
                                                  int64_t hsp_cap, UngappedCounters *__restrict__ ctr) {
    unsigned long long n_ext = 0, n_cols = 0;
    int my_unit = 0;
    // The blocks are dealt to the run-length classes, longest runs first (they start early, and a wave holds runs of similar
    // length instead of waiting for its longest lane): class c owns ceil(n_c / 256) blocks.
    unsigned n_heads = 0;
    uint64_t list_off = 0;
    {
        const uint64_t n = (uint64_t)n_hits;
        unsigned blk = blockIdx.x;
        bool found = false;
#pragma unroll
        for (int c = kRunClasses - 1; c >= 0; c--) {
            const unsigned nc = n_heads_p[c], nb = (nc + blockDim.x - 1) / blockDim.x;
            if (!found && blk < nb) { found = true; n_heads = nc; list_off = c == 0 ? 0 : c == 1 ? n : c == 2 ? n + n / 2 : n + n / 2 + n / 4; }
            if (!found) blk -= nb;
        }
        if (!found) return;
        heads += list_off;
        n_heads = min(n_heads, (blk + 1) * blockDim.x);
        n_heads = blk * blockDim.x + threadIdx.x < n_heads ? n_heads : 0;
        list_off = blk;                                                   // (reused: block index inside the class)
    }
    // one diagonal run per thread (a long HSP must not delay further runs queued behind it in the same lane)
    for (unsigned h = (unsigned)list_off * blockDim.x + threadIdx.x; h < n_heads; h = n_heads) {
        int64_t k = heads[h];
        unsigned long long key = keys[k];
        const uint32_t dq = (uint32_t)(key >> 32);
        const UnitRef un = unit_of(ut, dq);
        const uint8_t *tc = un.tc, *qc = un.qc;
        my_unit = un.id;
        int32_t ext = extent_get(extent, ut, dq);
        while (true) {
            const int32_t q_end = (int32_t)(uint32_t)key;
            if (q_end > ext) {
                const int64_t t_end = (int64_t)dq - un.qoff + q_end;
                // Separators (0xFF) bound every contig on both sides; device buffers carry kDevPad pad bytes, so the
                // 8-byte loads may overrun harmlessly.  Left covers the seed, then beyond; right starts at the seed end.
                int bestL, bl, bestR, br;
                constexpr int kPreL = 5, kPreR = 3;                      // 40 columns to the left (the seed is 19 of them), 24 to the right
                unsigned long long aL[kPreL], bL[kPreL], aR[kPreR], bR[kPreR];
                xdrop_preload<-1, kPreL>(tc + t_end, qc + q_end, aL, bL);
                xdrop_preload<+1, kPreR>(tc + t_end, qc + q_end, aR, bR);
                xdrop_dir<-1, kPreL>(tc + t_end, qc + q_end, xdrop, aL, bL, bestL, bl, n_cols);
                xdrop_dir<+1, kPreR>(tc + t_end, qc + q_end, xdrop, aR, bR, bestR, br, n_cols);
                n_ext++;
                ext = q_end + br;
                const int score = bestL + bestR;
                if (score >= K) {
                    const unsigned long long slot = atomicAdd(&ctr->hsps, 1ull);      // (the slot counter of the launch: unit 0's)
                    if ((int64_t)slot < hsp_cap) {
                        DevHsp hs;
                        hs.t_start = (int32_t)(t_end - bl);
                        hs.q_start = q_end - bl;
                        hs.len = bl + br;
                        hs.score = score;
                        hs.seed_t_end = (int32_t)t_end;
                        hs.seed_q_end = q_end;
                        hs.unit = un.id;
                        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                        for (int kk = 0; kk < hs.len; kk += 8) {          // identical-base census, 8 columns per load
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
                        hsps[slot] = hs;
                    }
                }
            }
            k++;
            if (k >= n_hits) break;
            key = keys[k];
            if ((uint32_t)(key >> 32) != dq) break;
        }
        extent_put(extent, ut, dq, ext);
    }
    unit_count(ctr, my_unit, n_ext, n_cols);
}

// ---- wave-per-run variant for busy diagonals --------------------------------------------------------------------

// One x-drop direction evaluated by a whole wave, 64 columns per step: running score = prefix sum, "best so far" =
// exclusive prefix max, the first lane where run < best - xdrop ends the extension (ballot).  Bit-identical to the
// sequential loop: the stopping column is examined (counted), never a new best, and nothing behind it is looked at.
template <int DIR>
__device__ __forceinline__ void xdrop_dir_wave(const uint8_t *__restrict__ tp, const uint8_t *__restrict__ qp, const int xdrop,
                                               const int lane, int &best_out, int &pos_out, unsigned long long &ncols) {
    constexpr int kMin = -2147483647 - 1;
    int run_base = 0, best = 0, bpos = 0;
    for (int base = 0;; base += 64) {
        const int k = base + lane;
        const unsigned a = DIR > 0 ? tp[k] : tp[-1 - k], b = DIR > 0 ? qp[k] : qp[-1 - k];
        const unsigned long long sepm = __ballot((a == kSep) | (b == kSep));
        const int nvalid = sepm ? (int)__ffsll((long long)sepm) - 1 : 64;
        const int sc = lane < nvalid ? sub_score(a, b) : 0;
        const int incl = dpp_scan_add(sc) + run_base;
        const int pmi = dpp_scan_max(lane < nvalid ? incl : kMin);
        const int pme = max(best, dpp_shr1(pmi, kMin));                 // best before this column
        const unsigned long long stopm = __ballot((lane < nvalid) & (incl <= pme) & (incl < pme - xdrop));
        const int first_stop = stopm ? (int)__ffsll((long long)stopm) - 1 : 64;
        const int lim = min(nvalid, first_stop + 1);                    // columns examined in this step
        ncols += (unsigned long long)lim;
        const int segbest = __builtin_amdgcn_readlane(pmi, 63 < lim - 1 ? 63 : (lim > 0 ? lim - 1 : 0));
        if (lim > 0 && segbest > best) {
            const unsigned long long w = __ballot((lane < lim) & (incl == segbest));
            bpos = base + (int)__ffsll((long long)w);                  // first column attaining it, 1-based length
            best = segbest;
        }
        if (stopm | sepm) break;                                        // x-drop or end of the contig inside this step
        run_base = __builtin_amdgcn_readlane(incl, 63);
    }
    best_out = best; pos_out = bpos;
}

__global__ __launch_bounds__(256) void k_ungapped_long(const unsigned long long *__restrict__ keys, int64_t n_hits,
                                                       const unsigned *__restrict__ heads, const unsigned *__restrict__ n_heads_p,
                                                       const UnitTab ut,
                                                       int32_t *__restrict__ extent, int xdrop, int K, DevHsp *__restrict__ hsps,
                                                       int64_t hsp_cap, UngappedCounters *__restrict__ ctr) {
    const int lane = threadIdx.x & 63;
    const unsigned n_heads = *n_heads_p;
    // a run per wave and turn; the waves of the grid stride over the list (the list is sized for the worst case, n_hits / (kLongRun + 1)
    // runs: a block per four POSSIBLE runs were 6 x 10^5 blocks on the 4 x 10^7 chance hits of an 8 Mb x 8 Mb strand, nearly all of them
    // empty -- 140 us of block dispatch)
    const unsigned n_waves = gridDim.x * (blockDim.x >> 6);
    for (unsigned h = uni((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6))); h < n_heads; h += n_waves) {
    unsigned long long n_ext = 0, n_cols = 0;
    int64_t k0 = heads[h];
    const uint32_t dq = (uint32_t)(keys[k0] >> 32);
    const UnitRef un = unit_of(ut, dq);                              // (one run per wave: uniform)
    const uint8_t *tc = un.tc, *qc = un.qc;
    int32_t ext = extent_get(extent, ut, dq);
    bool run_done = false;
    while (!run_done) {
        // 64 hits of the run at a time; all suppressed hits of the block are skipped with one ballot
        const int64_t k = k0 + lane;
        unsigned long long key = k < n_hits ? keys[k] : ~0ull;
        const bool mine = (uint32_t)(key >> 32) == dq;                   // still the same diagonal
        const unsigned long long inrun = __ballot(mine);
        const int n_in = inrun == ~0ull ? 64 : (int)__ffsll((long long)~inrun) - 1;
        int from = 0;
        while (true) {
            const unsigned long long todo = __ballot(mine & (lane >= from) & ((int32_t)(uint32_t)key > ext));
            if (!todo) break;
            const int l = (int)__ffsll((long long)todo) - 1;            // next hit that is not inside an extended stretch
            const int32_t q_end = (int32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, l);
            const int64_t t_end = (int64_t)dq - un.qoff + q_end;
            int bestL, bl, bestR, br;
            xdrop_dir_wave<-1>(tc + t_end, qc + q_end, xdrop, lane, bestL, bl, n_cols);
            xdrop_dir_wave<+1>(tc + t_end, qc + q_end, xdrop, lane, bestR, br, n_cols);
            n_ext++;
            ext = q_end + br;
            const int score = bestL + bestR;
            if (score >= K) {
                unsigned long long slot = 0;
                if (lane == 0) slot = atomicAdd(&ctr->hsps, 1ull);
                slot = uni64(slot);
                if ((int64_t)slot < hsp_cap) {
                    const int t_start = (int)(t_end - bl), q_start = q_end - bl, len = bl + br;
                    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                    for (int kk = lane; kk < len; kk += 64) {          // identical-base census, one column per lane
                        const unsigned a = tc[t_start + kk] & 7u, b = qc[q_start + kk] & 7u;
                        c0 += (a == b) & (a == 0u); c1 += (a == b) & (a == 1u); c2 += (a == b) & (a == 2u); c3 += (a == b) & (a == 3u);
                    }
                    for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_down(c0, o); c1 += __shfl_down(c1, o); c2 += __shfl_down(c2, o); c3 += __shfl_down(c3, o); }
                    if (lane == 0) {
                        DevHsp hs;
                        hs.t_start = t_start; hs.q_start = q_start; hs.len = len; hs.score = score;
                        hs.seed_t_end = (int32_t)t_end; hs.seed_q_end = q_end;
                        hs.cnt[0] = c0; hs.cnt[1] = c1; hs.cnt[2] = c2; hs.cnt[3] = c3;
                        hs.unit = un.id;
                        hsps[slot] = hs;
                    }
                }
            }
            from = l + 1;
        }
        if (n_in < 64) run_done = true;
        k0 += 64;
    }
    if (lane == 0) {
        extent_put(extent, ut, dq, ext);
        atomicAdd(&ctr[un.id].extended, n_ext);
        atomicAdd(&ctr[un.id].cols, n_cols);
    }
    }
}

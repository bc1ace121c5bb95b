// mb_hash16.h -- diagonal suppression through lastz's 16-bit diagonal hash (miblast_params.diag_hash16; SURVEY A.4 / A.9 #4: the
// state that tells whether a seed hit lies inside an already extended stretch is indexed by (t_end - q_end) & 0xFFFF, so diagonals
// 65536 apart share it and a hit may be dropped because of an extension on ANOTHER diagonal).  Included by mb_kernels.hip inside
// namespace mb after mb_ungapped_ux.h.
//
// The rule is sequential per hash class, in the order the search generates the hits: q_end ascending, then the word variant (exact,
// then the transition at care position 0, 1, ...), then the target position descending.  What makes it parallel is what makes the
// level-synchronous pipeline work: a walk depends on the sequences only, never on other hits.  So EVERY hit is extended first
// (k_ux_extend / k_ux_tail leave one record per hit: length to the right, columns, candidate HSP), the hits are brought into
// (unit, hash class, generation order) with two stable radix sorts, and one lane per class walks its hits applying the rule to the
// records.  A comparison mode: exactness first, speed second (busy diagonals are extended hit by hit).
#pragma once

// rank of the word variant that produced a hit: 0 exact, 1 + k transition at care position k
__device__ __forceinline__ int h16_variant_rank(const uint8_t *tc, const uint8_t *qc, const int64_t t_end, const int32_t q_end) {
    uint32_t wt, wq;
    (void)window_word(tc, t_end - kSeedSpan, wt);
    (void)window_word(qc, (int64_t)q_end - kSeedSpan, wq);
    const uint32_t dx = wt ^ wq;
    if (!dx) return 0;
    int r = 15;
#pragma unroll
    for (int k = 0; k < kSeedWeight; k++) if (dx == (2u << (2 * (kSeedWeight - 1 - k)))) r = 1 + k;
    return r;
}

// first sort: target position descending (the last criterion of the generation order)
__global__ __launch_bounds__(256) void k_h16_tkeys(const unsigned long long *__restrict__ keys, const int64_t n_hits, const UnitTab ut,
                                                    unsigned long long *__restrict__ k2, uint32_t *__restrict__ val) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hits) return;
    const unsigned long long key = keys[i];
    const uint32_t dq = (uint32_t)(key >> 32);
    const UnitRef un = unit_of(ut, dq);
    const int64_t t_end = (int64_t)dq - un.qoff + (int32_t)(uint32_t)key;
    k2[i] = (unsigned long long)(0x7FFFFFFFu - (uint32_t)t_end);
    val[i] = (uint32_t)i;
}

// second sort (stable): unit | hash class | q_end | variant rank, of the hit at every position of the first sort's order
constexpr int kH16ClassShift = 35;            // key bits: [0,4) rank, [4,35) q_end, [35,51) hash class, [51,59) unit
__global__ __launch_bounds__(256) void k_h16_ckeys(const unsigned long long *__restrict__ keys, const int64_t n_hits, const UnitTab ut,
                                                    const uint32_t *__restrict__ val, unsigned long long *__restrict__ k1) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_hits) return;
    const unsigned long long key = keys[val[p]];
    const uint32_t dq = (uint32_t)(key >> 32);
    const int32_t q_end = (int32_t)(uint32_t)key;
    const UnitRef un = unit_of(ut, dq);
    const int64_t t_end = (int64_t)dq - un.qoff + q_end;
    const unsigned long long h = (unsigned long long)((t_end - (int64_t)q_end) & 0xFFFF);
    const unsigned long long rank = (unsigned long long)h16_variant_rank(un.tc, un.qc, t_end, q_end);
    k1[p] = ((unsigned long long)(unsigned)un.id << 51) | (h << kH16ClassShift) | ((unsigned long long)(uint32_t)q_end << 4) | rank;
}

// the rule, one lane per (unit, hash class): a hit with q_end <= extent is dropped, else extent = end of its extension
__global__ __launch_bounds__(256) void k_h16_resolve(const unsigned long long *__restrict__ k1, const uint32_t *__restrict__ val, const int64_t n_hits,
                                                      const unsigned long long *__restrict__ rec, DevHsp *__restrict__ hsps,
                                                      UngappedCounters *__restrict__ ctr) {
    const int64_t p0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_ext = 0, n_cols = 0;
    int unit = 0;
    if (p0 < n_hits) {
        const unsigned long long cls = k1[p0] >> kH16ClassShift;
        if (p0 == 0 || (k1[p0 - 1] >> kH16ClassShift) != cls) {              // the first hit of a class walks the class
            unit = (int)(cls >> 16);
            int32_t ext = 0;
            for (int64_t p = p0; p < n_hits && (k1[p] >> kH16ClassShift) == cls; p++) {
                const int32_t q_end = (int32_t)((k1[p] >> 4) & 0x7FFFFFFFu);
                if (q_end <= ext) continue;
                const unsigned long long rc = rec[val[p]];
                const uint32_t x = (uint32_t)(rc >> 32);
                uint32_t cols = x;
                if (x & ux::kCand) {
                    DevHsp *hs = hsps + (x & ~ux::kCand);
                    cols = (uint32_t)hs->cnt[0];
                    hs->cnt[1] = 1;                                          // kept by the rule
                }
                ext = q_end + (int32_t)(uint32_t)rc;
                n_ext++; n_cols += cols;
            }
        }
    }
    unit_count(ctr, unit, n_ext, n_cols);
}

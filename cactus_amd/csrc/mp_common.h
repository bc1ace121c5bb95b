// mp_common.h -- device-side records and launch wrappers of the chaining stage (what
// `paffy chain | tile | trim` compute between the blast phase and cactus_consolidated).  Internal to libmiblast.so; the
// C ABI is include/mipaf.h.  Rules R-C*, R-T*, R-R* are listed in DESIGN.md section 11.
#pragma once

#include "mb_common.h"

namespace mb {

// cigar op in HBM: length << 3 | code; codes 0 '=', 1 'X', 2 'M' (aligned columns), 3 'I' (query only), 4 'D' (target only)
constexpr uint32_t kOpEq = 0, kOpX = 1, kOpM = 2, kOpI = 3, kOpD = 4;

struct ChainRec {                      // one alignment as the chain DP sees it (R-C2: ends pulled in by the trim fraction)
    int64_t qs;                        // untrimmed query start: the sort key the window search runs on
    int64_t tqs, tqe, tts, tte;
    int64_t score;
    int32_t same, pad;
};

struct TileRec {                       // one alignment as the tiling sees it
    uint64_t ops_off;                  // first op (and first entry of the per-op query offsets)
    uint32_t n_ops;
    int32_t same;
    int64_t qs, qe;
    uint32_t rec;                      // index of the record the level belongs to
    uint32_t pad;
};

struct TrimOut {                       // R-R2 / R-R3, all in the record's own op order
    long long cols, pre, suf;          // alignment columns; columns cut at the front / at the back
    long long qa, ta, qb, tb;          // query / target bases inside the two cuts
    long long nm, nb;                  // matches and columns that remain
    uint32_t first_op, first_len;      // first op that survives and what is left of it
    uint32_t last_op, last_len;        // last op that survives and what is left of it (== first when they coincide)
};

size_t sort_pairs_temp_bytes(int64_t n, int end_bit);   // the largest of the sizes a call sequence needs is allocated once
// stable ascending sort of (u64 key, u32 value) pairs on bits [0, end_bit)
void sort_pairs(void *temp, size_t temp_bytes, const unsigned long long *kin, unsigned long long *kout, const uint32_t *vin,
                uint32_t *vout, int64_t n, int end_bit, hipStream_t s);
void launch_iota(uint32_t *v, int64_t n, hipStream_t s);
void launch_gather_u64(const unsigned long long *src, const uint32_t *perm, unsigned long long *dst, int64_t n, hipStream_t s);
void launch_desc_keys(const long long *v, unsigned long long *key, int64_t n, hipStream_t s);
void launch_gather_chain(const ChainRec *src, const uint32_t *perm, ChainRec *dst, long long *tqe, long long *tend, int64_t n, hipStream_t s);
void launch_chain_dp(const ChainRec *recs, const long long *tqe, const long long *tend, const uint32_t *gstart, const int64_t *glmax,
                     int n_groups, long long max_gap, long long gap_open, long long gap_extend, long long *cs, int32_t *pred, int threads,
                     hipStream_t s);
void launch_tile(const TileRec *recs, const uint32_t *qstart, const uint64_t *cnt_off, int n_queries, uint16_t *cnt,
                 const uint32_t *ops, const uint32_t *qoff, int hist_bins, int32_t *level, hipStream_t s);
// sort-based tiling (mp_kernels.hip "Tiling without the walk")
size_t sort_keys64_temp_bytes(int64_t n, int end_bit);
void sort_keys64(void *temp, size_t temp_bytes, const unsigned long long *in, unsigned long long *out, int64_t n, int end_bit, hipStream_t s);
size_t scan_u64_temp_bytes(int64_t n);
void scan_u64(void *temp, size_t temp_bytes, const unsigned long long *in, unsigned long long *out, int64_t n, bool inclusive, hipStream_t s);
void launch_tile_heads(const unsigned long long *b, int64_t n, unsigned long long *flag, hipStream_t s);
void launch_tile_unique(const unsigned long long *b, const unsigned long long *flag, const unsigned long long *pos, int64_t n, unsigned long long *u, hipStream_t s);
void launch_tile_span(const unsigned long long *rs, const unsigned long long *re, int64_t n_runs, const unsigned long long *u, int64_t n_u, uint32_t *lo,
                      unsigned long long *cnt, hipStream_t s);
void launch_tile_expand(const uint32_t *lo, const unsigned long long *cnt, const unsigned long long *off, const uint32_t *rrank, int64_t n_runs,
                        unsigned long long *key, hipStream_t s);
void launch_tile_cover(const unsigned long long *key, int64_t n_pieces, const unsigned long long *u, unsigned long long *key2, uint32_t *weight, hipStream_t s);
void launch_widen(const uint32_t *in, unsigned long long *out, int64_t n, hipStream_t s);
void launch_tile_median(const unsigned long long *key2, const unsigned long long *wsum, int64_t n_pieces, int64_t n_recs, int32_t *level_by_rank, hipStream_t s);
void launch_trim_values(const uint32_t *ops, int64_t n_ops, unsigned long long *vc, unsigned long long *vm, unsigned long long *vq,
                        unsigned long long *vt, hipStream_t s);             // n_ops + 1 entries each (the last one 0)
void launch_trim_ops(const uint32_t *ops, int64_t n_ops, const unsigned long long *rec_start, const uint32_t *rec_n, int64_t n_recs,
                     const unsigned long long *Pc, const unsigned long long *Pm, long long num, long long den, unsigned long long *pre,
                     unsigned long long *suf, hipStream_t s);
void launch_trim_finish(const uint32_t *ops, const unsigned long long *rec_start, const uint32_t *rec_n, int64_t n_recs, const unsigned long long *Pc,
                        const unsigned long long *Pm, const unsigned long long *Pq, const unsigned long long *Pt, const unsigned long long *pre,
                        const unsigned long long *suf, TrimOut *out, hipStream_t s);

}  // namespace mb

// mb_common.h -- internal declarations shared by the host pipeline and the gfx950 kernels of
// libmiblast.so.  Nothing here is part of the C ABI (see include/miblast.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/miblast.h"

namespace mb {

constexpr int kSeedSpan = 19;                 // 12of19 = 1110100110010101111
constexpr int kSeedWeight = 12;
constexpr uint32_t kBuckets = 1u << 24;
constexpr uint8_t kSep = 0xFF;                // contig separator in code arrays
constexpr int32_t kNeg = -(1 << 29);
constexpr int kDevPad = 128;                  // separator bytes around device code arrays (8-byte loads may overrun)

// ---- device-side records ---------------------------------------------------------------------
struct DevHsp {                               // written by k_ungapped for every HSP with score >= K
    int32_t t_start, q_start, len, score;
    int32_t seed_t_end, seed_q_end;
    int32_t cnt[4];
};

struct DpProb {                               // one one-sided Y-drop DP (SURVEY A.7 ONE_SIDED)
    int32_t t0, q0;                           // anchor (concatenated coords on the searched strand)
    int32_t na, nb;                           // columns (target) / rows (query) available
    int32_t dir;                              // +1 forward from (t0,q0); -1 backward from (t0-1,q0-1)
    int32_t strand;                           // selects the query code array
    int32_t pad0, pad1;                       // pad0 = index of the chunk pair in the PairPtrs table
    uint64_t row_off;                         // index of this side's first row-chunk directory entry
    uint64_t ops_off;                         // traceback: index of this side's first run-length op (u32)
};

struct PairPtrs {                              // device pointers of one chunk pair's code arrays
    const uint8_t *tc, *qf, *qr;              // target, query '+', query '-'
};

struct DpOut {
    int32_t best, bi, bj, rows;
    int64_t cells;                            // cells evaluated (oracle counter dp_cells)
    int64_t clocks;                           // shader clocks spent in the row sweep (diagnostics)
    int32_t overflow;                         // 1: row wider than the LDS ring (rerun with HBM rows); 3: trace arena exhausted
    int32_t n_ops;                            // traceback: number of ops written
    long long prof[6];                        // MIBLAST_DP_PROFILE: shader clocks per phase of the row loop
};

struct UngappedCounters {
    unsigned long long extended, cols, hsps;
};

// ---- host-side sequence set --------------------------------------------------------------------
struct SeqSet {
    std::vector<std::string> names;
    std::vector<int64_t> starts, lens;
    int64_t total = 0;                        // concatenated length (one separator between contigs)
    std::vector<uint8_t> codes;               // [SEP] codes[0..total) [SEP]  -> codes.data()+1 is position 0
    int device = -1;
    uint8_t *d_buf = nullptr;                 // device copy of `codes`, kDevPad separator bytes on both sides
    int64_t *d_starts = nullptr;              // contig starts / lens on the device (revcomp kernel)
    int64_t *d_lens = nullptr;
    const uint8_t *host() const { return codes.data() + 1; }
    const uint8_t *dev() const { return d_buf + kDevPad; }
    int contig_of(int64_t pos) const;
};

int parse_fasta(const char *buf, size_t len, SeqSet &out);   // mb_seq.cpp

// ---- error plumbing ----------------------------------------------------------------------------
void set_error(const std::string &msg);
struct HipFailure { hipError_t code; const char *what; const char *file; int line; };

#define MB_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) throw ::mb::HipFailure{_e, #expr, __FILE__, __LINE__};              \
    } while (0)

// ---- kernel launch wrappers (mb_kernels.hip) ---------------------------------------------------
void launch_revcomp(const uint8_t *src, uint8_t *dst, const int64_t *starts, const int64_t *lens, int n_contigs,
                    int64_t total, hipStream_t s);
void launch_index_words(const uint8_t *codes, int64_t n, int step, uint32_t *words, int64_t n_slots, uint32_t *counts,
                        hipStream_t s);
void launch_index_scatter(const uint32_t *words, int64_t n_slots, int step, const uint32_t *offsets, uint32_t *cursor,
                          uint32_t *positions, hipStream_t s);
// exclusive scan of n u32 values; out may alias in; block_sums (u64, one per 2048 inputs) is scratch of
// ceil(n/2048)+1 entries and on return holds the exclusive prefix of the per-block totals, total last.
void launch_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, unsigned long long *block_sums, hipStream_t s);
void launch_block_sums(const uint32_t *in, int64_t n, unsigned long long *block_sums, hipStream_t s);
void launch_seed_count(const uint8_t *qcodes, int64_t qn, const uint32_t *offsets, int transitions, uint32_t *qcnt,
                       hipStream_t s);
void launch_seed_fill(const uint8_t *qcodes, int64_t q0, int64_t q1, int64_t qtot, const uint32_t *offsets,
                      const uint32_t *positions, int transitions, const uint32_t *hit_off, unsigned long long *keys,
                      hipStream_t s);
void launch_ungapped(const unsigned long long *keys, int64_t n_hits, unsigned *heads, unsigned *n_heads, const uint8_t *tcodes,
                     const uint8_t *qcodes, int64_t qtot, int32_t *extent, int xdrop, int K, DevHsp *hsps, int64_t hsp_cap,
                     UngappedCounters *ctr, hipStream_t s);
void launch_ydrop(bool global_rows, const DpProb *probs, DpOut *outs, int n, const PairPtrs *pairs,
                  int O, int E, int Y, int32_t *grows, uint8_t *arena, unsigned long long arena_bytes,
                  unsigned long long *arena_next, unsigned blk_bytes, unsigned long long *rowdir, hipStream_t s);
void launch_traceback(const DpProb *probs, DpOut *outs, const int *which, int n, const uint8_t *arena,
                      unsigned long long arena_bytes, const unsigned long long *rowdir, uint32_t *ops, hipStream_t s);
void launch_pack_ops(const DpProb *probs, const int *which, int n, const unsigned long long *coff, const uint32_t *ops,
                     uint32_t *packed, hipStream_t s);
size_t sort_keys_temp_bytes(int64_t n, int end_bit);
void sort_keys(void *temp, size_t temp_bytes, unsigned long long *in, unsigned long long *out, int64_t n, int end_bit,
               hipStream_t s);

constexpr int kLdsRowCap = 2048;              // LDS ring columns per DP problem (power of two)
constexpr int kGlobalRowCap = 1 << 20;        // fallback ring in HBM

}  // namespace mb

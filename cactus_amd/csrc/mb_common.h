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
struct DevHsp {                               // written by the ungapped kernels for every HSP with score >= K
    int32_t t_start, q_start, len, score;
    int32_t seed_t_end, seed_q_end;
    int32_t cnt[4];
    int32_t anchor_off;                       // k_hsp_anchor: middle of the best-scoring 31-column window, first on ties
    int32_t unit;                             // seed unit (pair, strand) of the launch the HSP belongs to; coordinates are the unit's own
};

// A seed unit = one (chunk pair, query strand) of a seed-stage launch.  The ungapped kernels work on the sorted hit keys of ALL units
// of a launch at once: unit u owns the diagonals [dbase, dbase + ttot + qtot + 2) of the launch's diagonal space (extent[], bit
// planes), a hit of the unit has the key (dbase + t_end - q_end + qtot) << 32 | q_end with t_end / q_end in the unit's own
// coordinates, and the units' ranges ascend with the unit index, so the sorted keys fall into one stretch per unit.
struct SeedUnit {
    const uint8_t *tc, *qc;                   // position 0 of the target / of the searched strand of the query
    int32_t qtot, ttot;                       // bases (concatenated contigs) of the query / target set
    uint32_t dbase;                           // first diagonal of the unit
    int32_t index;                            // batched seed search: the unit's target index (SparseIndex table)
    int64_t qpos0;                            // batched seed search: first slot of the unit's query positions in the launch's q space
};
struct UnitTab {                              // by value in the kernel arguments: a launch of ONE unit never touches memory for it
    SeedUnit one;
    const SeedUnit *tab;
    int32_t n;
    // where a diagonal's entry of extent[] lies: slot (dq * ext_mul) & ext_mask, or dq itself when ext_mul is 0 (the default: a zeroed table).
    // A strand in several q batches whose keys are sorted by the SCRAMBLED diagonal (mb_seed_dense.h) keeps its extents in the same order:
    // the kernels that read and leave them walk the array as they walk the keys, instead of touching a random line per run (round 6).
    uint32_t ext_mul, ext_mask;
};

struct DpProb {                               // one one-sided Y-drop DP (SURVEY A.7 ONE_SIDED)
    int32_t t0, q0;                           // anchor (concatenated coords on the searched strand)
    int32_t na, nb;                           // columns (target) / rows (query) available
    int32_t dir;                              // +1 forward from (t0,q0); -1 backward from (t0-1,q0-1)
    int32_t strand;                           // selects the query code array
    int32_t pad0;                             // index of the chunk pair in the PairPtrs table
    int32_t row_lo;                           // 0: fresh DP from the origin; > 0: continue after row row_lo from snapshot init_snap
    uint64_t row_off;                         // index of this piece's first row-chunk directory entry (record 0 <-> row row_lo)
    uint64_t ops_off;                         // traceback: index of this side's first run-length op (u32)
    int32_t stop_row;                         // > 0: stop after this row if the DP is still alive and write the exit snapshot
    int32_t snap_row;                         // > 0: write the entry snapshot after this row
    int32_t init_snap;                        // snapshot to continue from (row_lo > 0)
    int32_t snap_idx;                         // entry snapshot slot; the exit snapshot is slot snap_idx + 1 (-1: none)
    int32_t snap_row2, snap_row3;             // > 0: later entry snapshots (slots snap_idx + 2 / + 3): a hand-over rejected at snap_row is retried there
    // hand-over inside the launch (k_ydrop2, mb_ydrop2.h): a piece that reaches its stop row checks its state against the aimed relay's
    // entry snapshot itself and, rejected, goes on to the relay's next snapshot / the relay after -- the continuation the host would
    // otherwise queue as a launch of its own.  All three are 0 when unused (a zeroed record is a piece without an aim).
    int32_t aim1;                             // 1 + the aimed relay's piece in the round's table (probs of the launch = table + first); 0: none
    int32_t vjob1;                            // 1 + this piece's check among the launch's VerifyJobs (rewritten to the hand-over it ended at); 0: none
    int32_t cap_row;                          // the last row the piece may reach on its own (row-chunk directory entries are reserved up to it)
    int32_t ck0;                              // which entry snapshot of the aimed relay the first check is made against (0: after snap_row, 1 / 2: snap_row2 / 3)
};
constexpr int kSnapSlots = 4;                 // snapshot slots per piece: entry, exit, entry 2, entry 3

// DP state after a row (relay hand-over and continuation): header + C and D of the window [LY, RY)
struct SnapHdr {
    int32_t valid, LY, RY, best, bi, bj, row, rows;
    int64_t cells;
    int32_t stamp;                            // k_ydrop2: the round's stamp, written LAST (release): a reader inside the same launch trusts a header that carries it
    int32_t pad[5];
};
constexpr int kSnapCols = 2048;               // = kLdsRowCap: only the LDS-ring variant takes snapshots
constexpr size_t kSnapBytes = sizeof(SnapHdr) + 2 * (size_t)kSnapCols * sizeof(int32_t);

struct PairPtrs {                              // device pointers of one chunk pair's code arrays
    const uint8_t *tc, *qf, *qr;              // target, query '+', query '-'
};

struct DpOut {
    int32_t best, bi, bj, rows;
    int64_t cells;                            // cells evaluated (oracle counter dp_cells)
    int64_t clocks;                           // shader clocks spent in the row sweep (diagnostics)
    int32_t overflow;                         // 1: row wider than the LDS ring (rerun with HBM rows); 3: trace arena exhausted
    int32_t n_ops;                            // traceback: number of ops written
    int32_t stopped, exit_j;                  // 1: stopped at stop_row with live cells (exit snapshot written); best column of that row
    int32_t fin_stop, fin_aim1, fin_ck, fin_checks;   // k_ydrop2: the stop row, aim (1 + piece) and entry snapshot (0..2) the piece ended with; hand-overs it checked itself
    long long prof[6];                        // MIBLAST_DP_PROFILE: shader clocks per phase of the row loop
};

// ---- traceback --------------------------------------------------------------------------
// A side's path runs through the pieces of its validated chain.  Every piece is walked by its own wave at the same
// time (k_trace_walk): the piece that holds the best cell from that cell, every other piece from a GUESS, the best
// cell of its last row.  Paths that differ in their start merge after a few rows, so k_trace_join only walks from the
// true entry cell of a piece until it meets the guessed walk (same row, column and state = same remainder), then
// splices that walk's remaining runs.  The result is the run list of the sequential walk.
struct TbWalk {
    uint64_t row_off;                         // first row-chunk directory entry of the piece
    int32_t row_lo;                           // record 0 <-> row row_lo
    int32_t floor;                            // rows <= floor belong to the next walker of the side (-1: this is the head)
    int32_t si, sj;                           // start cell in the piece's coordinates (state: aligned pair)
    int32_t dr, dc;                           // a cell of this piece + (dr, dc) = the same cell in the next walker's coordinates
    uint64_t ops_off;                         // run buffer of the walk
    uint64_t rec_off;                         // one record (3 x u32: column, runs written, pending length << 2 | state) per entered row
    int32_t n_runs, ei, ej, estate;           // out: runs written; cell and state on reaching the floor
    int32_t pad[2];                           // the piece's share of the side's join buffer: first run slot (64 bit; all ones: head of its side)
};
// k_trace_prejoin's join walk of a piece: made from the PREDICTED entry (pi, pj, pstate); n_runs runs (-1: none made), then either the
// guessed walk's runs from its run nr on (the first shortened by sub) if the two met, or the cell and state it reached the floor with
struct TbJoin { int32_t pi, pj, pstate, n_runs, joined, ei, ej, estate, nr, sub; };
struct TbSeg { uint64_t src; int32_t n_runs, first_sub; };       // n_runs runs from ops[src], the first one shortened by first_sub
struct TbSide {
    int32_t first_walk, n_walks;              // walkers from the best cell's piece back to the head
    uint64_t jops_off;                        // run buffer of the join walk
    uint64_t seg_off;                         // segment list of the side (<= 2 * n_walks + 1 entries)
    int32_t n_segs, pad;
};

// relay hand-over check (k_verify): exit snapshot `eslot` of the upstream piece against entry snapshot `nslot` of the relay
struct VerifyJob { int32_t eslot, nslot, shift, drow; };     // E column = N column + shift; E row = N row + drow
struct VerifyOut {
    int32_t ok, n_rows;
    int32_t n_best, c;                        // relay's best at its entry row; E score = N score + c
    int64_t n_cells;
};

struct UngappedCounters {
    unsigned long long extended, cols, hsps;
};

// level-synchronous ungapped extension (mb_ungapped_ux.h)
struct UxEntry {                              // a hit whose walks are not finished after level 2
    uint32_t i;                               // index in the sorted key array
    int32_t cl, cr;                           // next chunk (8 columns) of the left / right walk; -1: that walk is finished
    int32_t run_l, best_l, bpos_l, run_r, best_r, bpos_r;
    uint32_t cols;                            // columns counted so far
};
struct UxScratch {                            // device scratch of one launch_ungapped call (the pipeline owns the memory)
    unsigned long long *rec;                  // n_hits records: low word = length to the right, high word = columns, or bit 31 + candidate slot
    UxEntry *blk_entries;                     // unfinished hits: 16 slots per block of k_ux_extend (12 of its wave 0, 4 of its wave 1),
    unsigned *blk_cnt;                        // ... slots in use, per (block, wave 0 / 1),
    unsigned n_blk;                           // ... blocks;
    UxEntry *entries;                         // and a shared list for what does not fit there
    unsigned entry_cap;
    unsigned *n_entries;                      // [0] list entries written (may exceed entry_cap: the overflowing lanes finish their hit themselves)
    uint32_t *long_bits;                      // one bit per diagonal: the diagonal's run belongs to k_ungapped_long
    uint32_t *dirty_bits;                     // one bit per diagonal: some walk of the run reaches the run's next hit (sequential rule needed)
    // where a diagonal's bit lies in the two planes: bit (dq * plane_mul) & plane_mask.  The dense seed stage sorts its hits by the
    // SCRAMBLED diagonal (mb_seed_dense.h): with the same multiplier here neighbours in the sorted order are neighbours in the planes
    // (a wave of k_ux_accept looks up two cache lines instead of 128).  1 / ~0: the diagonal itself.
    uint32_t plane_mul, plane_mask;
    unsigned *dirty_runs;                     // first hits of the dirty short runs (the entry list's memory, free after k_ux_tail); n_entries[1] counts them
    unsigned dirty_cap;
    const int32_t *extent;                    // extent[] of the launch, read by the first hit of a run when
    int extent_live;                          // ... an earlier q batch may have left extents (0: extent[] is all zero)
    // what a launch zeroes first: the two bit planes lie one behind the other from long_bits on, the counters (n_entries, blk_cnt) from
    // n_entries on; byte counts in multiples of 16 (a fill of any other size is two launches)
    unsigned long long zero_bits_bytes, zero_cnt_bytes;
    // level 1 of k_ux_extend from the packed strands of the launch's ONE unit (mb_ungapped_ux.h, round 6): the extension's 12-byte records
    // (32 bases: 2-bit codes + a flag per base that is N or a separator) of the target and of the searched query strand; t_n / q_n = bases.
    // Null / 0: windows from the code bytes.
    const uint32_t *t_px, *q_px;
    long long t_n, q_n;
};
inline size_t up16(size_t bytes) { return (bytes + 15) & ~(size_t)15; }

// ---- host-side sequence set --------------------------------------------------------------------
struct SeqSet {
    std::vector<std::string> names;
    std::vector<int64_t> starts, lens;
    int64_t total = 0;                        // concatenated length (one separator between contigs)
    std::vector<uint8_t> codes;               // [SEP] codes[0..total) [SEP]  -> codes.data()+1 is position 0
    const uint8_t *view = nullptr;            // a block of another set (mb_multi.cpp): position 0 inside the parent's codes, nothing owned
    int64_t origin = 0;                       // concatenated position of this set's position 0 in the file it is a block of (--step phase)
    int device = -1;
    uint8_t *d_buf = nullptr;                 // device copy of `codes`, kDevPad separator bytes on both sides
    int64_t *d_starts = nullptr;              // contig starts / lens on the device (revcomp kernel)
    int64_t *d_lens = nullptr;
    size_t d_cap = 0;                         // bytes of the device block d_buf heads (d_starts and d_lens live in its tail)
    const uint8_t *host() const { return view ? view : codes.data() + 1; }
    const uint8_t *dev() const { return d_buf + kDevPad; }
    int contig_of(int64_t pos) const;
};

int parse_fasta(const char *buf, size_t len, SeqSet &out);   // mb_seq.cpp

// ---- error plumbing ----------------------------------------------------------------------------
void set_error(const std::string &msg);
const std::string &last_error_text();       // of the calling thread
struct HipFailure { hipError_t code; const char *what; const char *file; int line; };

#define MB_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) throw ::mb::HipFailure{_e, #expr, __FILE__, __LINE__};              \
    } while (0)

// ---- kernel launch wrappers (mb_kernels.hip) ---------------------------------------------------
void launch_revcomp(const uint8_t *src, uint8_t *dst, const int64_t *starts, const int64_t *lens, int n_contigs,
                    int64_t total, hipStream_t s);
// slot k of the index <-> position first + k * step (first = the block's --step phase, 0 for a whole file)
void launch_index_words(const uint8_t *codes, int64_t n, int step, int64_t first, uint32_t *words, int64_t n_slots, uint32_t *counts,
                        hipStream_t s);
void launch_index_scatter(const uint32_t *words, int64_t n_slots, int step, int64_t first, const uint32_t *offsets, uint32_t *cursor,
                          uint32_t *positions, hipStream_t s);
// exclusive scan of n u32 values; out may alias in; block_sums (u64, one per 2048 inputs) is scratch of
// ceil(n/2048)+1 entries and on return holds the exclusive prefix of the per-block totals, total last.
void launch_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, unsigned long long *block_sums, hipStream_t s);
void launch_block_sums(const uint32_t *in, int64_t n, unsigned long long *block_sums, hipStream_t s);
void launch_seed_count(const uint8_t *qcodes, int64_t qn, const uint32_t *offsets, const uint32_t *occ, int transitions, uint32_t *qcnt,
                       hipStream_t s);
void launch_seed_fill(const uint8_t *qcodes, int64_t q0, int64_t q1, int64_t qtot, const uint32_t *offsets, const uint32_t *occ,
                      const uint32_t *positions, int transitions, const uint32_t *hit_off, unsigned long long *keys,
                      hipStream_t s, uint32_t hmul = 1u, uint32_t hmask = 0xFFFFFFFFu);      // key diagonal = (d * hmul) & hmask (mb_seed_dense.h)
void launch_index_clear(const uint32_t *words, int64_t n_slots, uint32_t *cursor, hipStream_t s);
// ---- seed stage of a large pair (mb_seed_dense.h) ----
inline size_t packed_wordsm(int64_t n) { return (size_t)((n + 63) / 64 + 2); }      // u64 words of the mask plane of n bases (k_pack2bit_mask runs one thread per word)
inline size_t packed_words2(int64_t n) { return 2 * packed_wordsm(n); }              // ... of the 2-bit plane: every thread stores the two words of its 64 bases, the padding words included
inline size_t packed_dwordsx(int64_t n) { return 6 * packed_wordsm(n); }             // ... of the ungapped extension's records (12 bytes per 32 bases: mb_ungapped_ux.h)
void launch_pack2bit(const uint8_t *codes, int64_t n, unsigned long long *p2, unsigned long long *pm, hipStream_t s,
                     uint32_t *px = nullptr);                           // px (packed_dwordsx(n) dwords): the ungapped extension's records as well
void launch_index_words_packed(const unsigned long long *p2, const unsigned long long *pm, int64_t n, int step, int64_t first, uint32_t *words, int64_t n_slots,
                               uint32_t *counts, hipStream_t s);
int64_t seed_ord_state_words(int64_t qtot);
void launch_seed_search_ord(const uint8_t *qcodes, const unsigned long long *p2, const unsigned long long *pm, int64_t qtot, const uint32_t *offsets, const uint32_t *occ,
                            const uint32_t *positions, int transitions, uint32_t hmul, uint32_t hmask, unsigned long long *keys, unsigned long long *scratch,
                            unsigned long long cap, unsigned long long *state, hipStream_t s);      // state[0] = hits of the strand afterwards
void launch_keys_unhash(unsigned long long *keys, int64_t n, uint32_t hinv, uint32_t hmask, hipStream_t s);
// grouping the keys by diagonal without the device-wide sort (mb_seed_bin.h)
int64_t bin_state_words();
int bin_cap_big();
unsigned long long bin_keys_max();          // more keys than this get no plan (mb_seed_bin.h)
int64_t bin_matrix_words_for(unsigned long long cap, int diag_bits, int mean);
void launch_bin_plan(const unsigned long long *keys, const unsigned long long *n_ptr, unsigned long long cap, int diag_bits, int mean, uint32_t *state, uint32_t *matrix,
                     hipStream_t s);
void launch_bin_group(const unsigned long long *in, unsigned long long *out, int64_t n, int diag_bits, int nbits, int n_small, int n_big, const uint32_t *state, const uint32_t *matrix,
                      uint32_t hinv, uint32_t hmask, hipStream_t s);
void launch_scan_index(uint32_t *counts, uint32_t *offsets, unsigned long long *block_sums, uint32_t *occ, hipStream_t s);
void launch_seed_search(const uint8_t *qcodes, int64_t qtot, const uint32_t *offsets, const uint32_t *occ, const uint32_t *positions, int transitions,
                        unsigned long long *keys, unsigned long long cap, unsigned long long *total, hipStream_t s);
// ut: the seed units of the launch (one for a strand of a pair, all (pair, strand) units of a batched call); ctr: one
// UngappedCounters per unit -- extended / cols per unit, hsps of entry 0 = the slot counter of the whole launch
void launch_ungapped(const unsigned long long *keys, int64_t n_hits, unsigned *heads, unsigned *n_heads, const UnitTab &ut,
                     int64_t n_diagonals, int32_t *extent, int xdrop, int K, DevHsp *hsps, int64_t hsp_cap,
                     UngappedCounters *ctr, const UxScratch *ux, bool extent_clean, hipStream_t s,        // ux: scratch of the level-synchronous pipeline, or nullptr; extent_clean: extent[] is all zero
                     bool n_heads_clean = false);                                                      // n_heads_clean: the caller has zeroed the 8 counters of n_heads
// wsegs / walns / wref: walls mode (miblast_params.walls) -- WallSeg runs of the earlier alignments, int2 run ranges per alignment, int2
// alignment ranges per problem; nullptr without walls
void launch_ydrop(bool global_rows, const DpProb *probs, DpOut *outs, int n, const PairPtrs *pairs,
                  int O, int E, int Y, int32_t *grows, uint8_t *arena, unsigned long long arena_bytes,
                  unsigned long long *arena_next, unsigned blk_bytes, unsigned long long *rowdir, uint8_t *snaps, hipStream_t s,
                  const void *wsegs = nullptr, const void *walns = nullptr, const void *wref = nullptr, uint8_t *wflags = nullptr);
void launch_ydrop1(int K, const DpProb *probs, DpOut *outs, int n, const PairPtrs *pairs, int O, int E, int Y, uint8_t *arena,
                   unsigned long long arena_bytes, unsigned long long *arena_next, unsigned blk_bytes, unsigned long long *rowdir,
                   uint8_t *snaps, const int *order, hipStream_t s,       // order: piece of block b (k_ydrop2 only), or nullptr
                   int first = 0, struct VerifyJob *vjobs = nullptr, int stamp = 0, int force_mod = 0);   // probs = the round's table, the launch's pieces [first, first + n); k_ydrop2's hand-over inside the launch
void launch_trace_walk(TbWalk *walks, int n, const uint8_t *arena, unsigned long long arena_bytes,
                       const unsigned long long *rowdir, uint32_t *ops, uint32_t *recs, hipStream_t s);
void launch_trace_prejoin(const TbWalk *walks, int n, TbJoin *joins, const uint8_t *arena, unsigned long long arena_bytes,
                          const unsigned long long *rowdir, uint32_t *ops, const uint32_t *recs, int poison, hipStream_t s);
void launch_trace_join(TbSide *sides, int n, const TbWalk *walks, TbSeg *segs, const uint8_t *arena, unsigned long long arena_bytes,
                       const unsigned long long *rowdir, uint32_t *ops, const uint32_t *recs, const TbJoin *joins, hipStream_t s);
void launch_pack_segs(const TbSeg *segs, const unsigned long long *dst, int n, const uint32_t *ops, uint32_t *packed, hipStream_t s);
void launch_verify(const VerifyJob *jobs, VerifyOut *res, int n, const uint8_t *snaps, int Y, int E, hipStream_t s);
// ---- batched seed stage (mb_seed_batch.h) ----
struct BatchTarget {                          // one distinct target of the call
    const uint8_t *codes;                     // position 0
    int64_t n;                                // bases
    int32_t step, pad;
    int64_t first;                            // slot s <-> position first + s * step  (--step phase of a block of a larger file)
    int64_t n_slots;
    int64_t slot0;                            // first slot in words[] / positions[]
    int64_t cbase;                            // first entry in cnt[] / starts[] / cursor[]  (n_slots + 1 entries)
    int64_t blk0;                             // first block of the per-slot kernels (256 slots per block)
};
constexpr int kBxWordsPerTarget = 1 << 18;    // 64-bit words of a target's bucket bitmap
constexpr int kBxDirBlocks = 128;             // blocks per target of k_bx_popc / k_bx_dir: 2048 bitmap words each

constexpr int kBsTile = 2048;                 // a unit's share of the q space of a batched seed search is a whole number of scan tiles
void launch_batch_index(const BatchTarget *tg, int n_targets, int64_t slot_blocks, int64_t n_cnt, uint32_t *words, unsigned long long *bits,
                        uint32_t *dir, uint32_t *bsum, uint32_t *cnt, uint32_t *cursor, uint32_t *starts, unsigned long long *scan_sums, uint32_t *positions,
                        hipStream_t s);
void launch_batch_seed_count(const SeedUnit *units, int n_units, const BatchTarget *tg, const unsigned long long *bits, const uint32_t *dir,
                             const uint32_t *starts, int transitions, int64_t q_slots, uint32_t *qcnt, uint32_t *hit_off, unsigned long long *scan_sums,
                             hipStream_t s);
void launch_batch_seed_fill(const SeedUnit *units, int n_units, const BatchTarget *tg, const unsigned long long *bits, const uint32_t *dir,
                            const uint32_t *starts, const uint32_t *positions, int transitions, int64_t q_slots, const uint32_t *hit_off,
                            unsigned long long *keys, hipStream_t s);
struct RcItem { const uint8_t *src; const int64_t *starts, *lens; long long total, grid_off; int n_contigs, pad; };
void launch_revcomp_sets(const RcItem *items, int n_items, int64_t grid_bytes, uint8_t *dst, hipStream_t s);
// outgroup trimming (k_cov_*, k_gather_stretches): the query sets of one call share every launch
struct CovItem { const uint8_t *codes; long long total, off_depth, off_edges; unsigned cap, pad; };     // off_depth: a multiple of 256
struct GatherItem { const uint8_t *src; uint8_t *dst; int64_t *d_starts, *d_lens; long long grid_off, seq_bytes, total, iv_off; int n_iv, pad; };
void launch_cov_mark(const long long *spans, int n, uint32_t *diff, hipStream_t s);
void launch_cov_edges(const uint32_t *depth, const CovItem *items, int n_items, int64_t n_depth, unsigned *n_edges, long long *first, long long *last, hipStream_t s);
void launch_gather_stretches(const GatherItem *items, int n_items, int64_t grid_bytes, const long long *iv, hipStream_t s);
size_t sort_pairs_temp_bytes(int64_t n);
void launch_ungapped_hash16(const unsigned long long *keys, int64_t n_hits, const UnitTab &ut, int64_t n_diagonals, int xdrop, int K, DevHsp *hsps, int64_t hsp_cap,
                            UngappedCounters *ctr, const UxScratch *ux, unsigned long long *ka, unsigned long long *kb, uint32_t *va, uint32_t *vb, void *temp,
                            size_t temp_bytes, int32_t *extent, hipStream_t s);
size_t sort_keys_temp_bytes(int64_t n, int end_bit);
void sort_keys(void *temp, size_t temp_bytes, unsigned long long *in, unsigned long long *out, int64_t n, int begin_bit, int end_bit,
               hipStream_t s);

constexpr int kLdsRowCap = 2048;              // LDS ring columns per DP problem (power of two)
constexpr int kGlobalRowCap = 1 << 20;        // fallback ring in HBM

}  // namespace mb

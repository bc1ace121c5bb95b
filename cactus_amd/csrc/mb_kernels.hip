// mb_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the MI355X-native blast phase.
//
// Stage map (SURVEY.md section 8a rows a6..a10, Appendix A.10 rules):
//   k_revcomp            '-' strand of the query set (A.1)
//   k_index_words/_scatter + scan kernels   target seed position table, CSR over 2^24 words (A.3, row a6)
//   k_scan_apply<true>   ... whose last scan pass also writes the occupancy bitmap of the seed table (keeps most lookups in L2)
//   k_seed_search        seed search in one pass: 12of19 word + 12 one-transition variants -> (diagonal,q) hit keys
//                        (A.4, row a7); k_seed_count/_fill: the two passes with exact sizes and q-ordered keys
//   k_run_heads, k_ungapped / k_ux_* (mb_ungapped_ux.h) / k_ungapped_grp (mb_ungapped_grp.h), k_ungapped_long, k_hsp_anchor
//                        per-diagonal suppression + x-drop extension, HSP emission, anchors (A.4/A.5/A.6, row a8)
//   k_ydrop2, k_ydrop1<K>   one-sided Y-drop affine DP (A.7, row a10), ONE wave per piece, previous row in registers
//   k_ydrop<HBM,...>     the same DP with four waves per piece and the row ring in LDS or HBM (wide windows)
//   k_verify             relay hand-over check: exit state of a piece against the entry state of the next
//   k_trace_walk/_join, k_pack_segs   traceback: every piece walked at once, then stitched
// All scoring is int32; integer results are bit-identical to the CPU oracle by construction
// (see DESIGN.md "Why the row sweep is exact").
#include "mb_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace mb {

#include "mb_xdrop.h"
#include "mb_units.h"

#include "mb_seedword.h"

// Pointers read from a table in memory are generic for the compiler: it then emits FLAT loads and, because those may
// complete out of order with LDS, waits for every outstanding memory operation around them.  The sequence pointers of
// PairPtrs are global memory: say so.
typedef const uint8_t __attribute__((address_space(1))) *gbytes;
__device__ __forceinline__ gbytes as_global(const uint8_t *p) { return (gbytes)(unsigned long long)p; }

// ---- wave-level helpers (DPP scans, SGPR pinning) --------------------------------------------------------
constexpr int kNeg2 = -(1 << 30);            // below every real score: fill value for shifted-in lanes
constexpr int kRowChunk = 4096;              // rows per row-info chunk (16 B each = one 64 KiB arena block)

__device__ __forceinline__ int dpp_shr1(int v, int fill) {            // lane l <- lane l-1, lane 0 <- fill
    return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_scan_max(int v) {                  // inclusive prefix max over the wave
    constexpr int kId = -2147483647 - 1;                                        // identity of max: lets the DPP fold into v_max
    v = max(v, __builtin_amdgcn_update_dpp(kId, v, 0x111, 0xf, 0xf, false));    // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(kId, v, 0x112, 0xf, 0xf, false));    // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(kId, v, 0x114, 0xf, 0xf, false));    // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(kId, v, 0x118, 0xf, 0xf, false));    // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(kId, v, 0x142, 0xa, 0xf, false));    // row_bcast:15 -> rows 1,3
    v = max(v, __builtin_amdgcn_update_dpp(kId, v, 0x143, 0xc, 0xf, false));    // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ int dpp_scan_add(int v) {                   // inclusive prefix sum over the wave
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }          // pin a wave-uniform value to an SGPR

__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    return ((unsigned long long)(unsigned)uni((int)(v >> 32)) << 32) | (unsigned)uni((int)(unsigned)v);
}

// ------------------------------------------------------------------------------------------------
__global__ void k_revcomp(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int64_t *__restrict__ starts,
                          const int64_t *__restrict__ lens, int n_contigs, int64_t total) {
    int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= total) return;
    int lo = 0, hi = n_contigs - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (starts[mid] <= pos) lo = mid; else hi = mid - 1;
    }
    int64_t st = starts[lo], n = lens[lo];
    if (pos >= st + n) { dst[pos] = kSep; return; }
    unsigned v = src[st + n - 1 - (pos - st)];
    if ((v & 7u) < 4u) v = (v & 8u) | (3u - (v & 7u));
    dst[pos] = (uint8_t)v;
}

void launch_revcomp(const uint8_t *src, uint8_t *dst, const int64_t *starts, const int64_t *lens, int n_contigs,
                    int64_t total, hipStream_t s) {
    if (total <= 0) return;
    int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(k_revcomp, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, starts, lens, n_contigs, total);
}

#include "mb_sets.h"

void launch_revcomp_sets(const RcItem *items, int n_items, int64_t grid_bytes, uint8_t *dst, hipStream_t s) {
    if (grid_bytes > 0 && n_items > 0) hipLaunchKernelGGL(k_revcomp_sets, dim3((unsigned)((grid_bytes + 255) / 256)), dim3(256), 0, s, items, n_items, dst);
}

#include "mb_seed_index.h"

// ---- the seed stage of a large pair: packed strands, q-ordered one-pass search, scrambled diagonals (mb_seed_dense.h) ------------
#include "mb_seed_dense.h"

void launch_pack2bit(const uint8_t *codes, int64_t n, unsigned long long *p2, unsigned long long *pm, hipStream_t s, uint32_t *px) {
    const int64_t nm = (int64_t)packed_wordsm(n);
    hipLaunchKernelGGL(k_pack2bit_mask, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, s, codes, n, p2, pm, nm, px);
}

void launch_index_words_packed(const unsigned long long *p2, const unsigned long long *pm, int64_t n, int step, int64_t first, uint32_t *words, int64_t n_slots,
                               uint32_t *counts, hipStream_t s) {
    if (n_slots <= 0) return;
    hipLaunchKernelGGL(k_index_words_packed, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, s, p2, pm, n, step, first, words, n_slots, counts);
}

// state (u64 words, zeroed by the caller): [0] hits of the strand (also when they did not fit `cap`), [8 ..) one word per tile = its
// stretch of the scratch, then u32 tile counts, their scan and the scan's block sums -- seed_ord_state_words(qtot) words in all.
// (tiles of 4096 positions with one word variant, of 512 with thirteen: sized for the smaller tile)
static int64_t ord_tiles_max(int64_t qtot) { return (((qtot + kOrdThreadsMin - 1) / kOrdThreadsMin + 1) + 3) & ~(int64_t)3; }      // (a multiple of 4: the u32 arrays behind it stay 16-byte aligned)
int64_t seed_ord_state_words(int64_t qtot) {
    const int64_t t = ord_tiles_max(qtot);
    return (8 + t + t / 2 + t / 2 + (t / 2048 + 4) + 8 + 1) & ~(int64_t)1;
}

// scratch: `cap` words (the sort's output buffer serves); keys: `cap` words; the kernels are queued on s, nothing is waited for
void launch_seed_search_ord(const uint8_t *qcodes, const unsigned long long *p2, const unsigned long long *pm, int64_t qtot, const uint32_t *offsets, const uint32_t *occ,
                            const uint32_t *positions, int transitions, uint32_t hmul, uint32_t hmask, unsigned long long *keys, unsigned long long *scratch,
                            unsigned long long cap, unsigned long long *state, hipStream_t s) {
    if (qtot <= 0) return;
    const int threads = transitions ? 512 : 1024, per_tile = threads * (transitions ? 1 : 4);
    const int n_tiles = (int)((qtot + per_tile - 1) / per_tile);
    const int64_t t = ord_tiles_max(qtot);
    unsigned long long *tile_base = state + 8;
    uint32_t *tile_cnt = (uint32_t *)(state + 8 + t), *tile_off = (uint32_t *)(state + 8 + t + t / 2);
    unsigned long long *sums = state + 8 + t + t / 2 + t / 2;
    const unsigned grid = (unsigned)std::min(n_tiles, transitions ? 3072 : 1024);
#define MB_ORD(P, R, NV, T) hipLaunchKernelGGL((k_seed_hits<P, R, NV, T>), dim3(grid), dim3(T), 0, s, qcodes, p2, pm, qtot, offsets, occ, scratch, cap, state, tile_base, tile_cnt, n_tiles)
    if (transitions) { if (p2) MB_ORD(true, 1, 1 + kSeedWeight, 512); else MB_ORD(false, 1, 1 + kSeedWeight, 512); }
    else { if (p2) MB_ORD(true, 4, 1, 1024); else MB_ORD(false, 4, 1, 1024); }
#undef MB_ORD
    launch_scan_u32(tile_cnt, tile_off, n_tiles, sums, s);
    hipLaunchKernelGGL(k_seed_keys, dim3((unsigned)std::min(n_tiles, 8192)), dim3(256), 0, s, scratch, tile_base, tile_cnt, tile_off, positions, keys, cap, qtot, per_tile, hmul, hmask, n_tiles);
    MB_HIP(hipGetLastError());
}

void launch_keys_unhash(unsigned long long *keys, int64_t n, uint32_t hinv, uint32_t hmask, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_keys_unhash, dim3((unsigned)(((n + 1) / 2 + 255) / 256)), dim3(256), 0, s, keys, n, hinv, hmask);
}

// ---- the keys of a strand grouped by diagonal: bins by the top bits of the scrambled diagonal, a work-group per bin in LDS (mb_seed_bin.h)
#include "mb_seed_bin.h"

int64_t bin_state_words() { return (kBinStateWords + 3) & ~3; }
int bin_cap_big() { return kBinCapBig; }
unsigned long long bin_keys_max() { return kBinKeysMax; }
int64_t bin_matrix_words_for(unsigned long long cap, int diag_bits, int mean) { return (int64_t)((bin_matrix_words(cap, diag_bits, mean) + 3ull) & ~3ull); }

// queued behind the kernels that write the keys and their number (*n_ptr); state: bin_state_words() zeroed u32 words; matrix:
// bin_matrix_words_for(cap, ..) words (not initialised).  The plan -- state[0 .. 8) -- is what the host reads back together with the number of keys.
void launch_bin_plan(const unsigned long long *keys, const unsigned long long *n_ptr, unsigned long long cap, int diag_bits, int mean, uint32_t *state, uint32_t *matrix,
                     hipStream_t s) {
    const unsigned long long room = std::min<unsigned long long>(cap, kBinKeysMax);
    hipLaunchKernelGGL(k_bin_count, dim3((unsigned)std::max<unsigned long long>(1, (room + kBinChunk - 1) / kBinChunk)), dim3(256), 0, s, keys, n_ptr, cap, diag_bits, mean, state, matrix);
    hipLaunchKernelGGL(k_bin_scan, dim3((1u << kBinBitsMax) / 256u), dim3(256), 0, s, n_ptr, cap, diag_bits, mean, state, matrix);
    MB_HIP(hipGetLastError());
}

// in: n keys (scrambled diagonal << 32 | q end) in any order; out: the same keys grouped by diagonal (unscrambled), q ascending inside a
// diagonal -- the array sort_keys(.., 32, 32 + diag_bits) + launch_keys_unhash give.  `in` is left as it was (with more than one bin `out` holds the keys
// bin by bin in between).  nbits / n_small / n_big (plan words 0, 6, 2): the plan of launch_bin_plan for these keys (largest bin <= bin_cap_big()).
void launch_bin_group(const unsigned long long *in, unsigned long long *out, int64_t n, int diag_bits, int nbits, int n_small, int n_big, const uint32_t *state, const uint32_t *matrix,
                      uint32_t hinv, uint32_t hmask, hipStream_t s) {
    if (n <= 0) return;
    const unsigned long long *binned = in;
    if (nbits > 0) {
        if (nbits <= kBinStagedBits) hipLaunchKernelGGL(k_bin_scatter_staged, dim3((unsigned)((n + kBinChunk - 1) / kBinChunk)), dim3(1024), 0, s, in, out, n, diag_bits, nbits, state, matrix);
        else hipLaunchKernelGGL(k_bin_scatter, dim3((unsigned)((n + kBinChunk - 1) / kBinChunk)), dim3(1024), 0, s, in, out, n, diag_bits, nbits, state, matrix);
        binned = out;
    }
    if (n_small > 0) hipLaunchKernelGGL((k_bin_sort<kBinCapSmall, 11, 512>), dim3(1u << nbits), dim3(512), 0, s, binned, out, state, diag_bits, nbits, hinv, hmask);
    if (n_big > 0) hipLaunchKernelGGL((k_bin_sort<kBinCapBig, 12, 1024>), dim3(1u << nbits), dim3(1024), 0, s, binned, out, state, diag_bits, nbits, hinv, hmask);
    MB_HIP(hipGetLastError());
}

// ---- the seed stage of all pairs of a call in shared launches (mb_seed_batch.h) ------------------------------------------------
#include "mb_seed_batch.h"

// sparse seed position tables of the n_targets distinct targets of a call.  bits / dir: n_targets x 2^18 entries; words, positions:
// one entry per slot; cnt, starts: n_cnt = slots + n_targets entries (cnt is scratch and ends as the zeroed cursor array);
// bsum: n_targets x 128; scan_sums: ceil(n_cnt / 2048) + 2
// cnt: the positions per occupied bucket; cursor: as many zeroed counters again for the scatter.  When the two lie right behind the
// bitmaps (cnt = end of bits, cursor = cnt + n_cnt rounded up to 4 entries: the pipeline's layout) ONE fill zeroes all three.
void launch_batch_index(const BatchTarget *tg, int n_targets, int64_t slot_blocks, int64_t n_cnt, uint32_t *words, unsigned long long *bits,
                        uint32_t *dir, uint32_t *bsum, uint32_t *cnt, uint32_t *cursor, uint32_t *starts, unsigned long long *scan_sums, uint32_t *positions,
                        hipStream_t s) {
    const size_t bits_bytes = (size_t)n_targets * kBxWordsPerTarget * 8, cnt_bytes = up16((size_t)n_cnt * 4);
    if ((uint8_t *)cnt == (uint8_t *)bits + bits_bytes && (uint8_t *)cursor == (uint8_t *)cnt + cnt_bytes) MB_HIP(hipMemsetAsync(bits, 0, bits_bytes + 2 * cnt_bytes, s));
    else {
        MB_HIP(hipMemsetAsync(bits, 0, bits_bytes, s));
        MB_HIP(hipMemsetAsync(cnt, 0, cnt_bytes, s));
        MB_HIP(hipMemsetAsync(cursor, 0, cnt_bytes, s));
    }
    if (slot_blocks > 0) hipLaunchKernelGGL(k_bx_words, dim3((unsigned)slot_blocks), dim3(256), 0, s, tg, n_targets, words, bits);
    hipLaunchKernelGGL(k_bx_popc, dim3((unsigned)(n_targets * kBxDirBlocks)), dim3(256), 0, s, bits, bsum);
    hipLaunchKernelGGL(k_bx_dir, dim3((unsigned)(n_targets * kBxDirBlocks)), dim3(256), 0, s, bits, bsum, dir);
    if (slot_blocks > 0) hipLaunchKernelGGL(k_bx_count, dim3((unsigned)slot_blocks), dim3(256), 0, s, tg, n_targets, words, bits, dir, cnt);
    launch_scan_u32(cnt, starts, n_cnt, scan_sums, s);
    if (slot_blocks > 0) hipLaunchKernelGGL(k_bx_scatter, dim3((unsigned)slot_blocks), dim3(256), 0, s, tg, n_targets, words, bits, dir, starts, cursor, positions);
    MB_HIP(hipGetLastError());
}

// hits per query slot of every unit (q_slots = the launch's q space, a multiple of 2048), then their exclusive scan -> hit_off;
// scan_sums[k] = hits before tile k (2048 slots), scan_sums[q_slots / 2048] = all hits
void launch_batch_seed_count(const SeedUnit *units, int n_units, const BatchTarget *tg, const unsigned long long *bits, const uint32_t *dir,
                             const uint32_t *starts, int transitions, int64_t q_slots, uint32_t *qcnt, uint32_t *hit_off, unsigned long long *scan_sums,
                             hipStream_t s) {
    if (q_slots <= 0) return;
    hipLaunchKernelGGL(k_bs_count, dim3((unsigned)(q_slots / 256)), dim3(256), 0, s, units, n_units, tg, bits, dir, starts, transitions ? 1 + kSeedWeight : 1, qcnt);
    launch_scan_u32(qcnt, hit_off, q_slots, scan_sums, s);
    MB_HIP(hipGetLastError());
}

void launch_batch_seed_fill(const SeedUnit *units, int n_units, const BatchTarget *tg, const unsigned long long *bits, const uint32_t *dir,
                            const uint32_t *starts, const uint32_t *positions, int transitions, int64_t q_slots, const uint32_t *hit_off,
                            unsigned long long *keys, hipStream_t s) {
    if (q_slots <= 0) return;
    hipLaunchKernelGGL(k_bs_fill, dim3((unsigned)(q_slots / 256)), dim3(256), 0, s, units, n_units, tg, bits, dir, starts, positions,
                       transitions ? 1 + kSeedWeight : 1, hit_off, keys);
    MB_HIP(hipGetLastError());
}

void launch_cov_mark(const long long *spans, int n, uint32_t *diff, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_cov_mark, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, spans, n, diff);
}
void launch_cov_edges(const uint32_t *depth, const CovItem *items, int n_items, int64_t n_depth, unsigned *n_edges, long long *first, long long *last, hipStream_t s) {
    if (n_depth > 0 && n_items > 0) hipLaunchKernelGGL(k_cov_edges, dim3((unsigned)((n_depth + 255) / 256)), dim3(256), 0, s, depth, items, n_items, n_edges, first, last);
}
void launch_gather_stretches(const GatherItem *items, int n_items, int64_t grid_bytes, const long long *iv, hipStream_t s) {
    if (grid_bytes > 0 && n_items > 0) hipLaunchKernelGGL(k_gather_stretches, dim3((unsigned)((grid_bytes + 255) / 256)), dim3(256), 0, s, items, n_items, iv);
}

size_t sort_keys_temp_bytes(int64_t n, int end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)n, 0,
                                   end_bit, (hipStream_t)0);
    return bytes;
}

// begin_bit = 32 when the keys already come in q order (k_seed_fill writes them so): the stable sort then only has to order
// the diagonals -- three 8-bit passes instead of seven on an 8 Mb pair.
void sort_keys(void *temp, size_t temp_bytes, unsigned long long *in, unsigned long long *out, int64_t n, int begin_bit, int end_bit,
               hipStream_t s) {
    MB_HIP(rocprim::radix_sort_keys(temp, temp_bytes, in, out, (size_t)n, begin_bit, end_bit, s));
}

// ------------------------------------------------------------------------------------------------
// ungapped x-drop extension with exact per-diagonal suppression (A.4, A.5).
// keys are sorted by (diagonal, q_end); the thread owning the first hit of a diagonal run walks
// the run in q order carrying extent[d], exactly the sequential rule "skip iff q_end <= extent[d]".
#include "mb_runs.h"

#include "mb_ungapped_lane.h"

// ---- eight-lanes-per-run variant of k_ungapped (the default for the short-run classes) -------------------------------
#include "mb_ungapped_grp.h"

// ---- level-synchronous pipeline for dense hit sets ------------------------------------------------------------------
#include "mb_ungapped_ux.h"

void launch_ungapped(const unsigned long long *keys, int64_t n_hits, unsigned *heads, unsigned *n_heads, const UnitTab &ut,
                     int64_t n_diagonals, int32_t *extent, int xdrop, int K, DevHsp *hsps, int64_t hsp_cap,
                     UngappedCounters *ctr, const UxScratch *ux, bool extent_clean, hipStream_t s, bool n_heads_clean) {
    if (n_hits <= 0) return;
    // runs longer than this go to the wave-per-run kernel: a few times the chance hits a diagonal holds on average
    int kLongRun = (int)std::min<int64_t>(kLongRunMax, 6 + 4 * n_hits / std::max<int64_t>(1, n_diagonals));
    if (const char *e = getenv("MIBLAST_LONG_RUN")) kLongRun = std::max(4, std::min(kLongRunMax, atoi(e)));      // (the long-run head list holds n_hits / 4 entries)
    // heads: the four short-run lists by run length and the long-run list (offsets in k_run_heads; 2.1 n_hits + 16 entries);
    // n_heads: five counters
    const uint64_t n = (uint64_t)n_hits;
    unsigned *heads_long = heads + (n + n / 2 + n / 4 + n / 8 + 8);
    if (!n_heads_clean) MB_HIP(hipMemsetAsync(n_heads, 0, up16((kRunClasses + 1) * sizeof(unsigned)), s));      // (n_heads: 8 counters)
    const int64_t max_long = n_hits / (kLongRun + 1) + 1;                        // a long run has more than kLongRun hits
    // Short runs.  Three kernels give the same results:
    //   lane  k_ungapped: a run per lane (the default for sparse hit sets: the phase's 0.6 Mb pairs, where a launch is as long as
    //         its longest chain of real extensions);
    //   ux    the level-synchronous pipeline of mb_ungapped_ux.h (the default for dense hit sets: a hit per four diagonals or more and at
    //         least 2^19 hits, where the work is chance hits -- the 1 Mb pair: 0.56 against 0.69 ms);
    //   grp   k_ungapped_grp: eight lanes per run (MIBLAST_UNGAPPED=grp only).
    // MIBLAST_UNGAPPED=lane|ux|grp forces one.
    const char *fe = getenv("MIBLAST_UNGAPPED");                                  // (read per launch: the tests switch it)
    const int forced = !fe ? 0 : !strcmp(fe, "lane") ? 1 : !strcmp(fe, "ux") ? 2 : !strcmp(fe, "grp") ? 3 : 0;
    // (the pooled hits of a batched call -- several units -- are dense enough for the pipeline whenever they are many: the 9-pair call
    //  of the evolver phase, 4.9 x 10^6 hits on 2.2 x 10^7 diagonals, 1.28 ms against 2.08 ms with a run per lane)
    // (and for the millions of hits of a chunk pair whatever their density: a 30 Mb x 30 Mb pair without homology, 1.5 x 10^7 chance hits per
    //  strand on 6 x 10^7 diagonals, 1.7 ms against 4.8 ms with a run per lane)
    int mode = forced ? forced : (ux && n_hits >= (1 << 19) && (ut.n > 1 || n_hits >= n_diagonals / 4 || n_hits >= (1 << 22))) ? 2 : 1;
    if (mode == 2 && (!ux || xdrop >= (1 << 24))) mode = 1;
    {   // (the pipeline takes the short runs hit by hit: it wants the list of the long runs only)
        const dim3 hg((unsigned)((n_hits + 1024 * kHeadsPerThread - 1) / (1024 * kHeadsPerThread)));
        static const bool all_lists = [] { const char *e = getenv("MIBLAST_UX_ALL_LISTS"); return e && atoi(e) != 0; }();      // (A/B switch: the four short lists as well)
        if (mode == 2 && !all_lists) hipLaunchKernelGGL(k_run_heads_long, dim3((unsigned)std::min<int64_t>((n_hits + 1023) / 1024, 4096)), dim3(256), 0, s, keys, n_hits, kLongRun, heads, n_heads);
        else hipLaunchKernelGGL(k_run_heads, hg, dim3(1024), 0, s, keys, n_hits, kLongRun, heads, n_heads);
    }
    if (mode == 1) {
        const int64_t blocks = (n_hits + 255) / 256 + kRunClasses;               // upper bound: sum over classes of ceil(runs / 256)
        hipLaunchKernelGGL(k_ungapped, dim3((unsigned)blocks), dim3(256), 0, s, keys, n_hits, heads, n_heads, ut,
                           extent, xdrop, K, hsps, hsp_cap, ctr);
    } else if (mode == 3) {
        // the groups are persistent: at most MIBLAST_UNGAPPED_BLOCKS blocks of 32 groups walk the runs in a strided order
        static const int64_t grp_blocks = [] { const char *e = getenv("MIBLAST_UNGAPPED_BLOCKS"); return e ? std::max(1, atoi(e)) : 4096; }();
        const int64_t blocks = std::min<int64_t>((n_hits + 31) / 32, grp_blocks); // (a run has at least one hit)
        hipLaunchKernelGGL(k_ungapped_grp<5>, dim3((unsigned)blocks), dim3(256), 0, s, keys, n_hits, heads, n_heads, ut,
                           extent, xdrop, K, hsps, hsp_cap, ctr);
    } else {
        MB_HIP(hipMemsetAsync(ux->long_bits, 0, (size_t)ux->zero_bits_bytes, s));
        MB_HIP(hipMemsetAsync(ux->n_entries, 0, (size_t)ux->zero_cnt_bytes, s));
        UxScratch sc = *ux;
        sc.extent = extent; sc.extent_live = extent_clean ? 0 : 1;
        ux = &sc;
        hipLaunchKernelGGL(k_ux_mark_long, dim3((unsigned)((max_long + 255) / 256)), dim3(256), 0, s, keys, heads_long, n_heads + kRunClasses, *ux);
        // level 1 from the packed strands when the caller has them (one unit: the dense path of a large pair): mb_ungapped_ux.h
        if (ux->t_px && ux->q_px && ut.n <= 1)
            hipLaunchKernelGGL(k_ux_extend_pk, dim3((unsigned)((n_hits + ux::kBlock - 1) / ux::kBlock)), dim3(ux::kBlock), 0, s, keys, n_hits, ut,
                               xdrop, K, *ux, hsps, hsp_cap, ctr);
        else
        hipLaunchKernelGGL(k_ux_extend, dim3((unsigned)((n_hits + ux::kBlock - 1) / ux::kBlock)), dim3(ux::kBlock), 0, s, keys, n_hits, ut,
                           xdrop, K, *ux, hsps, hsp_cap, ctr);
        const unsigned tail_blocks = (unsigned)std::min<int64_t>(2048, ((int64_t)ux->entry_cap + 2 * (int64_t)ux->n_blk + 31) / 32);
        hipLaunchKernelGGL(k_ux_tail, dim3(tail_blocks), dim3(256), 0, s, keys, n_hits, ut, xdrop, K, *ux, hsps, hsp_cap, ctr);
        hipLaunchKernelGGL(k_ux_accept, dim3((unsigned)std::min<int64_t>(4096, (n_hits + 255) / 256)), dim3(256), 0, s, keys, n_hits, ut, extent, *ux, hsps, ctr);
        hipLaunchKernelGGL(k_ux_resolve, dim3(256), dim3(256), 0, s, keys, n_hits, ut, extent, *ux, hsps, ctr);
        hipLaunchKernelGGL(k_ux_census, dim3(256), dim3(256), 0, s, ut, hsps, hsp_cap, ctr);
    }
    hipLaunchKernelGGL(k_ungapped_long, dim3((unsigned)std::min<int64_t>(8192, (max_long + 3) / 4)), dim3(256), 0, s, keys, n_hits, heads_long, n_heads + kRunClasses,
                       ut, extent, xdrop, K, hsps, hsp_cap, ctr);
    hipLaunchKernelGGL(k_hsp_anchor, dim3(512), dim3(256), 0, s, ut, hsps, hsp_cap, ctr);
    MB_HIP(hipGetLastError());                                                   // (a launch that was refused -- grid size, LDS -- must not pass as "no HSPs")
}

// ---- diagonal suppression through the 16-bit diagonal hash (miblast_params.diag_hash16) ------------------------------------------
#include "mb_hash16.h"

size_t sort_pairs_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (size_t)n, 0, 64, (hipStream_t)0);
    return bytes;
}

// every hit extended (level-synchronous pipeline without long diagonals), then the rule per (unit, hash class) in generation order.
// ka / kb: n_hits u64 each, va / vb: n_hits u32 each, temp: sort_pairs_temp_bytes(n_hits)
void launch_ungapped_hash16(const unsigned long long *keys, int64_t n_hits, const UnitTab &ut, int64_t n_diagonals, int xdrop, int K, DevHsp *hsps, int64_t hsp_cap,
                            UngappedCounters *ctr, const UxScratch *ux, unsigned long long *ka, unsigned long long *kb, uint32_t *va, uint32_t *vb, void *temp,
                            size_t temp_bytes, int32_t *extent, hipStream_t s) {
    if (n_hits <= 0) return;
    MB_HIP(hipMemsetAsync(ux->long_bits, 0, (size_t)ux->zero_bits_bytes, s));
    MB_HIP(hipMemsetAsync(ux->n_entries, 0, (size_t)ux->zero_cnt_bytes, s));
    UxScratch sc = *ux;
    sc.extent = extent; sc.extent_live = 0;
    hipLaunchKernelGGL(k_ux_extend, dim3((unsigned)((n_hits + ux::kBlock - 1) / ux::kBlock)), dim3(ux::kBlock), 0, s, keys, n_hits, ut, xdrop, K, sc, hsps, hsp_cap, ctr);
    const unsigned tail_blocks = (unsigned)std::min<int64_t>(2048, ((int64_t)sc.entry_cap + 2 * (int64_t)sc.n_blk + 31) / 32);
    hipLaunchKernelGGL(k_ux_tail, dim3(tail_blocks), dim3(256), 0, s, keys, n_hits, ut, xdrop, K, sc, hsps, hsp_cap, ctr);
    const unsigned blocks = (unsigned)((n_hits + 255) / 256);
    hipLaunchKernelGGL(k_h16_tkeys, dim3(blocks), dim3(256), 0, s, keys, n_hits, ut, ka, va);
    MB_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, ka, kb, va, vb, (size_t)n_hits, 0, 31, s));
    hipLaunchKernelGGL(k_h16_ckeys, dim3(blocks), dim3(256), 0, s, keys, n_hits, ut, vb, ka);
    MB_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, ka, kb, vb, va, (size_t)n_hits, 0, 59, s));
    hipLaunchKernelGGL(k_h16_resolve, dim3(blocks), dim3(256), 0, s, kb, va, n_hits, sc.rec, hsps, ctr);
    hipLaunchKernelGGL(k_ux_census, dim3(256), dim3(256), 0, s, ut, hsps, hsp_cap, ctr);
    hipLaunchKernelGGL(k_hsp_anchor, dim3(512), dim3(256), 0, s, ut, hsps, hsp_cap, ctr);
    MB_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// One-sided Y-drop DP (A.7 / A.10 ONE_SIDED), single pass with trace.
//
// One wave64 per problem.  Rows are swept sequentially (the y-drop rule prunes against the running
// best in ROW-MAJOR order, so row i+1 needs all of row i); inside a row the columns are evaluated
// 64 per segment, G segments per group with independent instruction streams:
//   * vertical gap D and the diagonal come from the previous row's C/D, kept in an LDS ring indexed
//     by column and overwritten in place;
//   * the horizontal gap I is a max-plus prefix scan over the row, the running best a prefix max --
//     both done with DPP row_shr / row_bcast steps (no LDS traffic), segment carries through SGPRs;
//   * target bases are staged in an LDS byte ring ahead of the window, query bases 64 rows at a time
//     in a register (v_readlane per row), so no global load sits on the row-to-row critical path;
//   * one 4-bit trace code per evaluated cell (two columns per byte) goes to 64 KiB blocks bump-allocated from an HBM arena, with a
//     16-byte (offset, LY) record per row in 4096-row chunks found through a per-problem directory.
struct RowInfo { unsigned long long off; uint32_t ly; uint32_t pad; };

// per-row packed score table: byte k = HOXD70[k][bq] + 128 for k = A,C,G,T (N handled separately)
__device__ __forceinline__ uint32_t row_score_lut(unsigned bq) {
    const unsigned b = bq & 7u;
    constexpr uint32_t LA = (91 + 128) | ((-114 + 128) << 8) | ((-31 + 128) << 16) | ((unsigned)(-123 + 128) << 24);
    constexpr uint32_t LC = (-114 + 128) | ((100 + 128) << 8) | ((-125 + 128) << 16) | ((unsigned)(-31 + 128) << 24);
    constexpr uint32_t LG = (-31 + 128) | ((-125 + 128) << 8) | ((100 + 128) << 16) | ((unsigned)(-114 + 128) << 24);
    constexpr uint32_t LT = (-123 + 128) | ((-31 + 128) << 8) | ((-114 + 128) << 16) | ((unsigned)(91 + 128) << 24);
    constexpr uint32_t LN = 28u | (28u << 8) | (28u << 16) | (28u << 24);            // -100 + 128
    return b == 0u ? LA : b == 1u ? LC : b == 2u ? LG : b == 3u ? LT : LN;
}
__device__ __forceinline__ int lut_score(uint32_t lut, unsigned at) {
    const unsigned a = at & 7u;
    const int v = (int)((lut >> ((a & 3u) * 8u)) & 0xFFu) - 128;
    return (a & 4u) ? -100 : v;
}

template <int CTRL, int ROWS>
__device__ __forceinline__ unsigned long long dpp_max64_step(const unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROWS, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROWS, 0xf, false);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned long long dpp_scan_max64(unsigned long long v) {      // inclusive prefix max over the wave (keys >= 0: 0 is the identity)
    v = dpp_max64_step<0x111, 0xf>(v); v = dpp_max64_step<0x112, 0xf>(v); v = dpp_max64_step<0x114, 0xf>(v);
    v = dpp_max64_step<0x118, 0xf>(v); v = dpp_max64_step<0x142, 0xa>(v); v = dpp_max64_step<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ unsigned long long dpp_shr1_64(unsigned long long v, unsigned long long fill) {      // lane l <- lane l-1, lane 0 <- fill
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)fill, (int)(unsigned)v, 0x138, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(fill >> 32), (int)(unsigned)(v >> 32), 0x138, 0xf, 0xf, false);
    return ((unsigned long long)hi << 32) | lo;
}

#include "mb_ydrop_lds.h"

template <bool GLOBAL_ROWS, bool PROF, bool WALLS>
__global__ __launch_bounds__(kYdThreads) void k_ydrop(const DpProb *__restrict__ probs, DpOut *__restrict__ outs, int n,
                                                      const PairPtrs *__restrict__ pairs, int O, int E, int Y, int32_t *grows,
                                                      uint8_t *__restrict__ arena, unsigned long long arena_bytes,
                                                      unsigned long long *__restrict__ arena_next, unsigned blk_bytes,
                                                      unsigned long long *__restrict__ rowdir, uint8_t *__restrict__ snaps,
                                                      const WallSeg *__restrict__ wsegs, const int2 *__restrict__ walns, const int2 *__restrict__ wref,
                                                      uint8_t *__restrict__ wflags) {
    int pi = blockIdx.x;
    if (pi >= n) return;
    DpProb pr = probs[pi];
    const PairPtrs pp = pairs[pr.pad0];
    const gbytes tc = as_global(pp.tc);
    const gbytes qc = as_global(pr.strand ? pp.qr : pp.qf);
    __shared__ YdShared sh;
    const int2 wr = WALLS ? wref[pi] : make_int2(0, 0);
    if (GLOBAL_ROWS) {
        int2 *CD = (int2 *)(grows + (size_t)pi * 2 * kGlobalRowCap);
        ydrop_body<true, PROF, WALLS>(pr, &outs[pi], tc, qc, O, E, Y, CD, nullptr, kGlobalRowCap, &sh, arena, arena_bytes, arena_next,
                                      blk_bytes, rowdir, snaps, wsegs, walns, wr.x, wr.y, WALLS ? wflags + (size_t)pi * kGlobalRowCap : nullptr);
    } else {
        __shared__ int2 sCD[kLdsRowCap];
        __shared__ uint8_t sT[kLdsRowCap];
        ydrop_body<false, PROF, WALLS>(pr, &outs[pi], tc, qc, O, E, Y, sCD, sT, kLdsRowCap, &sh, arena, arena_bytes, arena_next,
                                       blk_bytes, rowdir, snaps, wsegs, walns, wr.x, wr.y);
    }
}

// walls: the gap-free runs of the earlier alignments (wsegs), per alignment its runs [x, y) (walns), per problem the alignments
// [x, y) of its unit (wref, indexed like probs), with the HBM ring n x kGlobalRowCap zeroed flag bytes (wflags); all nullptr without walls
void launch_ydrop(bool global_rows, const DpProb *probs, DpOut *outs, int n, const PairPtrs *pairs,
                  int O, int E, int Y, int32_t *grows, uint8_t *arena, unsigned long long arena_bytes,
                  unsigned long long *arena_next, unsigned blk_bytes, unsigned long long *rowdir, uint8_t *snaps, hipStream_t s,
                  const void *wsegs, const void *walns, const void *wref, uint8_t *wflags) {
    if (n <= 0) return;
    dim3 g((unsigned)n), b(kYdThreads);
    static const bool prof = getenv("MIBLAST_DP_PROFILE") != nullptr;
    const WallSeg *ws = (const WallSeg *)wsegs; const int2 *wa = (const int2 *)walns, *wr = (const int2 *)wref;
    if (wref) {
        if (global_rows) hipLaunchKernelGGL((k_ydrop<true, false, true>), g, b, 0, s, probs, outs, n, pairs, O, E, Y, grows, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps, ws, wa, wr, wflags);
        else hipLaunchKernelGGL((k_ydrop<false, false, true>), g, b, 0, s, probs, outs, n, pairs, O, E, Y, grows, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps, ws, wa, wr, wflags);
    }
    else if (global_rows) hipLaunchKernelGGL((k_ydrop<true, false, false>), g, b, 0, s, probs, outs, n, pairs, O, E, Y, grows, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps, ws, wa, wr, wflags);
    else if (prof) hipLaunchKernelGGL((k_ydrop<false, true, false>), g, b, 0, s, probs, outs, n, pairs, O, E, Y, grows, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps, ws, wa, wr, wflags);
    else hipLaunchKernelGGL((k_ydrop<false, false, false>), g, b, 0, s, probs, outs, n, pairs, O, E, Y, grows, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps, ws, wa, wr, wflags);
}

// ------------------------------------------------------------------------------------------------
// k_ydrop1: the same one-sided Y-drop DP as k_ydrop, ONE wave per piece.  Lane l owns the K adjacent columns
// jb + K l .. jb + K l + K-1 of the window (jb = first column, a multiple of K); C and D of the previous row, the target
// bases of its columns and everything else live in registers -- no LDS, no barriers.  With relays a launch
// holds thousands of pieces, so what counts is instructions per row per CU: one wave x ~200 instructions instead of four
// waves x ~290.  Per row: K cells per lane (scores of 4 bases with one v_perm), thread-local prefixes, two DPP max-scans
// (horizontal gap X = M + rel and running best M, as in k_ydrop), y-drop test with sign-bit arithmetic, ballots for
// {first break, first alive, last alive}, K trace bytes per lane in one store.  When the window's left edge has moved K
// columns the lanes shift by one (wave_shl DPP moves); the target bases entering on the right come from a register
// prefetch two window widths ahead.  Columns outside the previous row's window need no masking (their C is dead, kNeg,
// and their D is below any threshold that can matter again: k_ydrop's note).  Windows wider than 64 K columns make the
// piece overflow (1): the host reruns it with k_ydrop (LDS ring), then with the HBM ring.
template <int K> struct BaseVec;
template <> struct BaseVec<4> { using T = uint32_t; };
template <> struct BaseVec<8> { using T = unsigned long long; };

__device__ __forceinline__ int dpp_shl1(int v, int fill) {            // lane l <- lane l+1, lane 63 <- fill
    return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false);
}

template <int K>
__global__ __launch_bounds__(64) void k_ydrop1(const DpProb *__restrict__ probs, DpOut *__restrict__ outs, int n,
                                               const PairPtrs *__restrict__ pairs, const int O, const int E, const int Y,
                                               uint8_t *__restrict__ arena, const unsigned long long arena_bytes,
                                               unsigned long long *__restrict__ arena_next, const unsigned blk_bytes,
                                               unsigned long long *__restrict__ rowdir, uint8_t *__restrict__ snaps) {
    const int pi = blockIdx.x;
    if (pi >= n) return;
    using TV = typename BaseVec<K>::T;
    constexpr int kCap = 64 * K;
    constexpr unsigned kAll = (1u << K) - 1u;
    const DpProb pr = probs[pi];
    const PairPtrs pp = pairs[pr.pad0];
    const gbytes tc = as_global(pp.tc);
    const gbytes qc = as_global(pr.strand ? pp.qr : pp.qf);
    DpOut *out = &outs[pi];
    const int lane = threadIdx.x & 63;
    const int na = pr.na, nb = pr.nb, dir = pr.dir;
    const int64_t t0 = pr.t0, q0 = pr.q0;
    const int row_lo = pr.row_lo;
    const long long clk0 = clock64();
    const int OE = O + E;
    const int grow = (Y >= O ? (Y - O) / E : 0) + 2;          // a row can outgrow the previous window by at most this
    int overflow = 0;
    int R0 = 0;
    if (Y >= O) { R0 = (Y - O) / E; if (R0 > na) R0 = na; }
    // target bases of the K columns c0 .. c0+K-1 (byte k = column c0 + k); columns beyond the contig are never alive
    auto load_t = [&](int c0) -> TV {
        TV v = 0;
        if (c0 <= na) {
            typedef const TV __attribute__((address_space(1), aligned(1))) *gvec;
            if (dir > 0) v = *(gvec)(tc + (t0 + c0 - 1));
            else {
                const TV w = *(gvec)(tc + (t0 - c0 - (K - 1)));
                if (K == 4) v = (TV)__builtin_bswap32((uint32_t)w); else v = (TV)__builtin_bswap64((unsigned long long)w);
            }
        }
        return v;
    };
    // ---- trace arena bookkeeping (lane 0 does the atomics) ----
    unsigned long long blk_off = 0, chunk_off = 0;
    unsigned blk_used = 0;
    auto arena_take = [&](unsigned nblk) -> unsigned long long {          // returns ~0 when the arena is exhausted
        unsigned long long o1 = 0;
        if (lane == 0) o1 = atomicAdd(arena_next, (unsigned long long)nblk * blk_bytes);
        o1 = uni64(o1);
        return o1 + (unsigned long long)nblk * blk_bytes > arena_bytes ? ~0ull : o1;
    };
    if (R0 + 1 + 2 * K > kCap) overflow = 1;
    if (!overflow) {
        const unsigned long long o1 = arena_take(2);
        if (o1 == ~0ull) overflow = 3;
        else { blk_off = o1; chunk_off = o1 + blk_bytes; }
    }
    unsigned rb_lo = 0, rb_hi = 0, rb_ly = 0;                             // row records buffered 64 at a time (lane = record & 63)
    auto flush_rows = [&](int last_rec) {
        const int r = (last_rec & ~63) + lane;
        if (r <= last_rec) {
            RowInfo ri; ri.off = ((unsigned long long)rb_hi << 32) | rb_lo; ri.ly = rb_ly; ri.pad = 0;
            ((RowInfo *)(arena + chunk_off))[r & (kRowChunk - 1)] = ri;
        }
    };
    int C[K], D[K];
    int jb = 0, LY = 0, RY = R0 + 1, best = 0, bi = 0, bj = 0, rows = 1;
    long long cells = R0 + 1;
    if (!overflow && row_lo == 0) {
        // ---- row 0: C = -(O + jE) while within ydrop of 0, every cell reached by a horizontal gap from the origin
        uint32_t tb0 = 0;                                                  // K 4-bit trace codes, column k in nibble k
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int j = K * lane + k;
            C[k] = j == 0 ? 0 : (j <= R0 ? -(O + j * E) : kNeg);
            D[k] = kNeg;
            const uint32_t b = j == 0 ? 3u : (2u | (j >= 2 ? 8u : 0u));
            tb0 |= b << (4 * k);
        }
        if (K * lane <= R0) __builtin_memcpy(arena + blk_off + (K / 2) * lane, &tb0, K / 2);
        if (lane == 0) { rb_lo = (unsigned)blk_off; rb_hi = (unsigned)(blk_off >> 32); rb_ly = 0; }
        blk_used = ((unsigned)(R0 + K) & ~(unsigned)(K - 1)) >> 1;
    } else if (!overflow) {
        // ---- continuation: the state after row row_lo comes from a snapshot (record 0 of this piece stays unused)
        const uint8_t *sp = snaps + (size_t)pr.init_snap * kSnapBytes;
        const SnapHdr *h = (const SnapHdr *)sp;
        const int *sC = (const int *)(sp + sizeof(SnapHdr)), *sD = sC + kSnapCols;
        LY = uni(h->LY); RY = uni(h->RY); best = uni(h->best); bi = uni(h->bi); bj = uni(h->bj); rows = uni(h->rows);
        cells = (long long)uni64((unsigned long long)h->cells);
        jb = LY & ~(K - 1);
        if (RY - jb + 2 * K > kCap) overflow = 1;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int j = jb + K * lane + k;
            const bool in = j >= LY && j < RY && !overflow;
            C[k] = in ? sC[j - LY] : kNeg;
            D[k] = in ? sD[j - LY] : kNeg;
        }
    }
    if (!overflow && lane == 0) rowdir[pr.row_off] = chunk_off;
    TV tw = load_t(jb + K * lane);                                       // bases of this lane's columns
    TV tfa = load_t(jb + kCap + K * lane), tfb = load_t(jb + 2 * kCap + K * lane);   // the next two window widths
    int tf_used = 0, tf_base = jb + 3 * kCap;                            // lanes of tfa consumed; first column not yet requested
    int qblk0 = 1 + (row_lo & ~255);                                     // first row of the 256-row block held in qv (4 rows per lane)
    auto load_q = [&](int r0) -> unsigned {
        typedef const uint32_t __attribute__((address_space(1), aligned(1))) *gword;
        const int r = r0 + 4 * lane;                                      // rows r .. r+3 (rows beyond nb are never evaluated)
        if (r > nb) return 0x04040404u;
        return dir > 0 ? *(gword)(qc + (q0 + r - 1)) : __builtin_bswap32(*(gword)(qc + (q0 - r - 3)));
    };
    unsigned qv = load_q(qblk0);
    const uint32_t lutv = row_score_lut((unsigned)min(lane, 4));     // lane k holds the packed score row of query base k
    const int laneKE = lane * K * E;
    // everything loaded so far is waited for HERE: a load still in flight at the loop entry would put a wait for all
    // outstanding memory operations (trace stores included) at the top of every row
    asm volatile("" : "+v"(qv), "+v"(tw), "+v"(tfa), "+v"(tfb));
#pragma unroll
    for (int k = 0; k < K; k++) asm volatile("" : "+v"(C[k]), "+v"(D[k]));
    int i = row_lo + 1;
    int stopped = 0, exit_j = 0;
    for (; i <= nb && !overflow; i++) {
        const int rho = i - row_lo;
        // No load may be in flight when the loop turns around: the compiler would otherwise wait for ALL outstanding
        // memory operations -- including the previous row's trace store -- at the top of every row.
        if (i - qblk0 >= 256) { qblk0 += 256; qv = load_q(qblk0); asm volatile("" : "+v"(qv)); }      // (every 256 rows: waited for on the spot)
        const unsigned qword = (unsigned)__builtin_amdgcn_readlane((int)qv, (i - qblk0) >> 2);
        const uint32_t lut = (uint32_t)__builtin_amdgcn_readlane((int)lutv, min((int)((qword >> (8 * ((i - qblk0) & 3))) & 7u), 4));
        // ---- the window's left edge moved K columns or more: shift the lanes
        while (LY - jb >= K) {
#pragma unroll
            for (int k = 0; k < K; k++) { C[k] = dpp_shl1(C[k], kNeg); D[k] = dpp_shl1(D[k], kNeg); }
            const uint32_t in_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tfa, tf_used);
            uint32_t lo = (uint32_t)dpp_shl1((int)(uint32_t)tw, (int)in_lo);
            if (K == 8) {
                const uint32_t in_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)tfa >> 32), tf_used);
                const uint32_t hi = (uint32_t)dpp_shl1((int)(uint32_t)((unsigned long long)tw >> 32), (int)in_hi);
                tw = (TV)(((unsigned long long)hi << 32) | lo);
            } else tw = (TV)lo;
            jb += K;
            if (++tf_used == 64) {                                        // (every 64 K columns: waited for on the spot, see above)
                tfa = tfb; tfb = load_t(tf_base + K * lane); tf_base += kCap; tf_used = 0;
                asm volatile("" : "+v"(tfb));
            }
        }
        // The row can reach at most column RY + grow, but it rarely does: the lanes only have to hold the old window here; a
        // row whose break is not found within the lanes makes the piece overflow AFTER the fact (it is rerun elsewhere).
        if (RY - jb + K > kCap) { overflow = 1; break; }
        const int reach = min(min(na, RY + grow), jb + kCap - 1);
        const int need = reach - jb + 1 + 2 * K;
        const bool new_blk = blk_used + (unsigned)need > blk_bytes;
        const bool new_chunk = (rho & (kRowChunk - 1)) == 0;
        if ((rho & 63) == 0) flush_rows(rho - 1);
        if (new_blk || new_chunk) {
            const unsigned nblk = (new_blk ? 1u : 0u) + (new_chunk ? 1u : 0u);
            unsigned long long o1 = arena_take(nblk);
            if (o1 == ~0ull) { overflow = 3; break; }
            if (new_blk) { blk_off = o1; blk_used = 0; o1 += blk_bytes; }
            if (new_chunk) { chunk_off = o1; if (lane == 0) rowdir[pr.row_off + (unsigned)(rho / kRowChunk)] = chunk_off; }
        }
        if (lane == (rho & 63)) { const unsigned long long ro = blk_off + blk_used; rb_lo = (unsigned)ro; rb_hi = (unsigned)(ro >> 32); rb_ly = (unsigned)jb; }
        // ---- the cells of the row
        const int j0 = jb + K * lane;
        const bool edge = jb + kCap - 1 > na;                             // some column lies beyond the contig: those are dead and count as breaks
        const int kna = na - j0;
        uint32_t sc[K / 4];
        sc[0] = __builtin_amdgcn_perm(0x1c1c1c1cu, lut, (uint32_t)tw & 0x07070707u);
        if (K == 8) sc[K / 4 - 1] = __builtin_amdgcn_perm(0x1c1c1c1cu, lut, (uint32_t)((unsigned long long)tw >> 32) & 0x07070707u);
        const int cpl = dpp_shr1(C[K - 1], kNeg);                         // C of the column left of this lane's first
        int diag[K], Dv[K], X[K], Mm[K];
        bool dex[K];
        int prev = cpl;
#pragma unroll
        for (int k = 0; k < K; k++) {
            diag[k] = prev + (int)((sc[k / 4] >> (8 * (k & 3))) & 0xFFu) - 128;
            const int de = D[k] - E, dn = C[k] - OE;
            Dv[k] = max(de, dn);
            dex[k] = de >= dn;
            const int Mv = max(diag[k], Dv[k]);
            X[k] = Mv + laneKE + k * E;
            Mm[k] = (!edge || k <= kna) ? Mv : kNeg;
            prev = C[k];
        }
        int lp[K], mi[K];                                                 // max X over the lane's columns before k; max M up to and including k
        lp[0] = kNeg2; mi[0] = Mm[0];
#pragma unroll
        for (int k = 1; k < K; k++) { lp[k] = max(lp[k - 1], X[k - 1]); mi[k] = max(mi[k - 1], Mm[k]); }
        const int PX = dpp_scan_max(max(lp[K - 1], X[K - 1]));
        const int PM = dpp_scan_max(mi[K - 1]);
        const int ex = dpp_shr1(PX, kNeg2);                               // max X over every column left of this lane
        const int emY = max(dpp_shr1(PM, kNeg2), best) - Y;               // (running best before this lane's columns) - Y
        const int allm = uni(__builtin_amdgcn_readlane(PM, 63));
        int pex[K], Iv[K], gm[K];
        unsigned dm = 0;                                                  // bit k: column k is dead
        const int ORel = O + laneKE;
#pragma unroll
        for (int k = 0; k < K; k++) {
            pex[k] = max(ex, lp[k]);
            Iv[k] = pex[k] - (ORel + k * E);
            gm[k] = max(Dv[k], Iv[k]);
            const int Cv = max(diag[k], gm[k]);
            int q = (Cv - max(emY, mi[k] - Y)) >> 31;                     // all ones iff below (running best incl. this cell) - Y
            if (edge) q |= (kna - k) >> 31;
            C[k] = (Cv & ~q) | (kNeg & q);
            D[k] = Dv[k];
            dm |= (unsigned)q & (1u << k);
        }
        const int fnext = pex[K - 1] >= X[K - 1] ? 1 : 0;                // the gap into the next lane's first column extends
        const int fprev = dpp_shr1(fnext, 0);
        // ---- first break, first and last alive column
        const int hi = RY - j0;                                           // column k is right of the old window iff k >= hi
        const unsigned am = ~dm & kAll;
        unsigned bm = dm & (kAll << min(max(hi, 0), K));
        if (edge) bm |= kAll << min(max(kna + 1, 0), K);
        bm &= kAll;
        const unsigned long long bl = __ballot(bm != 0u), al = __ballot(am != 0u);
        const int lb = (int)__ffsll((long long)bl) - 1;
        const unsigned vb = (unsigned)__builtin_amdgcn_readlane((int)bm, lb & 63);
        if (!bl) { overflow = 1; break; }                               // every column up to the last lane is still alive
        const int pbrk = K * lb + (__ffs((int)vb) - 1);                   // relative to jb
        const int nvalid = min(pbrk + ((jb + pbrk) <= na ? 1 : 0), kCap);
        int first_alive = -1, last_alive = -1;
        if (al) {
            const int lf = (int)__ffsll((long long)al) - 1, ll = 63 - (int)__clzll((long long)al);
            const unsigned vf = (unsigned)__builtin_amdgcn_readlane((int)am, lf), vl = (unsigned)__builtin_amdgcn_readlane((int)am, ll);
            first_alive = jb + K * lf + (__ffs((int)vf) - 1);
            last_alive = jb + K * ll + (31 - __clz((int)vl));
        }
        if (allm > best) {
            // the first cell of the row that reaches the new best
            unsigned wm = 0;
#pragma unroll
            for (int k = 0; k < K; k++) wm |= ((!edge || k <= kna) && max(diag[k], gm[k]) == allm) ? (1u << k) : 0u;
            const unsigned long long wl = __ballot(wm != 0u);
            const int lw = (int)__ffsll((long long)wl) - 1;
            const unsigned vw = (unsigned)__builtin_amdgcn_readlane((int)wm, lw & 63);
            best = allm; bi = i; bj = jb + K * lw + (__ffs((int)vw) - 1);
        }
        // ---- trace codes of the lane's columns (4 bits each), one store
        if (K * lane < nvalid) {
            uint32_t tb = 0;                                              // K 4-bit codes: K / 2 bytes per lane
#pragma unroll
            for (int k = 0; k < K; k++) {
                const bool iex = k == 0 ? fprev != 0 : pex[k - 1] >= X[k - 1];
                const unsigned src = diag[k] >= gm[k] ? 0u : (Dv[k] >= Iv[k] ? 1u : 2u);   // tie preference diag > D > I
                tb |= (src | (dex[k] ? 4u : 0u) | (iex ? 8u : 0u)) << (4 * k);
            }
            __builtin_memcpy(arena + blk_off + blk_used + (unsigned)((K / 2) * lane), &tb, K / 2);
        }
        blk_used += ((unsigned)(nvalid + K - 1) & ~(unsigned)(K - 1)) >> 1;
        cells += nvalid - (LY - jb);
        rows++;
        if (first_alive < 0) { i++; break; }
        LY = first_alive;
        RY = last_alive + 1;
        if (i == pr.snap_row || i == pr.stop_row || i == pr.snap_row2 || i == pr.snap_row3) {
            // state after row i
            uint8_t *sp = snaps + (size_t)(pr.snap_idx + (i == pr.stop_row ? 1 : i == pr.snap_row ? 0 : i == pr.snap_row2 ? 2 : 3)) * kSnapBytes;
            int *sC = (int *)(sp + sizeof(SnapHdr)), *sD = sC + kSnapCols;
            int lmax = kNeg2, lk = 0;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int j = j0 + k;
                if (j >= LY && j < RY) { sC[j - LY] = C[k]; sD[j - LY] = D[k]; if (C[k] > lmax) { lmax = C[k]; lk = k; } }
            }
            const int wmax = uni(__builtin_amdgcn_readlane(dpp_scan_max(lmax), 63));
            const unsigned long long ml = __ballot(lmax == wmax);
            const int lm = (int)__ffsll((long long)ml) - 1;
            exit_j = jb + K * lm + uni(__builtin_amdgcn_readlane(lk, lm & 63));
            if (lane == 0) {
                SnapHdr *h = (SnapHdr *)sp;
                h->LY = LY; h->RY = RY; h->best = best; h->bi = bi; h->bj = bj; h->row = i; h->rows = rows; h->cells = cells;
                h->valid = 1;
            }
            if (i == pr.stop_row) { stopped = 1; i++; break; }
        }
    }
    if (!overflow) flush_rows(i - 1 - row_lo);
    if (lane == 0) {
        out->best = best; out->bi = bi; out->bj = bj; out->rows = rows;
        out->cells = cells; out->clocks = clock64() - clk0; out->overflow = overflow; out->n_ops = 0; out->stopped = stopped; out->exit_j = exit_j;
    }
}

// ------------------------------------------------------------------------------------------------
// ---- k_ydrop2: the piece evaluator (mb_ydrop2.h)
#include "mb_ydrop2.h"

// The piece evaluator needs about 100 VGPRs when the compiler is left alone: 4 waves per SIMD (k_ydrop2).  k_ydrop2_w5 is the same
// instruction stream held to 96 VGPRs = 5 waves per SIMD.  It used to be launched when the pieces outnumbered the 4 x 1024 wave
// slots of k_ydrop2; since the row bookkeeping became straight-line code the squeezed build spills inside the row loop and loses
// (16 x 1 Mb pairs in one call: 33.5 ms against 31.3 ms, 315 against 357 Gcell/s in the kernel): MIBLAST_DP_WAVES=5 only.  (Before the trace codes were collected as sign bits the evaluator
// took 129 VGPRs, 3 waves per SIMD, and the squeezed build 128.)
__global__ __launch_bounds__(64)
void k_ydrop2(const DpProb *__restrict__ probs, DpOut *__restrict__ outs, int n, const PairPtrs *__restrict__ pairs, const int O,
              const int E, const int Y, uint8_t *__restrict__ arena, const unsigned long long arena_bytes,
              unsigned long long *__restrict__ arena_next, const unsigned blk_bytes, unsigned long long *__restrict__ rowdir,
              uint8_t *snaps, const int *__restrict__ order, const int first, VerifyJob *__restrict__ vjobs, const int stamp, const int force_mod) {
    ydrop2_piece(probs, outs, n, pairs, O, E, Y, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps, order, first, vjobs, stamp, force_mod);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5)))
void k_ydrop2_w5(const DpProb *__restrict__ probs, DpOut *__restrict__ outs, int n, const PairPtrs *__restrict__ pairs, const int O,
                 const int E, const int Y, uint8_t *__restrict__ arena, const unsigned long long arena_bytes,
                 unsigned long long *__restrict__ arena_next, const unsigned blk_bytes, unsigned long long *__restrict__ rowdir,
                 uint8_t *snaps, const int *__restrict__ order, const int first, VerifyJob *__restrict__ vjobs, const int stamp, const int force_mod) {
    ydrop2_piece(probs, outs, n, pairs, O, E, Y, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps, order, first, vjobs, stamp, force_mod);
}

void launch_ydrop1(int K, const DpProb *probs, DpOut *outs, int n, const PairPtrs *pairs, int O, int E, int Y, uint8_t *arena,
                   unsigned long long arena_bytes, unsigned long long *arena_next, unsigned blk_bytes, unsigned long long *rowdir,
                   uint8_t *snaps, const int *order, hipStream_t s, int first, VerifyJob *vjobs, int stamp, int force_mod) {
    if (n <= 0) return;
    dim3 g((unsigned)n), b(64);
    const char *we = getenv("MIBLAST_DP_WAVES");
    const int waves = we ? atoi(we) : 0;
    const bool five = waves == 5;                                       // (round 3: the 4-wave build is the faster one at every size, see above)
    if (K == 2 && five) hipLaunchKernelGGL(k_ydrop2_w5, g, b, 0, s, probs, outs, n, pairs, O, E, Y, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps, order, first, vjobs, stamp, force_mod);
    else if (K == 2) hipLaunchKernelGGL(k_ydrop2, g, b, 0, s, probs, outs, n, pairs, O, E, Y, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps, order, first, vjobs, stamp, force_mod);
    else if (K == 4) hipLaunchKernelGGL((k_ydrop1<4>), g, b, 0, s, probs + first, outs, n, pairs, O, E, Y, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps);
    else hipLaunchKernelGGL((k_ydrop1<8>), g, b, 0, s, probs + first, outs, n, pairs, O, E, Y, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps);
}

#include "mb_trace.h"

void launch_trace_walk(TbWalk *walks, int n, const uint8_t *arena, unsigned long long arena_bytes,
                       const unsigned long long *rowdir, uint32_t *ops, uint32_t *recs, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_trace_walk, dim3((unsigned)n), dim3(64), 0, s, walks, n, arena, arena_bytes, rowdir, ops, recs);
}
void launch_trace_prejoin(const TbWalk *walks, int n, TbJoin *joins, const uint8_t *arena, unsigned long long arena_bytes,
                          const unsigned long long *rowdir, uint32_t *ops, const uint32_t *recs, int poison, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_trace_prejoin, dim3((unsigned)n), dim3(64), 0, s, walks, n, joins, arena, arena_bytes, rowdir, ops, recs, poison);
}
void launch_trace_join(TbSide *sides, int n, const TbWalk *walks, TbSeg *segs, const uint8_t *arena, unsigned long long arena_bytes,
                       const unsigned long long *rowdir, uint32_t *ops, const uint32_t *recs, const TbJoin *joins, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_trace_join, dim3((unsigned)n), dim3(64), 0, s, sides, n, walks, segs, arena, arena_bytes, rowdir, ops, recs, joins);
}

#include "mb_verify.h"

void launch_verify(const VerifyJob *jobs, VerifyOut *res, int n, const uint8_t *snaps, int Y, int E, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_verify, dim3((unsigned)n), dim3(256), 0, s, jobs, res, n, snaps, Y, E);
}

// the segments of all sides back to back in walk order (each walk was given worst-case room, most of it unused)
__global__ __launch_bounds__(256) void k_pack_segs(const TbSeg *__restrict__ segs, const unsigned long long *__restrict__ dst, int n,
                                                   const uint32_t *__restrict__ ops, uint32_t *__restrict__ packed) {
    const int slot = blockIdx.x;
    if (slot >= n) return;
    const TbSeg sg = segs[slot];
    const uint32_t *src = ops + sg.src;
    const unsigned long long d0 = dst[slot];
    for (int x = threadIdx.x; x < sg.n_runs; x += blockDim.x) {
        uint32_t v = src[x];
        if (x == 0) v -= (uint32_t)sg.first_sub << 2;             // spliced in the middle of a run
        packed[d0 + (unsigned)x] = v;
    }
}

void launch_pack_segs(const TbSeg *segs, const unsigned long long *dst, int n, const uint32_t *ops, uint32_t *packed, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pack_segs, dim3((unsigned)n), dim3(256), 0, s, segs, dst, n, ops, packed);
}

}  // namespace mb

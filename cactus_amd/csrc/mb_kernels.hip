// mb_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the MI355X-native blast phase.
//
// Stage map (SURVEY.md section 8a rows a6..a10, Appendix A.10 rules):
//   k_revcomp            '-' strand of the query set (A.1)
//   k_index_words/_scatter + scan kernels   target seed position table, CSR over 2^24 words (A.3, row a6)
//   k_seed_count/_fill   seed search: 12of19 word + 12 one-transition variants -> (diagonal,q) hit keys (A.4, row a7)
//   k_ungapped           per-diagonal suppression + x-drop extension, HSP emission (A.4/A.5, row a8)
//   k_ydrop<TRACE,...>   one-sided Y-drop affine DP, one wave64 per problem, row ring in LDS (A.7, row a10)
//   k_traceback          walk the stored trace back to the anchor
// All scoring is int32; integer results are bit-identical to the CPU oracle by construction
// (see DESIGN.md "Why the row sweep is exact").
#include "mb_common.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace mb {

// ------------------------------------------------------------------------------------------------
// substitution score on code bytes: HOXD70, N (code 4) scores -100 against anything (A.2)
__device__ __forceinline__ int sub_score(unsigned a, unsigned b) {
    unsigned x = a & 7u, y = b & 7u;
    if ((x | y) & 4u) return -100;
    unsigned d = x ^ y;
    bool at = (x == 0u) || (x == 3u);
    if (d == 0u) return at ? 91 : 100;
    if (d == 2u) return -31;
    if (d == 1u) return -114;
    return at ? -123 : -125;
}

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

// ------------------------------------------------------------------------------------------------
// seed index
__global__ void k_index_words(const uint8_t *__restrict__ codes, int64_t n, int step, uint32_t *__restrict__ words,
                              int64_t n_slots, uint32_t *__restrict__ counts) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    int64_t p = s * step;
    uint32_t w = 0xFFFFFFFFu;
    if (p + kSeedSpan <= n) {
        uint32_t ww;
        if (window_word(codes, p, ww)) { w = ww; atomicAdd(&counts[ww], 1u); }
    }
    words[s] = w;
}

void launch_index_words(const uint8_t *codes, int64_t n, int step, uint32_t *words, int64_t n_slots, uint32_t *counts,
                        hipStream_t s) {
    if (n_slots <= 0) return;
    hipLaunchKernelGGL(k_index_words, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, s, codes, n, step, words,
                       n_slots, counts);
}

__global__ void k_index_scatter(const uint32_t *__restrict__ words, int64_t n_slots, int step,
                                const uint32_t *__restrict__ offsets, uint32_t *__restrict__ cursor,
                                uint32_t *__restrict__ positions) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    uint32_t w = words[s];
    if (w == 0xFFFFFFFFu) return;
    uint32_t k = atomicAdd(&cursor[w], 1u);
    positions[offsets[w] + k] = (uint32_t)(s * step);
}

void launch_index_scatter(const uint32_t *words, int64_t n_slots, int step, const uint32_t *offsets, uint32_t *cursor,
                          uint32_t *positions, hipStream_t s) {
    if (n_slots <= 0) return;
    hipLaunchKernelGGL(k_index_scatter, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, s, words, n_slots, step,
                       offsets, cursor, positions);
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of u32 (three launches: block totals, scan of totals, apply)
constexpr int kScanBlock = 256;
constexpr int kScanPerThread = 8;
constexpr int kScanTile = kScanBlock * kScanPerThread;    // 2048

__global__ void k_block_sums(const uint32_t *__restrict__ in, int64_t n, unsigned long long *__restrict__ bsum) {
    __shared__ unsigned long long red[kScanBlock / 64];
    int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPerThread;
    unsigned long long v = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; k++) if (base + k < n) v += in[base + k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int k = 0; k < kScanBlock / 64; k++) t += red[k];
        bsum[blockIdx.x] = t;
    }
}

__global__ void k_scan_bsums(unsigned long long *bsum, int64_t nb) {
    // single block of 1024 threads; serial over chunks with a running carry
    __shared__ unsigned long long tmp[1024];
    __shared__ unsigned long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        int64_t i = base + threadIdx.x;
        unsigned long long v = (i < nb) ? bsum[i] : 0;
        tmp[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            unsigned long long add = (threadIdx.x >= (unsigned)o) ? tmp[threadIdx.x - o] : 0;
            __syncthreads();
            tmp[threadIdx.x] += add;
            __syncthreads();
        }
        unsigned long long incl = tmp[threadIdx.x], carry = carry_s;
        if (i < nb) bsum[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[nb] = carry_s;
}

__global__ void k_scan_apply(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int64_t n,
                             const unsigned long long *__restrict__ bsum) {
    __shared__ uint32_t wsum[kScanBlock / 64];
    int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanPerThread;
    uint32_t v[kScanPerThread];
    uint32_t tsum = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; k++) { v[k] = (base + k < n) ? in[base + k] : 0u; tsum += v[k]; }
    // inclusive scan of tsum across the wave
    uint32_t incl = tsum;
    int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < (int)(threadIdx.x >> 6); k++) woff += wsum[k];
    uint32_t run = (uint32_t)bsum[blockIdx.x] + woff + incl - tsum;
#pragma unroll
    for (int k = 0; k < kScanPerThread; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
}

void launch_block_sums(const uint32_t *in, int64_t n, unsigned long long *block_sums, hipStream_t s) {
    if (n <= 0) return;
    int64_t nb = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_block_sums, dim3((unsigned)nb), dim3(kScanBlock), 0, s, in, n, block_sums);
}

void launch_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, unsigned long long *block_sums, hipStream_t s) {
    if (n <= 0) return;
    int64_t nb = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_block_sums, dim3((unsigned)nb), dim3(kScanBlock), 0, s, in, n, block_sums);
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(1024), 0, s, block_sums, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(kScanBlock), 0, s, in, out, n, block_sums);
}

// ------------------------------------------------------------------------------------------------
// seed search
__device__ __forceinline__ uint32_t variant_word(uint32_t w, int v) {
    // v = 0 exact ; v = 1..12 transition (xor 2) at care position v-1, first care base most significant
    return v == 0 ? w : (w ^ (2u << (2 * (kSeedWeight - v))));
}

__global__ void k_seed_count(const uint8_t *__restrict__ qcodes, int64_t qn, const uint32_t *__restrict__ offsets,
                             int nvar, uint32_t *__restrict__ qcnt) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= qn) return;
    uint32_t cnt = 0, w;
    if (q + kSeedSpan <= qn && window_word(qcodes, q, w)) {
        for (int v = 0; v < nvar; v++) {
            uint32_t wv = variant_word(w, v);
            cnt += offsets[wv + 1] - offsets[wv];
        }
    }
    qcnt[q] = cnt;
}

void launch_seed_count(const uint8_t *qcodes, int64_t qn, const uint32_t *offsets, int transitions, uint32_t *qcnt,
                       hipStream_t s) {
    if (qn <= 0) return;
    hipLaunchKernelGGL(k_seed_count, dim3((unsigned)((qn + 255) / 256)), dim3(256), 0, s, qcodes, qn, offsets,
                       transitions ? 1 + kSeedWeight : 1, qcnt);
}

__global__ void k_seed_fill(const uint8_t *__restrict__ qcodes, int64_t q0, int64_t q1, int64_t qn, int64_t qtot,
                            const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ positions, int nvar,
                            const uint32_t *__restrict__ hit_off, unsigned long long *__restrict__ keys) {
    int64_t q = q0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= q1) return;
    uint32_t w;
    if (!(q + kSeedSpan <= qn && window_word(qcodes, q, w))) return;
    uint32_t o = hit_off[q - q0];
    unsigned long long q_end = (unsigned long long)(q + kSeedSpan);
    for (int v = 0; v < nvar; v++) {
        uint32_t wv = variant_word(w, v);
        uint32_t b0 = offsets[wv], b1 = offsets[wv + 1];
        for (uint32_t k = b0; k < b1; k++) {
            // diagonal d = t_end - q_end = p - q ; stored biased by qtot so it is non-negative
            unsigned long long dq = (unsigned long long)((int64_t)positions[k] - q + qtot);
            keys[o++] = (dq << 32) | q_end;
        }
    }
}

void launch_seed_fill(const uint8_t *qcodes, int64_t q0, int64_t q1, int64_t qtot, const uint32_t *offsets,
                      const uint32_t *positions, int transitions, const uint32_t *hit_off, unsigned long long *keys,
                      hipStream_t s) {
    if (q1 <= q0) return;
    hipLaunchKernelGGL(k_seed_fill, dim3((unsigned)((q1 - q0 + 255) / 256)), dim3(256), 0, s, qcodes, q0, q1, qtot, qtot,
                       offsets, positions, transitions ? 1 + kSeedWeight : 1, hit_off, keys);
}

size_t sort_keys_temp_bytes(int64_t n, int end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)n, 0,
                                   end_bit, (hipStream_t)0);
    return bytes;
}

void sort_keys(void *temp, size_t temp_bytes, unsigned long long *in, unsigned long long *out, int64_t n, int end_bit,
               hipStream_t s) {
    MB_HIP(rocprim::radix_sort_keys(temp, temp_bytes, in, out, (size_t)n, 0, end_bit, s));
}

// ------------------------------------------------------------------------------------------------
// ungapped x-drop extension with exact per-diagonal suppression (A.4, A.5).
// keys are sorted by (diagonal, q_end); the thread owning the first hit of a diagonal run walks
// the run in q order carrying extent[d], exactly the sequential rule "skip iff q_end <= extent[d]".
__global__ void k_ungapped(const unsigned long long *__restrict__ keys, int64_t n_hits, const uint8_t *__restrict__ tc,
                           const uint8_t *__restrict__ qc, int64_t qtot, int32_t *__restrict__ extent, int xdrop, int K,
                           DevHsp *__restrict__ hsps, int64_t hsp_cap, UngappedCounters *__restrict__ ctr) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_ext = 0, n_cols = 0;
    if (i < n_hits) {
        unsigned long long key = keys[i];
        uint32_t dq = (uint32_t)(key >> 32);
        bool head = (i == 0) || ((uint32_t)(keys[i - 1] >> 32) != dq);
        if (head) {
            int32_t ext = extent[dq];
            int64_t k = i;
            while (true) {
                int32_t q_end = (int32_t)(uint32_t)key;
                if (q_end > ext) {
                    int64_t t_end = (int64_t)dq - qtot + q_end;
                    // left: covers the seed, then beyond; separators (0xFF) bound every contig on both sides
                    int run = 0, bestL = 0, bl = 0;
                    for (int kk = 1;; kk++) {
                        unsigned a = tc[t_end - kk], b = qc[q_end - kk];
                        if (a == kSep || b == kSep) break;
                        run += sub_score(a, b);
                        n_cols++;
                        if (run > bestL) { bestL = run; bl = kk; }
                        else if (run < bestL - xdrop) break;
                    }
                    run = 0;
                    int bestR = 0, br = 0;
                    for (int kk = 0;; kk++) {
                        unsigned a = tc[t_end + kk], b = qc[q_end + kk];
                        if (a == kSep || b == kSep) break;
                        run += sub_score(a, b);
                        n_cols++;
                        if (run > bestR) { bestR = run; br = kk + 1; }
                        else if (run < bestR - xdrop) break;
                    }
                    n_ext++;
                    ext = q_end + br;
                    int score = bestL + bestR;
                    if (score >= K) {
                        unsigned long long slot = atomicAdd(&ctr->hsps, 1ull);
                        if ((int64_t)slot < hsp_cap) {
                            DevHsp h;
                            h.t_start = (int32_t)(t_end - bl);
                            h.q_start = q_end - bl;
                            h.len = bl + br;
                            h.score = score;
                            h.seed_t_end = (int32_t)t_end;
                            h.seed_q_end = q_end;
                            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                            for (int kk = 0; kk < h.len; kk++) {
                                unsigned a = tc[h.t_start + kk] & 7u, b = qc[h.q_start + kk] & 7u;
                                if (a == b) { c0 += (a == 0u); c1 += (a == 1u); c2 += (a == 2u); c3 += (a == 3u); }
                            }
                            h.cnt[0] = c0; h.cnt[1] = c1; h.cnt[2] = c2; h.cnt[3] = c3;
                            hsps[slot] = h;
                        }
                    }
                }
                k++;
                if (k >= n_hits) break;
                key = keys[k];
                if ((uint32_t)(key >> 32) != dq) break;
            }
            extent[dq] = ext;
        }
    }
    // one atomic per wave for the counters
    for (int o = 32; o > 0; o >>= 1) { n_ext += __shfl_down(n_ext, o); n_cols += __shfl_down(n_cols, o); }
    if ((threadIdx.x & 63) == 0 && (n_ext | n_cols)) { atomicAdd(&ctr->extended, n_ext); atomicAdd(&ctr->cols, n_cols); }
}

void launch_ungapped(const unsigned long long *keys, int64_t n_hits, const uint8_t *tcodes, const uint8_t *qcodes,
                     int64_t qtot, int32_t *extent, int xdrop, int K, DevHsp *hsps, int64_t hsp_cap,
                     UngappedCounters *ctr, hipStream_t s) {
    if (n_hits <= 0) return;
    hipLaunchKernelGGL(k_ungapped, dim3((unsigned)((n_hits + 255) / 256)), dim3(256), 0, s, keys, n_hits, tcodes, qcodes,
                       qtot, extent, xdrop, K, hsps, hsp_cap, ctr);
}

// ------------------------------------------------------------------------------------------------
// One-sided Y-drop DP (A.7 / A.10 ONE_SIDED).  One wave64 per problem; rows are swept sequentially,
// 64 columns per segment.  The horizontal-gap recurrence becomes a max-plus prefix scan over the
// row; the running `best` that the y-drop test uses becomes a prefix max.  C and D of the previous
// row live in a ring indexed by column (LDS, or HBM for the rare row wider than the LDS ring) and
// are overwritten in place segment by segment (the diagonal input of the next segment's first
// lane is carried in a register).
__device__ __forceinline__ int wave_incl_max(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(v, o); if (lane >= o) v = max(v, t); }
    return v;
}

template <bool TRACE>
__device__ __forceinline__ void ydrop_body(const DpProb &pr, DpOut *out, const uint8_t *__restrict__ tc,
                                           const uint8_t *__restrict__ qc, int O, int E, int Y, int *Crow, int *Drow,
                                           int cap, uint8_t *__restrict__ trace, uint64_t *__restrict__ rowoff,
                                           uint32_t *__restrict__ rowly) {
    const int lane = threadIdx.x & 63;
    const int mask = cap - 1;
    const int na = pr.na, nb = pr.nb, dir = pr.dir;
    const int64_t t0 = pr.t0, q0 = pr.q0;
    // row 0
    int R0 = 0;
    if (Y >= O) { R0 = (Y - O) / E; if (R0 > na) R0 = na; }
    int overflow = 0;
    if (R0 + 2 > cap) overflow = 1;
    if (!overflow) {
        for (int j = lane; j <= R0; j += 64) {
            Crow[j & mask] = (j == 0) ? 0 : -(O + j * E);
            Drow[j & mask] = kNeg;
            if (TRACE) trace[pr.trace_off + j] = (j == 0) ? 3 : (uint8_t)(2 | (j >= 2 ? 8 : 0));
        }
        if (TRACE && lane == 0) { rowoff[pr.row_off] = 0; rowly[pr.row_off] = 0; }
    }
    int LY = 0, RY = R0 + 1;
    int best = 0, bi = 0, bj = 0;
    long long cells = R0 + 1, cells_to_bi = R0 + 1;
    int rows = 1;
    const int last_row = TRACE ? min(nb, pr.stop_row) : nb;
    for (int i = 1; i <= last_row && !overflow; i++) {
        const unsigned bq = qc[dir > 0 ? q0 + i - 1 : q0 - i];
        if (TRACE && lane == 0) { rowoff[pr.row_off + i] = (uint64_t)cells; rowly[pr.row_off + i] = (uint32_t)LY; }
        int carry_cp = kNeg;          // Cprev[base-1]
        int carry_x = kNeg;           // prefix max of M_k + (k-LY)*E over earlier segments
        int carry_iv = kNeg;          // Iv of column base-1
        int row_best = best;
        int first_alive = -1, last_alive = -1;
        bool done = false;
        for (int base = LY; !done; base += 64) {
            if (base + 64 - LY + 1 > cap) { overflow = 1; break; }
            const int j = base + lane;
            const bool active = j <= na;
            const bool inwin = j < RY;
            __builtin_amdgcn_wave_barrier();
            int cp = inwin ? Crow[j & mask] : kNeg;
            int dp = inwin ? Drow[j & mask] : kNeg;
            int cpl = __shfl_up(cp, 1);
            if (lane == 0) cpl = carry_cp;
            carry_cp = __shfl(cp, 63);
            unsigned at = 4u;
            if (active && j >= 1) at = tc[dir > 0 ? t0 + j - 1 : t0 - j];
            int diag = cpl + sub_score(at, bq);
            int dext_v = dp - E, dopn_v = cp - O - E;
            int Dv = max(dext_v, dopn_v);
            int Dext = dext_v >= dopn_v;
            int M = max(diag, Dv);
            int rel = (j - LY) * E;
            int X = M + rel;
            int pin = wave_incl_max(X, lane);
            int pex = __shfl_up(pin, 1);
            if (lane == 0) pex = carry_x; else pex = max(pex, carry_x);
            carry_x = max(carry_x, __shfl(pin, 63));
            int Iv = pex - O - rel;
            int iv_left = __shfl_up(Iv, 1);
            if (lane == 0) iv_left = carry_iv;
            carry_iv = __shfl(Iv, 63);
            int Iext = (Iv == iv_left - E);
            int Cv; int src;
            if (diag >= Dv && diag >= Iv) { Cv = diag; src = 0; }
            else if (Dv >= Iv) { Cv = Dv; src = 1; }
            else { Cv = Iv; src = 2; }
            int cact = active ? Cv : kNeg;
            int pb = wave_incl_max(cact, lane);
            int best_at = max(row_best, pb);
            bool alive = active && (Cv >= best_at - Y);
            unsigned long long brk = __ballot((j >= RY && !alive) || !active);
            bool valid = active;
            if (brk) {
                int fb = __ffsll((long long)brk) - 1;
                valid = active && lane <= fb;
                alive = alive && lane <= fb;
                done = true;
            }
            unsigned long long vmask = __ballot(valid);
            unsigned long long amask = __ballot(alive);
            cells += __popcll(vmask);
            // best update: strict >, first column attaining the row maximum (row-major first)
            int cand = valid ? Cv : kNeg;
            int segmax = cand;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) segmax = max(segmax, __shfl_xor(segmax, o));
            if (segmax > row_best) {
                unsigned long long w = __ballot(valid && Cv == segmax);
                bj = base + __ffsll((long long)w) - 1;
                bi = i;
                row_best = segmax;
            }
            if (valid) {
                Crow[j & mask] = alive ? Cv : kNeg;
                Drow[j & mask] = Dv;
                if (TRACE) trace[pr.trace_off + (uint64_t)(cells - __popcll(vmask)) + (j - base)] =
                    (uint8_t)(src | (Dext << 2) | (Iext << 3));
            }
            if (amask) {
                if (first_alive < 0) first_alive = base + __ffsll((long long)amask) - 1;
                last_alive = base + 63 - __clzll((long long)amask);
            }
        }
        if (overflow) break;
        rows++;
        best = row_best;
        if (bi == i) cells_to_bi = cells;
        if (first_alive < 0) break;
        LY = first_alive;
        RY = last_alive + 1;
    }
    if (lane == 0) {
        out->best = best; out->bi = bi; out->bj = bj; out->rows = rows;
        out->cells = cells; out->cells_to_bi = cells_to_bi; out->overflow = overflow; out->n_ops = 0;
    }
}

template <bool TRACE, bool GLOBAL_ROWS>
__global__ __launch_bounds__(64) void k_ydrop(const DpProb *__restrict__ probs, DpOut *__restrict__ outs, int n,
                                              const uint8_t *__restrict__ tc, const uint8_t *__restrict__ qf,
                                              const uint8_t *__restrict__ qr, int O, int E, int Y, int32_t *grows,
                                              uint8_t *__restrict__ trace, uint64_t *__restrict__ rowoff,
                                              uint32_t *__restrict__ rowly) {
    int pi = blockIdx.x;
    if (pi >= n) return;
    DpProb pr = probs[pi];
    const uint8_t *qc = pr.strand ? qr : qf;
    if (GLOBAL_ROWS) {
        int *C = grows + (size_t)pi * 2 * kGlobalRowCap;
        ydrop_body<TRACE>(pr, &outs[pi], tc, qc, O, E, Y, C, C + kGlobalRowCap, kGlobalRowCap, trace, rowoff, rowly);
    } else {
        __shared__ int sC[kLdsRowCap];
        __shared__ int sD[kLdsRowCap];
        ydrop_body<TRACE>(pr, &outs[pi], tc, qc, O, E, Y, sC, sD, kLdsRowCap, trace, rowoff, rowly);
    }
}

void launch_ydrop(bool trace, bool global_rows, const DpProb *probs, DpOut *outs, int n, const uint8_t *tc,
                  const uint8_t *qf, const uint8_t *qr, int O, int E, int Y, int32_t *grows, uint8_t *tracebuf,
                  uint64_t *rowoff, uint32_t *rowly, hipStream_t s) {
    if (n <= 0) return;
    dim3 g((unsigned)n), b(64);
    if (trace) {
        if (global_rows) hipLaunchKernelGGL((k_ydrop<true, true>), g, b, 0, s, probs, outs, n, tc, qf, qr, O, E, Y, grows, tracebuf, rowoff, rowly);
        else hipLaunchKernelGGL((k_ydrop<true, false>), g, b, 0, s, probs, outs, n, tc, qf, qr, O, E, Y, grows, tracebuf, rowoff, rowly);
    } else {
        if (global_rows) hipLaunchKernelGGL((k_ydrop<false, true>), g, b, 0, s, probs, outs, n, tc, qf, qr, O, E, Y, grows, tracebuf, rowoff, rowly);
        else hipLaunchKernelGGL((k_ydrop<false, false>), g, b, 0, s, probs, outs, n, tc, qf, qr, O, E, Y, grows, tracebuf, rowoff, rowly);
    }
}

// ------------------------------------------------------------------------------------------------
// traceback: one thread per DP side; emits one op byte per alignment column in walk-back order
// (0 aligned pair, 2 query-only, 3 target-only).
__global__ void k_traceback(const DpProb *__restrict__ probs, DpOut *__restrict__ outs, int n,
                            const uint8_t *__restrict__ trace, const uint64_t *__restrict__ rowoff,
                            const uint32_t *__restrict__ rowly, uint8_t *__restrict__ ops) {
    int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= n) return;
    DpProb pr = probs[pi];
    int i = outs[pi].bi, j = outs[pi].bj, state = 0;
    uint8_t *o = ops + pr.ops_off;
    int n_ops = 0;
    while (i > 0 || j > 0) {
        uint8_t tb = trace[pr.trace_off + rowoff[pr.row_off + i] + (uint64_t)(j - (int)rowly[pr.row_off + i])];
        if (state == 0) {
            int src = tb & 3;
            if (src == 0) { o[n_ops++] = 0; i--; j--; }
            else if (src == 1) state = 1;
            else if (src == 2) state = 2;
            else break;
        } else if (state == 1) {
            o[n_ops++] = 2; if (!(tb & 4)) state = 0; i--;
        } else {
            o[n_ops++] = 3; if (!(tb & 8)) state = 0; j--;
        }
    }
    outs[pi].n_ops = n_ops;
}

void launch_traceback(const DpProb *probs, DpOut *outs, int n, const uint8_t *tracebuf, const uint64_t *rowoff,
                      const uint32_t *rowly, uint8_t *ops, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_traceback, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, probs, outs, n, tracebuf, rowoff, rowly, ops);
}

}  // namespace mb

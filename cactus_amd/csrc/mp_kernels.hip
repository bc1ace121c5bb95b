// mp_kernels.hip -- gfx950 kernels of the chaining stage (rules in DESIGN.md section 11).
//
//   sort_pairs / k_iota / k_gather_*   ordering of alignment records (R-C1, R-C6, R-C7, R-T1): stable LSD radix sorts of
//                                       (key, index) pairs, one pass per sort key, keys re-gathered through the permutation
//   k_chain_dp                          R-C3..R-C5: one workgroup per (query, target, strand) group; the records of a group are
//                                       visited in R-C1 order, the 256 lanes share the scan of the predecessor window
//   k_tile                              R-T2..R-T4: one workgroup per query sequence; per alignment a histogram of the
//                                       per-base cover counters (LDS) gives the median, then the counters go up
//   k_trim_values/ops/finish + scans    R-R1..R-R3: one lane per OP (closed-form candidate cut from prefix sums), maximum per
//                                       record, then bisection for the ops the cuts end in
//
// All of it is integer work on coordinates, scores and cigar ops; nothing here touches sequence bytes.
#include "mp_common.h"

#include <climits>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace mb {

// ------------------------------------------------------------------------------------------------
size_t sort_pairs_temp_bytes(int64_t n, int end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0, end_bit, (hipStream_t)0);
    return bytes;
}

void sort_pairs(void *temp, size_t temp_bytes, const unsigned long long *kin, unsigned long long *kout, const uint32_t *vin,
                uint32_t *vout, int64_t n, int end_bit, hipStream_t s) {
    if (sort_pairs_temp_bytes(n, end_bit) > temp_bytes) throw HipFailure{hipErrorInvalidValue, "sort_pairs: temporary storage too small", __FILE__, __LINE__};
    MB_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, (size_t)n, 0, end_bit, s));
}

__global__ void k_iota(uint32_t *__restrict__ v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}
void launch_iota(uint32_t *v, int64_t n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, n);
}

__global__ void k_gather_u64(const unsigned long long *__restrict__ src, const uint32_t *__restrict__ perm,
                             unsigned long long *__restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
void launch_gather_u64(const unsigned long long *src, const uint32_t *perm, unsigned long long *dst, int64_t n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_gather_u64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, perm, dst, n);
}

// descending order of signed scores as an ascending u64 key
__global__ void k_desc_keys(const long long *__restrict__ v, unsigned long long *__restrict__ key, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = ~((unsigned long long)v[i] ^ 0x8000000000000000ull);
}
void launch_desc_keys(const long long *v, unsigned long long *key, int64_t n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_desc_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, key, n);
}

// records into R-C1 order; beside the records the two coordinates a candidate predecessor is asked for, as plain arrays (the
// window scan of k_chain_dp reads 24 B per candidate, coalesced, instead of the 64-byte record)
__global__ void k_gather_chain(const ChainRec *__restrict__ src, const uint32_t *__restrict__ perm, ChainRec *__restrict__ dst,
                               long long *__restrict__ tqe, long long *__restrict__ tend, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ChainRec r = src[perm[i]];
    dst[i] = r;
    tqe[i] = r.tqe;
    tend[i] = r.same ? r.tte : r.tts;                        // the end a successor's gap is measured from (R-C3)
}
void launch_gather_chain(const ChainRec *src, const uint32_t *perm, ChainRec *dst, long long *tqe, long long *tend, int64_t n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_gather_chain, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, perm, dst, tqe, tend, n);
}

// ------------------------------------------------------------------------------------------------
// Chain DP.  recs are in R-C1 order, group g = [gstart[g], gstart[g+1]); one workgroup per group.  cs_i needs cs_j of
// earlier records of the group, so a group is walked in order -- 64 records (a tile) at a time:
//   phase A  candidates BEFORE the tile (their cs is final): the waves share the tile's records, the 64 lanes of a wave stride
//            over the window of one record.  The window starts at the first j with qs_j >= tqs_i - maxGap - (longest query
//            span of the group): an earlier j ends too far back to satisfy gq <= maxGap.
//   phase B  candidates INSIDE the tile: one wave, lane l keeps record l and its cs in registers; record after record takes
//            the maximum over the lanes before it (one 64-bit max-reduction of value << 6 | 63 - lane) and the phase A result.
// Ties go to the smaller j everywhere (per lane ascending j with a strict ">", smaller j preferred across lanes, phase A
// before phase B): the first maximal j in R-C1 order (R-C5).  Chain scores must stay below 2^57 (checked by the host).
__global__ __launch_bounds__(1024) void k_chain_dp(const ChainRec *__restrict__ recs, const long long *__restrict__ tqe,
                                                   const long long *__restrict__ tend, const uint32_t *__restrict__ gstart,
                                                   const int64_t *__restrict__ glmax, const long long G, const long long gap_open,
                                                   const long long gap_extend, long long *cs, int32_t *__restrict__ pred) {
    const int g = blockIdx.x;
    const int lo = (int)gstart[g], hi = (int)gstart[g + 1];
    if (lo >= hi) return;
    const long long lmax = glmax[g];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (int)(blockDim.x >> 6);
    const bool same = recs[lo].same != 0;
    __shared__ long long s_tqs[64], s_tqe[64], s_tts[64], s_tte[64], s_score[64], s_ext[64];
    __shared__ int s_extj[64];
    for (int t0 = lo; t0 < hi; t0 += 64) {
        const int tn = hi - t0 < 64 ? hi - t0 : 64;
        if (tid < tn) {
            const ChainRec r = recs[t0 + tid];
            s_tqs[tid] = r.tqs; s_tqe[tid] = r.tqe; s_tts[tid] = r.tts; s_tte[tid] = r.tte; s_score[tid] = r.score;
        }
        __syncthreads();
        for (int ii = wave; ii < tn; ii += nw) {             // phase A
            const long long rtqs = s_tqs[ii], rt = same ? s_tts[ii] : s_tte[ii];
            const long long need = rtqs - G - lmax;
            int a = lo, b = t0;
            while (a < b) {
                const int m = (a + b) >> 1;
                if (recs[m].qs < need) a = m + 1; else b = m;
            }
            long long best = 0;
            int bj = INT_MAX;
            for (int j = a + lane; j < t0; j += 64) {
                const long long gq = rtqs - tqe[j];
                const long long gt = same ? rt - tend[j] : tend[j] - rt;
                if (gq >= 0 && gt >= 0 && gq <= G && gt <= G) {
                    const long long val = cs[j] - (gap_open + gap_extend * (gq + gt));
                    if (val > best) { best = val; bj = j; }
                }
            }
            for (int off = 32; off; off >>= 1) {
                const long long ov = __shfl_xor(best, off);
                const int oj = __shfl_xor(bj, off);
                if (ov > best || (ov == best && oj < bj)) { best = ov; bj = oj; }
            }
            if (lane == 0) { s_ext[ii] = best; s_extj[ii] = bj; }
        }
        __syncthreads();
        if (wave == 0) {                                     // phase B
            long long my_cs = 0;
            const long long q_tqe = lane < tn ? s_tqe[lane] : 0;
            const long long q_tend = lane < tn ? (same ? s_tte[lane] : s_tts[lane]) : 0;
            for (int ii = 0; ii < tn; ii++) {
                const long long rtqs = s_tqs[ii], rt = same ? s_tts[ii] : s_tte[ii];
                unsigned long long key = 0;
                if (lane < ii) {
                    const long long gq = rtqs - q_tqe;
                    const long long gt = same ? rt - q_tend : q_tend - rt;
                    if (gq >= 0 && gt >= 0 && gq <= G && gt <= G) {
                        const long long val = my_cs - (gap_open + gap_extend * (gq + gt));
                        if (val > 0) key = ((unsigned long long)val << 6) | (unsigned long long)(63 - lane);
                    }
                }
                for (int off = 32; off; off >>= 1) {
                    const unsigned long long o = __shfl_xor(key, off);
                    key = o > key ? o : key;
                }
                long long best = (long long)(key >> 6);
                int bj = best > 0 ? t0 + 63 - (int)(key & 63ull) : INT_MAX;
                const long long ev = s_ext[ii];
                if (ev > 0 && ev >= best) { best = ev; bj = s_extj[ii]; }
                const long long c = s_score[ii] + best;
                if (lane == ii) my_cs = c;
                if (lane == 0) { cs[t0 + ii] = c; pred[t0 + ii] = best > 0 ? bj : -1; }
            }
        }
        __syncthreads();                                     // the tile's cs are read by the tiles that follow
    }
}

void launch_chain_dp(const ChainRec *recs, const long long *tqe, const long long *tend, const uint32_t *gstart, const int64_t *glmax,
                     int n_groups, long long max_gap, long long gap_open, long long gap_extend, long long *cs, int32_t *pred, int threads,
                     hipStream_t s) {
    threads = threads < 64 ? 64 : threads > 1024 ? 1024 : threads & ~63;
    if (n_groups > 0)
        hipLaunchKernelGGL(k_chain_dp, dim3((unsigned)n_groups), dim3((unsigned)threads), 0, s, recs, tqe, tend, gstart, glmax, max_gap, gap_open,
                           gap_extend, cs, pred);
}

// ------------------------------------------------------------------------------------------------
// Tiling.  recs are grouped by query sequence (block q owns [qstart[q], qstart[q+1])), inside a group in R-T1 order.
// cnt + cnt_off[q] are the u16 cover counters of the sequence.  Wave w takes ops w, w+4, ... of the alignment, its lanes
// the bases of the op (qoff[k] = query bases before op k, in op order; on the '-' strand the ops run down the query).
// The median comes from a histogram of the counters in LDS: bins 0 .. bins-2 are exact, bin bins-1 lumps everything
// above; when the median lies in there it is found by bisection over re-walks of the alignment (never seen with real
// inputs -- it needs half of an alignment's bases covered by thousands of better ones -- and exercised by the tests with
// a tiny histogram).
template <typename F>
__device__ __forceinline__ void for_aligned_bases(const TileRec &r, const uint32_t *__restrict__ ops, const uint32_t *__restrict__ qoff,
                                                  int wave, int lane, F &&f) {
    for (uint32_t o = (uint32_t)wave; o < r.n_ops; o += 4) {
        const uint32_t op = ops[r.ops_off + o];
        if ((op & 7u) > kOpM) continue;
        const uint32_t len = op >> 3;
        const long long q0 = (long long)qoff[r.ops_off + o];
        for (uint32_t j = (uint32_t)lane; j < len; j += 64) f(r.same ? r.qs + q0 + j : r.qe - 1 - q0 - j);
    }
}

__global__ __launch_bounds__(256) void k_tile(const TileRec *__restrict__ recs, const uint32_t *__restrict__ qstart,
                                              const uint64_t *__restrict__ cnt_off, uint16_t *cnt, const uint32_t *__restrict__ ops,
                                              const uint32_t *__restrict__ qoff, const int bins, int32_t *__restrict__ level_out) {
    extern __shared__ unsigned hist[];
    __shared__ unsigned long long s_aligned, s_count;
    __shared__ unsigned s_max;
    __shared__ int s_level;
    const int qi = blockIdx.x;
    uint16_t *c = cnt + cnt_off[qi];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned top = (unsigned)bins - 1u;
    unsigned dirty = top;                                   // highest bin that may be non-zero
    for (uint32_t k = qstart[qi]; k < qstart[qi + 1]; k++) {
        const TileRec r = recs[k];
        for (unsigned b = (unsigned)tid; b <= dirty; b += 256) hist[b] = 0;
        if (tid == 0) { s_aligned = 0; s_max = 0; s_level = -1; }
        __syncthreads();
        unsigned long long al = 0;
        unsigned mx = 0;
        for_aligned_bases(r, ops, qoff, wave, lane, [&](long long b) {
            const unsigned v = c[b];
            atomicAdd(&hist[v < top ? v : top], 1u);
            mx = v > mx ? v : mx;
            al++;
        });
        for (int off = 32; off; off >>= 1) {
            al += __shfl_xor(al, off);
            const unsigned om = __shfl_xor(mx, off);
            mx = om > mx ? om : mx;
        }
        if (lane == 0) { atomicAdd(&s_aligned, al); atomicMax(&s_max, mx); }
        __syncthreads();
        const unsigned long long aligned = s_aligned;
        const unsigned maxl = s_max;
        if (tid == 0) {
            int level = 0;
            if (aligned > 0) {
                level = -1;
                unsigned long long cum = 0;
                const unsigned last = maxl < top ? maxl : top - 1u;      // bins 0 .. top-1 are exact (bins >= 2)
                for (unsigned l = 0; l <= last; l++) {
                    cum += hist[l];
                    if (2 * cum >= aligned) { level = (int)l; break; }
                }
            }
            s_level = level;
        }
        __syncthreads();
        int level = s_level;
        if (level < 0) {                                    // the median is >= top: bisect on #(counter <= mid)
            unsigned a = top, b = maxl;
            while (a < b) {
                const unsigned mid = a + (b - a) / 2;
                if (tid == 0) s_count = 0;
                __syncthreads();
                unsigned long long n_le = 0;
                for_aligned_bases(r, ops, qoff, wave, lane, [&](long long x) { n_le += c[x] <= mid; });
                for (int off = 32; off; off >>= 1) n_le += __shfl_xor(n_le, off);
                if (lane == 0) atomicAdd(&s_count, n_le);
                __syncthreads();
                const unsigned long long tot = s_count;
                __syncthreads();
                if (2 * tot >= aligned) b = mid; else a = mid + 1;
            }
            level = (int)a;
        }
        if (tid == 0) level_out[r.rec] = level + 1;
        for_aligned_bases(r, ops, qoff, wave, lane, [&](long long b) {
            const unsigned v = c[b];
            if (v < 32767u) c[b] = (uint16_t)(v + 1u);
        });
        dirty = maxl < top ? maxl : top;
        __syncthreads();                                    // the counters are read by the next alignment of the sequence
    }
}

void launch_tile(const TileRec *recs, const uint32_t *qstart, const uint64_t *cnt_off, int n_queries, uint16_t *cnt,
                 const uint32_t *ops, const uint32_t *qoff, int hist_bins, int32_t *level, hipStream_t s) {
    if (n_queries > 0)
        hipLaunchKernelGGL(k_tile, dim3((unsigned)n_queries), dim3(256), (size_t)hist_bins * sizeof(unsigned), s, recs, qstart, cnt_off, cnt,
                           ops, qoff, hist_bins, level);
}

// ------------------------------------------------------------------------------------------------
// Tiling without the walk (the default).  The counter an alignment reads at a base is just the number of BETTER-ranked
// alignments whose aligned columns hold that base -- a function of the input, not of an evaluation order -- so every
// alignment can be levelled at once:
//   runs       maximal stretches of aligned query bases of every alignment (host, from the ops; global base coordinates)
//   intervals  the sorted, de-duplicated run boundaries cut the bases into elementary intervals with a constant cover set
//   pieces     a run is cut into the intervals it spans: key (interval << 32 | rank); after a sort by that key the position
//              of a piece inside its interval's segment is its cover count (k_tile_cover), saturated like the u16 counters
//   median     pieces re-sorted by (rank << 15 | cover) with the interval length as weight; a prefix sum of the weights
//              and two bisections per alignment give the smallest cover L with 2 * (bases with cover <= L) >= bases
// Work and memory are proportional to the number of pieces = sum over intervals of their depth; the host falls back to
// k_tile (bounded memory, sequential per sequence) when a pile-up would make that number explode.
size_t sort_keys64_temp_bytes(int64_t n, int end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)n, 0, end_bit,
                                   (hipStream_t)0);
    return bytes;
}
void sort_keys64(void *temp, size_t temp_bytes, const unsigned long long *in, unsigned long long *out, int64_t n, int end_bit, hipStream_t s) {
    if (sort_keys64_temp_bytes(n, end_bit) > temp_bytes) throw HipFailure{hipErrorInvalidValue, "sort_keys64: temporary storage too small", __FILE__, __LINE__};
    MB_HIP(rocprim::radix_sort_keys(temp, temp_bytes, in, out, (size_t)n, 0, end_bit, s));
}
size_t scan_u64_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)n,
                                  rocprim::plus<unsigned long long>(), (hipStream_t)0);
    size_t b2 = 0;
    (void)rocprim::exclusive_scan(nullptr, b2, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, 0ull, (size_t)n,
                                  rocprim::plus<unsigned long long>(), (hipStream_t)0);
    return bytes > b2 ? bytes : b2;
}
void scan_u64(void *temp, size_t temp_bytes, const unsigned long long *in, unsigned long long *out, int64_t n, bool inclusive, hipStream_t s) {
    if (n <= 0) return;
    if (scan_u64_temp_bytes(n) > temp_bytes) throw HipFailure{hipErrorInvalidValue, "scan_u64: temporary storage too small", __FILE__, __LINE__};
    if (inclusive) MB_HIP(rocprim::inclusive_scan(temp, temp_bytes, in, out, (size_t)n, rocprim::plus<unsigned long long>(), s));
    else MB_HIP(rocprim::exclusive_scan(temp, temp_bytes, in, out, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), s));
}

__device__ __forceinline__ int64_t lower_bound_u64(const unsigned long long *__restrict__ v, int64_t n, unsigned long long x) {
    int64_t a = 0, b = n;
    while (a < b) {
        const int64_t m = (a + b) >> 1;
        if (v[m] < x) a = m + 1; else b = m;
    }
    return a;
}

// 1 where a sorted boundary differs from its predecessor
__global__ void k_tile_heads(const unsigned long long *__restrict__ b, int64_t n, unsigned long long *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || b[i] != b[i - 1]) ? 1ull : 0ull;
}
// pos = exclusive scan of the head flags: the distinct boundaries, in order
__global__ void k_tile_unique(const unsigned long long *__restrict__ b, const unsigned long long *__restrict__ flag,
                              const unsigned long long *__restrict__ pos, int64_t n, unsigned long long *__restrict__ u) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) u[pos[i]] = b[i];
}
// the elementary intervals [lo, lo + cnt) a run spans
__global__ void k_tile_span(const unsigned long long *__restrict__ rs, const unsigned long long *__restrict__ re, int64_t n_runs,
                            const unsigned long long *__restrict__ u, int64_t n_u, uint32_t *__restrict__ lo, unsigned long long *__restrict__ cnt) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_runs) return;
    const int64_t a = lower_bound_u64(u, n_u, rs[k]), b = lower_bound_u64(u, n_u, re[k]);
    lo[k] = (uint32_t)a;
    cnt[k] = (unsigned long long)(b - a);
}
__global__ void k_tile_expand(const uint32_t *__restrict__ lo, const unsigned long long *__restrict__ cnt, const unsigned long long *__restrict__ off,
                              const uint32_t *__restrict__ rrank, int64_t n_runs, unsigned long long *__restrict__ key) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_runs) return;
    const unsigned long long base = off[k], e0 = lo[k], r = rrank[k];
    for (unsigned long long j = 0; j < cnt[k]; j++) key[base + j] = ((e0 + j) << 32) | r;
}
// position inside the interval's segment = number of better-ranked alignments on these bases
__global__ void k_tile_cover(const unsigned long long *__restrict__ key, int64_t n_pieces, const unsigned long long *__restrict__ u,
                             unsigned long long *__restrict__ key2, uint32_t *__restrict__ weight) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pieces) return;
    const unsigned long long k = key[i], e = k >> 32, rank = k & 0xffffffffull;
    const int64_t seg = lower_bound_u64(key, i + 1, e << 32);
    unsigned long long cover = (unsigned long long)(i - seg);
    if (cover > 32767ull) cover = 32767ull;
    key2[i] = (rank << 15) | cover;
    weight[i] = (uint32_t)(u[e + 1] - u[e]);
}
__global__ void k_widen(const uint32_t *__restrict__ in, unsigned long long *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
// wsum = inclusive prefix sum of the weights in key2 order; one lane per alignment (rank)
__global__ void k_tile_median(const unsigned long long *__restrict__ key2, const unsigned long long *__restrict__ wsum, int64_t n_pieces,
                              int64_t n_recs, int32_t *__restrict__ level_by_rank) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_recs) return;
    const int64_t a = lower_bound_u64(key2, n_pieces, (unsigned long long)r << 15);
    const int64_t b = lower_bound_u64(key2, n_pieces, (unsigned long long)(r + 1) << 15);
    int32_t level = 0;
    if (b > a) {
        const unsigned long long base = a ? wsum[a - 1] : 0ull, total = wsum[b - 1] - base;
        int64_t x = a, y = b - 1;                              // first t in [a, b) with 2 * (wsum[t] - base) >= total
        while (x < y) {
            const int64_t m = (x + y) >> 1;
            if (2 * (wsum[m] - base) >= total) y = m; else x = m + 1;
        }
        level = (int32_t)(key2[x] & 0x7fffull);
    }
    level_by_rank[r] = level + 1;
}

static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }
void launch_tile_heads(const unsigned long long *b, int64_t n, unsigned long long *flag, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_tile_heads, grid_for(n), dim3(256), 0, s, b, n, flag);
}
void launch_tile_unique(const unsigned long long *b, const unsigned long long *flag, const unsigned long long *pos, int64_t n, unsigned long long *u, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_tile_unique, grid_for(n), dim3(256), 0, s, b, flag, pos, n, u);
}
void launch_tile_span(const unsigned long long *rs, const unsigned long long *re, int64_t n_runs, const unsigned long long *u, int64_t n_u, uint32_t *lo,
                      unsigned long long *cnt, hipStream_t s) {
    if (n_runs > 0) hipLaunchKernelGGL(k_tile_span, grid_for(n_runs), dim3(256), 0, s, rs, re, n_runs, u, n_u, lo, cnt);
}
void launch_tile_expand(const uint32_t *lo, const unsigned long long *cnt, const unsigned long long *off, const uint32_t *rrank, int64_t n_runs,
                        unsigned long long *key, hipStream_t s) {
    if (n_runs > 0) hipLaunchKernelGGL(k_tile_expand, grid_for(n_runs), dim3(256), 0, s, lo, cnt, off, rrank, n_runs, key);
}
void launch_tile_cover(const unsigned long long *key, int64_t n_pieces, const unsigned long long *u, unsigned long long *key2, uint32_t *weight, hipStream_t s) {
    if (n_pieces > 0) hipLaunchKernelGGL(k_tile_cover, grid_for(n_pieces), dim3(256), 0, s, key, n_pieces, u, key2, weight);
}
void launch_widen(const uint32_t *in, unsigned long long *out, int64_t n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_widen, grid_for(n), dim3(256), 0, s, in, out, n);
}
void launch_tile_median(const unsigned long long *key2, const unsigned long long *wsum, int64_t n_pieces, int64_t n_recs, int32_t *level_by_rank, hipStream_t s) {
    if (n_recs > 0) hipLaunchKernelGGL(k_tile_median, grid_for(n_recs), dim3(256), 0, s, key2, wsum, n_pieces, n_recs, level_by_rank);
}

// ------------------------------------------------------------------------------------------------
// Trim by identity, parallel over OPS (a lastz alignment of a whole chunk has a million ops; one lane per record would walk
// them alone).  The longest prefix with matches / columns < num / den ends in the LAST op that has a qualifying column:
// inside a run of non-matching columns the identity only falls, so only the end of the run can qualify; inside a run of
// matches that starts after m matches in c columns, (m + t) / (c + t) < num / den  <=>  t (den - num) < num c - den m, a
// closed form for the last qualifying t.  With prefix sums of columns and matches every op evaluates its own candidate
// (forward, and backward from the record's totals) and the record takes the maximum; the cut positions are then turned
// into ops, remaining lengths and consumed bases by bisection on the same prefix sums.
// P* are exclusive prefix sums over the whole op arena with one extra entry (the total); rec_start is ascending.
__device__ __forceinline__ long long run_candidate(bool is_match, long long len, long long c, long long m, long long num, long long den) {
    if (is_match) {
        const long long rhs = num * c - den * m;
        long long t = 0;
        if (den == num) t = rhs > 0 ? len : 0;
        else if (rhs > 0) { t = (rhs + (den - num) - 1) / (den - num) - 1; if (t > len) t = len; }
        return t >= 1 ? c + t : 0;
    }
    const long long c1 = c + len;
    return m * den < num * c1 ? c1 : 0;
}

__global__ void k_trim_values(const uint32_t *__restrict__ ops, int64_t n_ops, unsigned long long *__restrict__ vc, unsigned long long *__restrict__ vm,
                              unsigned long long *__restrict__ vq, unsigned long long *__restrict__ vt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_ops) return;
    unsigned long long len = 0;
    uint32_t code = kOpD + 1;
    if (i < n_ops) { len = ops[i] >> 3; code = ops[i] & 7u; }
    vc[i] = len;
    vm[i] = (code == kOpEq || code == kOpM) ? len : 0ull;
    vq[i] = (code <= kOpI) ? len : 0ull;                      // = X M I move along the query
    vt[i] = (code <= kOpM || code == kOpD) ? len : 0ull;      // = X M D move along the target
}

__global__ void k_trim_ops(const uint32_t *__restrict__ ops, int64_t n_ops, const unsigned long long *__restrict__ rec_start,
                           const uint32_t *__restrict__ rec_n, int64_t n_recs, const unsigned long long *__restrict__ Pc,
                           const unsigned long long *__restrict__ Pm, const long long num, const long long den,
                           unsigned long long *__restrict__ pre, unsigned long long *__restrict__ suf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ops) return;
    int64_t a = 0, b = n_recs;                                 // the record whose op range holds i, if any
    while (a < b) {
        const int64_t m = (a + b) >> 1;
        if (rec_start[m] <= (unsigned long long)i) a = m + 1; else b = m;
    }
    const int64_t r = a - 1;
    if (r < 0) return;
    const unsigned long long s0 = rec_start[r], e0 = s0 + rec_n[r];
    if ((unsigned long long)i >= e0) return;
    const uint32_t op = ops[i], code = op & 7u;
    const bool is_match = code == kOpEq || code == kOpM;
    const long long len = op >> 3, mlen = is_match ? len : 0;
    const long long c0 = (long long)(Pc[i] - Pc[s0]), m0 = (long long)(Pm[i] - Pm[s0]);
    const long long C = (long long)(Pc[e0] - Pc[s0]), M = (long long)(Pm[e0] - Pm[s0]);
    const long long f = run_candidate(is_match, len, c0, m0, num, den);
    const long long g = run_candidate(is_match, len, C - c0 - len, M - m0 - mlen, num, den);
    if (f > 0) atomicMax(&pre[r], (unsigned long long)f);
    if (g > 0) atomicMax(&suf[r], (unsigned long long)g);
}

__global__ void k_trim_finish(const uint32_t *__restrict__ ops, const unsigned long long *__restrict__ rec_start, const uint32_t *__restrict__ rec_n,
                              int64_t n_recs, const unsigned long long *__restrict__ Pc, const unsigned long long *__restrict__ Pm,
                              const unsigned long long *__restrict__ Pq, const unsigned long long *__restrict__ Pt,
                              const unsigned long long *__restrict__ pre, const unsigned long long *__restrict__ suf, TrimOut *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_recs) return;
    const unsigned long long s0 = rec_start[r], e0 = s0 + rec_n[r];
    TrimOut t{};
    t.cols = (long long)(Pc[e0] - Pc[s0]);
    t.pre = (long long)pre[r];
    t.suf = (long long)suf[r];
    if (t.pre + t.suf < t.cols) {
        const long long M = (long long)(Pm[e0] - Pm[s0]), Q = (long long)(Pq[e0] - Pq[s0]), T = (long long)(Pt[e0] - Pt[s0]);
        // the op that holds column x (0-based, within the record): the last op whose first column is <= x
        auto op_of_column = [&](long long x) {
            unsigned long long a = s0, b = e0 - 1;
            while (a < b) {
                const unsigned long long m = (a + b + 1) >> 1;
                if ((long long)(Pc[m] - Pc[s0]) <= x) a = m; else b = m - 1;
            }
            return a;
        };
        const unsigned long long kf = op_of_column(t.pre);
        {
            const uint32_t code = ops[kf] & 7u;
            const long long used = t.pre - (long long)(Pc[kf] - Pc[s0]);
            t.first_op = (uint32_t)(kf - s0);
            t.first_len = (uint32_t)((long long)(ops[kf] >> 3) - used);
            t.qa = (long long)(Pq[kf] - Pq[s0]) + (code != kOpD ? used : 0);
            t.ta = (long long)(Pt[kf] - Pt[s0]) + (code != kOpI ? used : 0);
            t.nm = (long long)(Pm[kf] - Pm[s0]) + ((code == kOpEq || code == kOpM) ? used : 0);      // matches inside the prefix cut
        }
        const unsigned long long kl = op_of_column(t.cols - t.suf - 1);
        {
            const uint32_t code = ops[kl] & 7u;
            const long long kept = (t.cols - t.suf) - (long long)(Pc[kl] - Pc[s0]);
            t.last_op = (uint32_t)(kl - s0);
            t.last_len = (uint32_t)kept;
            t.qb = Q - (long long)(Pq[kl] - Pq[s0]) - (code != kOpD ? kept : 0);
            t.tb = T - (long long)(Pt[kl] - Pt[s0]) - (code != kOpI ? kept : 0);
            const long long mb_ = M - (long long)(Pm[kl] - Pm[s0]) - ((code == kOpEq || code == kOpM) ? kept : 0);
            t.nm = M - t.nm - mb_;
        }
        t.nb = t.cols - t.pre - t.suf;
    }
    out[r] = t;
}

void launch_trim_values(const uint32_t *ops, int64_t n_ops, unsigned long long *vc, unsigned long long *vm, unsigned long long *vq,
                        unsigned long long *vt, hipStream_t s) {
    hipLaunchKernelGGL(k_trim_values, grid_for(n_ops + 1), dim3(256), 0, s, ops, n_ops, vc, vm, vq, vt);
}
void launch_trim_ops(const uint32_t *ops, int64_t n_ops, const unsigned long long *rec_start, const uint32_t *rec_n, int64_t n_recs,
                     const unsigned long long *Pc, const unsigned long long *Pm, long long num, long long den, unsigned long long *pre,
                     unsigned long long *suf, hipStream_t s) {
    if (n_ops > 0 && n_recs > 0) hipLaunchKernelGGL(k_trim_ops, grid_for(n_ops), dim3(256), 0, s, ops, n_ops, rec_start, rec_n, n_recs, Pc, Pm, num, den, pre, suf);
}
void launch_trim_finish(const uint32_t *ops, const unsigned long long *rec_start, const uint32_t *rec_n, int64_t n_recs, const unsigned long long *Pc,
                        const unsigned long long *Pm, const unsigned long long *Pq, const unsigned long long *Pt, const unsigned long long *pre,
                        const unsigned long long *suf, TrimOut *out, hipStream_t s) {
    if (n_recs > 0) hipLaunchKernelGGL(k_trim_finish, grid_for(n_recs), dim3(256), 0, s, ops, rec_start, rec_n, n_recs, Pc, Pm, Pq, Pt, pre, suf, out);
}

}  // namespace mb

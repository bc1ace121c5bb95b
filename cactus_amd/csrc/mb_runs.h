// mb_runs.h -- run heads of the sorted seed hits and anchors of the HSPs (gfx950, wave64): the two kernels around the ungapped
// extension kernels that every kernel choice shares.  Included by mb_kernels.hip inside namespace mb after mb_xdrop.h (and, with
// MB_EMU defined, by the host-side emulation test under tests/emu).
#pragma once

// Heads of the diagonal runs of the sorted hit keys, compacted into two lists: runs of at most kLongRun hits go to
// the lane-per-run kernel, longer ones (busy diagonals of real homology: hundreds to millions of hits, almost all of
// them suppressed) to the wave-per-run kernel.  One atomic pair per 1024-key block; list order is irrelevant.
// The threshold follows the hit density (launch_ungapped): on dense random data a diagonal holds several chance hits and the
// lane-per-run kernel is the cheap one (32: equal cost to 12 there, 4 is 7x slower -- the wave-per-run kernel has a fixed
// cost); on a small or sparse pair the runs longer than a handful are real homology, where one lane walking 32 extensions
// of hundreds of columns each is the critical path of the whole launch.
constexpr int kLongRunMax = 32;

constexpr int kRunClasses = 4;                // lane-per-run lists by run length: 1, 2-3, 4-7, 8..kLongRun (a wave then holds runs of similar length)

constexpr int kHeadsPerThread = 4;            // keys per thread of k_run_heads: 4096 keys per block share one atomic per list (16: 88 -> 132 us per 1.7 x 10^7 keys)

// kLongOnly: the level-synchronous pipeline (mb_ungapped_ux.h) takes the short runs hit by hit and needs the list of the LONG runs only:
// one look-ahead per head and one ballot per key instead of four and five (37 x 10^6 keys: 290 -> ... us).
template <bool kLongOnly>
__device__ __forceinline__ void run_heads_body(const unsigned long long *__restrict__ keys, int64_t n_hits, const int kLongRun,
                                                unsigned *__restrict__ heads, unsigned *__restrict__ n_heads /* [0..3] short classes, [4] long */) {
    // list c of the short classes starts at heads + off(c): class 0 at 0 (<= n runs), class 1 at n (<= n/2), class 2 at 3n/2 (<= n/4),
    // class 3 at 7n/4 (<= n/8); the long-run list at 15n/8 + 8 (<= n/(kLongRun+1) <= n/5).
    // (A returning atomic on one address costs ~7 ns whoever issues it: with one key per thread the five atomics of a 1024-key
    //  block were the kernel's time.)
    __shared__ unsigned cnt[kRunClasses + 1][16 * kHeadsPerThread];
    __shared__ unsigned base[kRunClasses + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int cls[kHeadsPerThread];
    unsigned rank[kHeadsPerThread];
#pragma unroll
    for (int j = 0; j < kHeadsPerThread; j++) {
        const int64_t i = ((int64_t)blockIdx.x * kHeadsPerThread + j) * blockDim.x + threadIdx.x;
        cls[j] = -1;
        if (i < n_hits) {
            const uint32_t d = (uint32_t)(keys[i] >> 32);
            const bool head = (i == 0) || ((uint32_t)(keys[i - 1] >> 32) != d);
            auto same = [&](int k) -> bool { return (i + k < n_hits) && ((uint32_t)(keys[i + k] >> 32) == d); };
            if (head) cls[j] = kLongOnly ? (same(kLongRun) ? kRunClasses : -1) : same(kLongRun) ? kRunClasses : same(7) ? 3 : same(3) ? 2 : same(1) ? 1 : 0;
        }
        rank[j] = 0;
#pragma unroll
        for (int c = kLongOnly ? kRunClasses : 0; c <= kRunClasses; c++) {
            const unsigned long long m = wballot(cls[j] == c);
            if (lane == 0) cnt[c][16 * j + w] = (unsigned)__popcll(m);
            if (cls[j] == c) rank[j] = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        }
    }
    __syncthreads();
    if (threadIdx.x <= kRunClasses && (!kLongOnly || threadIdx.x == kRunClasses)) {
        const int c = threadIdx.x;
        unsigned t = 0;
        for (int k = 0; k < 16 * kHeadsPerThread; k++) { const unsigned v = cnt[c][k]; cnt[c][k] = t; t += v; }
        base[c] = t ? atomicAdd(&n_heads[c], t) : 0u;
    }
    __syncthreads();
    const uint64_t n = (uint64_t)n_hits;
#pragma unroll
    for (int j = 0; j < kHeadsPerThread; j++) {
        if (cls[j] < 0) continue;
        const int c = cls[j];
        const int64_t i = ((int64_t)blockIdx.x * kHeadsPerThread + j) * blockDim.x + threadIdx.x;
        const uint64_t off = c == 0 ? 0 : c == 1 ? n : c == 2 ? n + n / 2 : c == 3 ? n + n / 2 + n / 4 : n + n / 2 + n / 4 + n / 8 + 8;
        heads[off + base[c] + cnt[c][16 * j + w] + rank[j]] = (unsigned)i;
    }
}
__global__ __launch_bounds__(1024) void k_run_heads(const unsigned long long *__restrict__ keys, int64_t n_hits, const int kLongRun,
                                                    unsigned *__restrict__ heads, unsigned *__restrict__ n_heads) {
    run_heads_body<false>(keys, n_hits, kLongRun, heads, n_heads);
}
// The long runs only (what the level-synchronous pipeline asks for).  Round 5: a plain streaming pass -- a key looks kLongRun keys
// ahead first (the same diagonal there: it lies in a run of more than kLongRun hits; next to never true for chance hits), only then
// back for the run's head, and a head takes its slot of the list with an atomic of its own: long runs are the busy diagonals of real
// homology, a few hundred per strand, so the atomics are no traffic at all, while the block-wide counting of run_heads_body (LDS
// counters, two barriers, a ballot per key and class) made this kernel 7 % of the kernel time of a step of 42 small chunk pairs.
__global__ __launch_bounds__(256) void k_run_heads_long(const unsigned long long *__restrict__ keys, int64_t n_hits, const int kLongRun,
                                                        unsigned *__restrict__ heads, unsigned *__restrict__ n_heads) {
    const uint64_t n = (uint64_t)n_hits;
    const uint64_t off = n + n / 2 + n / 4 + n / 8 + 8;                // (where run_heads_body puts the long-run list)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + kLongRun < n_hits; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t d = (uint32_t)(keys[i] >> 32);
        if ((uint32_t)(keys[i + kLongRun] >> 32) != d) continue;
        if (i == 0 || (uint32_t)(keys[i - 1] >> 32) != d) heads[off + atomicAdd(&n_heads[kRunClasses], 1u)] = (unsigned)i;
    }
}

// ---- anchors of the HSPs: the gapped stage starts an alignment in the middle of an HSP's best-scoring window of 31
// columns, the first one on ties.  The host used to scan every column of every HSP for it (25 ms on a 30 Mb x 30 Mb pair at 1.3 %
// divergence, where the HSPs hold 10^8 columns); here a lane walks an HSP, 8 columns per turn: the window gains the scores of
// columns c + 30 .. c + 37 and loses those of c - 1 .. c + 6, both read as one unaligned 8-byte load per sequence.
__global__ __launch_bounds__(256) void k_hsp_anchor(const UnitTab ut, DevHsp *__restrict__ hsps,
                                                     const int64_t hsp_cap, const UngappedCounters *__restrict__ ctr) {
    const unsigned long long n = min((unsigned long long)hsp_cap, ctr->hsps);
    for (unsigned long long s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (unsigned long long)gridDim.x * blockDim.x) {
        const int len = hsps[s].len;
        if (hsps[s].score == -2147483647 - 1) continue;                  // (a candidate the suppression rule dropped)
        int off = len / 2;
        if (len > 31) {
            const UnitRef un = unit_by_id(ut, hsps[s].unit);
            const uint8_t *tp = un.tc + hsps[s].t_start, *qp = un.qc + hsps[s].q_start;
            int sum = 0;
            for (int k0 = 0; k0 < 32; k0 += 8) {                          // the first window: columns 0 .. 30
                const unsigned long long a8 = load8(tp + k0), b8 = load8(qp + k0);
                const uint32_t s_lo = scores4((uint32_t)a8, (uint32_t)b8), s_hi = scores4((uint32_t)(a8 >> 32), (uint32_t)(b8 >> 32));
#pragma unroll
                for (int m = 0; m < 8; m++)
                    if (k0 + m < 31) sum += (int)(((m < 4 ? s_lo : s_hi) >> (8 * (m & 3))) & 0xFFu) - 128;
            }
            int bestsum = sum, bestc = 0;
            for (int cc = 1; cc + 31 <= len; cc += 8) {                   // windows cc .. cc + 7
                const unsigned long long a_in = load8(tp + cc + 30), b_in = load8(qp + cc + 30), a_out = load8(tp + cc - 1), b_out = load8(qp + cc - 1);
                const uint32_t i_lo = scores4((uint32_t)a_in, (uint32_t)b_in), i_hi = scores4((uint32_t)(a_in >> 32), (uint32_t)(b_in >> 32));
                const uint32_t o_lo = scores4((uint32_t)a_out, (uint32_t)b_out), o_hi = scores4((uint32_t)(a_out >> 32), (uint32_t)(b_out >> 32));
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int in = (int)(((m < 4 ? i_lo : i_hi) >> (8 * (m & 3))) & 0xFFu), out = (int)(((m < 4 ? o_lo : o_hi) >> (8 * (m & 3))) & 0xFFu);
                    sum += in - out;
                    const bool better = (cc + m + 31 <= len) & (sum > bestsum);
                    bestsum = better ? sum : bestsum;
                    bestc = better ? cc + m : bestc;
                }
            }
            off = bestc + 15;
        }
        hsps[s].anchor_off = off;
    }
}


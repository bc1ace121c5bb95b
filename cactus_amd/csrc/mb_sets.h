// mb_sets.h -- kernels that work on whole resident sequence sets, all sets of a call in one launch: the '-' strands of a call's query
// sets (k_revcomp_sets) and outgroup trimming between two calls (k_cov_mark, k_cov_edges, k_gather_stretches).  Included by
// mb_kernels.hip inside namespace mb; tests/emu/emu_sets.cpp runs them on the host.

// the '-' strands of all distinct query sets of a call in one launch, separator bytes around each included: set k owns the bytes
// [grid_off, grid_off + span) of dst (span = total + 2 kDevPad rounded up to whole blocks), its strand begins kDevPad bytes in
__global__ __launch_bounds__(256) void k_revcomp_sets(const RcItem *__restrict__ items, const int n_items, uint8_t *__restrict__ dst) {
    const long long g = (long long)blockIdx.x * blockDim.x;
    int k = 0;
    while (k + 1 < n_items && items[k + 1].grid_off <= g) k++;
    const RcItem it = items[k];
    const long long at = g + threadIdx.x;
    const long long pos = at - it.grid_off - kDevPad;
    uint8_t v = kSep;
    if (pos >= 0 && pos < it.total) {
        int lo = 0, hi = it.n_contigs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (it.starts[mid] <= pos) lo = mid; else hi = mid - 1;
        }
        const long long st = it.starts[lo], n = it.lens[lo];
        if (pos < st + n) {
            unsigned b = it.src[st + n - 1 - (pos - st)];
            if ((b & 7u) < 4u) b = (b & 8u) | (3u - (b & 7u));
            v = (uint8_t)b;
        }
    }
    dst[at] = v;
}

// ---- outgroup trimming on the device (/root/reference/src/cactus/paf/local_alignment.py:460-499) -------------------
// Per-base coverage of a query set by the query intervals of a call's alignments, kept on the device between two blast calls: the
// intervals are marked as +1 / -1 in a difference array, its prefix sum is the depth of every base, and the maximal uncovered
// stretches come out as their first and last+1 positions (unordered; the handful of them is sorted on the host).
__global__ __launch_bounds__(256) void k_cov_mark(const long long *__restrict__ spans, const int n, uint32_t *__restrict__ diff) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    atomicAdd(&diff[spans[2 * k]], 1u);
    atomicAdd(&diff[spans[2 * k + 1]], 0xFFFFFFFFu);                     // -1 (the prefix sums never go below zero)
}

// depth[p + 1] = number of intervals over base p (exclusive scan of diff, one entry further on).  All query sets of a trimming call
// in one launch: their depth arrays lie one after the other in shares of whole blocks (CovItem::off_depth), a block finds its set.
__global__ __launch_bounds__(256) void k_cov_edges(const uint32_t *__restrict__ depth_all, const CovItem *__restrict__ items, const int n_items,
                                                    unsigned *__restrict__ n_edges_all, long long *__restrict__ first_all,
                                                    long long *__restrict__ last_all) {
    const long long g = (long long)blockIdx.x * blockDim.x;
    int k = 0;
    while (k + 1 < n_items && items[k + 1].off_depth <= g) k++;
    const CovItem it = items[k];
    const long long total = it.total;
    const long long p = g - it.off_depth + threadIdx.x;
    if (p >= total) return;
    const uint32_t *depth = depth_all + it.off_depth;
    const uint8_t *codes = it.codes;
    auto open = [&](long long x) -> bool { return x >= 0 && x < total && depth[x + 1] == 0u && codes[x] != kSep; };
    if (!open(p)) return;
    if (!open(p - 1)) { const unsigned at = atomicAdd(&n_edges_all[2 * k], 1u); if (at < it.cap) first_all[it.off_edges + at] = p; }
    if (!open(p + 1)) { const unsigned at = atomicAdd(&n_edges_all[2 * k + 1], 1u); if (at < it.cap) last_all[it.off_edges + at] = p + 1; }
}

// The device images of the new sets, all sets of a call in one launch (a set's image = a share of whole blocks of the grid): the
// bases of the kept stretches one after the other with a separator between two stretches (iv = (dst, src, len) triples by dst),
// separator bytes around them, and the contig tables behind the codes (start and length of stretch x = iv[3x] and iv[3x + 2]).
__global__ __launch_bounds__(256) void k_gather_stretches(const GatherItem *__restrict__ items, const int n_items, const long long *__restrict__ iv_all) {
    const long long g = (long long)blockIdx.x * blockDim.x;
    int k = 0;
    while (k + 1 < n_items && items[k + 1].grid_off <= g) k++;
    const GatherItem it = items[k];
    const long long at = g - it.grid_off + threadIdx.x;                  // byte of the image
    if (at >= it.seq_bytes) return;
    const long long *iv = iv_all + it.iv_off;
    if (at < it.n_iv) { it.d_starts[at] = iv[3 * at]; it.d_lens[at] = iv[3 * at + 2]; }
    const long long p = at - kDevPad;
    uint8_t v = kSep;
    if (p >= 0 && p < it.total) {
        int lo = 0, hi = it.n_iv - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (iv[3 * mid] <= p) lo = mid; else hi = mid - 1;
        }
        const long long off = p - iv[3 * lo];
        if (off < iv[3 * lo + 2]) v = it.src[iv[3 * lo + 1] + off];
    }
    it.dst[at] = v;
}


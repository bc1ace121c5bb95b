// TEST INFRASTRUCTURE ONLY: runs cactus_amd/csrc/mb_sets.h -- the '-' strands of a call's query sets and outgroup trimming between
// two calls, all sets of a call in one launch -- on the HOST, one pthread per work-item (see hip/hip_runtime.h), against plain loops:
//   * k_revcomp_sets: every set's contig-wise reverse complement with separator bytes around it, as the one-set rule
//     (complement of A, C, G, T, the soft-mask bit kept, everything else as it is) gives it;
//   * k_cov_mark + prefix sum + k_cov_edges: first / last+1 of every maximal stretch of bases no interval covers, set by set
//     (/root/reference/src/cactus/paf/local_alignment.py:476-488: `paffy to_bed --excludeAligned`);
//   * k_gather_stretches: the device image of the kept stretches (separator bytes, bases, contig tables) equal to the image the
//     upload of the same records would make.
// Nothing of this is shipped or measured.        emu_sets <seed> <n_cases>      exit status 0 iff every case is identical
#define MB_EMU 1
#include <hip/hip_runtime.h>
#undef __launch_bounds__
#define __launch_bounds__(...)

#include <algorithm>
#include <cstdio>
#include <random>

#include "mb_common.h"

namespace mb {
#include "mb_sets.h"
}  // namespace mb

int main(int argc, char **argv) {
    const unsigned seed0 = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const int n_cases = argc > 2 ? atoi(argv[2]) : 4;
    int bad = 0;
    for (int cs = 0; cs < n_cases; cs++) {
        std::mt19937 rng(seed0 * 7919u + (unsigned)cs);
        auto rnd = [&](int n) { return (int)(rng() % (unsigned)n); };
        const int n_sets = 1 + rnd(4);
        struct Set { std::vector<uint8_t> img; std::vector<int64_t> starts, lens; long long total; };
        std::vector<Set> sets((size_t)n_sets);
        for (Set &S : sets) {
            const int nc = 1 + rnd(4);
            long long at = 0;
            std::vector<uint8_t> codes;
            for (int c = 0; c < nc; c++) {
                const int len = (cs % 3 == 0 && c == 0) ? 1 + rnd(3) : 1 + rnd(700);
                if (c) { codes.push_back(mb::kSep); at++; }
                S.starts.push_back(at); S.lens.push_back(len);
                for (int x = 0; x < len; x++) { const int r = rnd(100); codes.push_back((uint8_t)(r < 88 ? rnd(4) : r < 93 ? 4 : 8 | rnd(4))); }
                at += len;
            }
            S.total = at;
            S.img.assign((size_t)at + 2 * mb::kDevPad, mb::kSep);
            std::copy(codes.begin(), codes.end(), S.img.begin() + mb::kDevPad);
        }
        // ---- '-' strands
        {
            std::vector<mb::RcItem> items;
            long long grid = 0;
            for (Set &S : sets) {
                items.push_back(mb::RcItem{S.img.data() + mb::kDevPad, S.starts.data(), S.lens.data(), S.total, grid, (int)S.starts.size(), 0});
                grid += (long long)(((size_t)S.total + 2 * mb::kDevPad + 255) & ~(size_t)255);
            }
            std::vector<uint8_t> dst((size_t)grid + 16, 0x55);
            hipLaunchKernelGGL(mb::k_revcomp_sets, dim3((unsigned)(grid / 256)), dim3(256), 0, nullptr, items.data(), (int)items.size(), dst.data());
            for (size_t k = 0; k < sets.size(); k++) {
                const Set &S = sets[k];
                const uint8_t *src = S.img.data() + mb::kDevPad;
                const long long span = k + 1 < sets.size() ? items[k + 1].grid_off - items[k].grid_off : grid - items[k].grid_off;
                std::vector<uint8_t> want((size_t)span, mb::kSep);
                for (size_t c = 0; c < S.starts.size(); c++)
                    for (int64_t x = 0; x < S.lens[c]; x++) {
                        unsigned v = src[S.starts[c] + S.lens[c] - 1 - x];
                        if ((v & 7u) < 4u) v = (v & 8u) | (3u - (v & 7u));
                        want[(size_t)(mb::kDevPad + S.starts[c] + x)] = (uint8_t)v;
                    }
                if (memcmp(want.data(), dst.data() + items[k].grid_off, (size_t)span) != 0) { printf("case %d: '-' strand of set %zu differs\n", cs, k); bad++; }
            }
        }
        // ---- coverage: intervals, depth, edges
        std::vector<mb::CovItem> cov;
        std::vector<long long> spans;
        std::vector<std::vector<std::pair<long long, long long>>> ivs(sets.size());
        size_t n_depth = 0, n_ed = 0;
        for (size_t k = 0; k < sets.size(); k++) {
            const Set &S = sets[k];
            const int n_iv = cs % 4 == 1 ? 0 : rnd(12);
            for (int x = 0; x < n_iv; x++) {
                const size_t c = (size_t)rnd((int)S.starts.size());
                long long a = rnd((int)S.lens[c] + 1), b = rnd((int)S.lens[c] + 1);
                if (a > b) std::swap(a, b);
                if (a == b) continue;
                ivs[k].push_back({S.starts[c] + a, S.starts[c] + b});
            }
            const unsigned cap = (unsigned)(ivs[k].size() + S.starts.size() + 2);
            cov.push_back(mb::CovItem{S.img.data() + mb::kDevPad, S.total, (long long)n_depth, (long long)n_ed, cap, 0});
            for (auto &iv : ivs[k]) { spans.push_back(iv.first + (long long)n_depth); spans.push_back(iv.second + (long long)n_depth); }
            n_depth += (((size_t)S.total + 2) + 255) & ~(size_t)255;
            n_ed += cap;
        }
        std::vector<uint32_t> diff(n_depth + 2 * sets.size() + 8, 0u), depth(n_depth + 8, 0u);
        std::vector<long long> edges(2 * n_ed + 1, -7);
        unsigned *counts = diff.data() + n_depth;
        if (!spans.empty()) hipLaunchKernelGGL(mb::k_cov_mark, dim3((unsigned)((spans.size() / 2 + 255) / 256)), dim3(256), 0, nullptr, spans.data(), (int)(spans.size() / 2), diff.data());
        { uint32_t run = 0; for (size_t x = 0; x < n_depth; x++) { depth[x] = run; run += diff[x]; } }      // (the product scans with k_scan_*: exclusive prefix sum)
        hipLaunchKernelGGL(mb::k_cov_edges, dim3((unsigned)(n_depth / 256)), dim3(256), 0, nullptr, depth.data(), cov.data(), (int)cov.size(), counts, edges.data(), edges.data() + n_ed);
        std::vector<std::vector<std::pair<long long, long long>>> open_runs(sets.size());
        for (size_t k = 0; k < sets.size(); k++) {
            const Set &S = sets[k];
            const uint8_t *codes = S.img.data() + mb::kDevPad;
            std::vector<int> covered((size_t)S.total + 1, 0);
            for (auto &iv : ivs[k]) for (long long x = iv.first; x < iv.second; x++) covered[(size_t)x]++;
            std::vector<long long> want_first, want_last;
            for (long long x = 0; x < S.total; x++) {
                const bool o = !covered[(size_t)x] && codes[x] != mb::kSep;
                const bool before = x > 0 && !covered[(size_t)(x - 1)] && codes[x - 1] != mb::kSep;
                const bool after = x + 1 < S.total && !covered[(size_t)(x + 1)] && codes[x + 1] != mb::kSep;
                if (o && !before) want_first.push_back(x);
                if (o && !after) want_last.push_back(x + 1);
            }
            std::vector<long long> got_first(edges.begin() + cov[k].off_edges, edges.begin() + cov[k].off_edges + counts[2 * k]);
            std::vector<long long> got_last(edges.begin() + (long)n_ed + cov[k].off_edges, edges.begin() + (long)n_ed + cov[k].off_edges + counts[2 * k + 1]);
            std::sort(got_first.begin(), got_first.end()); std::sort(got_last.begin(), got_last.end());
            if (got_first != want_first || got_last != want_last || counts[2 * k] > cov[k].cap) { printf("case %d: uncovered stretches of set %zu differ (%zu / %zu edges)\n", cs, k, got_first.size(), want_first.size()); bad++; }
            for (size_t x = 0; x < want_first.size(); x++) open_runs[k].push_back({want_first[x], want_last[x]});
        }
        // ---- gather: every uncovered stretch of at least 3 bases becomes a record of the new set
        {
            std::vector<mb::GatherItem> items;
            std::vector<long long> iv_all;
            std::vector<std::vector<uint8_t>> out(sets.size());
            std::vector<std::vector<uint8_t>> want(sets.size());
            std::vector<size_t> live;
            long long grid = 0;
            for (size_t k = 0; k < sets.size(); k++) {
                std::vector<std::pair<long long, long long>> keep;
                for (auto &r : open_runs[k]) if (r.second - r.first >= 3) keep.push_back(r);
                if (keep.empty()) continue;
                long long total = 0;
                for (size_t x = 0; x < keep.size(); x++) total += keep[x].second - keep[x].first + (x ? 1 : 0);
                const size_t nc = keep.size(), seq_bytes = ((size_t)total + 2 * mb::kDevPad + 255) & ~(size_t)255, image = seq_bytes + 2 * nc * sizeof(int64_t);
                out[k].assign(image + 8, 0x33);
                want[k].assign(image, mb::kSep);
                int64_t *ws = (int64_t *)(want[k].data() + seq_bytes), *wl = ws + nc;
                const long long iv_off = (long long)iv_all.size();
                long long at = 0;
                for (size_t x = 0; x < nc; x++) {
                    if (x) at++;
                    const long long len = keep[x].second - keep[x].first;
                    iv_all.push_back(at); iv_all.push_back(keep[x].first); iv_all.push_back(len);
                    memcpy(want[k].data() + mb::kDevPad + at, sets[k].img.data() + mb::kDevPad + keep[x].first, (size_t)len);
                    ws[x] = at; wl[x] = len;
                    at += len;
                }
                items.push_back(mb::GatherItem{sets[k].img.data() + mb::kDevPad, out[k].data(), (int64_t *)(out[k].data() + seq_bytes), (int64_t *)(out[k].data() + seq_bytes) + nc,
                                               grid, (long long)seq_bytes, total, iv_off, (int)nc, 0});
                grid += (long long)seq_bytes;
                live.push_back(k);
            }
            if (!items.empty()) {
                hipLaunchKernelGGL(mb::k_gather_stretches, dim3((unsigned)(grid / 256)), dim3(256), 0, nullptr, items.data(), (int)items.size(), iv_all.data());
                for (size_t k : live)
                    if (memcmp(out[k].data(), want[k].data(), want[k].size()) != 0) { printf("case %d: image of the set cut out of set %zu differs\n", cs, k); bad++; }
            }
        }
    }
    printf("emu_sets: %d cases, %d differences\n", n_cases, bad);
    return bad ? 1 : 0;
}

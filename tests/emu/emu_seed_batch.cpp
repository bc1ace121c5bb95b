// TEST INFRASTRUCTURE ONLY: runs cactus_amd/csrc/mb_seed_batch.h (sparse seed tables of several targets, the seed search over all
// units of a call) on the HOST -- one pthread per work-item (see hip/hip_runtime.h) -- against a plain restatement of SURVEY A.3 /
// A.4: the table of a target = for every word the set of indexed positions, the hits of a unit = for every valid query window, every
// word variant, every position of its bucket.  Nothing of this is shipped or measured.
//   emu_seed_batch <seed> <n_cases>      exit status 0 iff every case is identical
#define MB_EMU 1
#include <hip/hip_runtime.h>
#undef __launch_bounds__
#define __launch_bounds__(...)

#include <algorithm>
#include <cstdio>
#include <map>
#include <random>

#include "mb_common.h"

inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline unsigned long long atomicOr(unsigned long long *p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
// (mb_units.h wants the wave primitives; the kernels under test use none of them)
inline unsigned long long wballot(bool) { abort(); }
inline int wreadlane(int, int) { abort(); }
template <int C, int B> inline int wdpp(int, int) { abort(); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }

namespace mb {
#include "mb_units.h"
#include "mb_seedword.h"
#include "mb_seed_batch.h"
}  // namespace mb

int main(int argc, char **argv) {
    const unsigned seed0 = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const int n_cases = argc > 2 ? atoi(argv[2]) : 3;
    int bad = 0;
    for (int cs = 0; cs < n_cases; cs++) {
        std::mt19937 rng(seed0 * 104729u + (unsigned)cs);
        auto rnd = [&](int n) { return (int)(rng() % (unsigned)n); };
        const int n_targets = 1 + rnd(3), n_pairs = 1 + rnd(4), step = 1 + rnd(3), nvar = cs % 2 ? 1 : 13;
        auto make_seq = [&](int64_t n, std::vector<uint8_t> &buf) {
            buf.assign((size_t)n + 2 * mb::kDevPad + 32, mb::kSep);
            uint8_t *c = buf.data() + mb::kDevPad;
            // a low-complexity alphabet makes words repeat: buckets with several positions, hits in number
            for (int64_t i = 0; i < n; i++) c[i] = (uint8_t)(rnd(10) < 7 ? rnd(2) : rnd(4));
            for (int s = 0; s < 6; s++) c[rnd((int)n)] = 4;                     // N
            for (int s = 0; s < 3; s++) { const int a = rnd((int)n); for (int k = a; k < std::min<int64_t>(n, a + 30); k++) c[k] |= 8; }   // soft-masked stretch
            for (int s = 0, ns = rnd(3); s < ns; s++) c[1 + rnd((int)n - 2)] = mb::kSep;
        };
        std::vector<std::vector<uint8_t>> tbuf((size_t)n_targets), qbuf((size_t)n_pairs * 2);
        std::vector<int64_t> tn((size_t)n_targets), qn((size_t)n_pairs);
        std::vector<mb::BatchTarget> tg((size_t)n_targets);
        int64_t slots = 0, blocks = 0;
        for (int t = 0; t < n_targets; t++) {
            tn[(size_t)t] = 200 + rnd(3000);
            make_seq(tn[(size_t)t], tbuf[(size_t)t]);
            mb::BatchTarget &g = tg[(size_t)t];
            memset(&g, 0, sizeof g);
            g.codes = tbuf[(size_t)t].data() + mb::kDevPad; g.n = tn[(size_t)t]; g.step = step; g.first = rnd(step);
            g.n_slots = g.n > g.first ? (g.n - g.first + step - 1) / step : 0;
            g.slot0 = slots; g.cbase = slots + t; g.blk0 = blocks;
            slots += g.n_slots; blocks += (g.n_slots + 255) / 256;
        }
        const int64_t n_cnt = slots + n_targets;
        std::vector<mb::SeedUnit> units((size_t)n_pairs * 2);
        int64_t d0 = 0, q0 = 0;
        for (int k = 0; k < n_pairs; k++) {
            qn[(size_t)k] = rnd(8) == 0 ? rnd(25) : 100 + rnd(5000);
            const int t = rnd(n_targets);
            for (int strand = 0; strand < 2; strand++) {
                make_seq(std::max<int64_t>(1, qn[(size_t)k]), qbuf[(size_t)(2 * k + strand)]);
                mb::SeedUnit &u = units[(size_t)(2 * k + strand)];
                memset(&u, 0, sizeof u);
                u.tc = tg[(size_t)t].codes; u.qc = qbuf[(size_t)(2 * k + strand)].data() + mb::kDevPad;
                // stretches of the target inside the query, a transition here and there: hits through exact words and through variants
                for (int c = 0, nc = rnd(6); c < nc && qn[(size_t)k] > 200; c++) {
                    const int len = 40 + rnd(120), a = rnd((int)std::max<int64_t>(1, tn[(size_t)t] - len)), b = rnd((int)(qn[(size_t)k] - len));
                    uint8_t *qd = qbuf[(size_t)(2 * k + strand)].data() + mb::kDevPad;
                    for (int x = 0; x < len && a + x < tn[(size_t)t]; x++) { const uint8_t v = tg[(size_t)t].codes[a + x]; qd[b + x] = (v < 4 && rnd(25) == 0) ? (uint8_t)(v ^ 2) : v; }
                }
                if (qn[(size_t)k] == 0) qbuf[(size_t)(2 * k + strand)][mb::kDevPad] = mb::kSep;
                u.qtot = (int32_t)qn[(size_t)k]; u.ttot = (int32_t)tn[(size_t)t]; u.dbase = (uint32_t)d0; u.index = t; u.qpos0 = q0;
                d0 += tn[(size_t)t] + qn[(size_t)k] + 2;
                q0 += (qn[(size_t)k] + mb::kBsTile - 1) / mb::kBsTile * mb::kBsTile;
            }
        }
        const int64_t q_slots = q0;
        // ---- the kernels
        std::vector<unsigned long long> bits((size_t)n_targets * mb::kBxWordsPerTarget, 0ull);
        std::vector<uint32_t> dir((size_t)n_targets * mb::kBxWordsPerTarget, 0u), bsum((size_t)n_targets * mb::kBxDirBlocks, 0u);
        std::vector<uint32_t> words((size_t)slots + 1, 0u), positions((size_t)slots + 1, 0xFFFFFFFFu), cnt((size_t)n_cnt + 1, 0u), starts((size_t)n_cnt + 1, 0u);
        if (blocks) hipLaunchKernelGGL(mb::k_bx_words, dim3((unsigned)blocks), dim3(256), 0, nullptr, tg.data(), n_targets, words.data(), bits.data());
        hipLaunchKernelGGL(mb::k_bx_popc, dim3((unsigned)(n_targets * mb::kBxDirBlocks)), dim3(256), 0, nullptr, bits.data(), bsum.data());
        hipLaunchKernelGGL(mb::k_bx_dir, dim3((unsigned)(n_targets * mb::kBxDirBlocks)), dim3(256), 0, nullptr, bits.data(), bsum.data(), dir.data());
        if (blocks) hipLaunchKernelGGL(mb::k_bx_count, dim3((unsigned)blocks), dim3(256), 0, nullptr, tg.data(), n_targets, words.data(), bits.data(), dir.data(), cnt.data());
        { uint32_t run = 0; for (int64_t i = 0; i < n_cnt; i++) { starts[(size_t)i] = run; run += cnt[(size_t)i]; } }       // (launch_scan_u32 on the device)
        std::fill(cnt.begin(), cnt.end(), 0u);
        if (blocks) hipLaunchKernelGGL(mb::k_bx_scatter, dim3((unsigned)blocks), dim3(256), 0, nullptr, tg.data(), n_targets, words.data(), bits.data(), dir.data(), starts.data(), cnt.data(), positions.data());
        std::vector<uint32_t> qcnt((size_t)q_slots + 1, 0xFFFFFFFFu), hit_off((size_t)q_slots + 1, 0u);
        if (q_slots) hipLaunchKernelGGL(mb::k_bs_count, dim3((unsigned)(q_slots / 256)), dim3(256), 0, nullptr, units.data(), (int)units.size(), tg.data(), bits.data(), dir.data(), starts.data(), nvar, qcnt.data());
        uint64_t total = 0;
        for (int64_t i = 0; i < q_slots; i++) { hit_off[(size_t)i] = (uint32_t)total; total += qcnt[(size_t)i]; }
        std::vector<unsigned long long> keys((size_t)total + 1, ~0ull);
        if (q_slots) hipLaunchKernelGGL(mb::k_bs_fill, dim3((unsigned)(q_slots / 256)), dim3(256), 0, nullptr, units.data(), (int)units.size(), tg.data(), bits.data(), dir.data(), starts.data(), positions.data(), nvar, hit_off.data(), keys.data());
        keys.resize((size_t)total);
        // ---- the rule
        bool ok = true;
        std::vector<std::map<uint32_t, std::vector<uint32_t>>> table((size_t)n_targets);
        for (int t = 0; t < n_targets; t++) {
            const mb::BatchTarget &g = tg[(size_t)t];
            for (int64_t p = g.first; p + mb::kSeedSpan <= g.n; p += step) {
                uint32_t w;
                if (mb::window_word(g.codes, p, w)) table[(size_t)t][w].push_back((uint32_t)p);
            }
            // the device table: every occupied bucket, in rank order, holds exactly these positions
            uint32_t rank = 0;
            for (auto &kv : table[(size_t)t]) {
                const uint32_t w = kv.first;
                if (!((bits[(size_t)t * mb::kBxWordsPerTarget + (w >> 6)] >> (w & 63u)) & 1ull) || mb::bx_rank(bits.data(), dir.data(), t, w) != rank) { ok = false; break; }
                std::vector<uint32_t> got(positions.begin() + starts[(size_t)(g.cbase + rank)], positions.begin() + starts[(size_t)(g.cbase + rank + 1)]);
                std::sort(got.begin(), got.end());
                if (got != kv.second) { ok = false; break; }
                rank++;
            }
            size_t occupied = 0;
            for (size_t x = 0; x < (size_t)mb::kBxWordsPerTarget; x++) occupied += (size_t)__builtin_popcountll(bits[(size_t)t * mb::kBxWordsPerTarget + x]);
            if (occupied != table[(size_t)t].size()) ok = false;
        }
        std::vector<unsigned long long> want;
        size_t at = 0;
        for (size_t u = 0; ok && u < units.size(); u++) {
            const mb::SeedUnit &su = units[u];
            const size_t first = at;
            for (int64_t q = 0; q + mb::kSeedSpan <= su.qtot; q++) {
                uint32_t w;
                if (!mb::window_word(su.qc, q, w)) continue;
                std::vector<unsigned long long> mine;
                for (int v = 0; v < nvar; v++) {
                    auto it = table[(size_t)su.index].find(mb::variant_word(w, v));
                    if (it == table[(size_t)su.index].end()) continue;
                    for (uint32_t p : it->second) mine.push_back(((unsigned long long)((int64_t)su.dbase + (int64_t)p - q + su.qtot) << 32) | (unsigned long long)(q + mb::kSeedSpan));
                }
                // the keys of a query position are one stretch of the buffer, the positions in q order (the order inside the stretch is free)
                if (at + mine.size() > keys.size()) { ok = false; break; }
                std::vector<unsigned long long> got(keys.begin() + (long)at, keys.begin() + (long)(at + mine.size()));
                std::sort(got.begin(), got.end()); std::sort(mine.begin(), mine.end());
                if (got != mine) { ok = false; break; }
                at += mine.size();
            }
            if (ok && hit_off[(size_t)su.qpos0] != first) ok = false;          // the unit's keys start where its first tile says
        }
        if (ok && at != keys.size()) ok = false;
        printf("case %d: %d targets (step %d), %d pairs, %d variants, %lld slots, %llu hits  %s\n", cs, n_targets, step, n_pairs, nvar, (long long)slots,
               (unsigned long long)total, ok ? "ok" : "MISMATCH");
        if (!ok) bad++;
    }
    return bad ? 1 : 0;
}

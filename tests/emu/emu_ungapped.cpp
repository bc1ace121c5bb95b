// TEST INFRASTRUCTURE ONLY: runs cactus_amd/csrc/mb_runs.h, mb_ungapped_lane.h, mb_ungapped_grp.h and mb_ungapped_ux.h (run heads, the ungapped extension kernels
// of the short runs -- "lane", the default, "ux" --, k_ungapped_long for the long ones, anchors) on the
// HOST -- one pthread per work-item, DPP / ballot exchanged through per-wave barriers (see hip/hip_runtime.h) -- against a
// sequential restatement of the rule (oracle/lastz_oracle.c:227-262, :508-521) on random sequence sets with planted homology,
// separators, N bases and busy diagonals.  Nothing of this is shipped or measured.
//   emu_ungapped <seed> <n_cases> [lane|ux|h16]     exit status 0 iff every case is identical (HSP records, extent[], counters)
#define MB_EMU 1
#include <hip/hip_runtime.h>
#undef __launch_bounds__
#define __launch_bounds__(...)

#include <algorithm>
#include <cstdio>
#include <random>

#include "mb_common.h"
#define __builtin_memcpy memcpy

// ---- the wave primitives of the kernel, emulated ----------------------------------------------------------------
static inline unsigned emu_tid() { return emu::t_threadIdx.x; }

template <int CTRL, int BANK>
inline int wdpp(int old, int v) {
    emu::Group *g = emu::g_group;
    const unsigned tid = emu_tid(), w = tid >> 6, lane = tid & 63, rl = lane & 15;
    g->slot[tid] = (unsigned long long)(unsigned)v;
    pthread_barrier_wait(&g->wave[w]);
    int src = -1;
    if (CTRL <= 0xFF) src = (int)((lane & ~3u) | ((unsigned)(CTRL >> (2 * (lane & 3))) & 3u));             // quad_perm
    else if (CTRL >= 0x111 && CTRL <= 0x11F) src = rl >= (unsigned)(CTRL - 0x110) ? (int)lane - (CTRL - 0x110) : -1;   // row_shr:n
    else if (CTRL == 0x141) src = (int)((lane & ~7u) | (7u - (lane & 7u)));                               // row_half_mirror
    else abort();
    const bool bank = (BANK >> (rl >> 2)) & 1;
    const int out = (src >= 0 && bank) ? (int)(unsigned)g->slot[(tid & ~63u) | (unsigned)src] : old;
    pthread_barrier_wait(&g->wave[w]);
    return out;
}
inline unsigned long long wballot(bool p) {
    emu::Group *g = emu::g_group;
    const unsigned tid = emu_tid(), w = tid >> 6;
    g->slot[tid] = p ? 1ull : 0ull;
    pthread_barrier_wait(&g->wave[w]);
    unsigned long long m = 0;
    for (unsigned l = 0; l < 64; l++) m |= g->slot[(tid & ~63u) | l] << l;
    pthread_barrier_wait(&g->wave[w]);
    return m;
}
inline uint32_t wperm(uint32_t hi, uint32_t lo, uint32_t sel) {                                           // v_perm_b32, selectors 0..7
    uint32_t out = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned s = (sel >> (8 * i)) & 0xFFu;
        if (s > 7) abort();
        const uint32_t byte = s < 4 ? (lo >> (8 * s)) & 0xFFu : (hi >> (8 * (s - 4))) & 0xFFu;
        out |= byte << (8 * i);
    }
    return out;
}
inline int wsdot4(uint32_t a, uint32_t b, int c) {
    for (int i = 0; i < 4; i++) c += (int)(int8_t)(a >> (8 * i)) * (int)(int8_t)(b >> (8 * i));
    return c;
}
inline int wreadlane(int v, int l) {
    emu::Group *g = emu::g_group;
    const unsigned tid = emu_tid(), w = tid >> 6;
    g->slot[tid] = (unsigned long long)(unsigned)v;
    pthread_barrier_wait(&g->wave[w]);
    const int out = (int)(unsigned)g->slot[(tid & ~63u) | (unsigned)(l & 63)];
    pthread_barrier_wait(&g->wave[w]);
    return out;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline unsigned wsignin(unsigned acc, int d) { return (acc << 1) | ((unsigned)d >> 31); }
inline int __ffs(int v) { return __builtin_ffs(v); }
using std::max;
using std::min;

// (mb_ungapped_lane.h calls the wave primitives by the names mb_kernels.hip gives them)
inline unsigned long long __ballot(bool p) { return wballot(p); }
#define __builtin_amdgcn_readlane(v, l) wreadlane((v), (l))
template <typename T>
inline T __shfl_down(T v, int o) {
    emu::Group *g = emu::g_group;
    const unsigned tid = emu_tid(), w = tid >> 6, lane = tid & 63;
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    g->slot[tid] = raw;
    pthread_barrier_wait(&g->wave[w]);
    raw = g->slot[(tid & ~63u) | (lane + (unsigned)o < 64 ? lane + (unsigned)o : lane)];
    pthread_barrier_wait(&g->wave[w]);
    T out;
    memcpy(&out, &raw, sizeof(T));
    return out;
}

namespace mb {
inline int uni(int v) { return wreadlane(v, 0); }
inline unsigned long long uni64(unsigned long long v) { return ((unsigned long long)(unsigned)uni((int)(v >> 32)) << 32) | (unsigned)uni((int)(unsigned)v); }
inline int dpp_shr1(int v, int fill) {                                 // lane l <- lane l-1, lane 0 <- fill
    emu::Group *g = emu::g_group;
    const unsigned tid = emu_tid(), w = tid >> 6, lane = tid & 63;
    g->slot[tid] = (unsigned long long)(unsigned)v;
    pthread_barrier_wait(&g->wave[w]);
    const int o = lane ? (int)(unsigned)g->slot[tid - 1] : fill;
    pthread_barrier_wait(&g->wave[w]);
    return o;
}
template <bool kMax>
inline int emu_scan(int v) {                                           // inclusive prefix sum / prefix max over the wave
    emu::Group *g = emu::g_group;
    const unsigned tid = emu_tid(), w = tid >> 6, lane = tid & 63;
    g->slot[tid] = (unsigned long long)(unsigned)v;
    pthread_barrier_wait(&g->wave[w]);
    int acc = v;
    for (unsigned l = 0; l < lane; l++) { const int x = (int)(unsigned)g->slot[(tid & ~63u) | l]; acc = kMax ? std::max(acc, x) : acc + x; }
    pthread_barrier_wait(&g->wave[w]);
    return acc;
}
inline int dpp_scan_add(int v) { return emu_scan<false>(v); }
inline int dpp_scan_max(int v) { return emu_scan<true>(v); }
#include "mb_xdrop.h"
#include "mb_units.h"
#include "mb_runs.h"
#include "mb_ungapped_lane.h"
#include "mb_ungapped_grp.h"
#include "mb_ungapped_ux.h"
#include "mb_seedword.h"
#include "mb_hash16.h"
}  // namespace mb

// ---- the rule, sequentially ----------------------------------------------------------------------------------------
static int score_of(unsigned a, unsigned b) {
    static const int hox[4][4] = {{91, -114, -31, -123}, {-114, 100, -125, -31}, {-31, -125, 100, -114}, {-123, -31, -114, 91}};
    const unsigned x = a & 7u, y = b & 7u;
    if ((x | y) & 4u) return -100;
    return hox[x][y];
}

struct Ref { std::vector<mb::DevHsp> hsps; std::vector<int32_t> extent; unsigned long long extended = 0, cols = 0, crossed = 0; };

static void reference(const std::vector<unsigned long long> &keys, const uint8_t *tc, const uint8_t *qc, int64_t qtot, int xdrop, int K,
                      int /* long_run: the long runs are k_ungapped_long's, the rule is one */, Ref &out) {
    for (size_t i = 0; i < keys.size();) {
        const uint32_t dq = (uint32_t)(keys[i] >> 32);
        int32_t ext = out.extent[dq];
        size_t j = i;
        for (; j < keys.size() && (uint32_t)(keys[j] >> 32) == dq; j++) {
            const int32_t q_end = (int32_t)(uint32_t)keys[j];
            if (q_end <= ext) continue;
            const int64_t t_end = (int64_t)dq - qtot + q_end;
            int run = 0, bestL = 0, bestR = 0, bl = 0, br = 0;
            for (int k = 1;; k++) {
                const unsigned a = tc[t_end - k], b = qc[q_end - k];
                if (a == mb::kSep || b == mb::kSep) break;
                run += score_of(a, b); out.cols++;
                if (run > bestL) { bestL = run; bl = k; } else if (run < bestL - xdrop) break;
            }
            run = 0;
            for (int k = 0;; k++) {
                const unsigned a = tc[t_end + k], b = qc[q_end + k];
                if (a == mb::kSep || b == mb::kSep) break;
                run += score_of(a, b); out.cols++;
                if (run > bestR) { bestR = run; br = k + 1; } else if (run < bestR - xdrop) break;
            }
            out.extended++;
            ext = q_end + br;
            if (bestL + bestR >= K) {
                mb::DevHsp h;
                h.anchor_off = 0; h.unit = 0;                           // (anchor checked separately against anchor_ref)
                h.t_start = (int32_t)(t_end - bl); h.q_start = q_end - bl; h.len = bl + br; h.score = bestL + bestR;
                h.seed_t_end = (int32_t)t_end; h.seed_q_end = q_end;
                for (int c = 0; c < 4; c++) h.cnt[c] = 0;
                for (int c = 0; c < h.len; c++) {
                    const unsigned a = tc[h.t_start + c] & 7u, b = qc[h.q_start + c] & 7u;
                    if (a < 4 && a == b) h.cnt[a]++;
                }
                out.hsps.push_back(h);
            }
        }
        out.extent[dq] = ext;
        i = j;
    }
}

// the rule with lastz's 16-bit diagonal hash (oracle/lastz_oracle.c, diag_hash16): one extent per (t_end - q_end) & 0xFFFF, the hits in
// the order the search generates them -- q_end ascending, then the word variant's rank, then the target position descending
static void reference_h16(const std::vector<unsigned long long> &keys, const uint8_t *tc, const uint8_t *qc, int64_t qtot, int xdrop, int K, Ref &out) {
    struct H { int32_t q_end; int rank; int64_t t_end; };
    std::vector<H> hits;
    for (unsigned long long k : keys) {
        const int32_t q_end = (int32_t)(uint32_t)k;
        const int64_t t_end = (int64_t)(uint32_t)(k >> 32) - qtot + q_end;
        hits.push_back({q_end, mb::h16_variant_rank(tc, qc, t_end, q_end), t_end});
    }
    std::stable_sort(hits.begin(), hits.end(), [](const H &a, const H &b) { return a.q_end != b.q_end ? a.q_end < b.q_end : a.rank != b.rank ? a.rank < b.rank : a.t_end > b.t_end; });
    std::vector<int32_t> ext16(65536, 0);
    std::vector<int64_t> by16(65536, 0);                                 // the diagonal whose extension a class's extent comes from
    for (const H &h : hits) {
        int32_t &ext = ext16[(size_t)((h.t_end - h.q_end) & 0xFFFF)];
        if (h.q_end <= ext) { out.crossed += by16[(size_t)((h.t_end - h.q_end) & 0xFFFF)] != h.t_end - h.q_end; continue; }
        by16[(size_t)((h.t_end - h.q_end) & 0xFFFF)] = h.t_end - h.q_end;
        const int64_t t_end = h.t_end; const int32_t q_end = h.q_end;
        int run = 0, bestL = 0, bestR = 0, bl = 0, br = 0;
        for (int k = 1;; k++) {
            const unsigned a = tc[t_end - k], b = qc[q_end - k];
            if (a == mb::kSep || b == mb::kSep) break;
            run += score_of(a, b); out.cols++;
            if (run > bestL) { bestL = run; bl = k; } else if (run < bestL - xdrop) break;
        }
        run = 0;
        for (int k = 0;; k++) {
            const unsigned a = tc[t_end + k], b = qc[q_end + k];
            if (a == mb::kSep || b == mb::kSep) break;
            run += score_of(a, b); out.cols++;
            if (run > bestR) { bestR = run; br = k + 1; } else if (run < bestR - xdrop) break;
        }
        out.extended++;
        ext = q_end + br;
        if (bestL + bestR >= K) {
            mb::DevHsp hs;
            hs.anchor_off = 0; hs.unit = 0;
            hs.t_start = (int32_t)(t_end - bl); hs.q_start = q_end - bl; hs.len = bl + br; hs.score = bestL + bestR;
            hs.seed_t_end = (int32_t)t_end; hs.seed_q_end = q_end;
            for (int c = 0; c < 4; c++) hs.cnt[c] = 0;
            for (int c = 0; c < hs.len; c++) { const unsigned a = tc[hs.t_start + c] & 7u, b = qc[hs.q_start + c] & 7u; if (a < 4 && a == b) hs.cnt[a]++; }
            out.hsps.push_back(hs);
        }
    }
}

// the anchor rule of SURVEY A.6, column by column: middle of the best-scoring 31-column window, first on ties
static int anchor_ref(const mb::DevHsp &h, const uint8_t *tc, const uint8_t *qc) {
    if (h.len <= 31) return h.len / 2;
    const uint8_t *tp = tc + h.t_start, *qp = qc + h.q_start;
    int sum = 0;
    for (int k = 0; k < 31; k++) sum += score_of(tp[k], qp[k]);
    int bestsum = sum, bestc = 0;
    for (int cc = 1; cc + 31 <= h.len; cc++) {
        sum += score_of(tp[cc + 30], qp[cc + 30]) - score_of(tp[cc - 1], qp[cc - 1]);
        if (sum > bestsum) { bestsum = sum; bestc = cc; }
    }
    return bestc + 15;
}

static bool hsp_less(const mb::DevHsp &a, const mb::DevHsp &b) {
    return a.seed_t_end != b.seed_t_end ? a.seed_t_end < b.seed_t_end : a.seed_q_end < b.seed_q_end;
}

int main(int argc, char **argv) {
    const unsigned seed0 = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const int n_cases = argc > 2 ? atoi(argv[2]) : 4;
    const bool use_ux = argc > 3 && !strcmp(argv[3], "ux");             // the level-synchronous pipeline instead of k_ungapped_grp
    const bool use_lane = argc > 3 && !strcmp(argv[3], "lane");         // the run-per-lane kernel k_ungapped
    const bool use_h16 = argc > 3 && !strcmp(argv[3], "h16");           // lastz's 16-bit diagonal hash (mb_hash16.h): every hit extended, the rule per hash class
    int bad = 0;
    for (int cs = 0; cs < n_cases; cs++) {
        std::mt19937 rng(seed0 * 7919u + (unsigned)cs);
        auto rnd = [&](int n) { return (int)(rng() % (unsigned)n); };
        // one to three seed units (chunk pair x strand) share the launch: unit u owns the diagonals [dbase, dbase + tn + qn + 2)
        const int n_units = cs % 3 == 2 ? 1 : 1 + rnd(3);
        const int xdrop = cs % 3 == 0 ? 910 : cs % 3 == 1 ? 300 + rnd(400) : 1500 + rnd(3000);
        const int K = cs % 2 ? 3000 : 400 + rnd(1500);
        const int long_run = 6 + rnd(27);
        const bool extent_clean = use_h16 || cs % 2 == 0;                   // (odd cases: extents left by an earlier q batch)
        struct Unit { int64_t tn, qn; std::vector<uint8_t> tb, qb; uint8_t *tc, *qc; uint32_t dbase; std::vector<unsigned long long> keys; Ref ref; };
        std::vector<Unit> units((size_t)n_units);
        std::vector<unsigned long long> keys;
        uint32_t dnext = 0;
        for (Unit &u : units) {
            const int64_t tn = u.tn = use_h16 ? 70000 + rnd(30000) : (n_units > 1 ? 1500 : 3000) + rnd(6000), qn = u.qn = (n_units > 1 ? 1500 : 3000) + rnd(6000);      // (h16: target positions 65536 apart exist)
            u.tb.assign((size_t)tn + 2 * mb::kDevPad + 8, mb::kSep); u.qb.assign((size_t)qn + 2 * mb::kDevPad + 8, mb::kSep);
            uint8_t *tc = u.tc = u.tb.data() + mb::kDevPad, *qc = u.qc = u.qb.data() + mb::kDevPad;
            u.dbase = dnext; dnext += (uint32_t)(tn + qn + 2);
            for (int64_t i = 0; i < tn; i++) tc[i] = (uint8_t)rnd(4);
            for (int64_t i = 0; i < qn; i++) qc[i] = (uint8_t)rnd(4);
            // planted homology: stretches of Q copied from T with mutations
            std::vector<std::pair<int64_t, int64_t>> diag_seeds;
            for (int s = 0, ns = 3 + rnd(5); s < ns; s++) {
                const int len = 50 + rnd(cs % 4 == 3 ? 1500 : 400);
                const int64_t t0 = rnd((int)(tn - len)), q0 = rnd((int)(qn - len));
                const int div = 2 + rnd(25);
                for (int c = 0; c < len; c++) qc[q0 + c] = rnd(100) < div ? (uint8_t)rnd(4) : tc[t0 + c];
                for (int h = 0, nhh = 1 + rnd(20); h < nhh; h++) { const int c = 1 + rnd(len - 1); diag_seeds.push_back({t0 + c, q0 + c}); }
            }
            // N bases, soft-mask bits, contig separators
            for (int s = 0; s < 12; s++) { tc[rnd((int)tn)] = 4; qc[rnd((int)qn)] = 4; }
            for (int s = 0; s < 200; s++) { tc[rnd((int)tn)] |= 8; qc[rnd((int)qn)] |= 8; }
            for (int s = 0, ns = rnd(4); s < ns; s++) { tc[1 + rnd((int)tn - 2)] = mb::kSep; qc[1 + rnd((int)qn - 2)] = mb::kSep; }
            // hits: (t_end, q_end) with the base before either end inside a contig; chance hits, hits on the planted diagonals, a busy diagonal
            auto add = [&](int64_t t_end, int64_t q_end) {
                if (t_end < 1 || t_end > tn || q_end < 1 || q_end > qn) return;
                if (tc[t_end - 1] == mb::kSep || qc[q_end - 1] == mb::kSep) return;
                const uint64_t d = (uint64_t)(t_end - q_end + qn);
                u.keys.push_back((d << 32) | (uint64_t)(uint32_t)q_end);
            };
            for (int h = 0, nhh = 300 + rnd(1200); h < nhh; h++) add(1 + rnd((int)tn), 1 + rnd((int)qn));
            for (auto &ds : diag_seeds) add(ds.first, ds.second);
            { const int64_t d0 = rnd((int)tn / 2); for (int h = 0, nhh = rnd(40); h < nhh; h++) { const int q = 1 + rnd((int)std::min(qn, tn - d0) - 1); add(d0 + q, q); } }
            if (use_h16)                                                 // pairs of hits on diagonals 65536 apart: one hash class, the first one's extension may drop the second
                for (int h = 0; h < 80; h++) {
                    const int64_t q = 20 + rnd((int)qn - 60), t = 1 + rnd((int)(tn - 65536 - 1));
                    add(t, q);
                    add(t + 65536, q + rnd(30));                         // same hash class, a little further along: inside the first one's extension or not
                }
            std::sort(u.keys.begin(), u.keys.end());
            u.keys.erase(std::unique(u.keys.begin(), u.keys.end()), u.keys.end());
            u.ref.extent.assign((size_t)(tn + qn + 2), 0);
            if (!extent_clean) for (int s = 0; s < 20; s++) u.ref.extent[(size_t)rnd((int)(tn + qn + 2))] = rnd((int)qn);
            for (unsigned long long k : u.keys) keys.push_back(k + ((unsigned long long)u.dbase << 32));       // the launch's keys: sorted, one stretch per unit
        }
        const int64_t n_hits = (int64_t)keys.size();
        const int64_t ndiag = (int64_t)dnext;
        // every other one-unit case of the pipeline: the hits come sorted by the SCRAMBLED diagonal, as the dense seed stage leaves them
        // (mb_seed_dense.h), and the bit planes follow the same scramble (UxScratch.plane_mul / plane_mask)
        uint32_t plane_mul = 1u, plane_mask = 0xFFFFFFFFu;
        if (use_ux && n_units == 1 && (cs & 2)) {
            int bits = 1; while ((1ll << bits) < ndiag) bits++;
            plane_mul = 0x9E3779B1u; plane_mask = (uint32_t)((1ull << bits) - 1ull);
            std::stable_sort(keys.begin(), keys.end(), [&](unsigned long long a, unsigned long long b) {
                return (((uint32_t)(a >> 32) * plane_mul) & plane_mask) < (((uint32_t)(b >> 32) * plane_mul) & plane_mask);
            });
            units[0].keys = keys;                                        // (dbase = 0: the unit's keys are the launch's)
        }
        const int64_t nplane = plane_mul == 1u ? ndiag : (int64_t)plane_mask + 1;
        std::vector<int32_t> extent0((size_t)ndiag, 0);
        for (Unit &u : units) {
            std::copy(u.ref.extent.begin(), u.ref.extent.end(), extent0.begin() + (long)u.dbase);
            if (use_h16) reference_h16(u.keys, u.tc, u.qc, u.qn, xdrop, K, u.ref);
            else reference(u.keys, u.tc, u.qc, u.qn, xdrop, K, long_run, u.ref);
        }
        std::vector<mb::SeedUnit> tab((size_t)n_units);
        for (int x = 0; x < n_units; x++) {
            memset(&tab[(size_t)x], 0, sizeof(mb::SeedUnit));
            tab[(size_t)x].tc = units[(size_t)x].tc; tab[(size_t)x].qc = units[(size_t)x].qc; tab[(size_t)x].qtot = (int32_t)units[(size_t)x].qn; tab[(size_t)x].ttot = (int32_t)units[(size_t)x].tn;
            tab[(size_t)x].dbase = units[(size_t)x].dbase;
        }
        mb::UnitTab ut; ut.one = tab[0]; ut.tab = tab.data(); ut.n = n_units; ut.ext_mul = 0; ut.ext_mask = 0;
        // class lists as k_run_heads lays them out (every run is "short" here)
        const uint64_t n = (uint64_t)n_hits;
        std::vector<unsigned> heads((size_t)(2 * n + n / 4 + 64), 0u);
        unsigned n_heads[5] = {0, 0, 0, 0, 0};
        for (int64_t i = 0; i < n_hits;) {
            int64_t j = i; while (j < n_hits && (keys[j] >> 32) == (keys[i] >> 32)) j++;
            const int64_t len = j - i;
            const int cls = len > long_run ? 4 : len >= 8 ? 3 : len >= 4 ? 2 : len >= 2 ? 1 : 0;
            const uint64_t off = cls == 0 ? 0 : cls == 1 ? n : cls == 2 ? n + n / 2 : cls == 3 ? n + n / 2 + n / 4 : n + n / 2 + n / 4 + n / 8 + 8;
            heads[off + n_heads[cls]++] = (unsigned)i;
            i = j;
        }
        {   // the lists as k_run_heads itself makes them: the same runs in every list (their order inside a list is free)
            std::vector<unsigned> kheads(heads.size(), 0u);
            unsigned kn[5] = {0, 0, 0, 0, 0};
            if (use_ux) hipLaunchKernelGGL(mb::k_run_heads_long, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n_hits + 1023) / 1024, 3))), dim3(256), 0, nullptr, keys.data(), n_hits, long_run, kheads.data(), kn);      // (few blocks: the stride loop is walked)
            else hipLaunchKernelGGL(mb::k_run_heads, dim3((unsigned)((n_hits + 4095) / 4096)), dim3(1024), 0, nullptr, keys.data(), n_hits, long_run, kheads.data(), kn);
            bool same = true;
            for (int c = use_ux ? 4 : 0; c < 5 && same; c++) {
                const uint64_t off = c == 0 ? 0 : c == 1 ? n : c == 2 ? n + n / 2 : c == 3 ? n + n / 2 + n / 4 : n + n / 2 + n / 4 + n / 8 + 8;
                same = kn[c] == n_heads[c];
                if (same) {
                    std::vector<unsigned> x(kheads.begin() + (long)off, kheads.begin() + (long)off + kn[c]), y(heads.begin() + (long)off, heads.begin() + (long)off + n_heads[c]);
                    std::sort(x.begin(), x.end()); std::sort(y.begin(), y.end());
                    same = x == y;
                }
            }
            if (!same) { printf("case %d: k_run_heads lists differ from the host's  MISMATCH\n", cs); bad++; continue; }
            heads = kheads;
        }
        std::vector<int32_t> extent = extent0;
        // the scrambled one-unit cases keep extent[] in the order of their keys too (UnitTab::ext_mul: a strand in several q batches, round 6):
        // the kernels see the slots permuted, the comparison below reads them back by diagonal
        const bool ext_scrambled = plane_mul != 1u && n_units == 1;
        if (ext_scrambled) {
            extent.assign((size_t)plane_mask + 1, 0);
            for (size_t d = 0; d < extent0.size(); d++) extent[(size_t)(((uint32_t)d * plane_mul) & plane_mask)] = extent0[d];
            ut.ext_mul = plane_mul; ut.ext_mask = plane_mask;
            printf("  extent slots in the order of the scrambled keys\n");
        }
        std::vector<mb::DevHsp> hsps((size_t)n_hits + 8);
        std::vector<mb::UngappedCounters> ctrs((size_t)n_units, mb::UngappedCounters{0, 0, 0});
        mb::UngappedCounters *ctr = ctrs.data();
        const unsigned blocks = 1 + (unsigned)rnd(3);                    // few groups: every group walks many runs
        if (use_lane) {
            hipLaunchKernelGGL(mb::k_ungapped, dim3((unsigned)((n_hits + 255) / 256 + mb::kRunClasses)), dim3(256), 0, nullptr, keys.data(), n_hits, heads.data(), n_heads, ut,
                               extent.data(), xdrop, K, hsps.data(), (int64_t)hsps.size(), ctr);
        } else if (!use_ux && !use_h16) {
            hipLaunchKernelGGL(mb::k_ungapped_grp<5>, dim3(blocks), dim3(256), 0, nullptr, keys.data(), n_hits, heads.data(), n_heads, ut,
                               extent.data(), xdrop, K, hsps.data(), (int64_t)hsps.size(), ctr);
        } else {
            std::vector<unsigned long long> rec((size_t)n_hits + 1, 0xdeadbeefdeadbeefull);
            const unsigned cap = cs % 5 == 4 ? 3u : (unsigned)n_hits;      // (a tiny list: lanes finish their hits themselves)
            std::vector<mb::UxEntry> entries(cap + 1);
            const unsigned n_blk = (unsigned)((n_hits + 255) / 256);
            std::vector<mb::UxEntry> blk_entries((size_t)n_blk * 16 + 1);
            std::vector<unsigned> blk_cnt((size_t)n_blk * 2 + 1, 0u);
            std::vector<uint32_t> bits((size_t)(nplane + 31) / 32 + 1, 0u), dirty((size_t)(nplane + 31) / 32 + 1, 0u);
            unsigned n_entries[2] = {0, 0};
            mb::UxScratch sc; sc.rec = rec.data(); sc.blk_entries = blk_entries.data(); sc.blk_cnt = blk_cnt.data(); sc.n_blk = n_blk; sc.entries = entries.data(); sc.entry_cap = cap; sc.n_entries = n_entries; sc.long_bits = bits.data(); sc.dirty_bits = dirty.data();
            sc.plane_mul = plane_mul; sc.plane_mask = plane_mask;
            std::vector<unsigned> dirty_runs((size_t)n_hits + 1);
            sc.dirty_runs = dirty_runs.data(); sc.dirty_cap = (unsigned)n_hits; sc.extent = extent.data(); sc.extent_live = extent_clean ? 0 : 1;
            const unsigned *heads_long = heads.data() + (n + n / 2 + n / 4 + n / 8 + 8);
            if (use_h16) sc.extent_live = 0;
            if (!use_h16) hipLaunchKernelGGL(mb::k_ux_mark_long, dim3((n_heads[4] + 255) / 256 + 1), dim3(256), 0, nullptr, keys.data(), heads_long, n_heads + 4, sc);
            // every other one-unit case: level 1 from the PACKED strands (k_ux_extend_pk: the extension's 12-byte records, made here
            // by a plain loop) -- hits whose windows hold an N or a separator, or cross an end of a set, take the byte path inside it
            sc.t_px = sc.q_px = nullptr; sc.t_n = sc.q_n = 0;
            std::vector<uint32_t> tpx, qpx;
            const bool use_pk = n_units == 1 && !use_h16 && (cs % 2 == 1 || cs % 5 == 2);
            if (use_pk) {
                auto pack = [](const uint8_t *codes, int64_t nb, std::vector<uint32_t> &px) {      // (the layout k_pack2bit_mask writes: mb_seed_dense.h)
                    const size_t recs = 2 * (size_t)((nb + 63) / 64 + 2);
                    px.assign(3 * recs, 0u);
                    for (int64_t i = 0; i < (int64_t)recs * 32; i++) {
                        const unsigned c = i < nb ? codes[i] : 0xFFu;
                        const unsigned long long two = (unsigned long long)(c & 3u) << (62 - 2 * (i & 31));
                        uint32_t *r = px.data() + 3 * (i >> 5);
                        r[0] |= (uint32_t)two; r[1] |= (uint32_t)(two >> 32);
                        if (c & 0x84u) r[2] |= 1u << (31 - (i & 31));
                    }
                };
                pack(units[0].tc, units[0].tn, tpx); pack(units[0].qc, units[0].qn, qpx);
                sc.t_px = tpx.data(); sc.q_px = qpx.data(); sc.t_n = units[0].tn; sc.q_n = units[0].qn;
                hipLaunchKernelGGL(mb::k_ux_extend_pk, dim3((unsigned)((n_hits + 255) / 256)), dim3(256), 0, nullptr, keys.data(), n_hits, ut, xdrop, K, sc,
                                   hsps.data(), (int64_t)hsps.size(), ctr);
                printf("  ux: level 1 from the packed strands\n");
            } else
            hipLaunchKernelGGL(mb::k_ux_extend, dim3((unsigned)((n_hits + 255) / 256)), dim3(256), 0, nullptr, keys.data(), n_hits, ut, xdrop, K, sc,
                               hsps.data(), (int64_t)hsps.size(), ctr);
            hipLaunchKernelGGL(mb::k_ux_tail, dim3(blocks), dim3(256), 0, nullptr, keys.data(), n_hits, ut, xdrop, K, sc, hsps.data(), (int64_t)hsps.size(), ctr);
            if (use_h16) {
                // launch_ungapped_hash16 (mb_kernels.hip): target position descending, then (unit | hash class | q_end | variant rank), both stable;
                // one lane per class applies the rule to the records
                std::vector<unsigned long long> k2((size_t)n_hits), k1((size_t)n_hits);
                std::vector<uint32_t> val((size_t)n_hits);
                const unsigned nb = (unsigned)((n_hits + 255) / 256);
                hipLaunchKernelGGL(mb::k_h16_tkeys, dim3(nb), dim3(256), 0, nullptr, keys.data(), n_hits, ut, k2.data(), val.data());
                { std::vector<uint32_t> o(val); std::stable_sort(o.begin(), o.end(), [&](uint32_t a, uint32_t b) { return (k2[a] & 0x7FFFFFFFull) < (k2[b] & 0x7FFFFFFFull); }); val = o; }      // (val[i] = i before the sort)
                hipLaunchKernelGGL(mb::k_h16_ckeys, dim3(nb), dim3(256), 0, nullptr, keys.data(), n_hits, ut, val.data(), k1.data());
                {
                    std::vector<size_t> ord((size_t)n_hits);
                    for (size_t x = 0; x < ord.size(); x++) ord[x] = x;
                    std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return k1[a] < k1[b]; });
                    std::vector<unsigned long long> k1s((size_t)n_hits); std::vector<uint32_t> vs((size_t)n_hits);
                    for (size_t x = 0; x < ord.size(); x++) { k1s[x] = k1[ord[x]]; vs[x] = val[ord[x]]; }
                    k1 = k1s; val = vs;
                }
                hipLaunchKernelGGL(mb::k_h16_resolve, dim3(nb), dim3(256), 0, nullptr, k1.data(), val.data(), n_hits, sc.rec, hsps.data(), ctr);
            } else {
            // (few blocks: a block's stretch of the sorted hits then spans several units)
            hipLaunchKernelGGL(mb::k_ux_accept, dim3(cs % 2 ? 1u + (unsigned)rnd(2) : (unsigned)((n_hits + 255) / 256)), dim3(256), 0, nullptr, keys.data(), n_hits, ut, extent.data(), sc, hsps.data(), ctr);
            hipLaunchKernelGGL(mb::k_ux_resolve, dim3(blocks), dim3(256), 0, nullptr, keys.data(), n_hits, ut, extent.data(), sc, hsps.data(), ctr);
            { size_t nd = 0; for (uint32_t w : dirty) nd += (size_t)__builtin_popcount(w); printf("  ux: %zu dirty diagonals\n", nd); }
            }
            hipLaunchKernelGGL(mb::k_ux_census, dim3(1), dim3(256), 0, nullptr, ut, hsps.data(), (int64_t)hsps.size(), ctr);
            { unsigned nb = 0; for (unsigned v : blk_cnt) nb += v; printf("  ux: %u + %u of %lld hits left for the tail (block slots + list), %llu candidates, %u dirty runs\n", nb, n_entries[0], (long long)n_hits, ctr[0].hsps, n_entries[1]); }
        }
        // the long runs, whichever kernel took the short ones: a wave per run and turn, the grid's waves striding over the list
        if (!use_h16) hipLaunchKernelGGL(mb::k_ungapped_long, dim3(1u + (unsigned)rnd(2)), dim3(256), 0, nullptr, keys.data(), n_hits, heads.data() + (n + n / 2 + n / 4 + n / 8 + 8), n_heads + 4, ut,
                           extent.data(), xdrop, K, hsps.data(), (int64_t)hsps.size(), ctr);
        hipLaunchKernelGGL(mb::k_hsp_anchor, dim3(1), dim3(256), 0, nullptr, ut, hsps.data(), (int64_t)hsps.size(), ctr);
        hsps.resize((size_t)ctr[0].hsps);
        bool all_ok = true;
        size_t total_hsps = 0, total_ref = 0;
        for (int x = 0; x < n_units; x++) {
            Unit &u = units[(size_t)x];
            std::vector<mb::DevHsp> mine;
            bool anchors_ok = true;
            for (const mb::DevHsp &d : hsps)
                if (d.unit == x && d.score != INT32_MIN) { if (d.anchor_off != anchor_ref(d, u.tc, u.qc)) anchors_ok = false; mine.push_back(d); }
            for (mb::DevHsp &d : mine) { d.anchor_off = 0; d.unit = 0; }
            for (mb::DevHsp &d : u.ref.hsps) d.unit = 0;
            std::sort(mine.begin(), mine.end(), hsp_less);
            std::sort(u.ref.hsps.begin(), u.ref.hsps.end(), hsp_less);
            std::vector<int32_t> ext(extent.begin() + (long)(ext_scrambled ? 0 : u.dbase), extent.begin() + (long)(ext_scrambled ? 0 : u.dbase) + (long)(u.tn + u.qn + 2));
            if (ext_scrambled) for (size_t d = 0; d < ext.size(); d++) ext[d] = extent[(size_t)(((uint32_t)d * plane_mul) & plane_mask)];
            bool ok = anchors_ok && ctr[x].extended == u.ref.extended && ctr[x].cols == u.ref.cols && mine.size() == u.ref.hsps.size() && ext == u.ref.extent;
            for (size_t i = 0; ok && i < mine.size(); i++) ok = memcmp(&mine[i], &u.ref.hsps[i], sizeof(mb::DevHsp)) == 0;
            total_hsps += mine.size(); total_ref += u.ref.hsps.size();
            if (!ok) {
                all_ok = false;
                printf("  unit %d of %d: hsps %zu/%zu extended %llu/%llu cols %llu/%llu anchors %d\n", x, n_units, mine.size(), u.ref.hsps.size(), ctr[x].extended, u.ref.extended,
                       ctr[x].cols, u.ref.cols, (int)anchors_ok);
                for (size_t i = 0; i < std::min(mine.size(), u.ref.hsps.size()); i++)
                    if (memcmp(&mine[i], &u.ref.hsps[i], sizeof(mb::DevHsp)) != 0) {
                        printf("  first differing hsp %zu: got t %d q %d len %d score %d seed %d/%d | want t %d q %d len %d score %d seed %d/%d\n", i, mine[i].t_start,
                               mine[i].q_start, mine[i].len, mine[i].score, mine[i].seed_t_end, mine[i].seed_q_end, u.ref.hsps[i].t_start, u.ref.hsps[i].q_start,
                               u.ref.hsps[i].len, u.ref.hsps[i].score, u.ref.hsps[i].seed_t_end, u.ref.hsps[i].seed_q_end);
                        break;
                    }
                for (size_t d = 0; d < ext.size(); d++)
                    if (ext[d] != u.ref.extent[d]) { printf("  first differing extent: diagonal %zu got %d want %d\n", d, ext[d], u.ref.extent[d]); break; }
            }
        }
        if (use_h16) { unsigned long long cr = 0; for (Unit &u : units) cr += u.ref.crossed; printf("  h16: %llu hits dropped by an extension on ANOTHER diagonal of their hash class\n", cr); }
        printf("case %d: %d unit(s) hits %lld runs %u + %u long (> %d hits) xdrop %d K %d  hsps %zu/%zu  %s\n", cs, n_units, (long long)n_hits,
               n_heads[0] + n_heads[1] + n_heads[2] + n_heads[3], n_heads[4], long_run, xdrop, K, total_hsps, total_ref, all_ok ? "ok" : "MISMATCH");
        if (!all_ok) bad++;
    }
    return bad ? 1 : 0;
}

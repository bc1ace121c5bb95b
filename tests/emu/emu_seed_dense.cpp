// TEST INFRASTRUCTURE ONLY: runs cactus_amd/csrc/mb_seed_dense.h (packed strands, seed words from the packed form, the q-ordered
// one-pass seed search k_seed_hits / k_seed_keys, the diagonal scramble and k_keys_unhash) and mb_seed_index.h (the dense seed table through
// its kernels -- index words, the scan of the 2^24 bucket counts with the occupancy bitmap, scatter --, the two-pass search in q batches
// and the unordered one-pass search) on the HOST -- one pthread per work-item
// (see hip/hip_runtime.h) -- against a plain restatement of SURVEY A.3 / A.4: the table of the target = for every word the indexed
// positions, the hits of a strand = for every valid query window, every word variant, every position of its bucket, query position
// by query position.  Nothing of this is shipped or measured.
//   emu_seed_dense <seed> <n_cases> [table|bin]      exit status 0 iff every case is identical (bin: mb_seed_bin.h only)
#define MB_EMU 1
#include <hip/hip_runtime.h>
#undef __launch_bounds__
#define __launch_bounds__(...)

#include <algorithm>
#include <cstdio>
#include <map>
#include <random>

#include "mb_common.h"
#define __builtin_memcpy memcpy

struct ulonglong2 { unsigned long long x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline int emu_readlane(int v, int l) {
    emu::Group *g = emu::g_group;
    const unsigned tid = emu::t_threadIdx.x, w = tid >> 6;
    g->slot[tid] = (unsigned long long)(unsigned)v;
    pthread_barrier_wait(&g->wave[w]);
    const int o = (int)(unsigned)g->slot[(tid & ~63u) | ((unsigned)l & 63u)];
    pthread_barrier_wait(&g->wave[w]);
    return o;
}
#define __builtin_amdgcn_readlane(v, l) emu_readlane((v), (l))
static inline void emu_wave_sync() { pthread_barrier_wait(&emu::g_group->wave[emu::t_threadIdx.x >> 6]); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

namespace mb {
// inclusive prefix sum over the wave (mb_kernels.hip: six DPP steps), through the per-wave exchange slots
inline int dpp_scan_add(int v) {
    emu::Group *g = emu::g_group;
    const unsigned tid = emu::t_threadIdx.x, w = tid >> 6, lane = tid & 63;
    g->slot[tid] = (unsigned long long)(unsigned)v;
    pthread_barrier_wait(&g->wave[w]);
    int s = 0;
    for (unsigned l = 0; l <= lane; l++) s += (int)(unsigned)g->slot[(tid & ~63u) | l];
    pthread_barrier_wait(&g->wave[w]);
    return s;
}
#include "mb_seedword.h"
#include "mb_seed_index.h"
#include "mb_seed_dense.h"
#include "mb_seed_bin.h"
}  // namespace mb

// ---- mb_seed_bin.h: keys grouped by diagonal through bins and LDS buckets, against std::sort by (scrambled diagonal, q) + the unscrambling.
//      Cases: few keys (one bin), many small bins (a small `mean`), diagonals with hundreds / thousands of hits (long rank loops, the large
//      sorter), plain and scrambled diagonals, keys that did not fit (nothing planned).
static int bin_cases(unsigned seed0, int n_cases) {
    int bad = 0;
    for (int cs = 0; cs < n_cases; cs++) {
        std::mt19937_64 rng(seed0 * 2654435761u + (unsigned)cs);
        auto rnd = [&](unsigned long long n) { return (unsigned long long)(rng() % n); };
        const int diag_bits = 8 + (int)rnd(20);
        const uint32_t hmask = (1u << diag_bits) - 1u, hmul = cs % 4 == 3 ? 1u : 0x9E3779B1u;
        uint32_t hinv = 1u; for (int it = 0; it < 5; it++) hinv *= 2u - hmul * hinv;
        const int mean = cs % 3 == 0 ? (cs % 2 ? 2800 : 11000) : 40 + (int)rnd(400);
        const size_t n_chance = cs % 5 == 4 ? rnd(60) : cs % 11 == 10 ? 70000 + rnd(9000) : 200 + rnd(cs % 3 == 0 ? 30000 : 6000);      // (several chunks now and then)
        std::vector<unsigned long long> keys;
        std::map<uint32_t, std::vector<uint32_t>> per_diag;            // diagonal -> q ends (distinct)
        auto add = [&](uint32_t d, uint32_t q) { per_diag[d & hmask].push_back(q); };
        for (size_t i = 0; i < n_chance; i++) add((uint32_t)rnd(1ull << diag_bits), (uint32_t)rnd(1u << 24));
        const int n_heavy = (int)rnd(4);
        for (int h = 0; h < n_heavy; h++) {                             // a diagonal of real homology: many hits, neighbours too
            const uint32_t d = (uint32_t)rnd(1ull << diag_bits), len = cs % 7 == 6 && h == 0 ? 5000 + (uint32_t)rnd(6000) : h == 1 ? 1500 + (uint32_t)rnd(2500) : 30 + (uint32_t)rnd(900);
            for (uint32_t k = 0; k < len; k++) add(d + (rnd(20) == 0 ? 1u : 0u), 19 + 3 * k + (uint32_t)rnd(3));
        }
        for (auto &kv : per_diag) {
            std::sort(kv.second.begin(), kv.second.end());
            kv.second.erase(std::unique(kv.second.begin(), kv.second.end()), kv.second.end());
            for (uint32_t q : kv.second) keys.push_back(((unsigned long long)((kv.first * hmul) & hmask) << 32) | q);
        }
        std::shuffle(keys.begin(), keys.end(), rng);
        const unsigned long long n = keys.size();
        std::vector<unsigned long long> want(keys);
        std::sort(want.begin(), want.end());
        for (auto &k : want) k = ((unsigned long long)((((uint32_t)(k >> 32)) * hinv) & hmask) << 32) | (uint32_t)k;
        bool ok = true;
        const char *why = "";
        const size_t sw = (size_t)mb::kBinStateWords + 8;
        std::vector<uint32_t> state(sw, 0u);
        unsigned long long n_dev = n;
        // (1) keys that did not fit their buffer: no plan, nothing counted
        const size_t mw = (size_t)mb::bin_matrix_words(n + 8, diag_bits, mean) + 8;
        {
            std::vector<uint32_t> st2(sw, 0u), mx2(mw, 0x5A5A5A5Au);
            unsigned long long big = n + 5;
            hipLaunchKernelGGL(mb::k_bin_count, dim3(2), dim3(256), 0, nullptr, keys.data(), &big, n, diag_bits, mean, st2.data(), mx2.data());
            hipLaunchKernelGGL(mb::k_bin_scan, dim3((1u << mb::kBinBitsMax) / 256u), dim3(256), 0, nullptr, &big, n, diag_bits, mean, st2.data(), mx2.data());
            if (st2[4] != 1u) { ok = false; why = "overflow not flagged"; }
            for (size_t x = 5; x < sw && ok; x++) if (st2[x]) { ok = false; why = "overflow: something counted"; }
        }
        // (2) the plan
        std::vector<uint32_t> matrix(mw, 0x5A5A5A5Au);
        const unsigned long long cap = n + rnd(8);
        const unsigned n_chunks_cap = (unsigned)std::max<unsigned long long>(1, (cap + mb::kBinChunk - 1) / mb::kBinChunk);
        hipLaunchKernelGGL(mb::k_bin_count, dim3(n_chunks_cap), dim3(256), 0, nullptr, keys.data(), &n_dev, cap, diag_bits, mean, state.data(), matrix.data());
        hipLaunchKernelGGL(mb::k_bin_scan, dim3((1u << mb::kBinBitsMax) / 256u), dim3(256), 0, nullptr, &n_dev, cap, diag_bits, mean, state.data(), matrix.data());
        const int nbits = mb::bin_bits(n, diag_bits, mean), nb = 1 << nbits;
        uint32_t mx = 0, big = 0, small = 0, run = 0;
        {
            std::vector<uint32_t> cnt((size_t)nb, 0u);
            for (auto k : keys) cnt[nbits ? (uint32_t)(k >> 32) >> (diag_bits - nbits) : 0u]++;
            for (int b = 0; b < nb && ok; b++) {
                if (mb::bin_starts(state.data())[b] != run || mb::bin_counts(state.data())[b] != cnt[(size_t)b]) { ok = false; why = "plan: a bin's place"; }
                run += cnt[(size_t)b]; mx = std::max(mx, cnt[(size_t)b]); big += cnt[(size_t)b] > (uint32_t)mb::kBinCapSmall; small += cnt[(size_t)b] > 0 && cnt[(size_t)b] <= (uint32_t)mb::kBinCapSmall;
            }
            if (ok && n && (mb::bin_starts(state.data())[nb] != n || state[0] != (uint32_t)nbits || state[1] != mx || state[2] != big || state[3] != (uint32_t)n || state[4] != 0u || state[6] != small)) { ok = false; why = "plan: head"; }
            for (size_t x = mw - 8; x < mw; x++) if (matrix[x] != 0x5A5A5A5Au) { ok = false; why = "a store behind the matrix"; }
        }
        // (3) scatter + the sorter(s), as launch_bin_group queues them; the output buffer has guard words on both sides
        std::vector<unsigned long long> in(keys), outbuf((size_t)n + 16, 0xEEEEEEEEEEEEEEEEull);
        unsigned long long *out = outbuf.data() + 8;
        if (ok && n && mx <= (uint32_t)mb::kBinCapBig) {
            const unsigned long long *binned = in.data();
            if (nbits > 0) {
                if (nbits <= mb::kBinStagedBits && cs % 2 == 0) hipLaunchKernelGGL(mb::k_bin_scatter_staged, dim3((unsigned)((n + mb::kBinChunk - 1) / mb::kBinChunk)), dim3(1024), 0, nullptr, in.data(), out, (int64_t)n, diag_bits, nbits, state.data(), matrix.data());
                else hipLaunchKernelGGL(mb::k_bin_scatter, dim3((unsigned)((n + mb::kBinChunk - 1) / mb::kBinChunk)), dim3(1024), 0, nullptr, in.data(), out, (int64_t)n, diag_bits, nbits, state.data(), matrix.data());
                binned = out;
                for (int b = 0; b < nb && ok; b++)                          // every bin holds its own keys (in some order)
                    for (uint32_t x = mb::bin_starts(state.data())[b]; x < mb::bin_starts(state.data())[b + 1]; x++)
                        if ((uint32_t)(out[x] >> 32) >> (diag_bits - nbits) != (uint32_t)b) { ok = false; why = "scatter: a key in the wrong bin"; break; }
            }
            if (small) hipLaunchKernelGGL((mb::k_bin_sort<mb::kBinCapSmall, 11, 512>), dim3((unsigned)nb), dim3(512), 0, nullptr, binned, out, state.data(), diag_bits, nbits, hinv, hmask);
            if (big) hipLaunchKernelGGL((mb::k_bin_sort<mb::kBinCapBig, 12, 1024>), dim3((unsigned)nb), dim3(1024), 0, nullptr, binned, out, state.data(), diag_bits, nbits, hinv, hmask);
            for (int g = 0; g < 8; g++) if (outbuf[(size_t)g] != 0xEEEEEEEEEEEEEEEEull || outbuf[(size_t)n + 8 + (size_t)g] != 0xEEEEEEEEEEEEEEEEull) { ok = false; why = "a store outside the keys"; }
            if (ok && !std::equal(want.begin(), want.end(), out)) { ok = false; why = "grouped keys differ from sort + unscramble"; }
        }
        size_t longest = 0;
        for (auto &kv : per_diag) longest = std::max(longest, kv.second.size());
        printf("bin case %d: %llu keys, %d diagonal bits (%s), longest diagonal %zu, mean %d -> %d bins, largest %u, %u beyond the small sorter%s  %s%s\n", cs, n, diag_bits, hmul == 1u ? "plain" : "scrambled",
               longest, mean, nb, mx, big, mx > (uint32_t)mb::kBinCapBig ? " (beyond the large one too: rocprim's case)" : "", ok ? "ok" : "MISMATCH: ", ok ? "" : why);
        if (!ok) bad++;
    }
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    const unsigned seed0 = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const int n_cases = argc > 2 ? atoi(argv[2]) : 4;
    // "table": also build the table through the kernels (the emulated scan of the 2^24 + 1 bucket counts takes minutes: 8 193 groups, three launches)
    const bool with_table = argc > 3 && !strcmp(argv[3], "table");
    if (argc > 3 && !strcmp(argv[3], "bin")) return bin_cases(seed0, n_cases);
    int bad = 0;
    const size_t kBuckets = (size_t)1 << 24;
    std::vector<uint32_t> offsets(kBuckets + 1), occ(kBuckets / 32), counts(kBuckets + 1);
    for (int cs = 0; cs < n_cases; cs++) {
        std::mt19937 rng(seed0 * 15485863u + (unsigned)cs);
        auto rnd = [&](int n) { return (int)(rng() % (unsigned)n); };
        const int step = 1 + rnd(3), nvar = cs % 2 ? 1 : 13;
        const bool packed = cs % 4 < 2;                                       // (the byte-code form of the query is the other instantiation)
        const int64_t tn = 300 + rnd(4000), qn = cs % 5 == 4 ? rnd(40) : 200 + rnd(nvar == 1 ? 14000 : 5000), first = rnd(step);
        const int low = 5 + rnd(5);                                            // of 10 bases, how many come from a two-letter alphabet (words repeat: full buckets)
        auto make_seq = [&](int64_t n, std::vector<uint8_t> &buf) {
            buf.assign((size_t)n + 2 * mb::kDevPad + 160, mb::kSep);
            uint8_t *c = buf.data() + mb::kDevPad;
            for (int64_t i = 0; i < n; i++) c[i] = (uint8_t)(rnd(10) < low ? rnd(2) : rnd(4));
            for (int s = 0; s < 6 && n > 0; s++) c[rnd((int)n)] = 4;
            for (int s = 0; s < 3 && n > 0; s++) { const int a = rnd((int)n); for (int k = a; k < std::min<int64_t>(n, a + 30); k++) c[k] |= 8; }
            for (int s = 0, ns = rnd(3); s < ns && n > 2; s++) c[1 + rnd((int)n - 2)] = mb::kSep;
        };
        std::vector<uint8_t> tbuf, qbuf;
        make_seq(tn, tbuf); make_seq(qn, qbuf);
        uint8_t *tc = tbuf.data() + mb::kDevPad, *qc = qbuf.data() + mb::kDevPad;
        for (int c = 0, nc = rnd(8); c < nc && qn > 200; c++) {              // stretches of the target inside the query, transitions here and there
            const int len = 40 + rnd(120), a = rnd((int)std::max<int64_t>(1, tn - len)), b = rnd((int)(qn - len));
            for (int x = 0; x < len && a + x < tn; x++) { const uint8_t v = tc[a + x]; qc[b + x] = (v < 4 && rnd(25) == 0) ? (uint8_t)(v ^ 2) : v; }
        }
        bool ok = true;
        const char *why = "";
        // ---- packed strands: the kernel against a plain loop
        auto pack = [&](const uint8_t *codes, int64_t n, std::vector<unsigned long long> &p2, std::vector<unsigned long long> &pm) {
            const int64_t nm = (int64_t)mb::packed_wordsm(n);
            // exactly what the pipeline allocates (pack_strand: packed_words2 / packed_wordsm words), guard words behind both planes
            const size_t n2 = mb::packed_words2(n), n1 = mb::packed_wordsm(n);
            const unsigned long long guard = 0x1234567812345678ull;
            p2.assign(n2 + 4, guard); pm.assign(n1 + 4, guard);
            const size_t nx = mb::packed_dwordsx(n);                    // the ungapped extension's records (mb_ungapped_ux.h): 12 bytes per 32 bases
            std::vector<uint32_t> px(nx + 4, 0x12345678u);
            hipLaunchKernelGGL(mb::k_pack2bit_mask, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, nullptr, codes, n, p2.data(), pm.data(), nm, px.data());
            for (size_t x = 0; x < 4; x++) if (p2[n2 + x] != guard || pm[n1 + x] != guard || px[nx + x] != 0x12345678u) return false;      // a store past a plane
            for (int64_t i = 0; i < nm * 64; i++) {                      // base i of record i >> 5: codes as in p2, a flag for N / IUPAC / separator / beyond the end only
                const unsigned c = i < n ? codes[i] : 0xFFu;
                const uint32_t *r = px.data() + 3 * (i >> 5);
                const unsigned long long a = ((unsigned long long)r[1] << 32) | r[0];
                if (((unsigned)(a >> (62 - 2 * (i & 31))) & 3u) != (c & 3u) || ((r[2] >> (31 - (i & 31))) & 1u) != (unsigned)((c & 0x84u) != 0u)) return false;
            }
            for (int64_t i = 0; i < nm * 64; i++) {
                const unsigned c = i < n ? codes[i] : 0xFFu;
                const unsigned two = (unsigned)(p2[(size_t)(i >> 5)] >> (62 - 2 * (i & 31))) & 3u, m = (unsigned)(pm[(size_t)(i >> 6)] >> (63 - (i & 63))) & 1u;
                if (two != (c & 3u) || m != (unsigned)((c & 0xFCu) != 0u)) return false;
            }
            return true;
        };
        std::vector<unsigned long long> tp2, tpm, qp2, qpm;
        if (!pack(tc, tn, tp2, tpm) || !pack(qc, qn, qp2, qpm)) { ok = false; why = "k_pack2bit_mask"; }
        // ---- the target's table: words from the packed form against window_word, then a plain counting sort
        const int64_t n_slots = tn > first ? (tn - first + step - 1) / step : 0;
        std::fill(counts.begin(), counts.end(), 0u);
        std::vector<uint32_t> words((size_t)n_slots + 1, 0x77777777u);
        if (n_slots) hipLaunchKernelGGL(mb::k_index_words_packed, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, nullptr, tp2.data(), tpm.data(), tn, step, first, words.data(), n_slots, counts.data());
        std::map<uint32_t, std::vector<uint32_t>> table;                      // bucket -> positions, ascending
        for (int64_t sl = 0; sl < n_slots && ok; sl++) {
            const int64_t p = first + sl * step;
            uint32_t w, want = 0xFFFFFFFFu;
            if (p + mb::kSeedSpan <= tn && mb::window_word(tc, p, w)) { want = mb::dense_bucket(w); table[want].push_back((uint32_t)p); }
            if (words[(size_t)sl] != want) { ok = false; why = "k_index_words_packed"; }
        }
        uint32_t run = 0;
        std::fill(occ.begin(), occ.end(), 0u);
        std::vector<uint32_t> positions;
        {
            auto it = table.begin();
            // (offsets of all 2^24 buckets; the occupied ones are few: walk the map)
            std::fill(offsets.begin(), offsets.end(), 0u);
            for (; it != table.end(); ++it) {
                if (counts[it->first] != it->second.size()) { ok = false; why = "bucket counts"; }
                offsets[it->first] = (uint32_t)it->second.size();
                occ[it->first >> 5] |= 1u << (it->first & 31u);
            }
            for (size_t b = 0; b <= kBuckets; b++) { const uint32_t c = b < kBuckets ? offsets[b] : 0u; offsets[b] = run; run += c; }
            positions.resize((size_t)run + 1);
            for (auto &kv : table) std::copy(kv.second.begin(), kv.second.end(), positions.begin() + offsets[kv.first]);
        }
        // (the 13 buckets of a word: dense_variant of its bucket = dense_bucket of variant_word)
        for (int k = 0; k < 200 && ok; k++) {
            const uint32_t w = (uint32_t)rng() & 0xFFFFFFu;
            for (int v = 0; v < 13; v++) if (mb::dense_variant(mb::dense_bucket(w), v) != mb::dense_bucket(mb::variant_word(w, v))) { ok = false; why = "dense_variant"; }
        }
        // ---- the same table through the kernels of mb_seed_index.h: words from the byte codes (= the packed ones), the three-launch scan of
        //      the 2^24 + 1 bucket counts (offsets, occupancy bitmap, counts zeroed), the scatter; then the counters cleared
        if (ok && with_table && cs % 3 == 0) {
            std::vector<uint32_t> words_b((size_t)n_slots + 1, 0x55555555u), k_counts(kBuckets + 8, 0u), k_offsets(kBuckets + 8, 0xCCCCCCCCu), k_occ(kBuckets / 32, 0xCCCCCCCCu);
            std::vector<unsigned long long> bsum((kBuckets + 1) / 2048 + 4, 0ull);
            mb::launch_index_words(tc, tn, step, first, words_b.data(), n_slots, k_counts.data(), nullptr);
            for (int64_t sl = 0; sl < n_slots && ok; sl++) if (words_b[(size_t)sl] != words[(size_t)sl]) { ok = false; why = "k_index_words"; }
            mb::launch_scan_index(k_counts.data(), k_offsets.data(), bsum.data(), k_occ.data(), nullptr);
            for (size_t b = 0; b <= kBuckets && ok; b++) if (k_offsets[b] != offsets[b]) { ok = false; why = "launch_scan_index: offsets"; }
            for (size_t x = 0; x < kBuckets / 32 && ok; x++) if (k_occ[x] != occ[x]) { ok = false; why = "launch_scan_index: occupancy bitmap"; }
            for (size_t b = 0; b <= kBuckets && ok; b++) if (k_counts[b] != 0u) { ok = false; why = "launch_scan_index: counts not zeroed"; }
            std::vector<uint32_t> k_pos((size_t)run + 1, 0xFFFFFFFFu);
            mb::launch_index_scatter(words.data(), n_slots, step, first, k_offsets.data(), k_counts.data(), k_pos.data(), nullptr);
            for (auto &kv : table) {
                std::vector<uint32_t> got(k_pos.begin() + offsets[kv.first], k_pos.begin() + offsets[kv.first + 1]);
                std::sort(got.begin(), got.end());
                if (got != kv.second) { ok = false; why = "k_index_scatter"; break; }
            }
            mb::launch_index_clear(words.data(), n_slots, k_counts.data(), nullptr);
            for (auto &kv : table) if (k_counts[kv.first] != 0u) { ok = false; why = "k_index_clear"; }
        }
        // ---- the search: k_seed_hits, the scan of the tiles' counts, k_seed_keys, k_keys_unhash
        int diag_bits = 1; while ((1ll << diag_bits) < tn + qn + 2) diag_bits++;
        const uint32_t hmask = (1u << diag_bits) - 1u, hmul = cs % 3 == 2 ? 1u : 0x9E3779B1u;
        uint32_t hinv = 1u; for (int it = 0; it < 5; it++) hinv *= 2u - hmul * hinv;
        const int threads = nvar == 13 ? 512 : 1024, per_tile = threads * (nvar == 13 ? 1 : 4);
        const int n_tiles = (int)((qn + per_tile - 1) / per_tile);
        // the rule, query position by query position
        std::vector<std::vector<unsigned long long>> want_q((size_t)std::max<int64_t>(qn, 1));
        unsigned long long total_want = 0;
        for (int64_t q = 0; q + mb::kSeedSpan <= qn; q++) {
            uint32_t w;
            if (!mb::window_word(qc, q, w)) continue;
            for (int v = 0; v < nvar; v++) {
                auto it = table.find(mb::dense_bucket(mb::variant_word(w, v)));
                if (it == table.end()) continue;
                for (uint32_t p : it->second) want_q[(size_t)q].push_back(((unsigned long long)(uint32_t)((int64_t)p - q + qn) << 32) | (unsigned long long)(q + mb::kSeedSpan));
            }
            total_want += want_q[(size_t)q].size();
        }
        for (int pass = 0; pass < 2 && ok && n_tiles > 0; pass++) {
            // pass 0: room for everything; pass 1: too little room -- the total still comes out, nothing is written past the end
            const unsigned long long cap = pass == 0 ? total_want + 7 : total_want / 2;
            std::vector<unsigned long long> scratch((size_t)cap + 8, 0xABABABABABABABABull), keys((size_t)cap + 8, 0xCDCDCDCDCDCDCDCDull), tile_base((size_t)n_tiles + 1, 0ull);
            std::vector<uint32_t> tile_cnt((size_t)n_tiles + 1, 0u), tile_off((size_t)n_tiles + 1, 0u);
            unsigned long long total = 0;
            const unsigned grid = 1u + (unsigned)rnd(std::max(1, n_tiles));   // (fewer blocks than tiles: a block takes several)
#define EMU_ORD(P, R, NV, T) hipLaunchKernelGGL((mb::k_seed_hits<P, R, NV, T>), dim3(grid), dim3(T), 0, nullptr, qc, qp2.data(), qpm.data(), qn, offsets.data(), occ.data(), scratch.data(), cap, &total, tile_base.data(), tile_cnt.data(), n_tiles)
            if (nvar == 13) { if (packed) EMU_ORD(true, 1, 13, 512); else EMU_ORD(false, 1, 13, 512); }
            else { if (packed) EMU_ORD(true, 4, 1, 1024); else EMU_ORD(false, 4, 1, 1024); }
#undef EMU_ORD
            if (total != total_want) { ok = false; why = "k_seed_hits total"; break; }
            { uint32_t r2 = 0; for (int t = 0; t < n_tiles; t++) { tile_off[(size_t)t] = r2; r2 += tile_cnt[(size_t)t]; } }      // (launch_scan_u32 on the device)
            hipLaunchKernelGGL(mb::k_seed_keys, dim3(1u + (unsigned)rnd(std::max(1, n_tiles))), dim3(256), 0, nullptr, scratch.data(), tile_base.data(), tile_cnt.data(), tile_off.data(),
                               positions.data(), keys.data(), cap, qn, per_tile, hmul, hmask, n_tiles);
            for (size_t k = (size_t)cap; k < keys.size(); k++) if (keys[k] != 0xCDCDCDCDCDCDCDCDull || scratch[k] != 0xABABABABABABABABull) { ok = false; why = "write past the end"; }
            if (pass == 1 || !ok) continue;
            if (hmul != 1u) {
                // scrambled keys group the diagonals: unscramble and compare
                hipLaunchKernelGGL(mb::k_keys_unhash, dim3((unsigned)(((total + 1) / 2 + 255) / 256 + 1)), dim3(256), 0, nullptr, keys.data(), (int64_t)total, hinv, hmask);
            }
            size_t at = 0;
            for (int64_t q = 0; q < qn && ok; q++) {
                std::vector<unsigned long long> &mine = want_q[(size_t)q];
                if (mine.empty()) continue;
                std::vector<unsigned long long> got(keys.begin() + (long)at, keys.begin() + (long)(at + mine.size()));
                std::sort(got.begin(), got.end()); std::sort(mine.begin(), mine.end());
                if (got != mine) { ok = false; why = "keys of a query position"; }
                at += mine.size();
            }
            if (ok && at != total) { ok = false; why = "key count"; }
        }
        // ---- a strand in q batches (k_seed_count, scan, k_seed_fill per batch: the path of a strand whose hits do not fit one key buffer) and
        //      the one-pass search without q order (k_seed_search: keys in the order the blocks get there)
        if (ok && qn > 0) {
            std::vector<uint32_t> qcnt((size_t)qn + 8, 0xCCCCCCCCu), hit_off((size_t)qn + 8, 0u);
            std::vector<unsigned long long> bsum((size_t)qn / 2048 + 4, 0ull);
            mb::launch_seed_count(qc, qn, offsets.data(), occ.data(), nvar == 13, qcnt.data(), nullptr);
            for (int64_t q = 0; q < qn && ok; q++) if (qcnt[(size_t)q] != want_q[(size_t)q].size()) { ok = false; why = "k_seed_count"; }
            const int64_t cuts[3] = {0, qn / 3, qn};
            for (int bt = 0; bt < 2 && ok; bt++) {
                const int64_t q0 = cuts[bt], q1 = cuts[bt + 1];
                if (q1 <= q0) continue;
                mb::launch_scan_u32(qcnt.data() + q0, hit_off.data(), q1 - q0, bsum.data(), nullptr);
                unsigned long long n_b = 0;
                for (int64_t q = q0; q < q1; q++) n_b += want_q[(size_t)q].size();
                std::vector<unsigned long long> keys((size_t)n_b + 4, 0xCDCDCDCDCDCDCDCDull);
                mb::launch_seed_fill(qc, q0, q1, qn, offsets.data(), occ.data(), positions.data(), nvar == 13, hit_off.data(), keys.data(), nullptr, hmul, hmask);
                size_t at = 0;
                for (int64_t q = q0; q < q1 && ok; q++) {
                    std::vector<unsigned long long> mine;
                    for (unsigned long long k : want_q[(size_t)q]) mine.push_back(((unsigned long long)((((uint32_t)(k >> 32)) * hmul) & hmask) << 32) | (uint32_t)k);
                    if (hit_off[(size_t)(q - q0)] != at) { ok = false; why = "scan of a q batch"; break; }
                    std::vector<unsigned long long> got(keys.begin() + (long)at, keys.begin() + (long)(at + mine.size()));
                    std::sort(got.begin(), got.end()); std::sort(mine.begin(), mine.end());
                    if (got != mine) { ok = false; why = "k_seed_fill"; }
                    at += mine.size();
                }
                if (ok && keys[(size_t)n_b] != 0xCDCDCDCDCDCDCDCDull) { ok = false; why = "k_seed_fill wrote past its batch"; }
            }
            if (ok) {
                std::vector<unsigned long long> keys((size_t)total_want + 4, 0xCDCDCDCDCDCDCDCDull), all;
                unsigned long long total = 0;
                mb::launch_seed_search(qc, qn, offsets.data(), occ.data(), positions.data(), nvar == 13, keys.data(), total_want, &total, nullptr);
                for (auto &v : want_q) all.insert(all.end(), v.begin(), v.end());
                keys.resize((size_t)total_want);
                std::sort(keys.begin(), keys.end()); std::sort(all.begin(), all.end());
                if (total != total_want || keys != all) { ok = false; why = "k_seed_search"; }
            }
        }
        printf("case %d: T %lld (step %d, first %lld) x Q %lld, %d variants, %s query, %s diagonals, %llu hits in %d tiles  %s%s\n", cs, (long long)tn, step, (long long)first,
               (long long)qn, nvar, packed ? "packed" : "byte-code", hmul == 1u ? "plain" : "scrambled", total_want, n_tiles, ok ? "ok" : "MISMATCH: ", ok ? "" : why);
        if (!ok) bad++;
    }
    return bad ? 1 : 0;
}

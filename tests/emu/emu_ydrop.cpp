// TEST INFRASTRUCTURE ONLY: runs cactus_amd/csrc/mb_ydrop2.h -- the piece evaluator of k_ydrop2, the gapped stage's DP kernel -- on the
// HOST: one pthread per lane of the wave, DPP moves / scans, readlane, readfirstlane and ballot exchanged through the wave's slots (see
// hip/hip_runtime.h), against a plain restatement of SURVEY A.10 ONE_SIDED (rows = query, one cell at a time in row-major order, y-drop
// against the running best, ties diag > D > I, extension wins gap ties): best cell, cells and rows counted, and the alignment read back
// from the kernel's 4-bit trace codes through its row records; the same through the four-wave kernel body with the LDS ring
// (mb_ydrop_lds.h).  Also a side cut in two pieces: the second continues from the first
// one's exit snapshot and must end where the whole side ends; the relay hand-over check (mb_verify.h) on the states the evaluator
// writes; and the traceback kernels (mb_trace.h: walkers, predicted joins, stitch) over those chains of pieces.  Nothing of this is shipped or measured.
//   emu_ydrop <seed> <n_cases>      exit status 0 iff every case is identical
#define MB_EMU 1
#include <hip/hip_runtime.h>
#undef __launch_bounds__
#define __launch_bounds__(...)

#include <algorithm>
#include <cstdio>
#include <functional>
#include <random>

#include "mb_common.h"

// ---- the wave primitives of the evaluator, emulated (every lane of the wave calls them at the same point) ---------------------------
static inline unsigned long long emu_xchg_begin(unsigned long long mine) {
    emu::Group *g = emu::g_group;
    const unsigned tid = emu::t_threadIdx.x;
    g->slot[tid] = mine;
    pthread_barrier_wait(&g->wave[tid >> 6]);
    return 0;
}
static inline unsigned long long emu_slot(unsigned lane) { return emu::g_group->slot[(emu::t_threadIdx.x & ~63u) | (lane & 63u)]; }
static inline void emu_xchg_end() { pthread_barrier_wait(&emu::g_group->wave[emu::t_threadIdx.x >> 6]); }

static inline int yd_readlane(int v, int l) { emu_xchg_begin((unsigned)v); const int o = (int)(unsigned)emu_slot((unsigned)l); emu_xchg_end(); return o; }
static inline unsigned long long yd_ballot(bool p) {
    emu_xchg_begin(p ? 1ull : 0ull);
    unsigned long long m = 0;
    for (unsigned l = 0; l < 64; l++) m |= emu_slot(l) << l;
    emu_xchg_end();
    return m;
}
static inline uint32_t yd_perm(uint32_t hi, uint32_t lo, uint32_t sel) {                          // v_perm_b32, selectors 0..7
    uint32_t out = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned s = (sel >> (8 * i)) & 0xFFu;
        if (s > 7) abort();
        out |= (s < 4 ? (lo >> (8 * s)) & 0xFFu : (hi >> (8 * (s - 4))) & 0xFFu) << (8 * i);
    }
    return out;
}
static inline uint32_t yd_alignbit(uint32_t hi, uint32_t lo, uint32_t n) { return (uint32_t)((((unsigned long long)hi << 32) | lo) >> (n & 31u)); }
static inline long long yd_clock() { return 0; }
#define YD_PIN1(a) ((void)0)
#define YD_PIN2(a, b) ((void)0)
#define YD_PIN5(a, b, c, d, e) ((void)0)
#define YD_GLOBAL_UNALIGNED __attribute__((aligned(1)))
#define yd_ld_agent(p) __atomic_load_n((p), __ATOMIC_RELAXED)
#define yd_ld_acquire(p) __atomic_load_n((p), __ATOMIC_ACQUIRE)
#define yd_st_release(p, v) __atomic_store_n((p), (v), __ATOMIC_RELEASE)
#define yd_st_agent(p, v) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
static inline void yd_st_agent4(int *p, int a, int b, int c, int d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
#define yd_fence() __sync_synchronize()
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
using std::max;
using std::min;
static int score_of(unsigned a, unsigned b);
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
inline int2 make_int2(int x, int y) { int2 v; v.x = x; v.y = y; return v; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline long long clock64() { return 0; }
template <typename T>
inline T __shfl_down(T v, int o) {                                    // lane l <- lane l + o (its own value beyond the wave's end)
    static_assert(sizeof(T) <= 8, "exchange slot is 8 bytes");
    const unsigned lane = emu::t_threadIdx.x & 63;
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    emu_xchg_begin(raw);
    raw = emu_slot(lane + (unsigned)o < 64 ? lane + (unsigned)o : lane);
    emu_xchg_end();
    T out;
    memcpy(&out, &raw, sizeof(T));
    return out;
}

namespace mb {
typedef const uint8_t *gbytes;
inline gbytes as_global(const uint8_t *p) { return p; }
constexpr int kNeg2 = -(1 << 30);
constexpr int kRowChunk = 4096;
struct RowInfo { unsigned long long off; uint32_t ly; uint32_t pad; };
inline int dpp_shr1(int v, int fill) {                                 // lane l <- lane l-1, lane 0 <- fill
    const unsigned lane = emu::t_threadIdx.x & 63;
    emu_xchg_begin((unsigned)v);
    const int o = lane ? (int)(unsigned)emu_slot(lane - 1) : fill;
    emu_xchg_end();
    return o;
}
inline int dpp_shl1(int v, int fill) {                                 // lane l <- lane l+1, lane 63 <- fill
    const unsigned lane = emu::t_threadIdx.x & 63;
    emu_xchg_begin((unsigned)v);
    const int o = lane < 63 ? (int)(unsigned)emu_slot(lane + 1) : fill;
    emu_xchg_end();
    return o;
}
inline int dpp_scan_max(int v) {                                       // inclusive prefix max over the wave
    const unsigned lane = emu::t_threadIdx.x & 63;
    emu_xchg_begin((unsigned)v);
    int m = v;
    for (unsigned l = 0; l < lane; l++) m = std::max(m, (int)(unsigned)emu_slot(l));
    emu_xchg_end();
    return m;
}
inline int uni(int v) { return yd_readlane(v, 0); }                    // readfirstlane: every lane is active wherever the evaluator pins a value
inline unsigned long long uni64(unsigned long long v) { return ((unsigned long long)(unsigned)uni((int)(v >> 32)) << 32) | (unsigned)uni((int)(unsigned)v); }
// per-row packed score table (mb_kernels.hip): byte k = HOXD70[k][bq] + 128 for k = A, C, G, T; N scores -100
inline uint32_t row_score_lut(unsigned bq) {
    const unsigned b = bq & 7u;
    uint32_t v = 0;
    for (unsigned k = 0; k < 4; k++) v |= (uint32_t)(score_of(k, b) + 128) << (8 * k);      // (score_of: the rule's own table, below)
    return v;
}
inline int lut_score(uint32_t lut, unsigned at) {
    const unsigned a = at & 7u;
    const int v = (int)((lut >> ((a & 3u) * 8u)) & 0xFFu) - 128;
    return (a & 4u) ? -100 : v;
}
inline unsigned long long dpp_scan_max64(unsigned long long v) {      // inclusive prefix max over the wave (unsigned keys)
    const unsigned lane = emu::t_threadIdx.x & 63;
    emu_xchg_begin(v);
    unsigned long long m = v;
    for (unsigned l = 0; l < lane; l++) m = std::max(m, emu_slot(l));
    emu_xchg_end();
    return m;
}
inline unsigned long long dpp_shr1_64(unsigned long long v, unsigned long long fill) {
    const unsigned lane = emu::t_threadIdx.x & 63;
    emu_xchg_begin(v);
    const unsigned long long o = lane ? emu_slot(lane - 1) : fill;
    emu_xchg_end();
    return o;
}
#include "mb_ydrop2.h"
}  // namespace mb
// (block-wide OR of a predicate: every work-item publishes its flag, all read all)
inline int __syncthreads_or(int p) {
    emu::Group *g = emu::g_group;
    const unsigned tid = emu::t_threadIdx.x, nt = emu::t_blockDim.x;
    g->slot[tid] = p ? 1ull : 0ull;
    pthread_barrier_wait(&g->all);
    int any = 0;
    for (unsigned t = 0; t < nt; t++) any |= (int)g->slot[t];
    pthread_barrier_wait(&g->all);
    return any;
}
inline unsigned long long __ballot(bool p) { return yd_ballot(p); }
#define __builtin_amdgcn_readlane(v, l) yd_readlane((v), (l))
#define __builtin_memcpy memcpy
namespace mb {
#include "mb_trace.h"
#include "mb_verify.h"
#include "mb_ydrop_lds.h"
// (the kernel proper: k_ydrop<false, false, false> of mb_kernels.hip -- the LDS ring, no walls)
void k_ydrop_lds_emu(const DpProb *probs, DpOut *outs, int n, const PairPtrs *pairs, int O, int E, int Y, uint8_t *arena, unsigned long long arena_bytes,
                     unsigned long long *arena_next, unsigned blk_bytes, unsigned long long *rowdir, uint8_t *snaps) {
    const int pi = (int)blockIdx.x;
    if (pi >= n) return;
    const DpProb pr = probs[pi];
    const PairPtrs pp = pairs[pr.pad0];
    static YdShared sh;
    static int2 sCD[kLdsRowCap];
    static uint8_t sT[kLdsRowCap];
    ydrop_body<false, false, false>(pr, &outs[pi], pp.tc, pr.strand ? pp.qr : pp.qf, O, E, Y, sCD, sT, kLdsRowCap, &sh, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps);
}
// (k_ydrop<false, false, true>: the walls variant; wref = the problem's alignments [wa0, wa1))
void k_ydrop_walls_emu(const DpProb *probs, DpOut *outs, int n, const PairPtrs *pairs, int O, int E, int Y, uint8_t *arena, unsigned long long arena_bytes,
                       unsigned long long *arena_next, unsigned blk_bytes, unsigned long long *rowdir, uint8_t *snaps, const WallSeg *wsegs, const int2 *walns, int wa0, int wa1) {
    const int pi = (int)blockIdx.x;
    if (pi >= n) return;
    const DpProb pr = probs[pi];
    const PairPtrs pp = pairs[pr.pad0];
    static YdShared sh;
    static int2 sCD[kLdsRowCap];
    static uint8_t sT[kLdsRowCap];
    ydrop_body<false, false, true>(pr, &outs[pi], pp.tc, pr.strand ? pp.qr : pp.qf, O, E, Y, sCD, sT, kLdsRowCap, &sh, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps, wsegs,
                                   walns, wa0, wa1);
}
// (the kernel proper: the __global__ wrapper of mb_kernels.hip)
void k_ydrop2_emu(const DpProb *probs, DpOut *outs, int n, const PairPtrs *pairs, const int O, const int E, const int Y, uint8_t *arena,
                  const unsigned long long arena_bytes, unsigned long long *arena_next, const unsigned blk_bytes, unsigned long long *rowdir, uint8_t *snaps,
                  const int *order, int first = 0, VerifyJob *vjobs = nullptr, int stamp = 0, int force_mod = 0) {
    ydrop2_piece(probs, outs, n, pairs, O, E, Y, arena, arena_bytes, arena_next, blk_bytes, rowdir, snaps, order, first, vjobs, stamp, force_mod);
}
}  // namespace mb

// ---- the rule, one cell at a time -----------------------------------------------------------------------------------------------
static int score_of(unsigned a, unsigned b) {
    static const int hox[4][4] = {{91, -114, -31, -123}, {-114, 100, -125, -31}, {-31, -125, 100, -114}, {-123, -31, -114, 91}};
    const unsigned x = a & 7u, y = b & 7u;
    if ((x | y) & 4u) return -100;
    return hox[x][y];
}
struct Side { int best = 0, bi = 0, bj = 0; long long cells = 0, rows = 0; std::vector<uint8_t> ops; };

// wall: (target position, query position) -> the pair lies on the path of an earlier alignment (SURVEY A.7 with the walls switch: such a
// cell is dead and no gap state survives it)
static Side one_sided(const uint8_t *tc, const uint8_t *qc, int64_t t0, int64_t q0, int dir, int64_t na, int64_t nb, int O, int E, int Y,
                      const std::function<bool(int64_t, int64_t)> *wall = nullptr) {
    const int NEG = -(1 << 29);
    Side r;
    std::vector<int> Cp((size_t)na + 4, NEG), Dp((size_t)na + 4, NEG), Cc((size_t)na + 4, NEG), Dc((size_t)na + 4, NEG);
    std::vector<int64_t> row_off, row_ly;
    std::vector<uint8_t> tr;
    int64_t R0 = 0;
    if (Y >= O) { R0 = (Y - O) / E; if (R0 > na) R0 = na; }
    int64_t LY = 0, RY = R0 + 1;
    row_off.push_back(0); row_ly.push_back(0);
    for (int64_t j = 0; j <= R0; j++) { Cp[(size_t)j] = j == 0 ? 0 : -(O + (int)j * E); Dp[(size_t)j] = NEG; tr.push_back(j == 0 ? 3 : (uint8_t)(2 | (j >= 2 ? 8 : 0))); }
    r.cells = R0 + 1;
    int64_t nrows = 1;
    int best = 0; int64_t bi = 0, bj = 0;
    for (int64_t i = 1; i <= nb; i++) {
        const uint8_t bq = qc[dir > 0 ? q0 + i - 1 : q0 - i];
        row_off.push_back((int64_t)tr.size()); row_ly.push_back(LY);
        int Iv = NEG, Cleft = NEG;
        int64_t first_alive = -1, last_alive = -1;
        for (int64_t j = LY; j <= na; j++) {
            int diag = NEG, Dv = NEG, Dext = 0, Iext = 0;
            if (j - 1 >= LY && j - 1 < RY) diag = Cp[(size_t)(j - 1)] + score_of(tc[dir > 0 ? t0 + j - 1 : t0 - j], bq);
            if (j < RY) { const int ext = Dp[(size_t)j] - E, opn = Cp[(size_t)j] - O - E; if (ext >= opn) { Dv = ext; Dext = 1; } else Dv = opn; }
            { const int ext = Iv - E, opn = Cleft - O - E; if (ext >= opn) { Iv = ext; Iext = 1; } else Iv = opn; }
            int Cv; uint8_t src;
            if (diag >= Dv && diag >= Iv) { Cv = diag; src = 0; } else if (Dv >= Iv) { Cv = Dv; src = 1; } else { Cv = Iv; src = 2; }
            r.cells++;
            const bool blocked = wall && j >= 1 && (*wall)(dir > 0 ? t0 + j - 1 : t0 - j, dir > 0 ? q0 + i - 1 : q0 - i);
            if (blocked) { Cv = NEG; Dv = NEG; Iv = NEG; }
            if (Cv > best) { best = Cv; bi = i; bj = j; }
            const bool alive = !blocked && Cv >= best - Y;
            if (!alive) Cv = NEG;
            Cc[(size_t)j] = Cv; Dc[(size_t)j] = Dv; Cleft = Cv;
            tr.push_back((uint8_t)(src | (Dext ? 4 : 0) | (Iext ? 8 : 0)));
            if (alive) { if (first_alive < 0) first_alive = j; last_alive = j; }
            else if (j >= RY) break;
        }
        nrows++;
        if (first_alive < 0) break;
        for (int64_t j = first_alive; j <= last_alive; j++) { Cp[(size_t)j] = Cc[(size_t)j]; Dp[(size_t)j] = Dc[(size_t)j]; }
        LY = first_alive; RY = last_alive + 1;
    }
    r.rows = nrows; r.best = best; r.bi = (int)bi; r.bj = (int)bj;
    int64_t i = bi, j = bj; int state = 0;
    while (i > 0 || j > 0) {
        const uint8_t tb = tr[(size_t)(row_off[(size_t)i] + (j - row_ly[(size_t)i]))];
        if (state == 0) { const int src = tb & 3; if (src == 0) { r.ops.push_back(0); i--; j--; } else if (src == 1) state = 1; else if (src == 2) state = 2; else break; }
        else if (state == 1) { r.ops.push_back(2); if (!(tb & 4)) state = 0; i--; }
        else { r.ops.push_back(3); if (!(tb & 8)) state = 0; j--; }
    }
    return r;
}

// the alignment of a piece chain read back from the kernel's trace: rows of piece p are records 0 .. of its directory entries
struct PieceRows { uint64_t row_off; int row_lo; };
static bool walk_trace(const uint8_t *arena, const unsigned long long *rowdir, const std::vector<PieceRows> &chain, int bi, int bj, std::vector<uint8_t> &ops) {
    auto code = [&](int i, int j, bool &ok) -> unsigned {
        // the piece that holds row i: the last one whose row_lo < i (row row_lo itself belongs to the piece before; row 0 to the first)
        size_t p = 0;
        for (size_t k = 0; k < chain.size(); k++) if (chain[k].row_lo < i || (i == 0 && k == 0)) p = k;
        const int rho = i - chain[p].row_lo;
        const unsigned long long chunk = rowdir[chain[p].row_off + (unsigned)(rho / mb::kRowChunk)];
        const mb::RowInfo ri = ((const mb::RowInfo *)(arena + chunk))[rho & (mb::kRowChunk - 1)];
        const int rel = j - (int)ri.ly;
        if (rel < 0) { ok = false; return 0; }
        const uint8_t b = arena[ri.off + (unsigned)(rel >> 1)];
        return (rel & 1) ? b >> 4 : b & 15u;
    };
    int i = bi, j = bj, state = 0;
    bool ok = true;
    while ((i > 0 || j > 0) && ok) {
        const unsigned tb = code(i, j, ok);
        if (state == 0) { const unsigned src = tb & 3u; if (src == 0) { ops.push_back(0); i--; j--; } else if (src == 1) state = 1; else if (src == 2) state = 2; else break; }
        else if (state == 1) { ops.push_back(2); if (!(tb & 4u)) state = 0; i--; }
        else { ops.push_back(3); if (!(tb & 8u)) state = 0; j--; }
        if (i < 0 || j < 0) ok = false;
    }
    return ok;
}

// the side's alignment through the traceback KERNELS (mb_trace.h): one walker per piece from the best cell's piece back to the head, the
// guessed start of every other piece = the best cell of its last row (exit_j), join walks from predicted entries (k_trace_prejoin), the
// stitch (k_trace_join); the segments' runs expanded to one op per column.  pieces: head first; (dr, dc) = a piece's origin in the
// coordinates of the piece before it.
struct ChainPiece { uint64_t row_off; int row_lo, floor, stop_row, exit_j, dr, dc; };
static bool trace_kernels(const uint8_t *arena, unsigned long long arena_bytes, const unsigned long long *rowdir, const std::vector<ChainPiece> &pieces, int bi, int bj, int poison,
                          std::vector<uint8_t> &ops_out) {
    std::vector<mb::TbWalk> tbw;
    uint64_t ooff = 0, roff = 0, side_slots = 0;
    for (size_t x = pieces.size(); x-- > 0;) {
        const ChainPiece &pc = pieces[x];
        mb::TbWalk w;
        memset(&w, 0, sizeof w);
        w.row_off = pc.row_off; w.row_lo = pc.row_lo; w.floor = x > 0 ? pc.floor : -1;
        if (x + 1 == pieces.size()) { w.si = bi; w.sj = bj; } else { w.si = pc.stop_row; w.sj = pc.exit_j; }
        if (x > 0) { w.dr = pc.dr; w.dc = pc.dc; }
        const uint64_t rows = (uint64_t)(w.si - w.floor), slots = 2 * rows + 2 * 2048 + 8;
        w.ops_off = ooff; ooff += slots; side_slots += slots;
        w.rec_off = roff; roff += 3 * rows;
        tbw.push_back(w);
    }
    mb::TbSide ts;
    memset(&ts, 0, sizeof ts);
    ts.first_walk = 0; ts.n_walks = (int32_t)tbw.size(); ts.jops_off = ooff; ooff += side_slots; ts.seg_off = 0;
    for (size_t x = 0; x < tbw.size(); x++) { const unsigned long long jo = x == 0 ? ~0ull : ts.jops_off + (tbw[x].ops_off - tbw[0].ops_off); memcpy(tbw[x].pad, &jo, 8); }
    std::vector<uint32_t> ops((size_t)ooff + 64, 0xDDDDDDDDu), recs((size_t)roff + 64, 0xBBBBBBBBu);
    std::vector<mb::TbSeg> segs(2 * tbw.size() + 2);
    std::vector<mb::TbJoin> joins(tbw.size());
    hipLaunchKernelGGL(mb::k_trace_walk, dim3((unsigned)tbw.size()), dim3(64), 0, nullptr, tbw.data(), (int)tbw.size(), arena, arena_bytes, rowdir, ops.data(), recs.data());
    hipLaunchKernelGGL(mb::k_trace_prejoin, dim3((unsigned)tbw.size()), dim3(64), 0, nullptr, tbw.data(), (int)tbw.size(), joins.data(), arena, arena_bytes, rowdir, ops.data(), recs.data(), poison);
    hipLaunchKernelGGL(mb::k_trace_join, dim3(1), dim3(64), 0, nullptr, &ts, 1, tbw.data(), segs.data(), arena, arena_bytes, rowdir, ops.data(), recs.data(), joins.data());
    ops_out.clear();
    for (int q = 0; q < ts.n_segs; q++) {
        const mb::TbSeg &sg = segs[(size_t)q];
        for (int k = 0; k < sg.n_runs; k++) {
            const uint32_t o = ops[(size_t)sg.src + (size_t)k];
            int len = (int)(o >> 2) - (k == 0 ? sg.first_sub : 0);
            if (len < 0 || (o & 3u) == 1u) return false;
            for (int c = 0; c < len; c++) ops_out.push_back((uint8_t)(o & 3u));
        }
    }
    return true;
}

// ---- inline mode: the hand-over INSIDE the launch (mb_ydrop2.h).  A head piece aimed at relay A, A aimed at relay B, B running to its end --
//      both relays started cold on the side's own path, as the gapped stage plants them --, all three in ONE launch (the emulation runs the
//      blocks one after another: relays first, so that their entry snapshots are there when the upstream piece looks).  The pieces check their
//      hand-overs themselves and go on where one is rejected (every first check is rejected on purpose in the second pass).  Whatever they
//      decide, following the chain as the host does -- k_verify on the hand-over each piece says it ended at -- must give the side's best cell,
//      score and cell count.
static int inline_cases(unsigned seed0, int n_cases) {
    int bad = 0, concluded = 0, went_on = 0;
    const unsigned long long arena_bytes = 96ull << 20;
    std::vector<uint8_t> arena((size_t)arena_bytes);
    for (int cs = 0; cs < n_cases; cs++) {
        std::mt19937 rng(seed0 * 2654435761u + (unsigned)cs);
        auto rnd = [&](int n) { return (int)(rng() % (unsigned)n); };
        const int O = 400, E = 30;
        const int Y = cs % 3 == 0 ? 3000 : cs % 3 == 1 ? 4000 : 2000 + rnd(2500);
        const int dir = cs % 2 ? -1 : 1;
        const int len = 700 + rnd(300);
        const int64_t tn = len + rnd(40), qn = len + rnd(40);
        std::vector<uint8_t> tb((size_t)tn + 2 * mb::kDevPad + 16, mb::kSep), qb((size_t)qn + 2 * mb::kDevPad + 16, mb::kSep);
        uint8_t *tc = tb.data() + mb::kDevPad, *qc = qb.data() + mb::kDevPad;
        for (int64_t i = 0; i < tn; i++) tc[i] = (uint8_t)rnd(4);
        {
            const int div = 3 + rnd(14);
            int64_t ti = 0;
            for (int64_t qi = 0; qi < qn; qi++) {
                if (rnd(1000) < 8) ti += 1 + rnd(5);
                if (rnd(1000) < 8) { qc[qi] = (uint8_t)rnd(4); continue; }
                qc[qi] = ti < tn && rnd(100) >= div ? (tc[ti] & 3) : (uint8_t)rnd(4);
                ti++;
            }
        }
        for (int s = 0; s < 40; s++) { tc[rnd((int)tn)] |= 8; qc[rnd((int)qn)] |= 8; }
        const int64_t t0 = dir > 0 ? rnd(10) : tn - rnd(10), q0 = dir > 0 ? rnd(10) : qn - rnd(10);
        const int64_t na = dir > 0 ? tn - t0 : t0, nb = dir > 0 ? qn - q0 : q0;
        const Side want = one_sided(tc, qc, t0, q0, dir, na, nb, O, E, Y);
        mb::PairPtrs pp; pp.tc = tc; pp.qf = qc; pp.qr = qc;
        std::vector<std::pair<int, int>> diag_cells;
        { int i = 0, j = 0; for (size_t k = want.ops.size(); k-- > 0;) { const uint8_t o = want.ops[k]; if (o == 0) { i++; j++; diag_cells.push_back({i, j}); } else if (o == 2) i++; else j++; } }
        if (want.bi < 500) { printf("case %d: side too short (%d rows)\n", cs, want.bi); continue; }
        const int w = 40 + rnd(50);
        std::pair<int, int> atA{-1, -1}, atB{-1, -1};
        for (const auto &c : diag_cells) { if (c.first <= want.bi * 3 / 10) atA = c; if (c.first <= want.bi * 6 / 10) atB = c; }
        if (atA.first < 1 || atB.first <= atA.first + w) { printf("case %d: no room for two relays\n", cs); continue; }
        for (int pass = 0; pass < 3; pass++) {              // 1: every first check rejected on purpose; 2: the first three -- past the relay's last snapshot, on to the relay after
            std::fill(arena.begin(), arena.begin() + (64 << 20), (uint8_t)0xEE);
            std::vector<mb::DpProb> probs(3);
            std::vector<mb::DpOut> outs(3);
            std::vector<mb::VerifyJob> vjobs(2, mb::VerifyJob{-1, -1, 0, 0});
            std::vector<unsigned long long> rowdir(64, ~0ull);
            std::vector<uint8_t> snaps((size_t)3 * mb::kSnapSlots * mb::kSnapBytes, 0x11);      // (stale bytes: no header carries the stamp)
            unsigned long long arena_next = 0;
            memset(probs.data(), 0, sizeof(mb::DpProb) * 3);
            mb::DpProb &hd = probs[0], &ra = probs[1], &rb = probs[2];
            hd.t0 = (int32_t)t0; hd.q0 = (int32_t)q0; hd.na = (int32_t)na; hd.nb = (int32_t)nb; hd.dir = dir; hd.row_off = 0;
            hd.snap_row = 0; hd.init_snap = -1; hd.snap_idx = 0; hd.snap_row2 = hd.snap_row3 = 0;
            ra = hd; rb = hd;
            ra.t0 = (int32_t)(t0 + dir * atA.second); ra.q0 = (int32_t)(q0 + dir * atA.first); ra.na = (int32_t)(na - atA.second); ra.nb = (int32_t)(nb - atA.first);
            rb.t0 = (int32_t)(t0 + dir * atB.second); rb.q0 = (int32_t)(q0 + dir * atB.first); rb.na = (int32_t)(na - atB.second); rb.nb = (int32_t)(nb - atB.first);
            ra.row_off = 8; rb.row_off = 16; ra.snap_idx = mb::kSnapSlots; rb.snap_idx = 2 * mb::kSnapSlots;
            hd.stop_row = atA.first + w; hd.aim1 = 2; hd.vjob1 = 1; hd.cap_row = (int32_t)nb;
            ra.snap_row = w; ra.stop_row = (atB.first - atA.first) + w; ra.aim1 = 3; ra.vjob1 = 2; ra.cap_row = ra.nb;
            if (2 * w < ra.stop_row) ra.snap_row2 = 2 * w;
            if (4 * w < ra.stop_row) ra.snap_row3 = 4 * w;
            rb.snap_row = w; rb.snap_row2 = 2 * w; rb.snap_row3 = 4 * w; rb.stop_row = 0;
            const int order[3] = {2, 1, 0};
            const int stamp = 1000 + cs;
            hipLaunchKernelGGL(mb::k_ydrop2_emu, dim3(3), dim3(64), 0, nullptr, probs.data(), outs.data(), 3, &pp, O, E, Y, arena.data(), arena_bytes, &arena_next, 64u << 10,
                               rowdir.data(), snaps.data(), order, 0, vjobs.data(), stamp, pass == 0 ? 0 : pass == 1 ? 1 : -3);
            bool ok = true, concl = true;
            const char *why = "";
            int cur = 0, gbest = -1, gbi = 0, gbj = 0;
            long long c_off = 0, cells = 0, entry_cells = 0;
            int hops = 0, checks = outs[0].fin_checks + outs[1].fin_checks;
            for (;; hops++) {
                const mb::DpOut &o = outs[(size_t)cur];
                if (o.overflow) { ok = false; why = "overflow"; break; }
                const int dr = (probs[(size_t)cur].q0 - hd.q0) * dir, dc = (probs[(size_t)cur].t0 - hd.t0) * dir;
                if ((long long)o.best + c_off > gbest) { gbest = (int)(o.best + c_off); gbi = o.bi + dr; gbj = o.bj + dc; }
                cells += o.cells - entry_cells;
                if (!o.stopped) break;
                const int aim = o.fin_aim1 - 1, ck = o.fin_ck;
                if (aim <= cur || aim > 2 || probs[(size_t)cur].vjob1 == 0) { ok = false; why = "a piece stopped without a relay ahead"; break; }
                const mb::VerifyJob vj{mb::kSnapSlots * cur + 1, mb::kSnapSlots * aim + (ck == 0 ? 0 : ck + 1), (probs[(size_t)aim].t0 - probs[(size_t)cur].t0) * dir,
                                       (probs[(size_t)aim].q0 - probs[(size_t)cur].q0) * dir};
                if (memcmp(&vj, &vjobs[(size_t)probs[(size_t)cur].vjob1 - 1], sizeof vj)) { ok = false; why = "the piece's own VerifyJob is not the hand-over it ended at"; break; }
                mb::VerifyOut vo;
                hipLaunchKernelGGL(mb::k_verify, dim3(1), dim3(256), 0, nullptr, &vj, &vo, 1, snaps.data(), Y, E);
                // the piece's own verdict: it stopped after a check of its own (fin_checks > 0: the last one accepted) or because it had no room / no snapshot
                if (!vo.ok) { concl = false; break; }
                c_off += vo.c; entry_cells = vo.n_cells; cur = aim;
            }
            if (ok && concl) {
                concluded++;
                if (gbest != want.best || gbi != want.bi || gbj != want.bj) { ok = false; why = "best cell / score"; }
                else if (cells != want.cells) { ok = false; why = "cells"; }
            }
            if (pass >= 1 && outs[0].fin_checks > 0 && outs[0].fin_stop == hd.stop_row && outs[0].stopped) { ok = false; why = "a first check that was to be rejected ended the piece"; }
            if (outs[0].fin_stop > hd.stop_row || outs[1].fin_stop > ra.stop_row) went_on++;
            printf("case %d pass %d: dir %+d ydrop %d, side of %d rows, relays at rows %d and %d, warm-up %d: head ended at row %d (aim %d, snapshot %d), relay A at its row %d (aim %d, snapshot %d), %d checks inside the launch, chain of %d: %s  %s %s\n",
                   cs, pass, dir, Y, want.bi, atA.first, atB.first, w, outs[0].fin_stop, outs[0].fin_aim1 - 1, outs[0].fin_ck, outs[1].fin_stop, outs[1].fin_aim1 - 1, outs[1].fin_ck, checks, hops + 1,
                   concl ? "concluded" : "last hand-over rejected", ok ? "ok" : "MISMATCH", why);
            if (!ok) bad++;
        }
    }
    printf("inline: %d chains concluded, %d pieces went on past their first stop\n", concluded, went_on);
    if (!concluded || !went_on) { printf("nothing concluded or nothing went on: the test says nothing  MISMATCH\n"); bad++; }
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    const unsigned seed0 = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const int n_cases = argc > 2 ? atoi(argv[2]) : 4;
    const bool relay_mode = argc > 3 && !strcmp(argv[3], "relay");      // the exactness of an accepted relay hand-over instead of the cut / check tests
    const bool inline_mode = argc > 3 && !strcmp(argv[3], "inline");    // the hand-over checked (and, rejected, continued) inside the launch
    if (inline_mode) return inline_cases(seed0, n_cases);
    int bad = 0, relays_accepted = 0, relays_tried = 0, n_wall_cases = 0, n_wall_changed = 0;
    const unsigned long long arena_bytes = 96ull << 20;
    std::vector<uint8_t> arena((size_t)arena_bytes);
    for (int cs = 0; cs < n_cases; cs++) {
        std::mt19937 rng(seed0 * 32452843u + (unsigned)cs);
        auto rnd = [&](int n) { return (int)(rng() % (unsigned)n); };
        const int O = 400, E = 30;
        const int Y = cs % 4 == 0 ? 3000 : cs % 4 == 1 ? 9400 : cs % 4 == 2 ? 600 + rnd(1500) : 4000 + rnd(5000);      // (9400: rows that need the second group of columns)
        const int dir = cs % 2 ? -1 : 1;
        const int len = 120 + rnd(140);
        // two sequences with a common origin: substitutions, a few indels, an N, soft-masked stretches; the side runs into the contig ends
        const int64_t tn = len + rnd(60), qn = len + rnd(60);
        std::vector<uint8_t> tb((size_t)tn + 2 * mb::kDevPad + 16, mb::kSep), qb((size_t)qn + 2 * mb::kDevPad + 16, mb::kSep);
        uint8_t *tc = tb.data() + mb::kDevPad, *qc = qb.data() + mb::kDevPad;
        for (int64_t i = 0; i < tn; i++) tc[i] = (uint8_t)rnd(4);
        {
            const int div = 1 + rnd(cs % 3 == 0 ? 12 : 30);
            int64_t ti = 0;
            for (int64_t qi = 0; qi < qn; qi++) {
                if (rnd(100) < 2) ti += 1 + rnd(6);                      // deletion in the query
                if (rnd(100) < 2) { qc[qi] = (uint8_t)rnd(4); continue; } // insertion
                qc[qi] = ti < tn && rnd(100) >= div ? (tc[ti] & 3) : (uint8_t)rnd(4);
                ti++;
            }
        }
        if (rnd(2)) tc[rnd((int)tn)] = 4;
        for (int s = 0; s < 20; s++) { tc[rnd((int)tn)] |= 8; qc[rnd((int)qn)] |= 8; }
        // the side: from an anchor near one end towards the other
        const int64_t t0 = dir > 0 ? rnd(10) : tn - rnd(10), q0 = dir > 0 ? rnd(10) : qn - rnd(10);
        const int64_t na = dir > 0 ? tn - t0 : t0, nb = dir > 0 ? qn - q0 : q0;
        const Side want = one_sided(tc, qc, t0, q0, dir, na, nb, O, E, Y);
        // ---- the whole side as one piece, then cut in two at a row the DP is still alive at
        mb::PairPtrs pp; pp.tc = tc; pp.qf = qc; pp.qr = qc;
        bool ok = true;
        const char *why = "";
        // ---- relay mode: an ACCEPTED hand-over is exact.  The upstream piece runs from the side's origin and stops after row r (exit state);
        //      a relay starts COLD at a cell of the side's own path w rows before r, as the gapped stage starts one at a downstream anchor,
        //      and takes its entry snapshot after its local row w (the same row of the side); k_verify compares the two under the relay's
        //      offsets.  If it accepts, the relay continued from that snapshot must end at the side's best cell with the side's score.
        if (relay_mode) {
            // the path's cells from the origin outwards (ops come in walk-back order)
            std::vector<std::pair<int, int>> diag_cells;                 // cells reached by a diagonal step
            { int i = 0, j = 0; for (size_t k = want.ops.size(); k-- > 0;) { const uint8_t o = want.ops[k]; if (o == 0) { i++; j++; diag_cells.push_back({i, j}); } else if (o == 2) i++; else j++; } }
            const int r = (int)(want.bi * 6 / 10), w = std::min(r - 4, 60 + rnd(90));
            std::pair<int, int> at{-1, -1};
            for (const auto &c : diag_cells) if (c.first <= r - w) at = c;
            if (want.bi < 40 || w < 20 || at.first < 1) { printf("case %d: side too short for a relay\n", cs); continue; }
            relays_tried++;
            const int i0 = at.first, j0 = at.second;                        // the relay's origin: rows / columns of the side before it
            std::fill(arena.begin(), arena.begin() + (64 << 20), (uint8_t)0xEE);
            std::vector<mb::DpProb> probs(3);
            std::vector<mb::DpOut> outs(3);
            std::vector<unsigned long long> rowdir(64, ~0ull);
            std::vector<uint8_t> snaps((size_t)3 * mb::kSnapSlots * mb::kSnapBytes, 0);
            unsigned long long arena_next = 0;
            memset(probs.data(), 0, sizeof(mb::DpProb) * 3);
            mb::DpProb &up = probs[0], &rl = probs[1], &ct = probs[2];
            up.t0 = (int32_t)t0; up.q0 = (int32_t)q0; up.na = (int32_t)na; up.nb = (int32_t)nb; up.dir = dir; up.row_off = 0;
            up.stop_row = r; up.snap_row = -1; up.init_snap = -1; up.snap_idx = 0; up.snap_row2 = -1; up.snap_row3 = -1;          // exit: slot 1
            rl = up;
            rl.t0 = (int32_t)(t0 + dir * j0); rl.q0 = (int32_t)(q0 + dir * i0); rl.na = (int32_t)(na - j0); rl.nb = (int32_t)(nb - i0); rl.row_off = 8;
            rl.stop_row = r - i0; rl.snap_idx = mb::kSnapSlots;                                                                     // the relay stopped after the same row: slot 5
            hipLaunchKernelGGL(mb::k_ydrop2_emu, dim3(2), dim3(64), 0, nullptr, probs.data(), outs.data(), 2, &pp, O, E, Y, arena.data(), arena_bytes, &arena_next, 64u << 10,
                               rowdir.data(), snaps.data(), (const int *)nullptr);
            if (outs[0].overflow || outs[1].overflow || !outs[0].stopped) { printf("case %d: relay setup failed  MISMATCH\n", cs); bad++; continue; }
            mb::VerifyJob vj{1, mb::kSnapSlots + 1, j0, i0};
            mb::VerifyOut vo;
            hipLaunchKernelGGL(mb::k_verify, dim3(1), dim3(256), 0, nullptr, &vj, &vo, 1, snaps.data(), Y, E);
            bool exact = true;
            if (vo.ok && outs[1].stopped) {
                relays_accepted++;
                ct = rl; ct.row_lo = r - i0; ct.row_off = 16; ct.stop_row = -1; ct.init_snap = mb::kSnapSlots + 1; ct.snap_idx = 2 * mb::kSnapSlots;
                hipLaunchKernelGGL(mb::k_ydrop2_emu, dim3(1), dim3(64), 0, nullptr, probs.data() + 2, outs.data() + 2, 1, &pp, O, E, Y, arena.data(), arena_bytes, &arena_next,
                                   64u << 10, rowdir.data(), snaps.data(), (const int *)nullptr);
                const mb::DpOut &f = outs[2];
                // (the side's best lies beyond the hand-over row by the choice of r; the relay counts rows, columns and scores from its origin)
                exact = !f.overflow && f.best + vo.c == want.best && f.bi + i0 == want.bi && f.bj + j0 == want.bj;
                // the path from the best cell back to the hand-over row is the side's own
                if (exact) {
                    std::vector<uint8_t> ops;
                    std::vector<PieceRows> chain{{8, 0}, {16, r - i0}};
                    exact = walk_trace(arena.data(), rowdir.data(), chain, f.bi, f.bj, ops);
                    size_t n_after = 0;                                  // ops of the side's walk-back until it reaches row r
                    { int i = want.bi; for (; n_after < want.ops.size() && i > r; n_after++) if (want.ops[n_after] != 3) i--; }
                    exact = exact && ops.size() >= n_after && std::equal(ops.begin(), ops.begin() + (long)n_after, want.ops.begin());
                }
                // the whole alignment through the traceback kernels: the relay's continuation back to the hand-over row, then the upstream
                // piece in ITS coordinates (origin offset i0 rows / j0 columns), entered wherever the path crosses row r
                if (exact) {
                    std::vector<uint8_t> ops;
                    std::vector<ChainPiece> cps{{0, 0, -1, r, outs[0].exit_j, 0, 0}, {16, r - i0, r - i0, -1, 0, i0, j0}};
                    for (int poison = 0; poison < 2 && exact; poison++)
                        exact = trace_kernels(arena.data(), arena_bytes, rowdir.data(), cps, f.bi, f.bj, poison, ops) && ops == want.ops;
                }
            }
            printf("case %d: dir %+d, ydrop %d, side of %d rows, hand-over after row %d, relay from (%d, %d) with %d rows of warm-up: %s  %s\n", cs, dir, Y, want.bi, r, i0, j0,
                   r - i0, vo.ok ? "accepted" : "rejected", exact ? "ok" : "MISMATCH");
            if (!exact) bad++;
            continue;
        }
        for (int mode = 0; mode < 3 && ok; mode++) {
            if (mode == 2) {
                // the four-wave kernel body with the previous row in an LDS ring (mb_ydrop_lds.h: the rerun of rows that outgrow the one-wave
                // kernels): the whole side, same format of trace and row records
                std::fill(arena.begin(), arena.begin() + (64 << 20), (uint8_t)0xEE);
                std::vector<mb::DpProb> probs(1);
                std::vector<mb::DpOut> outs(1);
                std::vector<unsigned long long> rowdir(64, ~0ull);
                std::vector<uint8_t> snaps((size_t)mb::kSnapSlots * mb::kSnapBytes, 0x5A);
                unsigned long long arena_next = 0;
                memset(probs.data(), 0, sizeof(mb::DpProb)); memset(outs.data(), 0xAA, sizeof(mb::DpOut));
                mb::DpProb &a = probs[0];
                a.t0 = (int32_t)t0; a.q0 = (int32_t)q0; a.na = (int32_t)na; a.nb = (int32_t)nb; a.dir = dir; a.row_lo = 0; a.row_off = 0;
                a.stop_row = -1; a.snap_row = -1; a.init_snap = -1; a.snap_idx = -1; a.snap_row2 = -1; a.snap_row3 = -1;
                hipLaunchKernelGGL(mb::k_ydrop_lds_emu, dim3(1), dim3(256), 0, nullptr, probs.data(), outs.data(), 1, &pp, O, E, Y, arena.data(), arena_bytes, &arena_next, 64u << 10,
                                   rowdir.data(), snaps.data());
                const mb::DpOut &fin = outs[0];
                if (fin.overflow) { ok = false; why = "overflow (LDS kernel)"; break; }
                if (fin.best != want.best || fin.bi != want.bi || fin.bj != want.bj) { ok = false; why = "best cell (LDS kernel)"; break; }
                if (fin.cells != want.cells || fin.rows != want.rows) { ok = false; why = "cells / rows (LDS kernel)"; break; }
                std::vector<uint8_t> ops;
                std::vector<PieceRows> chain{{0, 0}};
                if (!walk_trace(arena.data(), rowdir.data(), chain, fin.bi, fin.bj, ops) || ops != want.ops) { ok = false; why = "trace (LDS kernel)"; }
                break;
            }
            const int cut = mode == 0 ? 0 : (int)std::max<long long>(1, std::min<long long>(want.rows - 2, 20 + rnd(80)));
            if (mode == 1 && want.rows < 8) continue;
            std::fill(arena.begin(), arena.begin() + (64 << 20), (uint8_t)0xEE);
            std::vector<mb::DpProb> probs(2);
            std::vector<mb::DpOut> outs(2);
            std::vector<unsigned long long> rowdir(64, ~0ull);
            std::vector<uint8_t> snaps((size_t)2 * mb::kSnapSlots * mb::kSnapBytes, 0x5A);
            unsigned long long arena_next = 0;
            memset(probs.data(), 0, sizeof(mb::DpProb) * 2); memset(outs.data(), 0xAA, sizeof(mb::DpOut) * 2);
            mb::DpProb &a = probs[0];
            a.t0 = (int32_t)t0; a.q0 = (int32_t)q0; a.na = (int32_t)na; a.nb = (int32_t)nb; a.dir = dir; a.strand = 0; a.pad0 = 0; a.row_lo = 0; a.row_off = 0;
            a.stop_row = mode == 1 ? cut : -1; a.snap_row = mode == 1 ? std::max(1, cut / 2) : -1; a.init_snap = -1; a.snap_idx = 0; a.snap_row2 = -1; a.snap_row3 = -1;
            hipLaunchKernelGGL(mb::k_ydrop2_emu, dim3(1), dim3(64), 0, nullptr, probs.data(), outs.data(), 1, &pp, O, E, Y, arena.data(), arena_bytes, &arena_next, 64u << 10,
                               rowdir.data(), snaps.data(), (const int *)nullptr);
            std::vector<PieceRows> chain{{0, 0}};
            mb::DpOut fin = outs[0];
            if (fin.overflow) { ok = false; why = "overflow"; break; }
            if (mode == 1) {
                if (!fin.stopped) { ok = false; why = "the first piece did not stop at its stop row"; break; }
                const mb::SnapHdr *h = (const mb::SnapHdr *)(snaps.data() + (size_t)1 * mb::kSnapBytes);
                if (!h->valid || h->row != cut) { ok = false; why = "exit snapshot"; break; }
                mb::DpProb &b = probs[1];
                b = a; b.row_lo = cut; b.row_off = 8; b.stop_row = -1; b.snap_row = -1; b.init_snap = 1; b.snap_idx = mb::kSnapSlots;
                hipLaunchKernelGGL(mb::k_ydrop2_emu, dim3(1), dim3(64), 0, nullptr, probs.data() + 1, outs.data() + 1, 1, &pp, O, E, Y, arena.data(), arena_bytes, &arena_next,
                                   64u << 10, rowdir.data(), snaps.data(), (const int *)nullptr);
                fin = outs[1];
                if (fin.overflow) { ok = false; why = "overflow (second piece)"; break; }
                chain.push_back({8, cut});
            }
            if (fin.best != want.best || fin.bi != want.bi || fin.bj != want.bj) {
                if (getenv("EMU_YDROP_DEBUG")) fprintf(stderr, "mode %d: got best %d at (%d, %d), %lld cells, %d rows; want %d at (%d, %d)\n", mode, fin.best, fin.bi, fin.bj, (long long)fin.cells, fin.rows, want.best, want.bi, want.bj);
                ok = false; why = "best cell"; break;
            }
            if (fin.cells != want.cells || fin.rows != want.rows) { ok = false; why = "cells / rows"; break; }
            std::vector<uint8_t> ops;
            if (!walk_trace(arena.data(), rowdir.data(), chain, fin.bi, fin.bj, ops) || ops != want.ops) { ok = false; why = "trace"; break; }
            // ... and the same alignment through the traceback kernels, with every other join prediction made wrong on purpose the second time
            std::vector<ChainPiece> cps{{0, 0, -1, cut, outs[0].exit_j, 0, 0}};
            if (mode == 1) cps.push_back({8, cut, cut, -1, 0, 0, 0});
            for (int poison = 0; poison < 2 && ok; poison++)
                if (!trace_kernels(arena.data(), arena_bytes, rowdir.data(), cps, fin.bi, fin.bj, poison, ops) || ops != want.ops) { ok = false; why = "traceback kernels"; }
        }
        // ---- walls (miblast_params.walls; the WALLS variant of the LDS body): earlier alignments = stretches of this side's own path moved a few
        //      columns aside, so that the DP has to run along them, across their gaps and around their ends
        if (ok && !relay_mode && want.ops.size() > 60) {
            std::vector<std::pair<int64_t, int64_t>> cells;                // (q, t) of the side's aligned pairs, absolute
            { int i = 0, j = 0; for (size_t k = want.ops.size(); k-- > 0;) { const uint8_t o = want.ops[k]; if (o == 0) { i++; j++; cells.push_back({dir > 0 ? q0 + i - 1 : q0 - i, dir > 0 ? t0 + j - 1 : t0 - j}); } else if (o == 2) i++; else j++; } }
            std::vector<mb::WallSeg> wsegs;
            std::vector<int2> walns;
            std::vector<std::vector<std::pair<int64_t, int64_t>>> wall_cells;
            const int n_walls = 1 + rnd(3);
            for (int a = 0; a < n_walls; a++) {
                const size_t c0 = (size_t)rnd((int)cells.size() / 2), c1 = std::min(cells.size(), c0 + 20 + (size_t)rnd((int)cells.size() / 2));
                const int shift = (rnd(2) ? 1 : -1) * (1 + rnd(6));
                std::vector<std::pair<int64_t, int64_t>> w;
                for (size_t k = c0; k < c1; k++) { const int64_t t = cells[k].second + shift; if (t >= 0 && t < tn) w.push_back({cells[k].first, t}); }
                std::sort(w.begin(), w.end());
                const int first = (int)wsegs.size();
                for (size_t k = 0; k < w.size();) {                      // gap-free runs, q ascending
                    size_t e = k + 1;
                    while (e < w.size() && w[e].first == w[e - 1].first + 1 && w[e].second == w[e - 1].second + 1) e++;
                    mb::WallSeg sg; sg.q0 = (int32_t)w[k].first; sg.t0 = (int32_t)w[k].second; sg.len = (int32_t)(e - k);
                    wsegs.push_back(sg);
                    k = e;
                }
                int2 al; al.x = first; al.y = (int)wsegs.size();
                walns.push_back(al);
                wall_cells.push_back(w);
            }
            const std::function<bool(int64_t, int64_t)> wall = [&](int64_t t, int64_t q) {
                for (const auto &w : wall_cells) if (std::binary_search(w.begin(), w.end(), std::make_pair(q, t))) return true;
                return false;
            };
            const Side ww = one_sided(tc, qc, t0, q0, dir, na, nb, O, E, Y, &wall);
            std::fill(arena.begin(), arena.begin() + (64 << 20), (uint8_t)0xEE);
            std::vector<mb::DpProb> probs(1);
            std::vector<mb::DpOut> outs(1);
            std::vector<unsigned long long> rowdir(64, ~0ull);
            std::vector<uint8_t> snaps((size_t)mb::kSnapSlots * mb::kSnapBytes, 0x5A);
            unsigned long long arena_next = 0;
            memset(probs.data(), 0, sizeof(mb::DpProb)); memset(outs.data(), 0xAA, sizeof(mb::DpOut));
            mb::DpProb &a = probs[0];
            a.t0 = (int32_t)t0; a.q0 = (int32_t)q0; a.na = (int32_t)na; a.nb = (int32_t)nb; a.dir = dir; a.row_lo = 0; a.row_off = 0;
            a.stop_row = -1; a.snap_row = -1; a.init_snap = -1; a.snap_idx = -1; a.snap_row2 = -1; a.snap_row3 = -1;
            hipLaunchKernelGGL(mb::k_ydrop_walls_emu, dim3(1), dim3(256), 0, nullptr, probs.data(), outs.data(), 1, &pp, O, E, Y, arena.data(), arena_bytes, &arena_next, 64u << 10,
                               rowdir.data(), snaps.data(), wsegs.data(), walns.data(), 0, (int)walns.size());
            const mb::DpOut &fin = outs[0];
            std::vector<uint8_t> ops;
            std::vector<PieceRows> chain{{0, 0}};
            if (fin.overflow || fin.best != ww.best || fin.bi != ww.bi || fin.bj != ww.bj || fin.cells != ww.cells || fin.rows != ww.rows ||
                !walk_trace(arena.data(), rowdir.data(), chain, fin.bi, fin.bj, ops) || ops != ww.ops) {
                if (getenv("EMU_YDROP_DEBUG")) fprintf(stderr, "walls: got best %d at (%d, %d), %lld cells, %d rows, overflow %d; want %d at (%d, %d), %lld cells, %lld rows\n", fin.best, fin.bi, fin.bj,
                                                        (long long)fin.cells, fin.rows, fin.overflow, ww.best, ww.bi, ww.bj, ww.cells, ww.rows);
                ok = false; why = "walls";
            }
            n_wall_cases++; n_wall_changed += ww.best != want.best || ww.ops != want.ops;
        }
        // ---- the hand-over check (k_verify) on states the evaluator writes: a piece's entry snapshot after row r against the exit snapshot
        //      of the same DP stopped at r (equal: accepted, c = 0); the same state with every live value and the best moved by one
        //      constant (accepted, c = that constant) and shifted by whole columns and rows (accepted under the job's shift / drow); one
        //      live C or one D that can still matter changed (rejected); a D below what can ever matter again changed (accepted)
        if (ok && want.rows >= 12) {
            const int r = (int)std::max<long long>(2, std::min<long long>(want.rows - 3, 10 + rnd(60)));
            std::fill(arena.begin(), arena.begin() + (64 << 20), (uint8_t)0xEE);
            std::vector<mb::DpProb> probs(1);
            std::vector<mb::DpOut> outs(1);
            std::vector<unsigned long long> rowdir(64, ~0ull);
            std::vector<uint8_t> snaps((size_t)8 * mb::kSnapBytes, 0);
            unsigned long long arena_next = 0;
            memset(probs.data(), 0, sizeof(mb::DpProb));
            mb::DpProb &a = probs[0];
            a.t0 = (int32_t)t0; a.q0 = (int32_t)q0; a.na = (int32_t)na; a.nb = (int32_t)nb; a.dir = dir; a.row_lo = 0; a.row_off = 0;
            a.stop_row = -1; a.snap_row = r; a.init_snap = -1; a.snap_idx = 0; a.snap_row2 = -1; a.snap_row3 = -1;      // the entry snapshot after row r: slot 0
            hipLaunchKernelGGL(mb::k_ydrop2_emu, dim3(1), dim3(64), 0, nullptr, probs.data(), outs.data(), 1, &pp, O, E, Y, arena.data(), arena_bytes, &arena_next, 64u << 10,
                               rowdir.data(), snaps.data(), (const int *)nullptr);
            a.stop_row = r; a.snap_row = -1; a.snap_idx = 4;                                                              // the same DP stopped at r: exit snapshot in slot 5
            hipLaunchKernelGGL(mb::k_ydrop2_emu, dim3(1), dim3(64), 0, nullptr, probs.data(), outs.data(), 1, &pp, O, E, Y, arena.data(), arena_bytes, &arena_next, 64u << 10,
                               rowdir.data(), snaps.data(), (const int *)nullptr);
            auto hdr = [&](int slot) { return (mb::SnapHdr *)(snaps.data() + (size_t)slot * mb::kSnapBytes); };
            auto Cof = [&](int slot) { return (int *)(snaps.data() + (size_t)slot * mb::kSnapBytes + sizeof(mb::SnapHdr)); };
            auto Dof = [&](int slot) { return Cof(slot) + mb::kSnapCols; };
            const int kE = 5;                                                // the upstream piece's exit state
            if (!hdr(0)->valid || !hdr(kE)->valid) { ok = false; why = "snapshots of the check"; }
            const int w = hdr(kE)->RY - hdr(kE)->LY, cst = 1000 + rnd(100000), dsh = rnd(500), dr = rnd(300);
            auto derive = [&](int slot, int dc, int dcol, int drow) {       // slot <- exit state moved by a constant / whole columns / rows
                memcpy(snaps.data() + (size_t)slot * mb::kSnapBytes, snaps.data() + (size_t)kE * mb::kSnapBytes, mb::kSnapBytes);
                mb::SnapHdr *h = hdr(slot);
                h->best -= dc; h->LY -= dcol; h->RY -= dcol; h->row -= drow;
                for (int x = 0; x < w; x++) { if (Cof(slot)[x] != mb::kNeg) Cof(slot)[x] -= dc; if (Dof(slot)[x] > mb::kNeg2) Dof(slot)[x] -= dc; }
            };
            derive(1, cst, 0, 0); derive(2, cst, dsh, dr);
            derive(3, 0, 0, 0); derive(6, 0, 0, 0); derive(7, 0, 0, 0);
            int live_c = -1, live_d = -1, dead_d = -1;
            const int thr = hdr(kE)->best - Y;
            for (int x = 0; x < w; x++) {
                if (live_c < 0 && Cof(kE)[x] != mb::kNeg) live_c = x;
                if (live_d < 0 && Dof(kE)[x] - E >= thr) live_d = x;
                if (dead_d < 0 && Dof(kE)[x] > mb::kNeg2 && Dof(kE)[x] - E < thr - 50) dead_d = x;
            }
            if (live_c >= 0) Cof(3)[live_c] -= 1;
            if (live_d >= 0) Dof(6)[live_d] -= 1;
            if (dead_d >= 0) Dof(7)[dead_d] -= 7;
            std::vector<mb::VerifyJob> vj = {{kE, 0, 0, 0}, {kE, 1, 0, 0}, {kE, 2, dsh, dr}, {kE, 3, 0, 0}, {kE, 6, 0, 0}, {kE, 7, 0, 0}, {kE, 2, dsh + 1, dr}};
            std::vector<mb::VerifyOut> vo(vj.size());
            hipLaunchKernelGGL(mb::k_verify, dim3((unsigned)vj.size()), dim3(256), 0, nullptr, vj.data(), vo.data(), (int)vj.size(), snaps.data(), Y, E);
            const bool v_ok = vo[0].ok && vo[0].c == 0 && vo[1].ok && vo[1].c == cst && vo[2].ok && vo[2].c == cst && (live_c < 0 || !vo[3].ok) && (live_d < 0 || !vo[4].ok) &&
                              (dead_d < 0 || vo[5].ok) && !vo[6].ok;
            if (!v_ok) { ok = false; why = "k_verify"; }
        }
        printf("case %d: dir %+d, %lld x %lld, ydrop %d: best %d at (%d, %d), %lld cells in %lld rows, %zu ops  %s%s\n", cs, dir, (long long)na, (long long)nb, Y, want.best, want.bi,
               want.bj, want.cells, want.rows, want.ops.size(), ok ? "ok" : "MISMATCH: ", ok ? "" : why);
        if (!ok) bad++;
    }
    if (!relay_mode) printf("walls: %d of %d sides changed by their walls\n", n_wall_changed, n_wall_cases);
    if (relay_mode) {
        printf("relays: %d of %d accepted\n", relays_accepted, relays_tried);
        if (relays_tried && !relays_accepted) { printf("no hand-over was accepted: the test says nothing  MISMATCH\n"); bad++; }
    }
    return bad ? 1 : 0;
}

// mb_ydrop_lds.h -- the four-wave Y-drop DP kernel body with the previous row in an LDS (or HBM) ring: the evaluator of the rows that outgrow
// the one-wave kernels and of the walls mode (gfx950, wave64).  Included by mb_kernels.hip inside namespace mb after the wave helpers
// (dpp_scan_max, dpp_scan_max64, dpp_shr1, dpp_shr1_64, uni, uni64), RowInfo, row_score_lut / lut_score -- and, with MB_EMU defined, by
// the host-side emulation under tests/emu (emu_ydrop.cpp).
#pragma once

constexpr int kYdWaves = 4;                  // one wave per SIMD of the CU
constexpr int kYdThreads = 64 * kYdWaves;
constexpr int kBig = 1 << 20;

struct YdShared {                            // LDS of one DP problem (LDS-ring variant: 16 + 2 + 2 KiB + exchange)
    int2 scan[kYdThreads];                   // per-thread inclusive scans {X = M + rel, M}; read at uniform slots 63,127,191,255
    int4 xb[kYdWaves];                       // per-wave {first break, first alive, last alive, best candidate} (pass-relative columns)
    int iv_last, cp_last;                    // last column of a pass, for the next pass of a wide row
    unsigned long long blk, chunk; int fail; // arena allocations made by thread 0
    unsigned long long smax;                 // best cell of the exit row (score, column) packed for atomicMax
    // walls (miblast_params.walls): one flag per ring column -- the column lies, in this row, on the path of an earlier alignment --,
    // their number in the row, and the 64-bit (cuts, value) scan slots of the horizontal-gap chain
    int n_blocked;
    uint8_t wflag[2048];
    unsigned long long scanx[kYdThreads];
};

// One workgroup (4 waves) per problem; a row is evaluated 256 columns per pass, wave w taking columns
// base+64w .. base+64w+63.  Two barriers per pass:
//   B1  after the wave-local DPP scans: the other waves' scan totals are read from LDS.  Two scans run
//       side by side: X = M + rel for the horizontal gap, and M itself for the running best -- the prefix
//       max of C equals the prefix max of M because every horizontal-gap value is strictly below an M
//       further left, so the y-drop test needs no scan after I is known;
//   B2  after the y-drop test: each wave publishes {first break, first alive, last alive, best candidate};
//       all waves reduce the four records identically, so no third barrier is needed.
// C/D of the previous row live in an LDS ring of int2 indexed by column, overwritten in place.
// WALLS (miblast_params.walls): a cell that pairs a target base with a query base lying on the path of an
// earlier alignment of the unit is dead, and neither gap state survives it (SURVEY A.10 ONE_SIDED with the walls switch: C = D = I = -inf).
// The earlier alignments come as their gap-free runs (WallSeg, sorted by q inside an alignment); a thread follows up to kWallPerThread
// alignments with a cursor each while the rows advance and flags the column a path crosses the row in (at most one per alignment).
// A dead cell cuts the horizontal-gap chain of the row.  The max-plus scan therefore runs on 64-bit keys (blocked columns up to and
// including this one) << 32 | value: the candidates behind the last blocked column left of a lane always win -- the blocked column
// itself is among them, with a dead value -- so what a lane reads is exactly the oracle's chain restarted after every dead cell,
// however many a row holds.  Rows without a flagged column (nearly all) take the plain 32-bit scan.
struct WallSeg { int32_t q0, t0, len; };      // cells (t0 + k, q0 + k), k < len, in the coordinates of the searched strand
constexpr int kWallPerThread = 4;             // alignments per thread: a unit may hold 4 x 256 earlier alignments

__device__ __forceinline__ unsigned long long wall_key(int cuts, int x) { return ((unsigned long long)(unsigned)cuts << 32) | (unsigned)(x ^ (int)0x80000000); }
__device__ __forceinline__ int wall_val(unsigned long long k) { return (int)((unsigned)k ^ 0x80000000u); }

template <bool GLOBAL, bool PROF, bool WALLS>
__device__ __forceinline__ void ydrop_body(const DpProb &pr, DpOut *out, const gbytes tc,
                                           const gbytes qc, const int O, const int E, const int Y,
                                           int2 *CD, uint8_t *Tb, const int cap, YdShared *sh,
                                           uint8_t *__restrict__ arena, const unsigned long long arena_bytes,
                                           unsigned long long *__restrict__ arena_next, const unsigned blk_bytes,
                                           unsigned long long *__restrict__ rowdir, uint8_t *__restrict__ snaps,
                                           const WallSeg *__restrict__ wsegs = nullptr, const int2 *__restrict__ walns = nullptr, const int wa0 = 0, const int wa1 = 0,
                                           uint8_t *__restrict__ gflags = nullptr) {
    const int tid = threadIdx.x;
    uint8_t *const wflag = WALLS ? (GLOBAL ? gflags : sh->wflag) : nullptr;       // one flag per ring column (HBM ring: zeroed by the host)
    const int lane = tid & 63;
    const int wv = uni(tid >> 6);
    const int mask = cap - 1;
    const int row_lo = GLOBAL ? 0 : pr.row_lo;                          // row records are indexed by rho = row - row_lo
    const int na = pr.na, nb = pr.nb, dir = pr.dir;
    const int64_t t0 = pr.t0, q0 = pr.q0;
    const long long clk0 = clock64();
    long long pf[6] = {0, 0, 0, 0, 0, 0}, pt = 0;
#define MB_TICK(k) do { if (PROF) { long long _n = clock64(); pf[k] += _n - pt; pt = _n; } } while (0)
    const int OE = O + E;
    const int grow = (Y >= O ? (Y - O) / E : 0) + 2;          // a row can outgrow the previous window by at most this
    int overflow = 0;
    int R0 = 0;
    if (Y >= O) { R0 = (Y - O) / E; if (R0 > na) R0 = na; }
    if (R0 + 1 + grow + 2 * kYdThreads + 64 > cap) overflow = 1;
    // ---- trace arena bookkeeping (identical in every wave; thread 0 does the atomics) ----
    unsigned long long blk_off = 0, chunk_off = 0;
    unsigned blk_used = 0;
    if (!overflow) {
        if (tid == 0) {
            unsigned long long o1 = atomicAdd(arena_next, 2ull * blk_bytes);
            sh->blk = o1; sh->chunk = o1 + blk_bytes; sh->fail = (o1 + 2ull * blk_bytes > arena_bytes);
        }
        __syncthreads();
        blk_off = uni64(sh->blk); chunk_off = uni64(sh->chunk);
        if (uni(sh->fail)) overflow = 3;
        __syncthreads();
    }
    // row records {offset, LY} are buffered 64 rows at a time in wave 0 (lane = row & 63)
    unsigned rb_lo = 0, rb_hi = 0, rb_ly = 0;
    auto flush_rows = [&](int last_row) {      // records (last_row & ~63) .. last_row  (record = row - row_lo)
        if (wv == 0) {
            const int r = (last_row & ~63) + lane;
            if (r <= last_row) {
                RowInfo ri; ri.off = ((unsigned long long)rb_hi << 32) | rb_lo; ri.ly = rb_ly; ri.pad = 0;
                ((RowInfo *)(arena + chunk_off))[r & (kRowChunk - 1)] = ri;
            }
        }
    };
    int LY = 0, RY = R0 + 1, best = 0, bi = 0, bj = 0;
    long long cells = R0 + 1;
    int rows = 1;
    if (!overflow && row_lo == 0) {
        // ---- row 0: C = -(O + jE) while within ydrop of 0, every cell reached by a horizontal gap from the origin
        uint8_t *tr = arena + blk_off;
        for (int j = tid; j <= R0; j += kYdThreads) CD[j & mask] = make_int2((j == 0) ? 0 : -(O + j * E), kNeg);
        for (int j = 2 * tid; j <= R0; j += 2 * kYdThreads) {           // two 4-bit trace codes per byte, even column in the low nibble
            const unsigned lo = (j == 0) ? 3u : (2u | (j >= 2 ? 8u : 0u)), hi = (j + 1 <= R0) ? (2u | (j + 1 >= 2 ? 8u : 0u)) : 0u;
            tr[j >> 1] = (uint8_t)(lo | (hi << 4));
        }
        if (lane == 0) { rb_lo = (unsigned)blk_off; rb_hi = (unsigned)(blk_off >> 32); rb_ly = 0; }
        blk_used = (unsigned)(R0 + 2) >> 1;
    } else if (!overflow) {
        // ---- continuation: the state after row row_lo comes from a snapshot (record 0 of this piece stays unused)
        const uint8_t *sp = snaps + (size_t)pr.init_snap * kSnapBytes;
        const SnapHdr *h = (const SnapHdr *)sp;
        const int *sC = (const int *)(sp + sizeof(SnapHdr)), *sD = sC + kSnapCols;
        LY = uni(h->LY); RY = uni(h->RY); best = uni(h->best); bi = uni(h->bi); bj = uni(h->bj); rows = uni(h->rows);
        cells = (long long)uni64((unsigned long long)h->cells);
        for (int j = LY + tid; j < RY; j += kYdThreads) CD[j & mask] = make_int2(sC[j - LY], sD[j - LY]);
    }
    if (!overflow && tid == 0) rowdir[pr.row_off] = chunk_off;
    int t_hi = max(LY - 1, 0);                                          // highest target column staged in Tb
    int qblk0 = 1 + (row_lo & ~255);                                    // first row of the 256-row block held in qv (4 rows per lane of a wave)
    auto load_q = [&](int r0) -> unsigned {
        typedef const uint32_t __attribute__((address_space(1), aligned(1))) *gword;
        const int r = r0 + 4 * lane;                                      // rows r .. r+3 (rows beyond nb are never evaluated)
        if (r > nb) return 0x04040404u;
        return dir > 0 ? *(gword)(qc + (q0 + r - 1)) : __builtin_bswap32(*(gword)(qc + (q0 - r - 3)));
    };
    unsigned qv = load_q(qblk0);
    // No load may be in flight when the row loop starts or turns around: the compiler would otherwise wait for ALL
    // outstanding memory operations -- including the previous row's trace stores -- at the top of every row.
    asm volatile("" : "+v"(qv));
    const uint32_t lutv = row_score_lut((unsigned)min(lane, 4));     // lane k holds the packed score row of query base k
    const int tidE = tid * E;
    __syncthreads();
    int i = row_lo + 1;
    int stopped = 0, exit_j = 0;
    // walls: this thread's alignment (threads beyond the unit's alignments have none), its runs [ws0, ws1) and the cursor
    int ws0[kWallPerThread], ws1[kWallPerThread], wcur[kWallPerThread], wjb[kWallPerThread];
    if (WALLS) {
        if (wa1 - wa0 > kWallPerThread * kYdThreads) overflow = 4;   // (more earlier alignments in a unit than this mode covers)
#pragma unroll
        for (int m = 0; m < kWallPerThread; m++) {
            ws0[m] = ws1[m] = wcur[m] = 0; wjb[m] = -1;
            const int a = wa0 + tid + m * kYdThreads;
            if (!overflow && a < wa1) { const int2 al = walns[a]; ws0[m] = al.x; ws1[m] = al.y; wcur[m] = dir > 0 ? ws0[m] : ws1[m] - 1; }
        }
        if (!GLOBAL) for (int j = tid; j < 2048; j += kYdThreads) sh->wflag[j] = 0;
    }
    for (; i <= nb && !overflow; i++) {
        if (PROF) pt = clock64();
        const int rho = i - row_lo;
        if (i - qblk0 >= 256) { qblk0 += 256; qv = load_q(qblk0); asm volatile("" : "+v"(qv)); }      // (every 256 rows: waited for on the spot)
        const unsigned qword = (unsigned)__builtin_amdgcn_readlane((int)qv, (i - qblk0) >> 2);
        const uint32_t lut = (uint32_t)__builtin_amdgcn_readlane((int)lutv, min((int)((qword >> (8 * ((i - qblk0) & 3))) & 7u), 4));
        // the row can reach at most column RY + grow; everything it may touch must be staged and fit the ring
        const int reach = min(na, RY + grow);
        if (reach - LY + 2 * kYdThreads + 64 > cap) { overflow = 1; break; }
        const int need = reach - LY + 1;
        const bool new_blk = blk_used + (unsigned)need > blk_bytes;
        const bool new_chunk = (rho & (kRowChunk - 1)) == 0;
        if ((rho & 63) == 0) flush_rows(rho - 1);
        if (new_blk || new_chunk) {
            __syncthreads();                                             // everyone is past the previous use of sh->blk/chunk
            if (tid == 0) {
                const unsigned nblk = (new_blk ? 1u : 0u) + (new_chunk ? 1u : 0u);
                unsigned long long o1 = atomicAdd(arena_next, (unsigned long long)nblk * blk_bytes);
                sh->fail = (o1 + (unsigned long long)nblk * blk_bytes > arena_bytes);
                if (new_blk) { sh->blk = o1; o1 += blk_bytes; }
                if (new_chunk) sh->chunk = o1;
            }
            __syncthreads();
            if (uni(sh->fail)) { overflow = 3; break; }
            if (new_blk) { blk_off = uni64(sh->blk); blk_used = 0; }
            if (new_chunk) { chunk_off = uni64(sh->chunk); if (tid == 0) rowdir[pr.row_off + (unsigned)(rho / kRowChunk)] = chunk_off; }
        }
        if (!GLOBAL && t_hi < min(na, reach + kYdThreads)) {
            // stage target columns ahead of the window, 256 at a time (visible to all waves after the barrier)
            while (t_hi < min(na, reach + kYdThreads)) {
                const int j = t_hi + 1 + tid;
                if (j <= na) Tb[j & mask] = tc[dir > 0 ? t0 + j - 1 : t0 - j];
                t_hi += kYdThreads;
            }
            __syncthreads();
        }
        if (lane == (rho & 63)) { const unsigned long long ro = blk_off + blk_used; rb_lo = (unsigned)ro; rb_hi = (unsigned)(ro >> 32); rb_ly = (unsigned)LY; }
        int n_blk_row = 0;
        if (WALLS) {
            if (tid == 0) sh->n_blocked = 0;
            __syncthreads();                                             // (also: the flags of the previous row are cleared by now)
            const int qrow = (int)(dir > 0 ? q0 + i - 1 : q0 - i);
#pragma unroll
            for (int m = 0; m < kWallPerThread; m++) {
                wjb[m] = -1;
                if (ws1[m] <= ws0[m]) continue;
                bool hit;
                int c = wcur[m];
                if (dir > 0) { while (c < ws1[m] && wsegs[c].q0 + wsegs[c].len <= qrow) c++; hit = c < ws1[m] && wsegs[c].q0 <= qrow; }
                else { while (c >= ws0[m] && wsegs[c].q0 > qrow) c--; hit = c >= ws0[m] && qrow < wsegs[c].q0 + wsegs[c].len; }
                wcur[m] = c;
                if (hit) {
                    const int tcol = wsegs[c].t0 + (qrow - wsegs[c].q0);
                    const int jb = (int)(dir > 0 ? tcol - t0 + 1 : t0 - tcol);
                    if (jb >= max(LY, 1) && jb <= reach) { wjb[m] = jb; wflag[jb & mask] = 1; atomicAdd(&sh->n_blocked, 1); }
                }
            }
            __syncthreads();
            n_blk_row = uni(sh->n_blocked);
        }
        MB_TICK(0);
        int carry_x = kNeg2, carry_m = best, carry_iv = kNeg, carry_cp = kNeg;   // pass-level carries (wave-uniform)
        int row_best = best, first_alive = -1, last_alive = -1, nrow = 0;
        bool done = false;
        for (int base = LY; !done; base += kYdThreads) {
            const int j = base + tid;
            const int idx = j & mask;
            const int2 cd = CD[idx];
            int cpl = CD[(j - 1) & mask].x;
            const int jc = min(max(j, 1), max(na, 1));                   // lanes past the contig end must not read past the buffer
            const unsigned t = GLOBAL ? (unsigned)tc[dir > 0 ? t0 + jc - 1 : t0 - jc] : (unsigned)Tb[idx];
            const bool inwin = j < RY;
            const int cp = inwin ? cd.x : kNeg;
            const int dp = inwin ? cd.y : kNeg;
            cpl = (j - 1 >= LY && j - 1 < RY) ? cpl : kNeg;
            if (base != LY && tid == 0) cpl = carry_cp;                   // column base-1 was overwritten by the previous pass
            const unsigned at = (j <= na && j >= 1) ? t : 4u;
            // walls: is this column blocked in this row; how many blocked columns of the pass lie left of it, in the whole pass, and
            // left of the column before this one (every wave reads the flags of all four waves' columns)
            bool wblocked = false;
            int wbefore = 0;
            if (WALLS && n_blk_row) {
                unsigned long long fb[kYdWaves];
#pragma unroll
                for (int v = 0; v < kYdWaves; v++) fb[v] = __ballot(wflag[(base + 64 * v + lane) & mask] != 0);
                const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
                for (int v = 0; v < kYdWaves; v++) {
                    const int c = (int)__popcll(fb[v]);
                    if (v < wv) wbefore += c;
                    if (v == wv) { wbefore += (int)__popcll(fb[v] & below); wblocked = (fb[v] >> lane) & 1ull; }
                }
            }
            const int diag = wblocked ? kNeg : cpl + lut_score(lut, at);
            const int de = dp - E, dn = cp - OE;
            const int Dv = wblocked ? kNeg : max(de, dn);
            const int dext = de >= dn ? 4 : 0;
            const int M = max(diag, Dv);
            const int rel = tidE + uni((base - LY) * E);
            const bool wrow = WALLS && n_blk_row;                         // (uniform) a row with dead cells: the chain runs on (cuts, value) keys
            const unsigned long long PX64 = wrow ? dpp_scan_max64(wall_key(wbefore + (wblocked ? 1 : 0), M + rel)) : 0ull;
            const int PX = wrow ? 0 : dpp_scan_max(M + rel);
            const int PM = dpp_scan_max(j <= na ? M : kNeg);
            sh->scan[tid] = make_int2(PX, PM);
            if (wrow) sh->scanx[tid] = PX64;
            MB_TICK(1);
            __syncthreads();                                            // ---- B1
            MB_TICK(2);
            const int2 t0s = sh->scan[63], t1s = sh->scan[127], t2s = sh->scan[191], t3s = sh->scan[255];
            const int x62 = sh->scan[(64 * wv + 254) & 255].x;            // lane 62 of the previous wave (unused for wave 0)
            const int cx = max(max(carry_x, wv > 0 ? t0s.x : kNeg2), max(wv > 1 ? t1s.x : kNeg2, wv > 2 ? t2s.x : kNeg2));
            const int cxm1 = max(carry_x, max(wv > 1 ? t0s.x : kNeg2, wv > 2 ? t1s.x : kNeg2));   // carry into the previous wave
            const int cm = max(max(carry_m, wv > 0 ? t0s.y : kNeg2), max(wv > 1 ? t1s.y : kNeg2, wv > 2 ? t2s.y : kNeg2));
            const int allm = max(max(carry_m, t0s.y), max(max(t1s.y, t2s.y), t3s.y));
            const int allx = max(max(carry_x, t0s.x), max(max(t1s.x, t2s.x), t3s.x));
            int ivl0 = wv == 0 ? carry_iv : max(cxm1, x62) - O - (rel - tidE + (64 * wv - 1) * E);
            const int pex = max(dpp_shr1(PX, kNeg2), cx);
            int Iv = pex - O - rel;
            unsigned long long allx64 = 0;
            if (wrow) {
                // the same combination on the keys: totals of the waves before this one, the carry of the passes before (no cut yet)
                const unsigned long long k0 = sh->scanx[63], k1 = sh->scanx[127], k2 = sh->scanx[191], k3 = sh->scanx[255];
                const unsigned long long kc = wall_key(0, carry_x);
                auto mx = [](unsigned long long a, unsigned long long b) { return a > b ? a : b; };
                const unsigned long long cx64 = mx(mx(kc, wv > 0 ? k0 : 0ull), mx(wv > 1 ? k1 : 0ull, wv > 2 ? k2 : 0ull));
                const unsigned long long cxm64 = mx(kc, mx(wv > 1 ? k0 : 0ull, wv > 2 ? k1 : 0ull));
                const unsigned long long k62 = sh->scanx[(64 * wv + 254) & 255];
                allx64 = mx(mx(kc, k0), mx(mx(k1, k2), k3));
                const unsigned long long pex64 = mx(dpp_shr1_64(PX64, 0ull), cx64);
                Iv = wblocked ? kNeg : wall_val(pex64) - O - rel;            // (the winning keys carry exactly this lane's count of cuts)
                if (wv > 0) ivl0 = wall_val(mx(cxm64, k62)) - O - (rel - tidE + (64 * wv - 1) * E);
            }
            const int ivl = dpp_shr1(Iv, ivl0);
            const int iext = (Iv == ivl - E) ? 8 : 0;
            const int gmax = max(Dv, Iv);
            const int Cv = max(diag, gmax);
            const int src = diag >= gmax ? 0 : (Dv >= Iv ? 1 : 2);       // tie preference diag > D > I
            const int best_at = max(PM, cm);                             // running best, row-major, incl. this cell
            const bool alive = (j <= na) & (Cv >= best_at - Y) & !wblocked;
            const unsigned long long am = __ballot(alive);
            const unsigned long long bm = __ballot(((j >= RY) & !alive) | (j > na));
            const unsigned long long wm = __ballot((j <= na) & (Cv == allm));
            // C/D of columns past the break are never read again, so the ring can be written before validity is known
            CD[idx] = make_int2(alive ? Cv : kNeg, Dv);
            {
                const int w64 = 64 * wv;
                int4 rec;
                rec.x = bm ? w64 + (int)__ffsll((long long)bm) - 1 : kBig;
                rec.y = am ? w64 + (int)__ffsll((long long)am) - 1 : kBig;
                rec.z = am ? w64 + 63 - (int)__clzll((long long)am) : -1;
                rec.w = wm ? w64 + (int)__ffsll((long long)wm) - 1 : kBig;
                sh->xb[wv] = rec;
            }
            if (tid == kYdThreads - 1) { sh->iv_last = Iv; sh->cp_last = cp; }   // last column of the pass, for the next pass
            MB_TICK(3);
            __syncthreads();                                            // ---- B2
            MB_TICK(4);
            const int4 r0 = sh->xb[0], r1 = sh->xb[1], r2 = sh->xb[2], r3 = sh->xb[3];
            const int pbrk = uni(min(min(r0.x, r1.x), min(r2.x, r3.x)));
            const int fa = uni(min(min(r0.y, r1.y), min(r2.y, r3.y)));
            const int la = uni(max(max(r0.z, r1.z), max(r2.z, r3.z)));
            const int cand = uni(min(min(r0.w, r1.w), min(r2.w, r3.w)));
            // no alive cell lies behind the first break (cells there are only reachable through the dead break cell)
            int nvalid = kYdThreads;
            if (pbrk < kYdThreads) { nvalid = pbrk + ((base + pbrk) <= na ? 1 : 0); done = true; }
            if (fa < kBig && first_alive < 0) first_alive = base + fa;
            if (la >= 0) last_alive = base + la;
            if (uni(allm) > row_best) { row_best = uni(allm); bi = i; bj = base + cand; }
            {
                // two codes per byte: the even thread of a pair stores both (nrow is a multiple of the pass width, so pairs never straddle passes)
                const unsigned code = tid < nvalid ? (unsigned)(src | dext | iext) : 0u;
                const unsigned next = (unsigned)__shfl_down((int)code, 1);
                if (!(tid & 1) && tid < nvalid) arena[blk_off + blk_used + (unsigned)((nrow + tid) >> 1)] = (uint8_t)(code | (next << 4));
            }
            nrow += nvalid;
            if (!done) { carry_x = wrow ? uni(wall_val(allx64)) : uni(allx); carry_m = uni(allm); carry_iv = uni(sh->iv_last); carry_cp = uni(sh->cp_last); }
            MB_TICK(5);
        }
        if (WALLS && n_blk_row) {
#pragma unroll
            for (int m = 0; m < kWallPerThread; m++) if (wjb[m] >= 0) wflag[wjb[m] & mask] = 0;      // (the next row's barrier orders this before its flags)
        }
        blk_used += (unsigned)(nrow + 1) >> 1;
        cells += nrow;
        rows++;
        best = row_best;
        if (first_alive < 0) { i++; break; }
        LY = first_alive;
        RY = last_alive + 1;
        if (!GLOBAL && (i == pr.snap_row || i == pr.stop_row || i == pr.snap_row2 || i == pr.snap_row3)) {
            // state after row i (the ring writes of the row are visible: they precede the row's last barrier)
            uint8_t *sp = snaps + (size_t)(pr.snap_idx + (i == pr.stop_row ? 1 : i == pr.snap_row ? 0 : i == pr.snap_row2 ? 2 : 3)) * kSnapBytes;
            int *sC = (int *)(sp + sizeof(SnapHdr)), *sD = sC + kSnapCols;
            if (tid == 0) sh->smax = 0;
            __syncthreads();
            unsigned long long mine = 0;
            for (int j = LY + tid; j < RY; j += kYdThreads) {
                const int2 cd = CD[j & mask]; sC[j - LY] = cd.x; sD[j - LY] = cd.y;
                // (score biased to unsigned) << 32 | (inverted column): the maximum is the best score, leftmost on ties
                const unsigned long long key = ((unsigned long long)(unsigned)(cd.x + (1 << 30)) << 32) | (unsigned)(0x7fffffff - j);
                mine = key > mine ? key : mine;
            }
            atomicMax(&sh->smax, mine);
            __syncthreads();
            exit_j = 0x7fffffff - (int)(unsigned)(sh->smax & 0xffffffffull);
            if (tid == 0) {
                SnapHdr *h = (SnapHdr *)sp;
                h->LY = LY; h->RY = RY; h->best = best; h->bi = bi; h->bj = bj; h->row = i; h->rows = rows; h->cells = cells;
                h->valid = 1;
            }
            if (i == pr.stop_row) { stopped = 1; i++; break; }
        }
    }
    if (!overflow) flush_rows(i - 1 - row_lo);
    if (tid == 0) {
        out->best = best; out->bi = bi; out->bj = bj; out->rows = rows;
        out->cells = cells; out->clocks = clock64() - clk0; out->overflow = overflow; out->n_ops = 0; out->stopped = stopped; out->exit_j = exit_j;
        if (PROF) for (int k = 0; k < 6; k++) out->prof[k] = pf[k];
    }
}

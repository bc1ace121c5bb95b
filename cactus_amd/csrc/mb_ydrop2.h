// mb_ydrop2.h -- the piece evaluator of k_ydrop2 / k_ydrop2_w5 (gfx950, wave64): one wave per piece of a one-sided Y-drop DP, the previous
// row in registers, 4 columns per lane, a second group of 256 columns for the rows that need it.  Included by mb_kernels.hip inside
// namespace mb after the wave helpers (dpp_shr1 / dpp_shl1 / dpp_scan_max / uni / uni64, row_score_lut, RowInfo, as_global) -- and, with
// MB_EMU defined, by the host-side emulation under tests/emu (emu_ydrop.cpp), which supplies those helpers and the yd_* wrappers below
// through per-wave exchange slots.
#pragma once

#ifndef MB_EMU
// the wave primitives and the "keep it in a VGPR here" pins of the evaluator, by name (the emulation has its own)
#define yd_readlane(v, l) __builtin_amdgcn_readlane((v), (l))
#define yd_perm(a, b, s) __builtin_amdgcn_perm((a), (b), (s))
#define yd_alignbit(hi, lo, n) __builtin_amdgcn_alignbit((hi), (lo), (n))
#define yd_ballot(p) __ballot(p)
#define yd_clock() clock64()
#define YD_PIN1(a) asm volatile("" : "+v"(a))
#define YD_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define YD_PIN5(a, b, c, d, e) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e))
#define YD_GLOBAL_UNALIGNED __attribute__((address_space(1), aligned(1)))
// what one wave of a launch reads of another's snapshot while both run: loads that bypass the caches that are not coherent across
// CUs / XCDs (agent scope), the stamp as the acquire / release pair around them
#define yd_ld_agent(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// (the reader: the stamp and everything it reads of the snapshot afterwards are agent-scope loads -- they go to the coherent level, past this
//  XCD's L2 --, issued in order behind a branch on the stamp's value; an agent-scope ACQUIRE would invalidate the L2 for every check)
static __device__ __forceinline__ int yd_ld_acquire_i32(const int *p) {
    const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return v;
}
#define yd_ld_acquire(p) yd_ld_acquire_i32(p)
// A snapshot is WRITTEN with agent-scope stores (write-through: nothing of it stays dirty in this XCD's L2), its lanes' stores are waited
// for (a workgroup-scope release fence: s_waitcnt, no cache maintenance), then the stamp goes out the same way.  An agent-scope release
// FENCE instead writes the whole L2 back -- every wave's half-filled trace lines, ten thousand times per launch: the first version of
// this did, and the DP launches' WRITE_SIZE went from 1.28 to 1.54 times their algorithmic bytes (profiles/r05_hbm_traffic_pmc.json).
#define yd_st_agent(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define yd_st_release(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define yd_fence() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0); } while (0)
// a lane's four columns of a snapshot in ONE 16-byte write-through store: the wave's stores of an instruction are then one contiguous
// kilobyte.  (Four dword stores per lane are four instructions that each touch every fourth dword of the same lines: written through,
// every 32-byte sector went to memory four times -- 170 MB of the 1 100 MB a step's DP launches wrote, profiles/README.md.)
typedef int yd_int4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void yd_st_agent4(int *p, int a, int b, int c, int d) {
    const yd_int4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
#endif

// k_ydrop2: k_ydrop1 with 4 columns per lane and a SECOND group of 256 columns that is only evaluated when a row needs
// it.  Lane l owns columns jb + 4 l .. +3 (group A) and jb + 256 + 4 l .. +3 (group B).  With Cactus's --ydrop=3000..4000 a
// window is ~170 columns wide, so almost every row is group A alone (the instruction count of k_ydrop1<4>), and the rows
// that reach further -- the ones that make k_ydrop1<4> overflow and rerun -- just evaluate group B as well, with the scan
// carries of group A.  B can be skipped whenever the previous row's window ended inside A and this row's break is found in
// A: no column of B was alive in the previous row then, so all of them already hold dead values (k_ydrop1's invariant).
//
// The hand-over inside the launch.  `probs` is the ROUND's piece table, the launch's pieces are [first, first + n).
// A piece that reaches its stop row alive and is aimed at a relay (DpProb.aim1) looks at the relay's entry snapshot itself -- the relay
// runs in the same launch or ran in an earlier one of the round; a snapshot counts when its header carries the round's `stamp`, which
// its writer stores last -- and compares the two states as k_verify does (mb_verify.h).  Equal: it stops; the host reads k_verify's
// verdict as before.  Not equal: the piece simply goes on -- to the same relay's next entry snapshot, then to the relay after -- which
// is the continuation the host would otherwise queue as a launch of its own, with the state already in registers.  The check itself
// decides nothing about results: k_verify, run after the launch on the hand-over the piece ended at (the piece rewrites its
// VerifyJob), stays the authority, and a snapshot that is not there yet just ends the piece as before.
static __device__ __forceinline__ void ydrop2_piece(const DpProb *__restrict__ probs, DpOut *__restrict__ outs, int n,
                                                    const PairPtrs *__restrict__ pairs, const int O, const int E, const int Y,
                                                    uint8_t *__restrict__ arena, const unsigned long long arena_bytes,
                                                    unsigned long long *__restrict__ arena_next, const unsigned blk_bytes,
                                                    unsigned long long *__restrict__ rowdir, uint8_t *snaps,
                                                    const int *__restrict__ order, const int first, VerifyJob *__restrict__ vjobs,
                                                    const int stamp, const int force_mod) {
    if ((int)blockIdx.x >= n) return;
    // (order: longest pieces first when a launch holds more pieces than wave slots, so that its tail is made of short ones)
    const int pi = order ? order[blockIdx.x] : (int)blockIdx.x;
    constexpr int K = 4, G = 2, kHalf = 64 * K, kCap = G * kHalf;
    constexpr unsigned kAll = (1u << K) - 1u;
    const DpProb pr = probs[first + pi];
    const PairPtrs pp = pairs[pr.pad0];
    const gbytes tc = as_global(pp.tc);
    const gbytes qc = as_global(pr.strand ? pp.qr : pp.qf);
    DpOut *out = &outs[pi];
    const int lane = threadIdx.x & 63;
    const int na = pr.na, nb = pr.nb, dir = pr.dir;
    const int64_t t0 = pr.t0, q0 = pr.q0;
    const int row_lo = pr.row_lo;
    const long long clk0 = yd_clock();
    // the piece's own snapshot slots start out invalid.  Since the hand-over moved into the launch, pieces of the SAME launch read these headers
    // (the upstream piece that is aimed at this one looks at its entry snapshots): what such a reader trusts is the `stamp`, unique per round
    // and written last by the snapshot's writer -- a slot that still carries an older round's stamp (or none) is "not there yet" whatever its
    // `valid` says -- so this store decides nothing inside the launch; it is an agent-scope store like every other write to a header, for the
    // readers that come after the launch in stream order (k_verify, continuations) and so that no header field is ever written two ways.
    if (pr.snap_idx >= 0 && lane < kSnapSlots) yd_st_agent(&((SnapHdr *)(snaps + (size_t)(pr.snap_idx + lane) * kSnapBytes))->valid, 0);
    const int OE = O + E;
    int overflow = 0;
    int R0 = 0;
    if (Y >= O) { R0 = (Y - O) / E; if (R0 > na) R0 = na; }
    // target bases of the 4 columns c0 .. c0+3 (byte k = column c0 + k); columns beyond the contig are never alive
    auto load_t = [&](int c0) -> uint32_t {
        typedef const uint32_t YD_GLOBAL_UNALIGNED *gword;
        if (c0 > na) return 0u;
        return dir > 0 ? *(gword)(tc + (t0 + c0 - 1)) : __builtin_bswap32(*(gword)(tc + (t0 - c0 - (K - 1))));
    };
    // ---- trace arena bookkeeping (lane 0 does the atomics) ----
    unsigned long long blk_off = 0, chunk_off = 0;
    unsigned blk_used = 0;
    auto arena_take = [&](unsigned nblk) -> unsigned long long {          // returns ~0 when the arena is exhausted
        unsigned long long o1 = 0;
        if (lane == 0) o1 = atomicAdd(arena_next, (unsigned long long)nblk * blk_bytes);
        o1 = uni64(o1);
        return o1 + (unsigned long long)nblk * blk_bytes > arena_bytes ? ~0ull : o1;
    };
    if (R0 + 1 + 2 * K > kCap) overflow = 1;
    if (!overflow) {
        const unsigned long long o1 = arena_take(2);
        if (o1 == ~0ull) overflow = 3;
        else { blk_off = o1; chunk_off = o1 + blk_bytes; }
    }
    unsigned rb_lo = 0, rb_hi = 0, rb_ly = 0;                             // row records buffered 64 at a time (lane = record & 63)
    auto flush_rows = [&](int last_rec) {
        const int r = (last_rec & ~63) + lane;
        if (r <= last_rec) {
            RowInfo ri; ri.off = ((unsigned long long)rb_hi << 32) | rb_lo; ri.ly = rb_ly; ri.pad = 0;
            ((RowInfo *)(arena + chunk_off))[r & (kRowChunk - 1)] = ri;
        }
    };
    int C[G][K], D[G][K];
    int jb = 0, LY = 0, RY = R0 + 1, best = 0, bi = 0, bj = 0, rows = 1;
    long long cells = R0 + 1;
    if (!overflow && row_lo == 0) {
        // ---- row 0: C = -(O + jE) while within ydrop of 0, every cell reached by a horizontal gap from the origin
#pragma unroll
        for (int g = 0; g < G; g++) {
            uint32_t tb0 = 0;                                              // four 4-bit trace codes = one 16-bit store per lane
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int j = g * kHalf + K * lane + k;
                C[g][k] = j == 0 ? 0 : (j <= R0 ? -(O + j * E) : kNeg);
                D[g][k] = kNeg;
                tb0 |= (j == 0 ? 3u : (2u | (j >= 2 ? 8u : 0u))) << (4 * k);
            }
            if (g * kHalf + K * lane <= R0) *(uint16_t *)(arena + blk_off + (g * kHalf + K * lane) / 2) = (uint16_t)tb0;
        }
        if (lane == 0) { rb_lo = (unsigned)blk_off; rb_hi = (unsigned)(blk_off >> 32); rb_ly = 0; }
        blk_used = ((unsigned)(R0 + K) & ~(unsigned)(K - 1)) >> 1;
    } else if (!overflow) {
        // ---- continuation: the state after row row_lo comes from a snapshot (record 0 of this piece stays unused)
        const uint8_t *sp = snaps + (size_t)pr.init_snap * kSnapBytes;
        const SnapHdr *h = (const SnapHdr *)sp;
        const int *sC = (const int *)(sp + sizeof(SnapHdr)), *sD = sC + kSnapCols;
        LY = uni(h->LY); RY = uni(h->RY); best = uni(h->best); bi = uni(h->bi); bj = uni(h->bj); rows = uni(h->rows);
        cells = (long long)uni64((unsigned long long)h->cells);
        jb = LY & ~(K - 1);
        if (RY - jb + 2 * K > kCap) overflow = 1;
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int j = jb + g * kHalf + K * lane + k;
                const bool in = j >= LY && j < RY && !overflow;
                C[g][k] = in ? sC[j - LY] : kNeg;
                D[g][k] = in ? sD[j - LY] : kNeg;
            }
    }
    if (!overflow && lane == 0) rowdir[pr.row_off] = chunk_off;
    uint32_t tw[G];
    tw[0] = load_t(jb + K * lane); tw[1] = load_t(jb + kHalf + K * lane);                 // bases of this lane's columns
    uint32_t tfa = load_t(jb + kCap + K * lane), tfb = load_t(jb + kCap + kHalf + K * lane);   // the next 2 x 256 columns
    int tf_used = 0, tf_base = jb + kCap + 2 * kHalf;                    // lanes of tfa consumed; first column not yet requested
    int qblk0 = 1 + (row_lo & ~255);                                     // first row of the 256-row block held in qv (4 rows per lane)
    auto load_q = [&](int r0) -> unsigned {
        typedef const uint32_t YD_GLOBAL_UNALIGNED *gword;
        const int r = r0 + 4 * lane;                                      // rows r .. r+3 (rows beyond nb are never evaluated)
        if (r > nb) return 0x04040404u;
        return dir > 0 ? *(gword)(qc + (q0 + r - 1)) : __builtin_bswap32(*(gword)(qc + (q0 - r - 3)));
    };
    unsigned qv = load_q(qblk0);
    const uint32_t lutv = row_score_lut((unsigned)min(lane, 4));     // lane k holds the packed score row of query base k
    const int laneKE = lane * K * E;
    // everything loaded so far is waited for HERE (see k_ydrop1)
    YD_PIN5(qv, tw[0], tw[1], tfa, tfb);
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < K; k++) YD_PIN2(C[g][k], D[g][k]);
    int i = row_lo + 1;
    int stopped = 0, exit_j = 0;
    // the next row after which a snapshot is due (entry snapshots of a relay, the exit snapshot at stop_row): one comparison per row
    int stop_at = pr.stop_row;                                           // moves on when a hand-over is rejected inside the launch
    int aim = pr.aim1 - 1, ck = pr.ck0, n_checks = 0;                         // the aimed relay (piece of the round's table), which of its entry snapshots, checks made here
    auto next_event = [&](int after) -> int {
        int e = 0x7fffffff;
        if (pr.snap_row > after) e = min(e, pr.snap_row);
        if (stop_at > after) e = min(e, stop_at);
        if (pr.snap_row2 > after) e = min(e, pr.snap_row2);
        if (pr.snap_row3 > after) e = min(e, pr.snap_row3);
        return e;
    };
    int evt = uni(next_event(row_lo));
    if (overflow) i = nb + 1;                                             // (nothing to evaluate: straight to the epilogue)
    for (; i <= nb; i++) {                                                // (whatever sets `overflow` inside leaves the loop at once)
        const int rho = i - row_lo;
        if (i - qblk0 >= 256) { qblk0 += 256; qv = load_q(qblk0); YD_PIN1(qv); }      // (every 256 rows: waited for on the spot)
        const unsigned qword = (unsigned)yd_readlane((int)qv, (i - qblk0) >> 2);
        const uint32_t lut = (uint32_t)yd_readlane((int)lutv, min((int)((qword >> (8 * ((i - qblk0) & 3))) & 7u), 4));
        // ---- the window's left edge moved K columns or more: shift the lanes (A's last lane takes B's first)
        while (LY - jb >= K) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                C[0][k] = dpp_shl1(C[0][k], yd_readlane(C[1][k], 0));
                D[0][k] = dpp_shl1(D[0][k], yd_readlane(D[1][k], 0));
                C[1][k] = dpp_shl1(C[1][k], kNeg);
                D[1][k] = dpp_shl1(D[1][k], kNeg);
            }
            tw[0] = (uint32_t)dpp_shl1((int)tw[0], yd_readlane((int)tw[1], 0));
            tw[1] = (uint32_t)dpp_shl1((int)tw[1], yd_readlane((int)tfa, tf_used));
            jb += K;
            if (++tf_used == 64) {                                        // (every 256 columns: waited for on the spot)
                tfa = tfb; tfb = load_t(tf_base + K * lane); tf_base += kHalf; tf_used = 0;
                YD_PIN1(tfb);
            }
        }
        if (RY - jb + K > kCap) { overflow = 1; break; }
        // Rarely: every 64 rows the buffered row records go out, a full chunk of them (every kRowChunk rows) or a trace block that may
        // not hold another row (the widest a row can be: kCap columns + padding) is replaced.  One test per row for all of it.
        const bool blk_full = blk_used + (unsigned)(kCap + 2 * K) > blk_bytes;
        if (__builtin_expect(((rho & 63) == 0) | blk_full, 0)) {
            const bool new_chunk = (rho & (kRowChunk - 1)) == 0;
            if ((rho & 63) == 0) flush_rows(rho - 1);
            if (blk_full || new_chunk) {
                const unsigned nblk = (blk_full ? 1u : 0u) + (new_chunk ? 1u : 0u);
                unsigned long long o1 = arena_take(nblk);
                if (o1 == ~0ull) { overflow = 3; break; }
                if (blk_full) { blk_off = o1; blk_used = 0; o1 += blk_bytes; }
                if (new_chunk) { chunk_off = o1; if (lane == 0) rowdir[pr.row_off + (unsigned)(rho / kRowChunk)] = chunk_off; }
            }
        }
        if (lane == (rho & 63)) { const unsigned long long ro = blk_off + blk_used; rb_lo = (unsigned)ro; rb_hi = (unsigned)(ro >> 32); rb_ly = (unsigned)jb; }
        // ---- one group of 256 columns; carries: cpl0 = old C of the column left of the group, cx / cm = max X / running best left of it,
        //      f0 = "the gap into the group's first column extends"
        int cvs[G][K];                                                    // new C before the y-drop test (to locate a new best)
        unsigned dmg[G], amg[G], bmg[G];
        uint32_t tbg[G];
        int totx[G], totm[G], flast[G];
        const int cplB0 = yd_readlane(C[0][K - 1], 63);    // (read before group A overwrites it)
        // (EDGE: some column of the group lies beyond the contig -- those are dead and count as breaks; the common case carries none of
        //  that masking.  The 4-bit trace codes are collected as sign bits, v_sub + v_alignbit per flag: no compare / select pairs.)
        auto group = [&](auto gtag, auto etag, const int cpl0, const int cx, const int cm, const int f0) {
            constexpr int g = decltype(gtag)::value;                      // (compile-time: C[g][k] must stay in registers)
            constexpr bool edge = decltype(etag)::value;
            const int j0 = jb + g * kHalf + K * lane;
            const int kna = na - j0;
            const uint32_t sc = yd_perm(0x1c1c1c1cu, lut, tw[g] & 0x07070707u);
            const int relg = laneKE + g * kHalf * E;
            int diag[K], Dv[K], X[K], Mm[K], ddx[K];
            int prev = dpp_shr1(C[g][K - 1], cpl0);
#pragma unroll
            for (int k = 0; k < K; k++) {
                diag[k] = prev + (int)((sc >> (8 * k)) & 0xFFu) - 128;
                const int de = D[g][k] - E, dn = C[g][k] - OE;
                Dv[k] = max(de, dn);
                ddx[k] = de - dn;                                         // sign: the vertical gap OPENS here (dex = de >= dn is its complement)
                const int Mv = max(diag[k], Dv[k]);
                X[k] = Mv + relg + k * E;
                Mm[k] = (!edge || k <= kna) ? Mv : kNeg;
                prev = C[g][k];
            }
            int lp[K], mi[K];
            lp[0] = kNeg2; mi[0] = Mm[0];
#pragma unroll
            for (int k = 1; k < K; k++) { lp[k] = max(lp[k - 1], X[k - 1]); mi[k] = max(mi[k - 1], Mm[k]); }
            const int PX = dpp_scan_max(max(lp[K - 1], X[K - 1]));
            const int PM = dpp_scan_max(mi[K - 1]);
            const int ex = max(dpp_shr1(PX, kNeg2), cx);
            const int emY = max(dpp_shr1(PM, kNeg2), cm) - Y;
            totx[g] = uni(max(yd_readlane(PX, 63), cx));
            totm[g] = uni(max(yd_readlane(PM, 63), cm));
            int pex[K], Iv[K], gm[K];
            unsigned dm = 0;
            const int ORel = O + relg;
#pragma unroll
            for (int k = 0; k < K; k++) {
                pex[k] = max(ex, lp[k]);
                Iv[k] = pex[k] - (ORel + k * E);
                gm[k] = max(Dv[k], Iv[k]);
                const int Cv = max(diag[k], gm[k]);
                cvs[g][k] = (!edge || k <= kna) ? Cv : kNeg;
                int q = (Cv - max(emY, mi[k] - Y)) >> 31;                 // all ones iff below (running best incl. this cell) - Y
                if (edge) q |= (kna - k) >> 31;
                C[g][k] = (Cv & ~q) | (kNeg & q);
                D[g][k] = Dv[k];
                dm |= (unsigned)q & (1u << k);
            }
            // sign of dnx: the gap into the NEXT column (the next lane's first) does not extend
            const int dnx = pex[K - 1] - X[K - 1];
            const int dpv = dpp_shr1(dnx, f0 - 1);                         // the same for this lane's first column (f0 = 1: it extends)
            flast[g] = uni(yd_readlane(dnx, 63)) >= 0 ? 1 : 0;
            const int hi = RY - j0;                                       // column k is right of the old window iff k >= hi
            unsigned bm = dm & (kAll << min(max(hi, 0), K));
            if (edge) bm |= kAll << min(max(kna + 1, 0), K);
            dmg[g] = dm; amg[g] = ~dm & kAll; bmg[g] = bm & kAll;
            // raw nibble of column k: bit 3 = gap into k does not extend, bit 2 = vertical gap opens, bit 1 = I beats D, bit 0 = a gap
            // beats the diagonal (tie preference diag > D > I: strict comparisons).  Two chains of two columns each.
            unsigned r01 = 0, r23 = 0;
#pragma unroll
            for (int k = 1; k >= 0; k--) {
                r01 = yd_alignbit(r01, (unsigned)(k == 0 ? dpv : pex[k - 1] - X[k - 1]), 31);
                r01 = yd_alignbit(r01, (unsigned)ddx[k], 31);
                r01 = yd_alignbit(r01, (unsigned)(Dv[k] - Iv[k]), 31);
                r01 = yd_alignbit(r01, (unsigned)(diag[k] - gm[k]), 31);
            }
#pragma unroll
            for (int k = 3; k >= 2; k--) {
                r23 = yd_alignbit(r23, (unsigned)(pex[k - 1] - X[k - 1]), 31);
                r23 = yd_alignbit(r23, (unsigned)ddx[k], 31);
                r23 = yd_alignbit(r23, (unsigned)(Dv[k] - Iv[k]), 31);
                r23 = yd_alignbit(r23, (unsigned)(diag[k] - gm[k]), 31);
            }
            const unsigned raw = (r01 | (r23 << 8)) ^ 0xCCCCu;           // bit 3 -> the gap extends (iex), bit 2 -> the vertical gap extends (dex)
            const unsigned ga = raw & 0x1111u, gi = (raw >> 1) & 0x1111u;
            tbg[g] = (raw & 0xCCCCu) | (ga & ~gi) | ((ga & gi) << 1);     // src: 0 diagonal, 1 D, 2 I
        };
        const bool edgeA = jb + kHalf - 1 > na;                            // some column of group A lies beyond the contig
        if (edgeA) group(std::integral_constant<int, 0>{}, std::true_type{}, kNeg, kNeg2, best, 0);
        else group(std::integral_constant<int, 0>{}, std::false_type{}, kNeg, kNeg2, best, 0);
        const unsigned long long blA = yd_ballot(bmg[0] != 0u);
        const bool need_b = (RY - jb > kHalf) || !blA;                    // the old window reaches into B, or no break inside A
        int allm, pbrk, first_alive, last_alive;                           // row maximum; first break (relative to jb); first / last alive column (-1: none)
        if (__builtin_expect(!need_b, 1)) {
            // the row lies inside group A (nearly every row): a break exists there, no column of B is or becomes alive.  Straight-line
            // scalar code -- for a lone wave every branch of the bookkeeping is a stall.
            allm = totm[0];
            const int lb = (int)__ffsll((long long)blA) - 1;
            pbrk = K * lb + (__ffs(yd_readlane((int)bmg[0], lb)) - 1);
            const unsigned long long alA = yd_ballot(amg[0] != 0u);
            const int lf = ((int)__ffsll((long long)alA) - 1) & 63, ll = (63 - (int)__clzll((long long)alA)) & 63;      // (alA == 0: any lane, masked below)
            const int fa = jb + K * lf + (__ffs(yd_readlane((int)amg[0], lf)) - 1);
            const int la = jb + K * ll + (31 - __clz(yd_readlane((int)amg[0], ll)));
            first_alive = alA ? fa : -1;
            last_alive = alA ? la : -1;
        } else {
            group(std::integral_constant<int, 1>{}, std::true_type{}, cplB0, totx[0], totm[0], flast[0]);   // (rare: always the masking form)
            const unsigned long long blB = yd_ballot(bmg[1] != 0u);
            if (!blA && !blB) { overflow = 1; break; }                    // every column up to the last lane is still alive
            allm = totm[1];                                               // (carries make the last total the overall one)
            if (blA) { const int lb = (int)__ffsll((long long)blA) - 1; pbrk = K * lb + (__ffs(yd_readlane((int)bmg[0], lb)) - 1); }
            else { const int lb = (int)__ffsll((long long)blB) - 1; pbrk = kHalf + K * lb + (__ffs(yd_readlane((int)bmg[1], lb)) - 1); }
            first_alive = -1; last_alive = -1;
            const unsigned long long alA = yd_ballot(amg[0] != 0u), alB = yd_ballot(amg[1] != 0u);
            if (alA) { const int lf = (int)__ffsll((long long)alA) - 1; first_alive = jb + K * lf + (__ffs(yd_readlane((int)amg[0], lf)) - 1); }
            else if (alB) { const int lf = (int)__ffsll((long long)alB) - 1; first_alive = jb + kHalf + K * lf + (__ffs(yd_readlane((int)amg[1], lf)) - 1); }
            if (alB) { const int ll = 63 - (int)__clzll((long long)alB); last_alive = jb + kHalf + K * ll + (31 - __clz(yd_readlane((int)amg[1], ll))); }
            else if (alA) { const int ll = 63 - (int)__clzll((long long)alA); last_alive = jb + K * ll + (31 - __clz(yd_readlane((int)amg[0], ll))); }
        }
        const int nvalid = min(pbrk + ((jb + pbrk) <= na ? 1 : 0), kCap);
        if (allm > best) {
            // the first cell of the row that reaches the new best
            unsigned wmA = 0, wmB = 0;
#pragma unroll
            for (int k = 0; k < K; k++) { wmA |= cvs[0][k] == allm ? (1u << k) : 0u; if (need_b) wmB |= cvs[1][k] == allm ? (1u << k) : 0u; }
            const unsigned long long wlA = yd_ballot(wmA != 0u), wlB = yd_ballot(wmB != 0u);
            int bjn;
            if (wlA) { const int lw = (int)__ffsll((long long)wlA) - 1; bjn = jb + K * lw + (__ffs(yd_readlane((int)wmA, lw)) - 1); }
            else { const int lw = (int)__ffsll((long long)wlB) - 1; bjn = jb + kHalf + K * (lw & 63) + (__ffs(yd_readlane((int)wmB, lw & 63)) - 1); }
            best = allm; bi = i; bj = bjn;
        }
        // ---- trace codes of the lane's columns
        // 4-bit codes, two columns per byte (0.5 B per cell): one 16-bit store per lane and group
        uint8_t *rowp = arena + blk_off + blk_used;
        if (K * lane < nvalid) *(uint16_t *)(rowp + (K / 2) * lane) = (uint16_t)tbg[0];
        if (need_b && kHalf + K * lane < nvalid) *(uint16_t *)(rowp + kHalf / 2 + (K / 2) * lane) = (uint16_t)tbg[1];
        blk_used += ((unsigned)(nvalid + K - 1) & ~(unsigned)(K - 1)) >> 1;
        cells += nvalid - (LY - jb);
        rows++;
        if (first_alive < 0) { i++; break; }
        LY = first_alive;
        RY = last_alive + 1;
        if (__builtin_expect(i == evt, 0)) {
            evt = uni(next_event(i));
            // state after row i
            uint8_t *sp = snaps + (size_t)(pr.snap_idx + (i == stop_at ? 1 : i == pr.snap_row ? 0 : i == pr.snap_row2 ? 2 : 3)) * kSnapBytes;
            int *sC = (int *)(sp + sizeof(SnapHdr)), *sD = sC + kSnapCols;
            int lmax = kNeg2, lj = 0;
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int j0 = jb + g * kHalf + K * lane;
                if (j0 >= LY && j0 + K <= RY) {                              // (all four columns inside the window: nearly every lane that writes at all)
                    yd_st_agent4(&sC[j0 - LY], C[g][0], C[g][1], C[g][2], C[g][3]);
                    yd_st_agent4(&sD[j0 - LY], D[g][0], D[g][1], D[g][2], D[g][3]);
                }
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const int j = j0 + k;
                    if (j >= LY && j < RY) {
                        if (!(j0 >= LY && j0 + K <= RY)) { yd_st_agent(&sC[j - LY], C[g][k]); yd_st_agent(&sD[j - LY], D[g][k]); }
                        if (C[g][k] > lmax) { lmax = C[g][k]; lj = j; }
                    }
                }
            }
            // best cell of the row, leftmost on ties: (score, -column) maximum over the wave
            const int wmax = uni(yd_readlane(dpp_scan_max(lmax), 63));
            const int cand_j = lmax == wmax ? lj : 0x7fffffff;
            exit_j = -uni(yd_readlane(dpp_scan_max(-cand_j), 63));
            yd_fence();                                                   // (every lane's C / D before the header that announces them)
            if (lane == 0) {
                SnapHdr *h = (SnapHdr *)sp;
                yd_st_agent(&h->LY, LY); yd_st_agent(&h->RY, RY); yd_st_agent(&h->best, best); yd_st_agent(&h->bi, bi); yd_st_agent(&h->bj, bj);
                yd_st_agent(&h->row, i); yd_st_agent(&h->rows, rows); yd_st_agent(&h->cells, (int64_t)cells);
                yd_st_agent(&h->valid, 1);
                yd_fence();                                               // (the header's fields before the stamp that vouches for them)
                yd_st_release(&h->stamp, stamp);
            }
            if (i == stop_at) {
                // ---- the hand-over, checked here (see the head of this file)
                int nstop = 0;
                if (aim >= 0 && pr.cap_row > stop_at) {
                    const DpProb *R = probs + aim;
                    const int r_t0 = uni(R->t0), r_q0 = uni(R->q0), r_s2 = uni(R->snap_row2), r_s3 = uni(R->snap_row3);
                    const int drow = (r_q0 - pr.q0) * dir, shift = (r_t0 - pr.t0) * dir;
                    SnapHdr *nh = (SnapHdr *)(snaps + (size_t)(kSnapSlots * aim + (ck == 0 ? 0 : ck + 1)) * kSnapBytes);
                    const int seen = uni(yd_ld_acquire(&nh->stamp));
                    if (seen == stamp && uni(yd_ld_agent(&nh->valid)) != 0) {
                        const int n_ly = uni(yd_ld_agent(&nh->LY)), n_ry = uni(yd_ld_agent(&nh->RY)), n_best = uni(yd_ld_agent(&nh->best)),
                                  n_row = uni(yd_ld_agent(&nh->row));
                        int bad = !(i == n_row + drow && LY == n_ly + shift && RY == n_ry + shift);
                        if (!bad) {
                            int *NC = (int *)((uint8_t *)nh + sizeof(SnapHdr)), *ND = NC + kSnapCols;
                            const int c = best - n_best, thr = best - Y;
#pragma unroll
                            for (int g = 0; g < G; g++)
#pragma unroll
                                for (int k = 0; k < K; k++) {
                                    const int j = jb + g * kHalf + K * lane + k;
                                    if (j >= LY && j < RY) {
                                        const int ec = C[g][k], ed = D[g][k], nc = yd_ld_agent(NC + (j - LY)), nd = yd_ld_agent(ND + (j - LY));
                                        const bool ea = ec != kNeg, na_ = nc != kNeg;
                                        if (ea != na_ || (ea && ec != nc + c)) bad = 1;
                                        const bool el = ed - E >= thr, nl = nd > kNeg2 && nd + c - E >= thr;
                                        if (el != nl || (el && ed != nd + c)) bad = 1;
                                    }
                                }
                        }
                        // (force_mod, a test knob: n > 0 rejects the first check of every n-th piece, -n the first n checks of every piece)
                        const bool rejected = yd_ballot(bad != 0) != 0ull || (force_mod > 0 && n_checks == 0 && (first + pi) % force_mod == 0) || (force_mod < 0 && n_checks < -force_mod);
                        n_checks++;
                        if (rejected) {
                            // where the host's continuation would go (gapped_phase, make_cont): the same relay's next entry snapshot, else the
                            // first relay further down the chain whose entry row is still ahead
                            int a2 = aim, c2 = ck;
                            if (ck < 2) {
                                const int sn = ck == 0 ? r_s2 : r_s3;
                                if (sn > 0 && drow + sn > stop_at) { nstop = drow + sn; c2 = ck + 1; }
                            }
                            if (!nstop) {
                                a2 = uni(R->aim1) - 1;
                                while (a2 >= 0) {
                                    const DpProb *R2 = probs + a2;
                                    const int er = (uni(R2->q0) - pr.q0) * dir + uni(R2->snap_row);
                                    if (er > stop_at + 64) { nstop = er; c2 = 0; break; }
                                    a2 = uni(R2->aim1) - 1;
                                }
                            }
                            if (nstop > 0 && nstop <= pr.cap_row) { aim = a2; ck = c2; }
                            else nstop = 0;
                        }
                    }
                }
                if (!nstop) { stopped = 1; i++; break; }
                stop_at = nstop;
                evt = uni(next_event(i));
            }
        }
    }
    if (!overflow) flush_rows(i - 1 - row_lo);
    if (lane == 0) {
        out->best = best; out->bi = bi; out->bj = bj; out->rows = rows;
        out->cells = cells; out->clocks = yd_clock() - clk0; out->overflow = overflow; out->n_ops = 0; out->stopped = stopped; out->exit_j = exit_j;
        out->fin_stop = stop_at; out->fin_aim1 = aim + 1; out->fin_ck = ck; out->fin_checks = n_checks;
        if (pr.vjob1 > 0 && aim >= 0 && vjobs && !overflow) {                          // the hand-over k_verify is to judge: the one the piece ended at
            const DpProb *R = probs + aim;
            VerifyJob vj;
            vj.eslot = pr.snap_idx + 1; vj.nslot = kSnapSlots * aim + (ck == 0 ? 0 : ck + 1);
            vj.shift = (R->t0 - pr.t0) * dir; vj.drow = (R->q0 - pr.q0) * dir;
            vjobs[pr.vjob1 - 1] = vj;
        }
    }
}

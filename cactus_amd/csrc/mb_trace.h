// mb_trace.h -- the traceback of the gapped stage (gfx950, wave64): every piece of a side walked at once by a wave of its own, then stitched.
// Included by mb_kernels.hip inside namespace mb after the wave helpers (uni, RowInfo, kRowChunk) -- and, with MB_EMU defined, by the
// host-side emulation under tests/emu (emu_ydrop.cpp).
#pragma once

// ------------------------------------------------------------------------------------------------
// traceback.  walk_piece moves one wave from a cell of a piece down to the piece's floor.  Memory latency is taken
// off the chain by fetching, for 64 rows at a time, each row's (offset, first column) record and the 8 trace bytes
// around the column a gap-free path would visit (lane l <-> row i-l, columns j-l-3 .. j-l+4).  Inside a block, whole
// runs of diagonal steps are recognised with one ballot (all lanes test "src == diag" at the current drift); only gap
// cells are stepped one at a time.  Output: run-length ops (len << 2 | op) in walk-back order; op 0 aligned pair,
// 2 query-only, 3 target-only.
//   MODE 0 (k_trace_walk): every entered row gets a record (column, runs written so far, pending run length, state);
//   MODE 1 (k_trace_join): every entered row is compared with the record another walk left there; on equality the
//          walk stops (`joined`): from an identical (row, column, state) both walks are identical.
struct RunOut {
    uint32_t *o; int n_runs, cur_op, cur_len;
    __device__ __forceinline__ void emit(int op, int len, int lane) {
        if (op == cur_op) cur_len += len;
        else {
            if (cur_len > 0 && lane == 0) o[n_runs] = ((uint32_t)cur_len << 2) | (uint32_t)cur_op;
            n_runs += cur_len > 0 ? 1 : 0;
            cur_op = op; cur_len = len;
        }
    }
};

template <int MODE>
__device__ __forceinline__ bool walk_piece(const TbWalk &P, int &i, int &j, int &state, RunOut &ro, uint32_t *__restrict__ rec,
                                           const uint8_t *__restrict__ arena, const unsigned long long arena_bytes,
                                           const unsigned long long *__restrict__ rowdir) {
    const int lane = threadIdx.x & 63;
    const int floor = uni(P.floor), si = uni(P.si);
    auto load_ri = [&](int r) -> RowInfo {
        RowInfo ri; ri.off = 0; ri.ly = 0; ri.pad = 0;
        if (r > floor) {
            const int rho = r - P.row_lo;
            ri = ((const RowInfo *)(arena + rowdir[P.row_off + (unsigned)(rho / kRowChunk)]))[rho & (kRowChunk - 1)];
        }
        return ri;
    };
    auto rec_at = [&](int r) -> uint32_t * { return rec + 3ull * (unsigned)(si - r); };
    // the start cell is an entered row too
    if (MODE == 0) { if (lane == 0) { uint32_t *q = rec_at(i); q[0] = (uint32_t)j; q[1] = 0; q[2] = (uint32_t)state; } }
    else {
        const uint32_t *q = rec_at(i);
        if (i <= si && i > floor && q[0] == (uint32_t)j && (q[2] & 3u) == (uint32_t)state) return true;
    }
    int pre_i0 = i;
    RowInfo ri_pre = load_ri(i - lane);
    while ((i > 0 || j > 0) && i > floor) {
        // fetch block: rows i .. i-63
        i = uni(i); j = uni(j);
        const int i0 = i, j0 = j;
        const int r = i0 - lane;
        const RowInfo ri = (i0 == pre_i0) ? ri_pre : load_ri(r);
        pre_i0 = i0 - 64;
        ri_pre = load_ri(pre_i0 - lane);
        unsigned long long win = 0;
        uint32_t rj = 0xFFFFFFFFu, rs = 0;                    // MODE 1: the other walk's record of this lane's row
        if (r > floor) {
            const int wly = (int)ri.ly;
            const int wc0 = j0 - lane - 3;                     // column of byte 0 of the window
            const uint8_t *rowp = arena + ri.off;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int c = wc0 + k;
                unsigned b = 0xFFu;                             // never a diagonal source: stops runs left of the row
                // codes right of the stored row are never consulted, but a long gap can put the predicted column far
                // beyond it: never read past the arena.  Two 4-bit codes per byte, even column of the row in the low nibble.
                if (c >= wly && ri.off + (unsigned long long)((c - wly) >> 1) < arena_bytes) b = ((unsigned)rowp[(c - wly) >> 1] >> (4 * ((c - wly) & 1))) & 0xFu;
                win |= (unsigned long long)b << (8 * k);
            }
            if (MODE == 1 && r <= si) { const uint32_t *q = rec_at(r); rj = q[0]; rs = q[2] & 3u; }
        } else win = ~0ull;
        int l = 0;
        while ((i > 0 || j > 0) && l < 64 && i > floor) {
            i = uni(i); j = uni(j); state = uni(state); l = uni(l);
            const int k = j - (j0 - l - 3);                    // byte of the window that holds column j of row i
            if (k < 0 || k >= 8) break;                        // drifted out of the prefetched window: refetch
            if (state == 0) {
                // how many consecutive rows, starting at lane l, continue diagonally at this drift?
                const unsigned tbl = (unsigned)(win >> (8 * k)) & 0xFFu;
                const unsigned long long stop = __ballot(lane >= l && (tbl & 3u) != 0u);
                int run = stop ? (int)__ffsll((long long)stop) - 1 - l : 64 - l;
                if (run > 0) {
                    // rows i-1 .. i-run are entered at columns j-1 .. j-run (lane l+x holds row i-x while l+x < 64)
                    const int x = lane - l;
                    const bool entered = x >= 1 && x <= run && (i - x) > floor;
                    if (MODE == 1) {
                        const unsigned long long hit = __ballot(entered && rj == (uint32_t)(j - x) && rs == 0u);
                        if (hit) { const int xs = (int)__ffsll((long long)hit) - 1 - l; ro.emit(0, xs, lane); i -= xs; j -= xs; return true; }
                        // row i-run may be held by no lane (l + run == 64): it is checked as the start of the next block
                    } else if (entered) {
                        const int base = ro.cur_op == 0 ? ro.cur_len : 0;
                        const int nr = ro.n_runs + ((ro.cur_op != 0 && ro.cur_len > 0) ? 1 : 0);
                        uint32_t *q = rec_at(i - x);
                        q[0] = (uint32_t)(j - x); q[1] = (uint32_t)nr; q[2] = ((uint32_t)(base + x) << 2);
                    }
                    ro.emit(0, run, lane); i -= run; j -= run; l += run;
                    if (l == 64 && i > floor) {
                        // the row just entered starts the next block: handle its record here
                        if (MODE == 0) { if (lane == 0) { uint32_t *q = rec_at(i); q[0] = (uint32_t)j; q[1] = (uint32_t)ro.n_runs; q[2] = ((uint32_t)ro.cur_len << 2); } }
                        else if (i <= si) { const uint32_t *q = rec_at(i); if (q[0] == (uint32_t)j && (q[2] & 3u) == 0u) return true; }
                    }
                    continue;
                }
                const unsigned src = (unsigned)__builtin_amdgcn_readlane((int)tbl, l) & 3u;
                if (src == 1u) state = 1;
                else if (src == 2u) state = 2;
                else { i = 0; j = 0; }
            } else {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)win, l);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(win >> 32), l);
                const unsigned tb = ((k < 4 ? lo >> (8 * k) : hi >> (8 * (k - 4)))) & 0xFFu;
                if (state == 1) {
                    ro.emit(2, 1, lane); if (!(tb & 4u)) state = 0; i--; l++;
                    if (i > floor) {                           // entered row i at column j in `state`
                        if (MODE == 0) {
                            if (lane == 0) { uint32_t *q = rec_at(i); q[0] = (uint32_t)j; q[1] = (uint32_t)ro.n_runs; q[2] = ((uint32_t)ro.cur_len << 2) | (uint32_t)state; }
                        } else if (i <= si) {
                            const uint32_t *q = rec_at(i);
                            if (q[0] == (uint32_t)j && (q[2] & 3u) == (uint32_t)state) return true;
                        }
                    }
                }
                else { ro.emit(3, 1, lane); if (!(tb & 8u)) state = 0; j--; }
            }
        }
    }
    return false;
}

__global__ __launch_bounds__(64) void k_trace_walk(TbWalk *__restrict__ walks, int n, const uint8_t *__restrict__ arena,
                                                   const unsigned long long arena_bytes, const unsigned long long *__restrict__ rowdir,
                                                   uint32_t *__restrict__ ops, uint32_t *__restrict__ recs) {
    const int slot = blockIdx.x;
    if (slot >= n) return;
    const TbWalk P = walks[slot];
    const int lane = threadIdx.x & 63;
    int i = uni(P.si), j = uni(P.sj), state = 0;
    RunOut ro{ops + P.ops_off, 0, -1, 0};
    walk_piece<0>(P, i, j, state, ro, recs + P.rec_off, arena, arena_bytes, rowdir);
    // the pending run stays open in the records (a splice shortens it); it is closed here
    ro.emit(-2, 0, lane);
    if (lane == 0) { walks[slot].n_runs = ro.n_runs; walks[slot].ei = i; walks[slot].ej = j; walks[slot].estate = state; }
}

// One wave per piece that is not the head of its side: the join walk of the piece, made BEFORE the side is stitched, from the cell the
// true path enters it through if the walk of the piece before it ends where the true path does -- which it does whenever that walk and
// the true path have met inside the piece, i.e. nearly always.  k_trace_join then only compares: a prediction that turns out wrong
// (the cell it reaches the piece with is another one) makes it walk itself, as it would without this kernel.  What is saved is the
// chain of dependent memory round trips of a side's join walks, one after the other in one wave.
// TbWalk::pad holds the piece's share of the side's join buffer (first run slot, 64 bit; all ones: head of a side, no join walk).
__global__ __launch_bounds__(64) void k_trace_prejoin(const TbWalk *__restrict__ walks, int n, TbJoin *__restrict__ joins,
                                                      const uint8_t *__restrict__ arena, const unsigned long long arena_bytes,
                                                      const unsigned long long *__restrict__ rowdir, uint32_t *__restrict__ ops,
                                                      const uint32_t *__restrict__ recs, const int poison) {
    const int slot = blockIdx.x;
    if (slot >= n) return;
    const int lane = threadIdx.x & 63;
    const TbWalk W = walks[slot];
    unsigned long long jo;
    __builtin_memcpy(&jo, W.pad, 8);
    TbJoin J;
    J.pi = (int32_t)0x80000000; J.pj = 0; J.pstate = 0; J.n_runs = -1; J.joined = 0; J.ei = J.ej = J.estate = 0; J.nr = J.sub = 0;
    if (jo != ~0ull) {
        const TbWalk Wp = walks[slot - 1];
        int i = uni(Wp.ei + Wp.dr), j = uni(Wp.ej + Wp.dc), state = uni(Wp.estate);
        if (poison && (slot & 1)) j = uni(j + 1);                         // (tests: a wrong prediction must cost nothing but time)
        J.pi = i; J.pj = j; J.pstate = state;
        if (!(i == W.si && j == W.sj && state == 0)) {
            RunOut ro{ops + jo, 0, -1, 0};
            const bool joined = walk_piece<1>(W, i, j, state, ro, const_cast<uint32_t *>(recs) + W.rec_off, arena, arena_bytes, rowdir);
            ro.emit(-2, 0, lane);
            J.n_runs = ro.n_runs; J.joined = joined ? 1 : 0; J.ei = i; J.ej = j; J.estate = state;
            if (joined) {
                const uint32_t *q = recs + W.rec_off + 3ull * (unsigned)(W.si - i);
                J.nr = (int32_t)q[1]; J.sub = (int32_t)(q[2] >> 2);
            }
        }
    }
    if (lane == 0) joins[slot] = J;
}

// one wave per side: stitches the walks of the side's pieces (see TbWalk); joins: k_trace_prejoin's walks (nullptr: none were made)
__global__ __launch_bounds__(64) void k_trace_join(TbSide *__restrict__ sides, int n, const TbWalk *__restrict__ walks,
                                                   TbSeg *__restrict__ segs, const uint8_t *__restrict__ arena,
                                                   const unsigned long long arena_bytes, const unsigned long long *__restrict__ rowdir,
                                                   uint32_t *__restrict__ ops, const uint32_t *__restrict__ recs,
                                                   const TbJoin *__restrict__ joins) {
    const int slot = blockIdx.x;
    if (slot >= n) return;
    const TbSide sd = sides[slot];
    const int lane = threadIdx.x & 63;
    TbSeg *sg = segs + sd.seg_off;
    int n_segs = 0;
    auto push_seg = [&](unsigned long long src, int n_runs, int first_sub) {
        if (n_runs <= 0) return;
        if (lane == 0) { TbSeg t; t.src = src; t.n_runs = n_runs; t.first_sub = first_sub; sg[n_segs] = t; }
        n_segs++;
    };
    const TbWalk W0 = walks[sd.first_walk];
    push_seg(W0.ops_off, W0.n_runs, 0);
    int i = uni(W0.ei + W0.dr), j = uni(W0.ej + W0.dc), state = uni(W0.estate);
    // The guessed start of a piece is sometimes the cell the true path enters it through: then the whole guessed walk is spliced.
    // Otherwise the join walk k_trace_prejoin made from its predicted entry is spliced if the prediction holds.  Either way nothing
    // has to be read but the records of the walker: those of 64 walkers are fetched at once (one per lane), so the chain of
    // hand-overs costs no memory latency.
    for (int k0 = 1; k0 < sd.n_walks; k0 += 64) {
        const int kk = k0 + lane;
        TbWalk Wl;
        TbJoin Jl;
        Jl.pi = (int32_t)0x80000000; Jl.pj = 0; Jl.pstate = 0; Jl.n_runs = -1; Jl.joined = 0; Jl.ei = Jl.ej = Jl.estate = 0; Jl.nr = Jl.sub = 0;
        if (kk < sd.n_walks) { Wl = walks[sd.first_walk + kk]; if (joins) Jl = joins[sd.first_walk + kk]; }
        else { Wl.si = -1; Wl.sj = -1; Wl.dr = Wl.dc = 0; Wl.ops_off = 0; Wl.n_runs = 0; Wl.ei = Wl.ej = Wl.estate = 0; Wl.pad[0] = Wl.pad[1] = 0; }
        const int cnt = min(64, sd.n_walks - k0);
        for (int t = 0; t < cnt; t++) {
            const int si = __builtin_amdgcn_readlane(Wl.si, t), sj = __builtin_amdgcn_readlane(Wl.sj, t);
            const int wdr = __builtin_amdgcn_readlane(Wl.dr, t), wdc = __builtin_amdgcn_readlane(Wl.dc, t);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)Wl.ops_off, t);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(Wl.ops_off >> 32), t);
            const unsigned long long w_ops = ((unsigned long long)hi << 32) | lo;
            const unsigned long long jo = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(Wl.pad[1], t) << 32) | (unsigned)__builtin_amdgcn_readlane(Wl.pad[0], t);
            if (i == si && j == sj && state == 0) {
                push_seg(w_ops, __builtin_amdgcn_readlane(Wl.n_runs, t), 0);
                i = __builtin_amdgcn_readlane(Wl.ei, t); j = __builtin_amdgcn_readlane(Wl.ej, t); state = __builtin_amdgcn_readlane(Wl.estate, t);
            } else if (__builtin_amdgcn_readlane(Jl.n_runs, t) >= 0 && i == __builtin_amdgcn_readlane(Jl.pi, t) && j == __builtin_amdgcn_readlane(Jl.pj, t) &&
                       state == __builtin_amdgcn_readlane(Jl.pstate, t)) {
                push_seg(jo, __builtin_amdgcn_readlane(Jl.n_runs, t), 0);
                if (__builtin_amdgcn_readlane(Jl.joined, t)) {
                    const int nr = __builtin_amdgcn_readlane(Jl.nr, t);
                    push_seg(w_ops + (unsigned)nr, __builtin_amdgcn_readlane(Wl.n_runs, t) - nr, __builtin_amdgcn_readlane(Jl.sub, t));
                    i = __builtin_amdgcn_readlane(Wl.ei, t); j = __builtin_amdgcn_readlane(Wl.ej, t); state = __builtin_amdgcn_readlane(Wl.estate, t);
                } else {
                    i = __builtin_amdgcn_readlane(Jl.ei, t); j = __builtin_amdgcn_readlane(Jl.ej, t); state = __builtin_amdgcn_readlane(Jl.estate, t);
                }
            } else {
                const TbWalk W = walks[sd.first_walk + k0 + t];
                RunOut ro{ops + jo, 0, -1, 0};
                const bool joined = walk_piece<1>(W, i, j, state, ro, const_cast<uint32_t *>(recs) + W.rec_off, arena, arena_bytes, rowdir);
                ro.emit(-2, 0, lane);
                push_seg(jo, ro.n_runs, 0);
                if (joined) {
                    const uint32_t *q = recs + W.rec_off + 3ull * (unsigned)(W.si - i);
                    const int nr = (int)q[1], sub = (int)(q[2] >> 2);
                    push_seg(W.ops_off + (unsigned)nr, W.n_runs - nr, sub);
                    i = W.ei; j = W.ej; state = W.estate;
                }
            }
            i = uni(i + wdr); j = uni(j + wdc); state = uni(state);
        }
    }
    if (lane == 0) sides[slot].n_segs = n_segs;
}

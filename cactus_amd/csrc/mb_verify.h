// mb_verify.h -- the relay hand-over check of the gapped stage (gfx950).  Included by mb_kernels.hip inside namespace mb (and, with MB_EMU
// defined, by the host-side emulation under tests/emu: emu_ydrop.cpp).
#pragma once

// relay hand-over check: the upstream piece's state after its exit row against the relay's state after the same row.
// Equal means: same window, and every value that can still matter equal up to ONE constant -- C of live cells (dead
// cells are stored as kNeg in both), D wherever D - E can still reach (best - Y); lower D values can never lift a cell
// over the threshold again, whatever they are.  The recurrence commutes with adding a constant to a state, so equal
// states evolve identically from here on.
__global__ __launch_bounds__(256) void k_verify(const VerifyJob *__restrict__ jobs, VerifyOut *__restrict__ res, int n,
                                                const uint8_t *__restrict__ snaps, int Y, int E) {
    const int ji = blockIdx.x;
    if (ji >= n) return;
    const VerifyJob jb = jobs[ji];
    const uint8_t *ep = snaps + (size_t)jb.eslot * kSnapBytes, *np = snaps + (size_t)jb.nslot * kSnapBytes;
    const SnapHdr eh = *(const SnapHdr *)ep, nh = *(const SnapHdr *)np;
    const int *EC = (const int *)(ep + sizeof(SnapHdr)), *ED = EC + kSnapCols;
    const int *NC = (const int *)(np + sizeof(SnapHdr)), *ND = NC + kSnapCols;
    int bad = !(eh.valid && nh.valid && eh.row == nh.row + jb.drow && eh.LY == nh.LY + jb.shift && eh.RY == nh.RY + jb.shift);
    const int c = eh.best - nh.best;
    if (!bad) {
        const int thr = eh.best - Y;
        const int w = eh.RY - eh.LY;
        for (int x = threadIdx.x; x < w; x += blockDim.x) {
            const int ec = EC[x], nc = NC[x];
            const bool ea = ec != kNeg, na = nc != kNeg;
            if (ea != na || (ea && ec != nc + c)) bad = 1;
            const int ed = ED[x], nd = ND[x];
            const bool el = ed - E >= thr, nl = nd > kNeg2 && nd + c - E >= thr;
            if (el != nl || (el && ed != nd + c)) bad = 1;
        }
    }
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) {
        VerifyOut o;
        o.ok = !bad; o.n_rows = nh.rows; o.n_best = nh.best; o.c = c; o.n_cells = nh.cells;
        res[ji] = o;
    }
}

// mb_units.h -- which seed unit (chunk pair, strand) a hit belongs to, and per-unit counters (gfx950, wave64).  Included inside
// namespace mb by mb_kernels.hip (and, with MB_EMU defined, by the host-side emulation under tests/emu) before the ungapped kernels.
#pragma once

// Pointers read from a table in memory are generic for the compiler, which then emits FLAT loads; the sequence pointers of a unit
// are global memory: a round trip through address space 1 says so (InferAddressSpaces rewrites the loads that use the result).
#ifndef MB_EMU
__device__ __forceinline__ const uint8_t *unit_glob(const uint8_t *p) {
    return (const uint8_t *)(const uint8_t __attribute__((address_space(1))) *)(unsigned long long)p;
}
#else
inline const uint8_t *unit_glob(const uint8_t *p) { return p; }
#endif
#ifndef MB_EMU
__device__ __forceinline__ const SeedUnit *unit_table(const UnitTab &ut) {
    return (const SeedUnit *)(const SeedUnit __attribute__((address_space(1))) *)(unsigned long long)ut.tab;
}
#else
inline const SeedUnit *unit_table(const UnitTab &ut) { return ut.tab; }
#endif

// extent[] of a launch -- the right end of the last extension on a diagonal, carried from one q-ordered batch of a strand's hits to the
// next.  A strand whose hits are ONE batch needs none: the pipeline passes nullptr (every diagonal starts at 0, nothing is kept), which
// saves the fill of 4 B per diagonal (240 MB per strand of a 30 Mb x 30 Mb chunk pair) and a random read + write per diagonal run.
__device__ __forceinline__ uint32_t extent_slot(const UnitTab &ut, const uint32_t dq) { return ut.ext_mul ? (dq * ut.ext_mul) & ut.ext_mask : dq; }
__device__ __forceinline__ int32_t extent_get(const int32_t *__restrict__ extent, const UnitTab &ut, const uint32_t dq) { return extent ? extent[extent_slot(ut, dq)] : 0; }
__device__ __forceinline__ void extent_put(int32_t *__restrict__ extent, const UnitTab &ut, const uint32_t dq, const int32_t v) { if (extent) extent[extent_slot(ut, dq)] = v; }

struct UnitRef {
    const uint8_t *tc, *qc;
    int64_t qoff;                             // t_end = (int64_t)dq - qoff + q_end      (qoff = dbase + qtot)
    int32_t id;
};

// the unit owning diagonal dq: the last one whose dbase is <= dq (the table ascends)
__device__ __forceinline__ int unit_index(const UnitTab &ut, const uint32_t dq) {
    const SeedUnit *tab = unit_table(ut);
    int lo = 0, hi = ut.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].dbase <= dq) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ UnitRef unit_of(const UnitTab &ut, const uint32_t dq) {
    UnitRef r;
    if (ut.n <= 1) { r.tc = unit_glob(ut.one.tc); r.qc = unit_glob(ut.one.qc); r.qoff = (int64_t)ut.one.dbase + ut.one.qtot; r.id = 0; return r; }
    const int u = unit_index(ut, dq);
    const SeedUnit &su = unit_table(ut)[u];
    r.tc = unit_glob(su.tc); r.qc = unit_glob(su.qc); r.qoff = (int64_t)su.dbase + su.qtot; r.id = u;
    return r;
}
__device__ __forceinline__ UnitRef unit_by_id(const UnitTab &ut, const int id) {
    const SeedUnit &su = ut.n <= 1 ? ut.one : unit_table(ut)[id];
    UnitRef r;
    r.tc = unit_glob(su.tc); r.qc = unit_glob(su.qc); r.qoff = (int64_t)su.dbase + su.qtot; r.id = id;
    return r;
}

// sum over the 64 lanes of the wave, in every lane
__device__ __forceinline__ int wave_sum(int v) {
    v += wdpp<0xB1, 0xf>(0, v);                                        // quad_perm [1,0,3,2]
    v += wdpp<0x4E, 0xf>(0, v);                                        // quad_perm [2,3,0,1]
    v += wdpp<0x141, 0xf>(0, v);                                       // row_half_mirror: every lane holds the sum of its 8-lane group
    int t = 0;
#pragma unroll
    for (int g = 0; g < 8; g++) t += wreadlane(v, 8 * g);
    return t;
}

// Adds a lane's numbers to the counters of its unit.  Every lane of the wave must call it (a lane with nothing to add passes
// zeros).  The lanes of a wave nearly always share the unit: one pair of atomics per wave then, else one per lane.
__device__ __forceinline__ void unit_count(UngappedCounters *__restrict__ ctr, const int unit, unsigned long long n_ext, unsigned long long n_cols) {
    const bool has = (n_ext | n_cols) != 0ull;
    const unsigned long long m = wballot(has);
    if (!m) return;
    const int u0 = wreadlane(unit, (int)__ffsll((long long)m) - 1);
    if (wballot(has && unit != u0)) {
        if (has) { atomicAdd(&ctr[unit].extended, n_ext); atomicAdd(&ctr[unit].cols, n_cols); }
        return;
    }
    // 24-bit slices so that 64 of them cannot overflow the 32-bit wave sums
    const int e0 = wave_sum((int)(n_ext & 0xFFFFFFu)), e1 = wave_sum((int)(n_ext >> 24));
    const int c0 = wave_sum((int)(n_cols & 0xFFFFFFu)), c1 = wave_sum((int)((n_cols >> 24) & 0xFFFFFFu)), c2 = wave_sum((int)(n_cols >> 48));
    if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)m) - 1)) {
        atomicAdd(&ctr[u0].extended, (unsigned long long)(unsigned)e0 + ((unsigned long long)(unsigned)e1 << 24));
        atomicAdd(&ctr[u0].cols, (unsigned long long)(unsigned)c0 + ((unsigned long long)(unsigned)c1 << 24) + ((unsigned long long)(unsigned)c2 << 48));
    }
}

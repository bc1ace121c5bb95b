// mb_guard.h -- MIBLAST_DEBUG_GUARD: every device allocation of the library at its EXACT size with a canary behind it.
//
// The normal allocators round sizes up (DevBuf::ensure adds a quarter, the block cache rounds to 4 096 bytes and hands out blocks up to
// twice the size asked for, the chaining stage's cache up to four times): a kernel that writes a few elements past what the host sized is
// absorbed by that slack on most runs and faults -- or corrupts a neighbour -- on the run whose layout leaves none.  Under
//   MIBLAST_DEBUG_GUARD=1   an allocation is hipMalloc(bytes + 4 KiB), the 4 KiB behind the last byte asked for are a pattern (0xA5) that is
//                           checked whenever the block is freed and by guard::check_all() (the pipeline calls it at the end of every stage:
//                           the device is synchronised first); no head-room, no recycling of blocks between owners;
//   MIBLAST_DEBUG_GUARD=2   as 1, and the block itself starts as 0xCD bytes: state that a kernel reads before anything wrote it shows as
//                           wild indices / scores at once instead of as whatever a recycled page held.
//   MIBLAST_DEBUG_GUARD=3   an electric fence: the block is mapped with the virtual-memory calls (hipMemAddressReserve / hipMemCreate / hipMemMap) so
//                           that its last byte (rounded up to 256, hipMalloc's alignment; MIBLAST_DEBUG_GUARD_ALIGN) is the LAST MAPPED byte -- the address range behind it is reserved and never mapped.
//                           A kernel that READS or writes one element past what the host sized faults at once, on every run, instead of on the
//                           run whose layout puts an unmapped page there (round 5's GPUTEST fault was of that kind: not reproduced in 400 runs
//                           of the same command).  The runtime then names the address; the SIGABRT handler of this mode prints the table of live
//                           blocks (tag, range) so that the address can be read as "N bytes behind block X".  Blocks start as 0xCD as in 2.
// A damaged canary is reported with the allocation's tag, size and the first damaged offset, and the process aborts.
// Debug facility: outputs are unchanged by it (the GPU suite runs under it once per round: profiles/r06_guard_suite.log).
#pragma once
#include <hip/hip_runtime.h>
#include <signal.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace mb {
// hipMalloc / hipFree calls made by the library since the process started (every site counts itself).  Either call in the middle of a step
// stalls every lane of a call -- hipFree waits for an idle device, and with the runtime polling the wait is 0.3 - 1 s --
// so bench.py reads this around its timed steps: device_allocs_in_timed_steps on the line must be 0 (miblast_debug_device_allocs()).
inline std::atomic<long long> &device_alloc_calls() { static std::atomic<long long> n{0}; return n; }
inline void count_device_alloc() { device_alloc_calls().fetch_add(1, std::memory_order_relaxed); }
// MIBLAST_DEBUG_ALLOC=2: every device allocation / release of the library on stderr (which buffer still grows in the middle of a job?)
inline void note_device_alloc(const char *what, size_t bytes) {
    static const int lv = [] { const char *v = getenv("MIBLAST_DEBUG_ALLOC"); return v && *v ? atoi(v) : 0; }();
    if (lv >= 2) {
        static const auto t0 = std::chrono::steady_clock::now();
        fprintf(stderr, "[miblast] %9.3f s  device allocation: %.3f MB  %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), bytes / 1e6, what);
    }
}

namespace guard {

constexpr size_t kCanary = 4096;

inline int level() {
    static const int lv = [] { const char *v = getenv("MIBLAST_DEBUG_GUARD"); return v && *v ? atoi(v) : 0; }();
    return lv;
}
inline bool on() { return level() > 0; }

#ifdef MB_EMU_HIP_STANDIN
struct Rec { size_t bytes; const char *tag; int device; void *base = nullptr; size_t mapped = 0, reserved = 0; };      // (host emulation: levels 1 and 2 only)
#else
struct Rec { size_t bytes; const char *tag; int device; void *base = nullptr; size_t mapped = 0, reserved = 0; hipMemGenericAllocationHandle_t handle = {}; };
#endif
struct Registry {
    std::mutex mu;
    std::unordered_map<void *, Rec> live;
    unsigned long long n_alloc = 0, n_check = 0;
};
inline void report();
inline Registry &registry() { static Registry *r = [] { atexit(report); return new Registry(); }(); return *r; }          // (never destroyed: frees may come after main)

[[noreturn]] inline void die(const char *what, void *p, const Rec &r, long off, const char *where) {
    if (const char *path = getenv("MIBLAST_DEBUG_GUARD_LOG")) if (FILE *f = fopen(path, "a")) { fprintf(f, "[miblast guard] pid %d: %s: allocation '%s' (%zu bytes), canary byte +%ld, seen at: %s\n", (int)getpid(), what, r.tag ? r.tag : "?", r.bytes, off, where ? where : "?"); fclose(f); }
    fprintf(stderr, "[miblast guard] %s: allocation '%s' (%zu bytes at %p, device %d): first damaged canary byte at +%ld behind its end (seen at: %s)\n",
            what, r.tag ? r.tag : "?", r.bytes, p, r.device, off, where ? where : "?");
    fflush(stderr);
    abort();
}

inline long first_damage(void *p, const Rec &r) {
    if (r.base) return -1;                                 // (fenced block: nothing mapped behind it to look at)
    static thread_local std::vector<unsigned char> host(kCanary);
    if (hipMemcpy(host.data(), (char *)p + r.bytes, kCanary, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    for (size_t i = 0; i < kCanary; i++) if (host[i] != 0xA5) return (long)i;
    return -1;
}

// hipMalloc's stand-in.  Returns hipSuccess / the error of hipMalloc (the trace arena retries smaller on failure).
inline void dump_live(int);
#ifdef MB_EMU_HIP_STANDIN
inline hipError_t fenced(void **, size_t, Rec &) { return hipErrorInvalidValue; }
#else
inline hipError_t fenced(void **out, size_t bytes, Rec &r) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = r.device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran < 4096) gran = 4096;
    // the block's start keeps the alignment hipMalloc gives every block (256 bytes: rocprim lays its temporary storage out from it, kernels
    // load 16 bytes at a time): an overrun of fewer bytes than the padding this costs goes unseen -- MIBLAST_DEBUG_GUARD_ALIGN=16 narrows it
    static const size_t align = [] { const char *v = getenv("MIBLAST_DEBUG_GUARD_ALIGN"); const long a = v && *v ? atol(v) : 256; return (size_t)(a >= 4 && (a & (a - 1)) == 0 ? a : 256); }();
    const size_t user = (bytes + align - 1) & ~(align - 1);
    r.mapped = (user + gran - 1) / gran * gran;
    if (r.mapped == 0) r.mapped = gran;
    r.reserved = r.mapped + gran;
    if ((e = hipMemAddressReserve(&r.base, r.reserved, gran, nullptr, 0)) != hipSuccess) return e;
    if ((e = hipMemCreate(&r.handle, r.mapped, &prop, 0)) != hipSuccess) { (void)hipMemAddressFree(r.base, r.reserved); return e; }
    if ((e = hipMemMap(r.base, r.mapped, 0, r.handle, 0)) != hipSuccess) { (void)hipMemRelease(r.handle); (void)hipMemAddressFree(r.base, r.reserved); return e; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(r.base, r.mapped, &acc, 1)) != hipSuccess) { (void)hipMemUnmap(r.base, r.mapped); (void)hipMemRelease(r.handle); (void)hipMemAddressFree(r.base, r.reserved); return e; }
    (void)hipMemset(r.base, 0xCD, r.mapped);
    *out = (char *)r.base + (r.mapped - user);
    return hipSuccess;
}

#endif

inline hipError_t alloc(void **out, size_t bytes, const char *tag) {
    void *p = nullptr;
    int dev = 0; (void)hipGetDevice(&dev);
    Rec r{bytes, tag, dev};
    // MIBLAST_DEBUG_GUARD_ONLY=<text>: only the blocks whose tag contains <text> get the fence, the others a canary (to find WHICH block a
    // kernel overruns when the runtime's fault report names no usable address)
    static const char *only = getenv("MIBLAST_DEBUG_GUARD_ONLY");
    if (level() >= 3 && (!only || !*only || (tag && strstr(tag, only)))) {
        static const bool handler = [] { signal(SIGABRT, dump_live); return true; }();
        (void)handler;
        hipError_t e = fenced(&p, bytes, r);
        if (e != hipSuccess) { *out = nullptr; return e; }
    } else {
        hipError_t e = hipMalloc(&p, bytes + kCanary);
        if (e != hipSuccess) { *out = nullptr; return e; }
        if (level() > 1 && bytes) (void)hipMemset(p, 0xCD, bytes);
        (void)hipMemset((char *)p + bytes, 0xA5, kCanary);
    }
    (void)hipDeviceSynchronize();
    Registry &g = registry();
    { std::lock_guard<std::mutex> lk(g.mu); g.live[p] = r; g.n_alloc++; }
    *out = p;
    return hipSuccess;
}

// SIGABRT in fence mode (the runtime aborts after naming the faulting address): the live blocks, so that the address can be placed
inline void dump_live(int) {
    Registry &g = registry();
    fprintf(stderr, "[miblast guard] live blocks at the abort (an address just behind an `end` is an overrun of that block):\n");
    for (auto &kv : g.live)
        fprintf(stderr, "[miblast guard]   %p .. end %p  %zu bytes  %s\n", kv.first, (void *)((char *)kv.first + kv.second.bytes), kv.second.bytes, kv.second.tag ? kv.second.tag : "?");
    fflush(stderr);
    if (const char *path = getenv("MIBLAST_DEBUG_GUARD_LOG")) if (FILE *f = fopen(path, "a")) {
        fprintf(f, "[miblast guard] pid %d: ABORT (device fault?) with %zu live blocks:\n", (int)getpid(), g.live.size());
        for (auto &kv : g.live) fprintf(f, "    %p .. end %p  %zu bytes  %s\n", kv.first, (void *)((char *)kv.first + kv.second.bytes), kv.second.bytes, kv.second.tag ? kv.second.tag : "?");
        fclose(f);
    }
    signal(SIGABRT, SIG_DFL);
    abort();
}

inline void free(void *p, const char *where = "free") {
    if (!p) return;
    Registry &g = registry();
    Rec r{0, nullptr, 0}; bool known = false;
    { std::lock_guard<std::mutex> lk(g.mu); auto it = g.live.find(p); if (it != g.live.end()) { r = it->second; known = true; g.live.erase(it); } }
    if (known) {
        int cur = 0; (void)hipGetDevice(&cur);
        if (cur != r.device) (void)hipSetDevice(r.device);
        (void)hipDeviceSynchronize();
        const long off = first_damage(p, r);
        if (off >= 0) die("overrun found on free", p, r, off, where);
#ifndef MB_EMU_HIP_STANDIN
        if (r.base) { (void)hipMemUnmap(r.base, r.mapped); (void)hipMemRelease(r.handle); (void)hipMemAddressFree(r.base, r.reserved); }
#endif
        if (cur != r.device) (void)hipSetDevice(cur);
        if (r.base) return;
    }
    (void)hipFree(p);
}

// every live allocation's canary (of the current device), after the device has finished what is queued
inline void check_all(const char *where) {
    if (!on()) return;
    int cur = 0; (void)hipGetDevice(&cur);
    (void)hipDeviceSynchronize();
    Registry &g = registry();
    std::lock_guard<std::mutex> lk(g.mu);                  // (held over the sweep: a block on the list cannot be freed under it -- free() takes the block off the list first)
    g.n_check++;
    for (auto &kv : g.live) {
        if (kv.second.device != cur) continue;
        const long off = first_damage(kv.first, kv.second);
        if (off >= 0) die("overrun", kv.first, kv.second, off, where);
    }
}

// at process exit, one line per process appended to the file MIBLAST_DEBUG_GUARD_LOG names (the front ends' stderr stays empty)
inline void report() {
    const char *path = getenv("MIBLAST_DEBUG_GUARD_LOG");
    if (!on() || !path || !*path) return;
    Registry &g = registry();
    std::lock_guard<std::mutex> lk(g.mu);
    if (FILE *f = fopen(path, "a")) {
        fprintf(f, "[miblast guard] pid %d level %d: %llu guarded allocations, %llu sweeps over the live ones, %zu live at exit, no canary damaged, no fault\n", (int)getpid(), level(), g.n_alloc,
                g.n_check, g.live.size());
        fclose(f);
    }
}

}  // namespace guard
}  // namespace mb

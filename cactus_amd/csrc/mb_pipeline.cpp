// mb_pipeline.cpp -- host orchestration of one blast job (= one `lastz target query` process of
// /root/reference/src/cactus/paf/local_alignment.py:65-73) on one MI355X.
//
//   index build  ->  per strand: seed search / sort / ungapped  ->  HSP filters (host, entropy in IEEE
//   double like lastz)  ->  anchors  ->  score-ordered gapped extension: speculative batches of Y-drop
//   DPs, long ones cut into concurrently evaluated pieces with verified hand-overs  ->  PAF text.
//
// The sequential rules of SURVEY.md A.10 that look order dependent are kept exact:
//   * diagonal suppression: hits are sorted by (diagonal, q) and each diagonal run is walked in order;
//   * "anchor covered by an earlier alignment": DPs are independent of each other, so they are run
//     speculatively in score-ranked batches and a host pass commits them strictly in anchor order;
//   * a one-sided DP is a chain of rows: it is cut at relays (fresh DPs started downstream) whose state
//     after a warm-up must equal the upstream state exactly, else the upstream piece is continued.
#include "mb_pipeline.h"
#include "mb_guard.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <future>
#include <memory>
#include <mutex>
#include <sched.h>
#include <thread>
#include <unordered_map>

namespace mb {

namespace {

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void release_idle_arenas();          // (the pool's free trace arenas of the current device back to the runtime: defined beside ArenaPool)

// A buffer that does not fit while the pool of trace arenas holds idle ones (up to a third of the device, plus what a call reserved): those go
// back to the runtime and the allocation is tried once more before the call gives up.
inline hipError_t device_malloc_or_trim(void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipErrorOutOfMemory) return e;
    (void)hipGetLastError();
    release_idle_arenas();
    return hipMalloc(p, bytes);
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        const double t0 = now_s();
        release();
        n = count;
        if (count) { count_device_alloc(); note_device_alloc(__PRETTY_FUNCTION__, count * sizeof(T)); }
        if (count && guard::on()) MB_HIP(guard::alloc((void **)&p, count * sizeof(T), __PRETTY_FUNCTION__));
        else if (count) MB_HIP(device_malloc_or_trim((void **)&p, count * sizeof(T)));
        if (getenv("MIBLAST_DEBUG_ALLOC") && now_s() - t0 > 0.02) fprintf(stderr, "[miblast] slow device allocation: %.1f MB in %.1f ms\n", count * sizeof(T) / 1e6, (now_s() - t0) * 1e3);
    }
    // hw: a high-water mark shared by the same buffer of every lane of a context (Workspace::gapped_hw).  Which lane meets the heaviest group of
    // pairs changes from call to call, and a lane that grew its tables in the middle of a later step stalled every lane (a device allocation beside
    // busy lanes: DESIGN.md section 6); with the mark a buffer that has to grow goes straight to the largest size any lane has needed, and
    // presize() -- at a lane's start, before it takes a pair -- brings it there without waiting for the need.
    std::atomic<size_t> *hw = nullptr;
    size_t marked(size_t count) {
        if (!hw) return count;
        size_t seen = hw->load(std::memory_order_relaxed);
        while (seen < count && !hw->compare_exchange_weak(seen, count, std::memory_order_relaxed)) {}
        return std::max(count, seen);
    }
    void ensure(size_t count) {
        if (count <= n) { (void)marked(count); return; }
        count = marked(count);
        alloc(guard::on() ? count : count + count / 4);
    }
    void presize() { if (hw && hw->load(std::memory_order_relaxed) > n) { const size_t c = hw->load(std::memory_order_relaxed); alloc(guard::on() ? c : c + c / 4); } }
    void ensure_keep(size_t count) {              // grow without losing the contents
        if (count <= n) { (void)marked(count); return; }
        count = marked(count);
        T *old = p; const size_t old_n = n;
        p = nullptr; n = guard::on() ? count : count + count / 2;
        count_device_alloc(); if (old) count_device_alloc();
        note_device_alloc(__PRETTY_FUNCTION__, n * sizeof(T));
        if (guard::on()) MB_HIP(guard::alloc((void **)&p, n * sizeof(T), __PRETTY_FUNCTION__));
        else MB_HIP(hipMalloc((void **)&p, n * sizeof(T)));
        if (old) { MB_HIP(hipMemcpy(p, old, old_n * sizeof(T), hipMemcpyDeviceToDevice)); if (guard::on()) guard::free(old, "DevBuf::ensure_keep"); else (void)hipFree(old); }
    }
    void release() { if (p) { count_device_alloc(); note_device_alloc("(release) DevBuf", n * sizeof(T)); } if (p) { if (guard::on()) guard::free(p, "DevBuf::release"); else (void)hipFree(p); } p = nullptr; n = 0; }
};

// pinned host staging buffer: one large device-to-host copy at link speed, no page faults
template <typename T>
struct PinBuf {
    T *p = nullptr;
    size_t n = 0;
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    ~PinBuf() { release(); }
    std::atomic<size_t> *hw = nullptr;           // (as DevBuf::hw: pinned allocations beside busy lanes stall as well)
    void ensure(size_t count) {
        if (hw) {
            size_t seen = hw->load(std::memory_order_relaxed);
            while (seen < count && !hw->compare_exchange_weak(seen, count, std::memory_order_relaxed)) {}
            if (count > n) count = std::max(count, seen);
        }
        if (count <= n) return;
        const double t0 = now_s();
        release();
        n = count + count / 4;
        MB_HIP(hipHostMalloc((void **)&p, n * sizeof(T), hipHostMallocDefault));
        if (getenv("MIBLAST_DEBUG_ALLOC") && now_s() - t0 > 0.02) fprintf(stderr, "[miblast] slow pinned allocation: %.1f MB in %.1f ms\n", n * sizeof(T) / 1e6, (now_s() - t0) * 1e3);
    }
    void presize() { if (hw && hw->load(std::memory_order_relaxed) > n) ensure(hw->load(std::memory_order_relaxed)); }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
};

// Pinned staging for the small host<->device copies of a round.  A hipMemcpyAsync from or to PAGEABLE memory blocks its caller
// until the copy is done and, measured on the MI355X box, now and then for 10-35 ms (the runtime pins or stages under a
// process-wide lock); copies from pinned memory are queued in microseconds.  h2d() copies the host data into the pinned area
// first; d2h() lands in the pinned area and done() -- called after the stream has been synchronised -- moves the bytes home.
struct Stager {
    std::vector<PinBuf<char> *> chunks;
    size_t used = 0;                              // of the last chunk
    struct Out { void *dst; const char *src; size_t n; };
    std::vector<Out> pending;
    Stager() = default;
    Stager(const Stager &) = delete;
    Stager &operator=(const Stager &) = delete;
    ~Stager() { for (PinBuf<char> *c : chunks) delete c; }
    char *take(size_t n) {
        n = (n + 255) & ~(size_t)255;
        if (chunks.empty() || used + n > chunks.back()->n) {                // (earlier chunks may still be in flight: they stay)
            size_t total = 0;
            for (PinBuf<char> *c : chunks) total += c->n;
            PinBuf<char> *c = new PinBuf<char>();
            c->ensure(std::max<size_t>(std::max<size_t>(n, total), 1u << 20));
            chunks.push_back(c); used = 0;
        }
        char *b = chunks.back()->p + used;
        used += n;
        return b;
    }
    void h2d(void *dev, const void *host, size_t n, hipStream_t s) {
        if (!n) return;
        char *b = take(n);
        memcpy(b, host, n);
        MB_HIP(hipMemcpyAsync(dev, b, n, hipMemcpyHostToDevice, s));
    }
    void d2h(void *host, const void *dev, size_t n, hipStream_t s) {
        if (!n) return;
        char *b = take(n);
        MB_HIP(hipMemcpyAsync(b, dev, n, hipMemcpyDeviceToHost, s));
        pending.push_back({host, b, n});
    }
    // two host arrays that lie one behind the other on the device (the second behind(na) bytes in): ONE copy each way
    static size_t behind(size_t n) { return (n + 255) & ~(size_t)255; }
    void h2d2(void *dev, const void *a, size_t na, const void *b, size_t nb, hipStream_t s) {
        const size_t off_b = behind(na), n = nb ? off_b + nb : na;
        if (!n) return;
        char *buf = take(n);
        if (na) memcpy(buf, a, na);
        if (nb) memcpy(buf + off_b, b, nb);
        MB_HIP(hipMemcpyAsync(dev, buf, n, hipMemcpyHostToDevice, s));
    }
    void d2h2(void *a, size_t na, void *b, size_t nb, const void *dev, hipStream_t s) {
        const size_t off_b = behind(na), n = nb ? off_b + nb : na;
        if (!n) return;
        char *buf = take(n);
        MB_HIP(hipMemcpyAsync(buf, dev, n, hipMemcpyDeviceToHost, s));
        if (na) pending.push_back({a, buf, na});
        if (nb) pending.push_back({b, buf + off_b, nb});
    }
    // A call that ended early (a HIP failure thrown between d2h() and done(), an early return) leaves `pending` entries whose
    // destinations were locals of frames that are gone: every entry point drops them before it queues anything.
    void abort() { pending.clear(); used = 0; }
    void done() {                                 // the stream is idle
        for (const Out &o : pending) memcpy(o.dst, o.src, o.n);
        pending.clear();
        while (chunks.size() > 1) { delete chunks.front(); chunks.erase(chunks.begin()); }      // the last one is as large as all before it together
        used = 0;
    }
};

// bit k set iff t[k] and q[k] are the same base of A, C, G, T (codes 0..3 in the low three bits; 4 = N, bit 3 = soft-masked)
inline unsigned match_bits8(const uint8_t *t, const uint8_t *q) {
    uint64_t a, b;
    memcpy(&a, t, 8); memcpy(&b, q, 8);
    a &= 0x0707070707070707ull; b &= 0x0707070707070707ull;
    const uint64_t x = a ^ b;
    const uint64_t eq = ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x | 0x7F7F7F7F7F7F7F7Full);      // 0x80 where the byte of x is zero
    const uint64_t acgt = (~a & 0x0404040404040404ull) << 5;                                                      // 0x80 where the code is below 4
    return (unsigned)((((eq & acgt) >> 7) * 0x0102040810204080ull) >> 56);                                         // byte k -> bit k
}

// host substitution score (HOXD70 + N = -100)
struct HostScoreTable {                     // 8 x 8 by the low three code bits: branch-free lookups in the anchor window scan
    int8_t s[64];
    HostScoreTable() {
        static const int M[4][4] = {{91, -114, -31, -123}, {-114, 100, -125, -31}, {-31, -125, 100, -114}, {-123, -31, -114, 91}};
        for (unsigned x = 0; x < 8; x++) for (unsigned y = 0; y < 8; y++) s[x * 8 + y] = (int8_t)((x > 3u || y > 3u) ? -100 : M[x][y]);
    }
};
inline int host_score(unsigned a, unsigned b) {
    static const HostScoreTable T;
    return T.s[(a & 7u) * 8u + (b & 7u)];
}

inline uint32_t host_word(const uint8_t *c, int64_t p) {
    static const int care[kSeedWeight] = {0, 1, 2, 4, 7, 8, 11, 13, 15, 16, 17, 18};
    uint32_t w = 0;
    for (int k = 0; k < kSeedWeight; k++) w = (w << 2) | (c[p + care[k]] & 3u);
    return w;
}

// rank of the word variant that produced a seed hit: 0 exact, 1+k transition at care position k.
// Needed only to order HSPs the way the sequential search finds them (--queryhspbest ties).
inline int variant_rank(const uint8_t *tc, const uint8_t *qc, int64_t t_end, int64_t q_end) {
    uint32_t dx = host_word(tc, t_end - kSeedSpan) ^ host_word(qc, q_end - kSeedSpan);
    if (!dx) return 0;
    for (int k = 0; k < kSeedWeight; k++) if (dx == (2u << (2 * (kSeedWeight - 1 - k)))) return 1 + k;
    return 99;
}

struct Anchor { int32_t t, q, score; };

struct Cached {                // result of one anchor's two one-sided DPs
    bool accepted = false;
    bool traced = false;       // ops / dmin / dmax are valid (an accepted result is traced unless an earlier one of its unit covers it)
    size_t epoch = 0;          // walls mode: alignments the unit had committed when the DPs ran (they were its walls)
    int32_t score = 0, t_lo = 0, t_hi = 0, q_lo = 0, q_hi = 0, dmin = 0, dmax = 0;
    int64_t cells = 0, rows = 0;
    std::vector<uint32_t> ops;           // merged run-length ops, forward order
};

template <typename It, typename Cmp> void parallel_sort(It first, It last, Cmp cmp);      // (defined with the worker pool below)

struct Unit {                  // one (pair, query contig, strand): anchors are committed strictly in order
    int pair = 0, strand = 0, q_contig = 0;
    std::vector<Anchor> anchors;
    size_t next = 0;           // first anchor not yet committed
    std::unordered_map<size_t, Cached> cache;
    std::vector<miblast_aln> kept;          // ops_off indexes unit_ops
    std::vector<int32_t> kept_anchor_score; // HSP score of the anchor of kept[k] (anchor order = (-score, t, q): the merge key of blocked runs)
    std::vector<uint32_t> unit_ops;
    // coverage bookkeeping: anchors indexed by q so that an alignment only visits the anchors inside its q range
    // (units of a 30 Mb chunk pair hold up to --queryhspbest=100000 anchors and thousands of alignments)
    std::vector<uint32_t> by_q;             // anchor indices sorted by q
    std::vector<uint8_t> cov;               // covered by a COMMITTED alignment (the rule of SURVEY A.10 GAPPED)
    std::vector<uint8_t> tent;              // would be covered by an accepted, not yet committed result (heuristic only)
    std::vector<uint32_t> comp;             // colinear group of the anchor (heuristic only): anchors one gapped alignment is likely to run through
    uint32_t n_comp = 0;

    // gap_q: how far apart in q two consecutive anchors of a group may lie; tol_d: how far apart their diagonals; a run of n_run
    // or more N between two anchors (in either sequence) ends the group as well: no extension survives it (N scores -100)
    // nt / nq: the runs of N bases of the target / of the searched strand of the query (n_runs_of), sorted; nullptr: none known
    void index_anchors(int32_t gap_q, int32_t tol_d, const std::vector<std::pair<int32_t, int32_t>> *nt, const std::vector<std::pair<int32_t, int32_t>> *nq,
                       int32_t n_run) {
        by_q.resize(anchors.size());
        for (size_t k = 0; k < by_q.size(); k++) by_q[k] = (uint32_t)k;
        parallel_sort(by_q.begin(), by_q.end(), [&](uint32_t x, uint32_t y) { return anchors[x].q != anchors[y].q ? anchors[x].q < anchors[y].q : x < y; });
        cov.assign(anchors.size(), 0);
        tent.assign(anchors.size(), 0);
        // Colinear groups: every anchor is linked to the next anchor in q order whose diagonal lies within tol_d (an indel larger
        // than that ends a Y-drop extension anyway) and whose q lies within gap_q.  One speculative head per group and round is
        // enough -- its relay chain covers the group in both directions; more heads on the same alignment only evaluate the
        // stretch between them twice.  Purely a scheduling hint: commits stay in anchor order.
        std::vector<uint32_t> parent(anchors.size());
        for (size_t k = 0; k < parent.size(); k++) parent[k] = (uint32_t)k;
        auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
        for (size_t p = 0; p < by_q.size(); p++) {
            const Anchor &a = anchors[by_q[p]];
            const int32_t da = a.t - a.q;
            for (size_t p2 = p + 1; p2 < by_q.size(); p2++) {
                const Anchor &b = anchors[by_q[p2]];
                if (b.q - a.q > gap_q) break;
                const int32_t dd = (b.t - b.q) - da;
                if (dd >= -tol_d && dd <= tol_d) {
                    // a run of >= n_run N inside [lo, hi): looked up in the sorted list of the sequence's N runs (scanning the bases
                    // between every pair of linked anchors cost 24 ms on a 30 Mb x 30 Mb pair -- the whole of both sequences, on one thread)
                    auto n_between = [&](const std::vector<std::pair<int32_t, int32_t>> *runs, int32_t lo, int32_t hi) -> bool {
                        if (!runs || hi - lo < n_run) return false;
                        auto it = std::lower_bound(runs->begin(), runs->end(), lo, [](const std::pair<int32_t, int32_t> &r, int32_t x) { return r.second <= x; });
                        for (; it != runs->end() && it->first < hi; ++it)
                            if (std::min(it->second, hi) - std::max(it->first, lo) >= n_run) return true;
                        return false;
                    };
                    if (!n_between(nq, a.q, b.q) && !n_between(nt, a.t, b.t)) parent[find(by_q[p2])] = find(by_q[p]);
                    break;
                }
            }
        }
        comp.assign(anchors.size(), 0);
        std::vector<uint32_t> id(anchors.size(), 0xFFFFFFFFu);
        n_comp = 0;
        for (size_t k = 0; k < anchors.size(); k++) {
            const uint32_t r = find((uint32_t)k);
            if (id[r] == 0xFFFFFFFFu) id[r] = n_comp++;
            comp[k] = id[r];
        }
    }
    // marks every anchor inside the bounding box and diagonal band of an alignment
    void mark(std::vector<uint8_t> &flags, int32_t t_lo, int32_t t_hi, int32_t q_lo, int32_t q_hi, int32_t dmin, int32_t dmax) {
        auto it = std::lower_bound(by_q.begin(), by_q.end(), q_lo, [&](uint32_t x, int32_t q) { return anchors[x].q < q; });
        for (; it != by_q.end() && anchors[*it].q < q_hi; ++it) {
            const Anchor &a = anchors[*it];
            const int32_t d = a.t - a.q;
            if (a.t >= t_lo && a.t < t_hi && d >= dmin && d <= dmax) flags[*it] = 1;
        }
    }
};

// independent work items on a few persistent host threads (merging traces, formatting PAF, anchors: all embarrassingly
// parallel).  The caller takes part; nested calls run inline.  Several parallel regions may be open at a time (the gapped stages of
// two groups of a call's pairs, calls on several contexts): the workers serve whichever has items left.
class Pool {
    struct Job {
        const std::function<void(size_t)> *fn;
        size_t n;
        std::atomic<size_t> next{0}, done{0};
        std::atomic<int> refs{0};                 // workers that hold a pointer to the job
    };
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv;
    std::vector<Job *> open_jobs;                // under m
    size_t sleepers = 0;                          // workers blocked on cv (under m)
    unsigned long long gen = 0;                   // bumped with every new job (under m)
    std::atomic<unsigned long long> gen_a{0};
    std::atomic<int> hot_a{0};                    // > 0 while an alignment call is in flight: idle workers spin a little before they sleep
    std::atomic<bool> stop_a{false};
    bool stop = false;
    std::atomic<int> regions{0};                  // parallel regions open right now
    void worker() {
        unsigned long long seen = 0;
        for (;;) {
            // While a call is in flight an idle worker spins for a few tens of microseconds (the parallel regions of a call follow
            // each other closely), then sleeps: a call is mostly GPU waits, and workers that spin through them burn the CPU quota
            // of the container (a 16-CPU cgroup throttled the process for 10 ms every few calls with 15 spinning workers).
            for (unsigned spins = 0;;) {
                if (gen_a.load(std::memory_order_acquire) != seen || stop_a.load(std::memory_order_relaxed)) break;
                if (hot_a.load(std::memory_order_relaxed) > 0 && spins < 1500) { __builtin_ia32_pause(); spins++; continue; }
                std::unique_lock<std::mutex> lk(m);
                sleepers++;
                cv.wait(lk, [&] { return stop || gen != seen; });
                sleepers--;
                spins = 0;
            }
            for (;;) {                                               // serve the open jobs until none has an item left
                Job *j = nullptr;
                {
                    std::lock_guard<std::mutex> lk(m);
                    if (stop) return;
                    seen = gen;
                    for (Job *c : open_jobs)
                        if (c->next.load(std::memory_order_relaxed) < c->n) { j = c; j->refs.fetch_add(1, std::memory_order_relaxed); break; }
                }
                if (!j) break;
                for (size_t i; (i = j->next.fetch_add(1, std::memory_order_relaxed)) < j->n;) { (*j->fn)(i); j->done.fetch_add(1, std::memory_order_release); }
                j->refs.fetch_sub(1, std::memory_order_release);
            }
        }
    }
    void join_all() {
        { std::lock_guard<std::mutex> lk(m); stop = true; stop_a = true; }
        cv.notify_all();
        for (std::thread &t : th) t.join();
        th.clear();
        { std::lock_guard<std::mutex> lk(m); stop = false; stop_a = false; }
    }
public:
    // host cores this process may use: the affinity mask, cut down to the cgroup CPU quota when there is one (a job of a
    // workflow engine or one rank of several on a node must not spin on cores it does not own)
    static unsigned cores_available() {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long quota = 0, period = 0;
            if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
                n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
            fclose(f);
        }
        return n;
    }
    static unsigned default_threads() {
        long v = 0;
        if (const char *e = getenv("MIBLAST_THREADS")) v = atol(e);
        if (v <= 0) v = std::min(16u, cores_available());
        return (unsigned)std::min(64l, std::max(1l, v));
    }
    Pool() { spawn(default_threads()); }
    ~Pool() { join_all(); }
    static Pool &get() { static Pool p; return p; }
    unsigned threads() const { return (unsigned)th.size() + 1; }
    void spawn(unsigned total) {                                   // `total` threads including the caller
        for (unsigned t = 1; t < total; t++) th.emplace_back([this] { worker(); });
    }
    // change the number of threads (caller included); refused while a call keeps the workers awake or a region is open
    bool resize(unsigned total) {
        if (hot_a.load() > 0 || regions.load() > 0) return false;
        if (total == 0) total = default_threads();
        total = std::min(64u, std::max(1u, total));
        if (total == threads()) return true;
        join_all();
        spawn(total);
        return true;
    }
    struct Hot {                              // keeps the workers awake for the duration of a call
        Hot() { Pool &p = get(); p.hot_a++; }
        ~Hot() { get().hot_a--; }
    };
    void run(size_t count, const std::function<void(size_t)> &f) {
        static thread_local bool inside = false;
        if (count <= 1 || th.empty() || inside) { for (size_t i = 0; i < count; i++) f(i); return; }
        inside = true;
        regions++;
        Job job;
        job.fn = &f; job.n = count;
        bool wake;
        {
            std::lock_guard<std::mutex> lk(m);
            open_jobs.push_back(&job);
            gen++;
            gen_a.store(gen, std::memory_order_release);
            wake = sleepers > 0;
        }
        if (wake) cv.notify_all();
        try {
            for (size_t i; (i = job.next.fetch_add(1, std::memory_order_relaxed)) < count;) { f(i); job.done.fetch_add(1, std::memory_order_release); }
        } catch (...) {
            // (the items are not supposed to throw; if one does, the job still has to leave the list before the stack unwinds)
            job.next.store(count, std::memory_order_relaxed);
            { std::lock_guard<std::mutex> lk(m); open_jobs.erase(std::find(open_jobs.begin(), open_jobs.end(), &job)); }
            while (job.refs.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
            regions--; inside = false;
            throw;
        }
        { std::lock_guard<std::mutex> lk(m); open_jobs.erase(std::find(open_jobs.begin(), open_jobs.end(), &job)); }
        // the stragglers finish within microseconds: spin
        while (job.done.load(std::memory_order_acquire) < count || job.refs.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
        regions--;
        inside = false;
    }
};

template <typename F>
void parallel_for(size_t n, F &&f) {
    const std::function<void(size_t)> fn = std::ref(f);
    Pool::get().run(n, fn);
}

// sort on the worker threads: chunks sorted independently, then merged pairwise (the comparators used with it are total
// orders, so the result does not depend on the split)
template <typename It, typename Cmp>
void parallel_sort(It first, It last, Cmp cmp) {
    const size_t n = (size_t)(last - first);
    const size_t kMin = 1024;
    if (n < 2 * kMin) { std::sort(first, last, cmp); return; }
    size_t parts = 1;
    while (parts < 16 && n / (parts * 2) >= kMin) parts *= 2;
    std::vector<size_t> cut(parts + 1);
    for (size_t p = 0; p <= parts; p++) cut[p] = n * p / parts;
    parallel_for(parts, [&](size_t p) { std::sort(first + (long)cut[p], first + (long)cut[p + 1], cmp); });
    for (size_t width = 1; width < parts; width *= 2)
        parallel_for(parts / (2 * width), [&](size_t m) {
            const size_t a = 2 * width * m;
            std::inplace_merge(first + (long)cut[a], first + (long)cut[a + width], first + (long)cut[a + 2 * width], cmp);
        });
}

}  // namespace

// the worker pool for the other translation units of the library (mp_chain.cpp)
void host_parallel_for(size_t n, const std::function<void(size_t)> &f) { Pool::get().run(n, f); }
HostHot::HostHot() { Pool::get(); new (&impl) Pool::Hot(); }
HostHot::~HostHot() { reinterpret_cast<Pool::Hot *>(&impl)->~Hot(); }

int set_host_threads(int n) { return Pool::get().resize(n < 0 ? 0u : (unsigned)n) ? (int)Pool::get().threads() : -1; }
int host_threads() { return (int)Pool::get().threads(); }

namespace {

long env_long(const char *name, long dflt) {
    const char *v = getenv(name);
    return v && *v ? atol(v) : dflt;
}

}  // namespace

// --------------------------------------------------------------------------------------------------
// Sequence sets come and go with every call of a phase (trimmed ingroups, chunk files): their device blocks are recycled through a
// small per-process cache, because hipMalloc / hipFree cost as much as the upload itself and hipFree synchronises the device; and
// the image of a set (separator padding, codes, contig table) is put together in pinned memory and travels in ONE copy.
namespace {
struct DeviceBlocks {
    struct Block { int device; void *p; size_t cap; };
    std::mutex mu;
    std::vector<Block> free_list;
    size_t cached = 0;
    static constexpr size_t kKeep = (size_t)8 << 30;        // (of 288 GB: the derived data of a genome pair's chunks comes and goes with every step of a bench)
    void *take(int device, size_t bytes, size_t &cap) {
        if (guard::on()) {                                   // exact size, canary behind it, no recycling
            void *p = nullptr;
            cap = bytes;
            MB_HIP(guard::alloc(&p, bytes, "DeviceBlocks::take"));
            return p;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t best = free_list.size();
            for (size_t i = 0; i < free_list.size(); i++)
                if (free_list[i].device == device && free_list[i].cap >= bytes && free_list[i].cap <= 2 * bytes + 4096 && (best == free_list.size() || free_list[i].cap < free_list[best].cap)) best = i;
            if (best < free_list.size()) {
                Block b = free_list[best];
                free_list.erase(free_list.begin() + (long)best);
                cached -= b.cap; cap = b.cap;
                return b.p;
            }
        }
        void *p = nullptr;
        cap = (bytes + 4095) & ~(size_t)4095;
        count_device_alloc(); note_device_alloc("block cache (sequence sets, seed tables, strands)", cap);
        MB_HIP(hipMalloc(&p, cap));
        return p;
    }
    void give(int device, void *p, size_t cap) {
        if (guard::on()) { guard::free(p, "DeviceBlocks::give"); return; }
        {
            std::lock_guard<std::mutex> lk(mu);
            if (cached + cap <= kKeep) { free_list.push_back({device, p, cap}); cached += cap; return; }
        }
        count_device_alloc(); note_device_alloc("(release) block cache over its budget", cap);
        (void)hipFree(p);
    }
    ~DeviceBlocks() { for (Block &b : free_list) (void)hipFree(b.p); }
};
DeviceBlocks &device_blocks() { static DeviceBlocks *d = new DeviceBlocks(); return *d; }       // (never destroyed: the HIP runtime may be gone first)

struct UploadStage {                               // pinned staging of set images up to kMax bytes; larger sets are copied from where they lie
    std::mutex mu;
    char *p = nullptr;
    size_t n = 0;
    static constexpr size_t kMax = (size_t)64 << 20;
    char *ensure(size_t bytes) {
        if (bytes > n) {
            if (p) (void)hipHostFree(p);
            p = nullptr; n = bytes + bytes / 4;
            MB_HIP(hipHostMalloc((void **)&p, n, hipHostMallocDefault));
        }
        return p;
    }
};
UploadStage &upload_stage() { static UploadStage *u = new UploadStage(); return *u; }
}  // namespace

// device memory out of the block cache (no hipMalloc / hipFree -- the latter waits for the device -- when a size comes round again)
template <typename T>
struct PoolBuf {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    int device = 0;
    PoolBuf() = default;
    PoolBuf(const PoolBuf &) = delete;
    PoolBuf &operator=(const PoolBuf &) = delete;
    ~PoolBuf() { release(); }
    void ensure(size_t count) {
        if (count <= n) return;
        release();
        MB_HIP(hipGetDevice(&device));
        p = (T *)device_blocks().take(device, count * sizeof(T), cap);
        n = cap / sizeof(T);
    }
    void release() { if (p) device_blocks().give(device, p, cap); p = nullptr; n = 0; cap = 0; }
};
namespace {
struct PinnedBlocks {                              // the same for pinned host memory (hipHostMalloc / hipHostFree of 30 MB cost milliseconds)
    struct Block { void *p; size_t cap; };
    std::mutex mu;
    std::vector<Block> free_list;
    void *take(size_t bytes, size_t &cap) {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < free_list.size(); i++)
                if (free_list[i].cap >= bytes && free_list[i].cap <= 2 * bytes + 4096) {
                    Block b = free_list[i];
                    free_list.erase(free_list.begin() + (long)i);
                    cap = b.cap;
                    return b.p;
                }
        }
        void *p = nullptr;
        cap = (bytes + 4095) & ~(size_t)4095;
        MB_HIP(hipHostMalloc(&p, cap, hipHostMallocDefault));
        return p;
    }
    void give(void *p, size_t cap) {
        std::lock_guard<std::mutex> lk(mu);
        if (free_list.size() < 64) { free_list.push_back({p, cap}); return; }
        (void)hipHostFree(p);
    }
};
PinnedBlocks &pinned_blocks() { static PinnedBlocks *b = new PinnedBlocks(); return *b; }
}  // namespace
template <typename T>
struct PoolPin {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    PoolPin() = default;
    PoolPin(const PoolPin &) = delete;
    PoolPin &operator=(const PoolPin &) = delete;
    ~PoolPin() { if (p) pinned_blocks().give(p, cap); }
    void ensure(size_t count) {
        if (count <= n) return;
        if (p) pinned_blocks().give(p, cap);
        p = (T *)pinned_blocks().take(count * sizeof(T), cap);
        n = cap / sizeof(T);
    }
};

void note_n_runs(const SeqSet &S);

void upload_seqset(SeqSet &s, int device) {
    MB_HIP(hipSetDevice(device));
    s.device = device;
    note_n_runs(s);                                                       // (part of making the set resident: see n_runs_cached)
    const size_t nc = std::max<size_t>(1, s.starts.size());
    const size_t seq_bytes = ((size_t)s.total + 2 * kDevPad + 255) & ~(size_t)255, image = seq_bytes + 2 * nc * sizeof(int64_t);
    s.d_buf = (uint8_t *)device_blocks().take(device, image, s.d_cap);
    s.d_starts = (int64_t *)(s.d_buf + seq_bytes);
    s.d_lens = s.d_starts + nc;
    if (image <= UploadStage::kMax) {
        UploadStage &u = upload_stage();
        std::lock_guard<std::mutex> lk(u.mu);
        char *h = u.ensure(image);
        memset(h, 0xFF, kDevPad);
        if (s.total) memcpy(h + kDevPad, s.host(), (size_t)s.total);
        memset(h + kDevPad + (size_t)s.total, 0xFF, seq_bytes - kDevPad - (size_t)s.total);
        memset(h + seq_bytes, 0, 2 * nc * sizeof(int64_t));
        if (!s.starts.empty()) {
            memcpy(h + seq_bytes, s.starts.data(), s.starts.size() * sizeof(int64_t));
            memcpy(h + seq_bytes + nc * sizeof(int64_t), s.lens.data(), s.lens.size() * sizeof(int64_t));
        }
        MB_HIP(hipMemcpy(s.d_buf, h, image, hipMemcpyHostToDevice));
        return;
    }
    MB_HIP(hipMemset(s.d_buf, 0xFF, seq_bytes));
    if (s.total) MB_HIP(hipMemcpy(s.d_buf + kDevPad, s.host(), (size_t)s.total, hipMemcpyHostToDevice));
    if (!s.starts.empty()) {
        MB_HIP(hipMemcpy(s.d_starts, s.starts.data(), s.starts.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        MB_HIP(hipMemcpy(s.d_lens, s.lens.data(), s.lens.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    }
}

void forget_n_runs(const SeqSet &S);
static void forget_derived(const SeqSet &S);

void release_seqset(SeqSet &s) {
    forget_n_runs(s);
    if (s.d_buf) forget_derived(s);
    if (s.d_buf) device_blocks().give(s.device, s.d_buf, s.d_cap);
    s.d_buf = nullptr; s.d_starts = nullptr; s.d_lens = nullptr; s.d_cap = 0;
}


// --------------------------------------------------------------------------------------------------
// Trace arenas are borrowed for the length of a gapped stage: a process-wide pool per device hands out the smallest free one that
// holds the stage's estimate (else the largest free one, else a new one) and takes it back when the stage is over.  A chunk pair with
// homology needs 5-15 GiB, one without a few MB; which lane meets which pair changes from call to call (align_pairs deals large pairs
// from a queue), and an arena owned by the lane made every lane grow its own in turn -- 0.7 s per 16 GiB hipMalloc and a repeated round,
// spread over the first steps of a job.  With the pool the large arenas exist once per concurrently running heavy pair.
namespace {
struct ArenaPool {
    struct A { int device; uint8_t *p; size_t n; };
    std::mutex mu;
    std::vector<A> free_list;
    bool take(int device, size_t want, uint8_t *&p, size_t &n) {        // false: none free (the caller allocates)
        std::lock_guard<std::mutex> lk(mu);
        size_t best = free_list.size();
        for (size_t i = 0; i < free_list.size(); i++) {
            if (free_list[i].device != device) continue;
            if (best == free_list.size()) { best = i; continue; }
            const size_t a = free_list[i].n, b = free_list[best].n;
            if ((a >= want && (b < want || a < b)) || (a < want && b < want && a > b)) best = i;
        }
        if (best == free_list.size()) return false;
        p = free_list[best].p; n = free_list[best].n;
        free_list.erase(free_list.begin() + (long)best);
        return true;
    }
    // What lies free in the pool is bounded (MIBLAST_ARENA_POOL_MB, default a third of the device's memory: 96 GB on an MI355X): a stage that outgrew its arena four times
    // over leaves arenas of 2, 8, 32, 128 GiB behind, and a job of many different pairs one per size class it ever met.  Beyond the bound
    // the smallest go first (the large ones are the expensive ones to make again).
    void give(int device, uint8_t *p, size_t n) {
        if (!p) return;
        std::vector<A> drop;
        {
            std::lock_guard<std::mutex> lk(mu);
            free_list.push_back({device, p, n});
            static const size_t cap = [] {                                  // (a third of the device's memory unless the switch says otherwise)
                if (getenv("MIBLAST_ARENA_POOL_MB")) return (size_t)env_long("MIBLAST_ARENA_POOL_MB", 64l << 10) << 20;
                size_t free_b = 0, total_b = 0;
                return hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b ? total_b / 3 : (size_t)64 << 30;
            }();
            size_t held = 0;
            for (const A &a : free_list) if (a.device == device) held += a.n;
            while (held > cap) {
                size_t smallest = free_list.size();
                for (size_t i = 0; i < free_list.size(); i++)
                    if (free_list[i].device == device && (smallest == free_list.size() || free_list[i].n < free_list[smallest].n)) smallest = i;
                if (smallest == free_list.size()) break;
                held -= free_list[smallest].n;
                drop.push_back(free_list[smallest]);
                free_list.erase(free_list.begin() + (long)smallest);
            }
        }
        for (const A &a : drop) { count_device_alloc(); note_device_alloc("(release) trace arena", a.n); (void)hipFree(a.p); }
    }
    size_t count_free(int device, size_t at_least) {
        std::lock_guard<std::mutex> lk(mu);
        size_t n = 0;
        for (const A &a : free_list) n += a.device == device && a.n >= at_least;
        return n;
    }
    // everything that lies free on the device back to the runtime (a stage that needs most of the device's memory for its arena)
    void trim(int device) {
        std::vector<A> drop;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < free_list.size();)
                if (free_list[i].device == device) { drop.push_back(free_list[i]); free_list.erase(free_list.begin() + (long)i); } else i++;
        }
        for (const A &a : drop) { count_device_alloc(); note_device_alloc("(release) trace arena", a.n); (void)hipFree(a.p); }
    }
};
ArenaPool &arena_pool() { static ArenaPool *a = new ArenaPool(); return *a; }
void release_idle_arenas() {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) arena_pool().trim(dev);
}
// estimate x this / 256 (grows when a stage had to grow its arena).  Per CONTEXT since round 6 (Workspace::arena_scale of the context that owns the
// call, its lanes point there): the factor was one per process, and in a process that runs several kinds of jobs -- the bench line: the phase's
// small pairs, then chr20, then 42 human-mouse chunk pairs, each leg in a context of its own -- what an outlier stage of one kind had taught (16 x)
// made every stage of the next kind ask for arenas of 16 GiB where 1 GiB holds its trace: new ones were allocated in the middle of timed steps
// (0.7 s each, a step of 620 ms among steps of 146).  The process-wide one stays as the fallback of a context without an owner.
std::atomic<unsigned> &arena_scale_q8() { static std::atomic<unsigned> s{256}; return s; }
std::atomic<unsigned> &arena_scale_of(Ctx &ctx) { return ctx.arena_scale ? *ctx.arena_scale : arena_scale_q8(); }
}  // namespace

// seed position table of a target (CSR over the 2^24 seed words + occupancy bitmap of the buckets) and the packed form of a strand
// (mb_seed_dense.h); both live with the set they are derived from (SetDerived below) or, as scratch, in a context's workspace
struct SeedTable {
    int step = 0; int64_t first = 0, n_slots = 0; uint32_t n_positions = 0;
    PoolBuf<uint32_t> offsets, occ, positions;
};
struct PackedStrand { PoolBuf<unsigned long long> p2, pm; PoolBuf<uint32_t> px; bool ready = false; };      // px: the ungapped extension's records (mb_ungapped_ux.h)

struct Workspace {                      // device buffers that persist across miblast_align() calls of one context
    // seed position table
    DevBuf<uint32_t> words, counts;
    bool counts_zero = false;                 // `counts` is all zero (build_index leaves it so)
    PackedStrand pack_t;                      // the packed target of the build in progress (scratch)
    std::shared_ptr<SeedTable> own_table;     // MIBLAST_RESIDENT_TABLES=0: the table of the call in progress
    DevBuf<unsigned long long> ord_state;     // q-ordered seed search: totals, the tiles' stretches of the scratch, their counts and the scan of those, both strands
    DevBuf<uint32_t> bin_state, bin_matrix;   // grouping by diagonal without the sort (mb_seed_bin.h): plan, sizes and places of the bins; keys per (chunk, bin) -- both strands
    DevBuf<unsigned long long> bsum;
    // seed search / ungapped
    DevBuf<uint8_t> rc;
    PinBuf<uint8_t> h_rc;
    DevBuf<RcItem> rc_items;
    DevBuf<int32_t> extent;
    DevBuf<uint32_t> qcnt, hit_off;
    DevBuf<unsigned long long> qbsum, scan_scratch, keys_a, keys_b;
    DevBuf<char> sort_temp;
    DevBuf<DevHsp> hsps;
    DevBuf<UngappedCounters> ctr;
    DevBuf<unsigned> heads, n_heads;
    DevBuf<UxEntry> ux_entries;               // level-synchronous ungapped extension (mb_ungapped_ux.h): unfinished hits,
    DevBuf<unsigned> ux_cnt;                  // their counter,
    DevBuf<uint32_t> ux_bits;                 // two bit planes, one bit per diagonal each (runs of k_ungapped_long; runs that need the sequential rule)
    // both strands of a pair in one go (seed_phase, fused path): events per strand, pinned read-back areas
    hipEvent_t ev_base = nullptr;             // start of the current call (DpSpans::base)
    hipEvent_t sev[2][6] = {{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}};
    unsigned long long last_strand_hits = 0;  // hits of the larger strand of the last pair seeded with this workspace
    std::atomic<unsigned long long> hits_hint{0};   // (of a context's own workspace) the largest strand its lanes have met: Ctx::hits_hint of the lanes points here
    PinBuf<unsigned long long> pin_u64;
    PinBuf<UngappedCounters> pin_ctr;
    PinBuf<DevHsp> pin_hsps;
    Stager stage;
    // batched seed stage (seed_phase_batched): sparse tables of the call's distinct targets, unit tables, per-unit counters
    DevBuf<unsigned long long> bx_bits, bx_scan;
    DevBuf<uint32_t> bx_dir, bx_bsum, bx_words, bx_starts, bx_positions;
    DevBuf<unsigned long long> h16_ka, h16_kb;               // diag_hash16: sort keys and values of the resolve pass
    DevBuf<uint32_t> h16_va, h16_vb;
    DevBuf<BatchTarget> bx_targets;
    DevBuf<SeedUnit> bx_units;
    PinBuf<unsigned long long> pin_scan;
    // outgroup trimming on the device (seqset_unaligned)
    DevBuf<uint32_t> cov_diff, cov_depth;
    DevBuf<long long> cov_spans, cov_first, cov_iv;
    // gapped
    DevBuf<DpProb> probs;
    DevBuf<int> dp_order;                     // a crowded DP launch: piece of block b (longest first)
    DevBuf<DpOut> outs;
    DevBuf<int32_t> grows;
    struct Arena { uint8_t *p = nullptr; size_t n = 0; } arena;      // borrowed from arena_pool() for the length of a gapped stage
    DevBuf<unsigned long long> arena_next;
    DevBuf<unsigned long long> rowdir;
    DevBuf<uint32_t> ops, ops_packed;
    DevBuf<unsigned long long> coff;
    PinBuf<uint32_t> hops;
    DevBuf<char> tb_blk;                      // traceback tables (walks, sides, segments) one behind the other
    DevBuf<uint32_t> recs;
    DevBuf<uint8_t> snaps;
    DevBuf<char> dp_up, dp_down;             // a DP launch's pieces + hand-over checks / their results: one copy each way
    DevBuf<char> round_tab;                   // the round's piece table (DpProb of every piece queued so far), the current launch's hand-over checks behind it
    DevBuf<PairPtrs> pair_ptrs;
    DevBuf<int32_t> wall_segs;                // walls mode: WallSeg runs (3 x int32), run ranges per alignment and alignment ranges per piece (int2 each)
    DevBuf<int32_t> wall_alns, wall_ref;
    DevBuf<uint8_t> wall_flags;               // ... and, for DPs on the HBM ring, one flag byte per ring column
    // batched calls: extra lanes (own stream, events and seed-stage buffers) so that the seed stages of several pairs are on
    // the device at the same time
    std::vector<Ctx *> lanes;
    // (of a context's own workspace) high-water marks of the gapped stage's tables over all lanes: DevBuf::hw of the lanes' buffers point here
    std::atomic<unsigned> arena_scale{256};          // (of a context's own workspace) see arena_scale_q8
    std::atomic<size_t> arena_class{0};              // ... the largest trace arena a stage of the context has asked for (reserve_arenas)
    std::atomic<size_t> gapped_hw[16] = {};
    void share_marks(Workspace &owner) {
        std::atomic<size_t> *m = owner.gapped_hw;
        probs.hw = m + 0; outs.hw = m + 1; dp_order.hw = m + 2; rowdir.hw = m + 3; ops.hw = m + 4; ops_packed.hw = m + 5; coff.hw = m + 6; tb_blk.hw = m + 7;
        recs.hw = m + 8; snaps.hw = m + 9; dp_up.hw = m + 10; dp_down.hw = m + 11; round_tab.hw = m + 12; hops.hw = m + 13; grows.hw = m + 14;
        // (the seed stage's buffers follow Ctx::hits_hint, presize_lane; giving them marks of this kind as well -- tried at the end of round 6 --
        //  left as many allocations in the timed steps and cost 3 - 5 % of the chunk legs' step: every lane then holds every buffer at its largest)
    }
    void presize_gapped() {
        probs.presize(); outs.presize(); dp_order.presize(); rowdir.presize(); ops.presize(); ops_packed.presize(); coff.presize(); tb_blk.presize();
        recs.presize(); snaps.presize(); dp_up.presize(); dp_down.presize(); round_tab.presize(); hops.presize(); grows.presize();
    }
};

Workspace *workspace_create() { return new Workspace(); }
static std::atomic<size_t> &arena_class_of(Ctx &ctx) {
    static std::atomic<size_t> process_wide{0};
    return ctx.arena_class ? *ctx.arena_class : process_wide;
}

static Ctx *lane_create(int device, int priority) {
    Ctx *c = new Ctx();
    c->device = device;
    c->priority = priority;
    c->ws = workspace_create();
    if (priority == 0) MB_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));      // (the default: as every other stream of the library)
    else MB_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, priority));
    MB_HIP(hipEventCreate(&c->ev0)); MB_HIP(hipEventCreate(&c->ev1)); MB_HIP(hipEventCreate(&c->ev2));
    MB_HIP(hipEventCreate(&c->ev3)); MB_HIP(hipEventCreate(&c->ev4));
    return c;
}

static void lanes_destroy(Workspace *w) {
    for (Ctx *c : w->lanes) {
        for (hipEvent_t e : {c->ev0, c->ev1, c->ev2, c->ev3, c->ev4}) if (e) (void)hipEventDestroy(e);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        workspace_destroy(c->ws);
        delete c;
    }
    w->lanes.clear();
}

void workspace_destroy(Workspace *w) {
    if (!w) return;
    for (auto &row : w->sev) for (hipEvent_t e : row) if (e) (void)hipEventDestroy(e);
    if (w->ev_base) (void)hipEventDestroy(w->ev_base);
    lanes_destroy(w);
    delete w;
}

// The runtime deals a process's streams to a few hardware queues in the order of their first use, and two streams on one queue run
// their kernels one after the other.  The stream of a context and that of the lane its batched calls run their second group of pairs
// on must overlap (16 x 1 Mb pairs in one call: 31 ms side by side, 34 ms one after the other), so they are made and used for
// the first time together, one right after the other: neighbours in that order never share a queue.
void ctx_pair_streams(Ctx &ctx) {
    if (env_long("MIBLAST_PAIR_STREAMS", 1) == 0) return;
    MB_HIP(hipSetDevice(ctx.device));
    Workspace &w = *ctx.ws;
    const size_t gapped_lanes = (size_t)std::min<long>(8, std::max(1l, env_long("MIBLAST_GAPPED_LANES", 2)));
    while (w.lanes.size() + 1 < gapped_lanes) { w.lanes.push_back(lane_create(ctx.device, ctx.priority)); w.lanes.back()->ws->share_marks(w); w.lanes.back()->arena_scale = ctx.arena_scale; w.lanes.back()->arena_class = ctx.arena_class; }
    MB_HIP(hipEventRecord(ctx.ev0, ctx.stream));
    for (Ctx *l : w.lanes) MB_HIP(hipEventRecord(l->ev0, l->stream));
    MB_HIP(hipStreamSynchronize(ctx.stream));
    for (Ctx *l : w.lanes) MB_HIP(hipStreamSynchronize(l->stream));
}

// The context's launches yield to (level < 0) or go before (level > 0) those of other contexts on the device: its stream, and the
// streams of the lanes its batched calls open, are made anew at the lowest / highest priority the device offers (0: the default).
int ctx_set_priority(Ctx &ctx, int level) {
    MB_HIP(hipSetDevice(ctx.device));
    int least = 0, greatest = 0;
    MB_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int priority = level < 0 ? least : level > 0 ? greatest : 0;
    MB_HIP(hipStreamSynchronize(ctx.stream));
    lanes_destroy(ctx.ws);
    hipStream_t fresh = nullptr;
    MB_HIP(hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, priority));
    (void)hipStreamDestroy(ctx.stream);
    ctx.stream = fresh;
    ctx.priority = priority;
    ctx_pair_streams(ctx);
    return MIBLAST_OK;
}

// ---- outgroup trimming between two blast calls, on the device --------------------------------------------------
// `paffy to_bed --excludeAligned --minSize N` + `faffy extract --flank F` (/root/reference/src/cactus/paf/local_alignment.py:460-499)
// without the round trip PAF text -> BED -> FASTA text -> parse -> upload: the per-base coverage of the resident query set by the
// alignments' query intervals is taken on the device (difference array, prefix sum), the maximal uncovered stretches come back as
// a few numbers, the rule of the reference's two tools is applied to them (at least min_size long; widened by flank; widened
// stretches that touch are one), and the kept stretches are gathered on the device into a new resident set whose records are named
// NAME|SEQLEN|START like faffy's.  Same set -- names, lengths, codes -- as mipaf_unaligned_fasta + miblast_seqset_from_fasta_mem.
// The chains of a dependency level are trimmed in ONE call: every item's kernels are queued before the first of the two
// synchronisations (edges back; gathers done).
int seqset_unaligned(Ctx &ctx, size_t n, const SeqSet *const *Qs, const char *const *pafs, const size_t *paf_lens, int64_t min_size, int64_t flank,
                     SeqSet *const *outs, bool *nothing_left) {
    MB_HIP(hipSetDevice(ctx.device));
    Workspace &w = *ctx.ws;
    hipStream_t s = ctx.stream;
    w.stage.abort();
    struct Item {
        std::vector<long long> spans, first, last, iv;
        unsigned cap = 0, n_edges[2] = {0, 0};
        size_t off_depth = 0, off_spans = 0, off_edges = 0, off_iv = 0;      // the item's share of the workspace arrays
        int rc = MIBLAST_OK;
        std::string err;
    };
    std::vector<Item> items(n);
    // ---- query intervals of the alignments, in each set's concatenated coordinates (text work: on the worker threads)
    parallel_for(n, [&](size_t k) {
        Item &it = items[k];
        const SeqSet &Q = *Qs[k];
        nothing_left[k] = true;
        if (Q.device != ctx.device) { it.rc = MIBLAST_EINVAL; it.err = "sequence set lives on another device"; return; }
        std::unordered_map<std::string, size_t> by_name;
        for (size_t c = 0; c < Q.names.size(); c++)
            if (!by_name.emplace(Q.names[c], c).second) { it.rc = MIBLAST_EINVAL; it.err = "to_bed: sequence name " + Q.names[c] + " occurs twice in the query set"; return; }
        const char *paf = pafs[k];
        const size_t paf_len = paf_lens[k];
        size_t line_no = 0;
        for (size_t pos = 0; pos < paf_len;) {
            const char *nl = (const char *)memchr(paf + pos, '\n', paf_len - pos);
            const size_t end = nl ? (size_t)(nl - paf) : paf_len;
            line_no++;
            const size_t p0 = pos;
            pos = end + 1;
            bool blank = true;
            for (size_t x = p0; x < end && blank; x++) blank = paf[x] == ' ' || paf[x] == '\t' || paf[x] == '\r';
            if (blank) continue;
            const char *t[4];
            bool ok = true;
            size_t c = p0;
            for (int f = 0; f < 4 && ok; f++) {
                t[f] = (const char *)memchr(paf + c, '\t', end - c);
                if (!t[f]) ok = false; else c = (size_t)(t[f] - paf) + 1;
            }
            if (!ok) { it.rc = MIBLAST_EINVAL; it.err = "to_bed: PAF line " + std::to_string(line_no) + " has fewer than 5 columns"; return; }
            const std::string name(paf + p0, (size_t)(t[0] - (paf + p0)));
            const auto f = by_name.find(name);
            if (f == by_name.end()) { it.rc = MIBLAST_EINVAL; it.err = "to_bed: PAF line " + std::to_string(line_no) + ": query " + name + " is not in the query set"; return; }
            const int64_t len = Q.lens[f->second];
            const int64_t s0 = std::min(len, std::max<int64_t>(0, strtoll(t[1] + 1, nullptr, 10))), e0 = std::min(len, std::max<int64_t>(0, strtoll(t[2] + 1, nullptr, 10)));
            if (e0 > s0) { it.spans.push_back(Q.starts[f->second] + s0); it.spans.push_back(Q.starts[f->second] + e0); }
        }
        it.cap = (unsigned)std::min<size_t>(it.spans.size() / 2 + Q.names.size() + 2, 0x7fffffffu);
    });
    for (const Item &it : items) if (it.rc != MIBLAST_OK) { set_error(it.err); return it.rc; }
    // ---- coverage on the device, edges of the uncovered stretches back: one difference array, one scan, one edge search for all
    //      items (an item's difference array sums to zero, so the running depth is zero where the next item's share begins)
    static_assert(sizeof(CovItem) == 5 * sizeof(long long) && sizeof(GatherItem) == 9 * sizeof(long long), "tables travel as long long");
    size_t n_depth = 0, n_sp = 0, n_ed = 0;
    for (size_t k = 0; k < n; k++) {
        Item &it = items[k];
        it.off_depth = n_depth; it.off_spans = n_sp; it.off_edges = n_ed;
        n_depth += (((size_t)std::max<int64_t>(0, Qs[k]->total) + 2) + 255) & ~(size_t)255;       // (shares of whole blocks: k_cov_edges)
        n_sp += it.spans.size(); n_ed += it.cap;
    }
    std::vector<long long> up(5 * n + n_sp);                                   // CovItem table, then every item's spans at its share of the array
    for (size_t k = 0; k < n; k++) {
        Item &it = items[k];
        CovItem ci{Qs[k]->d_buf ? Qs[k]->dev() : nullptr, std::max<int64_t>(0, Qs[k]->total), (long long)it.off_depth, (long long)it.off_edges, it.cap, 0};
        memcpy(&up[5 * k], &ci, sizeof ci);
        for (size_t x = 0; x < it.spans.size(); x++) up[5 * n + it.off_spans + x] = it.spans[x] + (long long)it.off_depth;
    }
    std::vector<long long> edges(2 * n_ed + 1);
    std::vector<unsigned> counts(2 * n + 2);
    w.cov_diff.ensure(n_depth + 2 * n + 8); w.cov_depth.ensure(n_depth + 8);
    w.cov_spans.ensure(up.size() + 2); w.cov_first.ensure(2 * n_ed + 1);
    w.bx_scan.ensure((n_depth + 2047) / 2048 + 2);
    unsigned *d_counts = w.cov_diff.p + n_depth;                                  // (zeroed with the difference array)
    MB_HIP(hipMemsetAsync(w.cov_diff.p, 0, up16((n_depth + 2 * n + 4) * 4), s));
    w.stage.h2d(w.cov_spans.p, up.data(), up.size() * sizeof(long long), s);
    launch_cov_mark(w.cov_spans.p + 5 * n, (int)(n_sp / 2), w.cov_diff.p, s);
    launch_scan_u32(w.cov_diff.p, w.cov_depth.p, (int64_t)n_depth, w.bx_scan.p, s);
    launch_cov_edges(w.cov_depth.p, (const CovItem *)w.cov_spans.p, (int)n, (int64_t)n_depth, d_counts, w.cov_first.p, w.cov_first.p + n_ed, s);
    w.stage.d2h(counts.data(), d_counts, 2 * n * sizeof(unsigned), s);
    if (n_ed) w.stage.d2h(edges.data(), w.cov_first.p, 2 * n_ed * sizeof(long long), s);
    MB_HIP(hipStreamSynchronize(s));                                           // (1) the edges of every item
    w.stage.done();
    for (size_t k = 0; k < n; k++) {
        Item &it = items[k];
        it.n_edges[0] = counts[2 * k]; it.n_edges[1] = counts[2 * k + 1];
        const size_t got = std::min<size_t>(it.cap, std::max(it.n_edges[0], it.n_edges[1]));
        it.first.assign(edges.begin() + (long)it.off_edges, edges.begin() + (long)(it.off_edges + got));
        it.last.assign(edges.begin() + (long)(n_ed + it.off_edges), edges.begin() + (long)(n_ed + it.off_edges + got));
    }
    // ---- the rule of the two tools on the uncovered stretches, contig by contig; host images of the new sets
    struct Iv { size_t contig; int64_t s, e; };
    std::vector<std::vector<Iv>> keeps(n);
    for (size_t k = 0; k < n; k++) {
        Item &it = items[k];
        const SeqSet &Q = *Qs[k];
        if (it.n_edges[0] != it.n_edges[1] || it.n_edges[0] > it.cap) { set_error("internal: coverage edges do not pair up"); return MIBLAST_EHIP; }
        it.first.resize(it.n_edges[0]); it.last.resize(it.n_edges[0]);
        std::sort(it.first.begin(), it.first.end()); std::sort(it.last.begin(), it.last.end());
        std::vector<Iv> &keep = keeps[k];
        for (size_t x = 0; x < it.first.size(); x++) {
            const int cg = Q.contig_of(it.first[x]);
            if (cg < 0) continue;
            const int64_t c0 = Q.starts[(size_t)cg], len = Q.lens[(size_t)cg];
            int64_t a = it.first[x] - c0, b = it.last[x] - c0;
            if (b - a < min_size || b <= a) continue;
            a = std::max<int64_t>(0, a - flank); b = std::min(len, b + flank);
            if (!keep.empty() && keep.back().contig == (size_t)cg && a <= keep.back().e) keep.back().e = std::max(keep.back().e, b);
            else keep.push_back(Iv{(size_t)cg, a, b});
        }
    }
    size_t n_iv = 0;
    for (size_t k = 0; k < n; k++) { items[k].off_iv = n_iv; n_iv += 3 * keeps[k].size(); }
    parallel_for(n, [&](size_t k) {
        const std::vector<Iv> &keep = keeps[k];
        if (keep.empty()) return;
        const SeqSet &Q = *Qs[k];
        SeqSet &out = *outs[k];
        out.names.clear(); out.starts.clear(); out.lens.clear(); out.codes.clear();
        out.view = nullptr; out.origin = 0;
        int64_t new_total = 0;
        for (size_t x = 0; x < keep.size(); x++) new_total += (keep[x].e - keep[x].s) + (x ? 1 : 0);
        out.codes.resize((size_t)new_total + 2);
        out.codes[0] = kSep;
        std::vector<long long> &iv = items[k].iv;
        iv.resize(3 * keep.size());
        int64_t at = 0;
        for (size_t x = 0; x < keep.size(); x++) {
            const Iv &v = keep[x];
            if (x) out.codes[(size_t)(1 + at++)] = kSep;
            out.names.push_back(Q.names[v.contig] + "|" + std::to_string(Q.lens[v.contig]) + "|" + std::to_string(v.s));
            out.starts.push_back(at); out.lens.push_back(v.e - v.s);
            iv[3 * x] = at; iv[3 * x + 1] = Q.starts[v.contig] + v.s; iv[3 * x + 2] = v.e - v.s;
            memcpy(out.codes.data() + 1 + at, Q.host() + Q.starts[v.contig] + v.s, (size_t)(v.e - v.s));
            at += v.e - v.s;
        }
        out.codes[(size_t)new_total + 1] = kSep;
        out.total = new_total;
    });
    // ---- device images gathered from the resident sets: one table, one launch
    std::vector<size_t> live;
    for (size_t k = 0; k < n; k++) if (!keeps[k].empty()) live.push_back(k);
    if (live.empty()) return MIBLAST_OK;
    try {
        std::vector<long long> tab(9 * live.size() + n_iv);
        long long grid = 0, iv_at = 0;
        for (size_t x = 0; x < live.size(); x++) {
            const size_t k = live[x];
            nothing_left[k] = false;
            SeqSet &out = *outs[k];
            out.device = ctx.device;
            const size_t nc = out.starts.size();
            const size_t seq_bytes = ((size_t)out.total + 2 * kDevPad + 255) & ~(size_t)255, image = seq_bytes + 2 * nc * sizeof(int64_t);
            out.d_buf = (uint8_t *)device_blocks().take(ctx.device, image, out.d_cap);
            out.d_starts = (int64_t *)(out.d_buf + seq_bytes);
            out.d_lens = out.d_starts + nc;
            GatherItem gi{Qs[k]->dev(), out.d_buf, out.d_starts, out.d_lens, grid, (long long)seq_bytes, out.total, iv_at, (int)keeps[k].size(), 0};
            memcpy(&tab[9 * x], &gi, sizeof gi);
            std::copy(items[k].iv.begin(), items[k].iv.end(), tab.begin() + (long)(9 * live.size()) + iv_at);
            grid += (long long)seq_bytes; iv_at += (long long)items[k].iv.size();
        }
        w.cov_iv.ensure(tab.size() + 3);
        w.stage.h2d(w.cov_iv.p, tab.data(), tab.size() * sizeof(long long), s);
        launch_gather_stretches((const GatherItem *)w.cov_iv.p, (int)live.size(), grid, w.cov_iv.p + 9 * live.size(), s);
        MB_HIP(hipStreamSynchronize(s));                                       // (2) the new sets are resident
        w.stage.done();
    } catch (...) {
        for (size_t k = 0; k < n; k++) if (outs[k]->d_buf) release_seqset(*outs[k]);
        throw;
    }
    return MIBLAST_OK;
}

struct Index { uint32_t n_positions = 0; };

// ---- data derived from a resident set, kept with it (SURVEY 8e "target-major": a target chunk's seed table is built once and stays
// resident while the query chunks stream through it; the '-' strand and the packed form of a query chunk likewise serve every target
// chunk it meets).  Keyed by the set's device image; dropped with the set (release_seqset) or on request (drop_derived: bench.py does
// so at the start of every step, so that a step pays for every table it uses once).  A chunk pair of the reference's CPU path
// (30 Mb + 10 kb) holds 64 MiB + 2 MiB + 4 B per indexed position of table, 30 MB of '-' strand and 23 MB of packed strands: three
// target and three query chunks of a chr20 pair are half a gigabyte of the 288.
struct SetDerived {
    std::mutex mu;                                  // a lane that needs something another lane is building waits here
    std::map<std::pair<int, int64_t>, std::shared_ptr<SeedTable>> tables;      // by (step, first)
    bool rc_ready = false;
    PoolBuf<uint8_t> d_rc;                          // '-' strand, kDevPad separator bytes either side
    PoolPin<uint8_t> h_rc;                          // ... and its host copy ([SEP] codes [SEP])
    PackedStrand packed[2];
    PackedStrand packed_t;                          // the set as a TARGET: its '+' strand packed for the ungapped extension's windows
};
namespace {
struct DerivedCache {
    std::mutex mu;
    std::unordered_map<const uint8_t *, std::shared_ptr<SetDerived>> sets;     // by SeqSet::d_buf
};
DerivedCache &derived_cache() { static DerivedCache *c = new DerivedCache(); return *c; }
}  // namespace
static std::shared_ptr<SetDerived> derived_of(const SeqSet &S) {
    DerivedCache &c = derived_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    std::shared_ptr<SetDerived> &e = c.sets[S.d_buf];
    if (!e) e = std::make_shared<SetDerived>();
    return e;
}
static void forget_derived(const SeqSet &S) {
    std::shared_ptr<SetDerived> gone;                // (freed outside the lock: hipFree waits for the device)
    DerivedCache &c = derived_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.sets.find(S.d_buf);
    if (it != c.sets.end()) { gone = it->second; c.sets.erase(it); }
}
void drop_derived() {
    std::vector<std::shared_ptr<SetDerived>> gone;
    DerivedCache &c = derived_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    for (auto &kv : c.sets) gone.push_back(kv.second);
    c.sets.clear();
}

// the packed form (2 bits + 1 mask bit per base, mb_seed_dense.h) of n code bytes
static void pack_strand(const uint8_t *codes, int64_t n, PackedStrand &ps, hipStream_t s) {
    ps.p2.ensure(packed_words2(n)); ps.pm.ensure(packed_wordsm(n)); ps.px.ensure(packed_dwordsx(n));
    launch_pack2bit(codes, n, ps.p2.p, ps.pm.p, s, ps.px.p);
    ps.ready = true;
}

// seed position table of T into tab (the scratch of the build -- words, bucket counts, scan sums, the packed target -- is the context's)
static void build_index(Ctx &ctx, const SeqSet &T, int step, SeedTable &tab) {
    hipStream_t s = ctx.stream;
    Workspace &w = *ctx.ws;
    // indexed positions are those with (origin + p) % step == 0: a block of a larger file keeps the file's phase
    const int64_t first = (step - T.origin % step) % step;
    int64_t n_slots = T.total > first ? (T.total - first + step - 1) / step : 0;
    tab.step = step; tab.first = first; tab.n_slots = n_slots;
    w.words.ensure((size_t)std::max<int64_t>(1, n_slots));
    w.counts.ensure((size_t)kBuckets + 1);
    tab.offsets.ensure((size_t)kBuckets + 1);
    tab.positions.ensure((size_t)std::max<int64_t>(1, n_slots));
    int64_t nblk = ((int64_t)kBuckets + 1 + 2047) / 2048;
    w.bsum.ensure((size_t)nblk + 2);
    static const long spike_ms = env_long("MIBLAST_DEBUG_SPIKE", 0);
    const double t0 = now_s();
    if (spike_ms) MB_HIP(hipEventRecord(ctx.ev0, s));
    // the bucket counts are zero between builds: the scan zeroes what the histogram counted, and the scatter's cursors (the same
    // array) are zeroed again by a pass over the indexed words -- 64 MiB memsets are most of a small target's build otherwise
    if (!w.counts_zero) MB_HIP(hipMemsetAsync(w.counts.p, 0, ((size_t)kBuckets + 1) * 4, s));
    w.counts_zero = false;                                              // (until the clearing pass below is queued: an error in between costs a memset)
    const long packed_mode = env_long("MIBLAST_SEED_PACKED", 1);       // 0: words from the code bytes; 1: from the packed form for sets of 64 kb and more; 2: always (tests)
    if (packed_mode == 2 || (packed_mode == 1 && T.total >= (1 << 16))) {
        // words from the packed target: 0.375 B per base read once instead of 19 code bytes per window
        pack_strand(T.dev(), T.total, w.pack_t, s);
        launch_index_words_packed(w.pack_t.p2.p, w.pack_t.pm.p, T.total, step, first, w.words.p, n_slots, w.counts.p, s);
    } else
    launch_index_words(T.dev(), T.total, step, first, w.words.p, n_slots, w.counts.p, s);
    tab.occ.ensure((size_t)kBuckets / 32);
    launch_scan_index(w.counts.p, tab.offsets.p, w.bsum.p, tab.occ.p, s);
    launch_index_scatter(w.words.p, n_slots, step, first, tab.offsets.p, w.counts.p, tab.positions.p, s);
    launch_index_clear(w.words.p, n_slots, w.counts.p, s);
    w.counts_zero = true;
    if (spike_ms) MB_HIP(hipEventRecord(ctx.ev1, s));
    const double t1 = now_s();
    w.stage.d2h(&tab.n_positions, tab.offsets.p + kBuckets, 4, s);
    const double t2 = now_s();
    MB_HIP(hipStreamSynchronize(s));
    w.stage.done();
    if (spike_ms && (now_s() - t0) * 1e3 > (double)spike_ms) {
        float ms = 0;
        MB_HIP(hipEventElapsedTime(&ms, ctx.ev0, ctx.ev1));
        fprintf(stderr, "[miblast] slow index build: queueing %.2f ms, copy call %.2f ms, wait %.2f ms; device time first..last kernel %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3,
                (now_s() - t2) * 1e3, ms);
    }
}

// the table of (T, step): the resident one (built by the first lane that asks, under the set's lock), or -- MIBLAST_RESIDENT_TABLES=0, or
// a set that is a view of a parsed file -- one of the context's own, built for this call
static std::shared_ptr<SeedTable> acquire_table(Ctx &ctx, const SeqSet &T, int step, bool &built) {
    built = false;
    const int64_t first = (step - T.origin % step) % step;
    if (env_long("MIBLAST_RESIDENT_TABLES", 1) == 0 || !T.d_buf) {
        if (!ctx.ws->own_table) ctx.ws->own_table = std::make_shared<SeedTable>();
        build_index(ctx, T, step, *ctx.ws->own_table);
        built = true;
        return ctx.ws->own_table;
    }
    std::shared_ptr<SetDerived> d = derived_of(T);
    std::lock_guard<std::mutex> lk(d->mu);
    std::shared_ptr<SeedTable> &t = d->tables[{step, first}];
    if (!t) {
        auto fresh = std::make_shared<SeedTable>();
        build_index(ctx, T, step, *fresh);                              // (synchronises the stream: the table is complete when the lock goes)
        t = fresh;
        built = true;
    }
    return t;
}

// '-' strand of Q (device + pinned host copy) and, on request, the packed form of both strands: resident with the set
static std::shared_ptr<SetDerived> acquire_strands(Ctx &ctx, const SeqSet &Q, bool packed) {
    std::shared_ptr<SetDerived> d = derived_of(Q);
    std::lock_guard<std::mutex> lk(d->mu);
    hipStream_t s = ctx.stream;
    const int64_t qtot = Q.total;
    bool queued = false;
    if (!d->rc_ready) {
        d->d_rc.ensure((size_t)qtot + 2 * kDevPad);
        MB_HIP(hipMemsetAsync(d->d_rc.p, 0xFF, (size_t)qtot + 2 * kDevPad, s));
        launch_revcomp(Q.dev(), d->d_rc.p + kDevPad, Q.d_starts, Q.d_lens, (int)Q.starts.size(), qtot, s);
        d->h_rc.ensure((size_t)qtot + 2);
        MB_HIP(hipMemcpyAsync(d->h_rc.p, d->d_rc.p + kDevPad - 1, (size_t)qtot + 2, hipMemcpyDeviceToHost, s));
        d->rc_ready = true; queued = true;
    }
    if (packed && qtot > 0) {
        if (!d->packed[0].ready) { pack_strand(Q.dev(), qtot, d->packed[0], s); queued = true; }
        if (!d->packed[1].ready) { pack_strand(d->d_rc.p + kDevPad, qtot, d->packed[1], s); queued = true; }
    }
    if (queued) MB_HIP(hipStreamSynchronize(s));                        // (complete before another lane's stream reads them)
    return d;
}

// the packed form of a target (the seed stage's planes + the ungapped extension's records): resident with the set like its seed table, for the windows of the
// ungapped extension (k_ux_extend_pk).  0.375 B per base.
static std::shared_ptr<SetDerived> acquire_packed_target(Ctx &ctx, const SeqSet &T) {
    std::shared_ptr<SetDerived> d = derived_of(T);
    std::lock_guard<std::mutex> lk(d->mu);
    if (!d->packed_t.ready && T.total > 0) {
        pack_strand(T.dev(), T.total, d->packed_t, ctx.stream);
        MB_HIP(hipStreamSynchronize(ctx.stream));                       // (complete before another lane's stream reads it)
    }
    return d;
}

int seed_variant_rank(const uint8_t *t19, const uint8_t *q19) { return variant_rank(t19, q19, kSeedSpan, kSeedSpan); }

int export_index(Ctx &ctx, const SeqSet &T, int step, uint32_t **offsets, uint32_t **positions) {
    MB_HIP(hipSetDevice(ctx.device));
    ctx.ws->stage.abort();
    SeedTable ix;
    build_index(ctx, T, step, ix);
    uint32_t *off = (uint32_t *)malloc(((size_t)kBuckets + 1) * 4);
    uint32_t *pos = (uint32_t *)malloc(((size_t)ix.n_positions + 1) * 4);
    MB_HIP(hipMemcpy(off, ix.offsets.p, ((size_t)kBuckets + 1) * 4, hipMemcpyDeviceToHost));
    if (ix.n_positions) MB_HIP(hipMemcpy(pos, ix.positions.p, (size_t)ix.n_positions * 4, hipMemcpyDeviceToHost));
    // The device numbers the buckets of the dense table by dense_bucket(word) (mb_seedword.h: the word's low base bits above its high ones)
    // and fills a bucket in arrival order; the exported table is the canonical one: buckets by word, positions ascending.
    auto dense_bucket_h = [](uint32_t w) -> uint32_t {
        uint32_t e = 0, o = 0;
        for (int i = 0; i < kSeedWeight; i++) { e |= ((w >> (2 * i)) & 1u) << i; o |= ((w >> (2 * i + 1)) & 1u) << i; }
        return (e << 12) | o;
    };
    uint32_t *off_w = (uint32_t *)malloc(((size_t)kBuckets + 1) * 4);
    uint32_t *pos_w = (uint32_t *)malloc(((size_t)ix.n_positions + 1) * 4);
    uint32_t at = 0;
    for (uint32_t w = 0; w < kBuckets; w++) {
        const uint32_t b = dense_bucket_h(w);
        off_w[w] = at;
        for (uint32_t k = off[b]; k < off[b + 1]; k++) pos_w[at++] = pos[k];
        if (at - off_w[w] > 1) std::sort(pos_w + off_w[w], pos_w + at);
    }
    off_w[kBuckets] = at;
    free(off); free(pos);
    *offsets = off_w; *positions = pos_w;
    return 0;
}

// kernel of a DP launch: one wave per piece with K columns per lane in registers (K = 4 or 8), the 4-wave kernel with the
// LDS ring (windows up to ~1400 columns), or the 4-wave kernel with the ring in HBM (any width)
enum DpKernel { kDpWave2x4 = 2, kDpWave4 = 4, kDpWave8 = 8, kDpLds = 100, kDpHbm = 101 };

struct DpSpans {
    std::mutex m;
    hipEvent_t base = nullptr;
    std::vector<std::pair<float, float>> v;
    double busy_ms() {                                                 // length of the union of the intervals
        std::sort(v.begin(), v.end());
        double busy = 0, end = -1e30;
        for (const auto &iv : v) {
            if (iv.first > end) { busy += iv.second - iv.first; end = iv.second; }
            else if (iv.second > end) { busy += iv.second - end; end = iv.second; }
        }
        return busy;
    }
};

static void collect_dp_time(Ctx &ctx, miblast_stats &st) {             // after the stream has been synchronised
    float ms = 0;
    MB_HIP(hipEventElapsedTime(&ms, ctx.ev0, ctx.ev1));
    st.t_dp_kernel_ms += ms;
    st.dp_kernel_launches++;
    if (ctx.spans && ctx.spans->base) {
        float a = 0, b = 0;
        if (hipEventElapsedTime(&a, ctx.spans->base, ctx.ev0) == hipSuccess && hipEventElapsedTime(&b, ctx.spans->base, ctx.ev1) == hipSuccess) {
            std::lock_guard<std::mutex> lk(ctx.spans->m);
            ctx.spans->v.emplace_back(a, b);
        } else (void)hipGetLastError();
    }
}

static void run_ydrop_timed(Ctx &ctx, miblast_stats &st, int kernel, const DpProb *probs, DpOut *outs, int n,
                            const PairPtrs *pairs, const miblast_params &p, unsigned blk_bytes, bool defer = false, const DpProb *host_probs = nullptr,
                            const int32_t *wall_ref = nullptr, int first = 0, VerifyJob *vjobs = nullptr, int stamp = 0, int force_mod = 0) {
    // (probs + first = the launch's pieces; k_ydrop2 is given the table itself: its pieces read the records of the relays they are aimed at)
    Workspace &g = *ctx.ws;
    // A launch with more pieces than wave slots: the blocks take the pieces longest first (counting sort by the rows a piece will
    // run at most), so that the launch ends with short pieces instead of a long one started late.  Scheduling only.
    const int *order = nullptr;
    if (kernel == kDpWave2x4 && host_probs && n > 4096 && env_long("MIBLAST_DP_LPT", 1) != 0) {
        constexpr int kBins = 256;
        std::vector<int> bin_of((size_t)n), start(kBins + 1, 0);
        for (int x = 0; x < n; x++) {
            const DpProb &pr = host_probs[x];
            const long rows = std::max(0l, (long)(pr.stop_row > 0 ? pr.stop_row : pr.nb) - (long)pr.row_lo);
            const int b = kBins - 1 - (int)std::min<long>(kBins - 1, rows / 16);          // bin 0: the longest
            bin_of[(size_t)x] = b; start[(size_t)b + 1]++;
        }
        for (int b = 0; b < kBins; b++) start[(size_t)b + 1] += start[(size_t)b];
        std::vector<int> ord((size_t)n);
        for (int x = 0; x < n; x++) ord[(size_t)start[(size_t)bin_of[(size_t)x]]++] = x;
        g.dp_order.ensure((size_t)n);
        g.stage.h2d(g.dp_order.p, ord.data(), (size_t)n * sizeof(int), ctx.stream);
        order = g.dp_order.p;
    }
    if (wall_ref && kernel == kDpHbm) {                                  // walls on the HBM ring: one zeroed flag byte per ring column and problem
        g.wall_flags.ensure((size_t)n * (size_t)kGlobalRowCap);
        MB_HIP(hipMemsetAsync(g.wall_flags.p, 0, (size_t)n * (size_t)kGlobalRowCap, ctx.stream));
    }
    MB_HIP(hipEventRecord(ctx.ev0, ctx.stream));
    if (kernel == kDpWave2x4 || kernel == kDpWave4 || kernel == kDpWave8)
        launch_ydrop1(kernel, probs, outs, n, pairs, p.gap_open, p.gap_extend, p.ydrop, g.arena.p, (unsigned long long)g.arena.n - 64, g.arena_next.p,
                      blk_bytes, g.rowdir.p, g.snaps.p, order, ctx.stream, first, vjobs, stamp, force_mod);
    else
        launch_ydrop(kernel == kDpHbm, probs + first, outs, n, pairs, p.gap_open, p.gap_extend, p.ydrop, g.grows.p, g.arena.p,
                     (unsigned long long)g.arena.n - 64, g.arena_next.p, blk_bytes, g.rowdir.p, g.snaps.p, ctx.stream,
                     wall_ref ? g.wall_segs.p : nullptr, wall_ref ? g.wall_alns.p : nullptr, wall_ref, wall_ref && kernel == kDpHbm ? g.wall_flags.p : nullptr);
    MB_HIP(hipEventRecord(ctx.ev1, ctx.stream));
    if (defer) return;                                                  // the caller synchronises once for several things
    MB_HIP(hipEventSynchronize(ctx.ev1));
    collect_dp_time(ctx, st);
}

struct PairJob {                          // one chunk pair of a (possibly batched) call
    const SeqSet *T = nullptr, *Q = nullptr;
    Result *res = nullptr;
    const uint8_t *tc_h = nullptr;
    const uint8_t *qc_h[2] = {nullptr, nullptr};
    const uint8_t *qc_d[2] = {nullptr, nullptr};
    std::vector<miblast_hsp> strand_hsps[2];
    std::vector<DevHsp> found[2];         // HSPs as the device found them, per strand
    std::vector<int32_t> strand_anchor[2];  // anchor offset (k_hsp_anchor) of every entry of strand_hsps[strand], same order
    bool anchor_mismatch = false;           // MIBLAST_CHECK_ANCHORS: the host scan found a different offset
    struct HostOut { int64_t lookups = 0, pre = 0, kept = 0; double seconds = 0; } host_out[2];
    bool defer_host = false;              // batched calls run the host half of the seed stage on worker threads
    int64_t valid_windows = -1;           // seed windows of the '+' strand without N / soft-masked bases (counter seed_lookups)
    std::vector<Unit> units;              // anchors of this pair (merged into the call's unit list in pair order)
    double t_begin = 0;
    std::shared_ptr<SeedTable> table;     // the target's seed table and the query's derived strands, held for the call (seed_phase)
    std::shared_ptr<SetDerived> strands;
    std::shared_ptr<SetDerived> target_packed;        // the target's packed form (k_ux_extend_pk), kept alive for the length of the job
};

// a launch of the ungapped kernels over ONE seed unit: a strand of a pair (the table lives in the kernel arguments)
static UnitTab one_unit(const uint8_t *tc, const uint8_t *qc, int64_t ttot, int64_t qtot) {
    UnitTab ut;
    memset(&ut, 0, sizeof ut);
    ut.one.tc = tc; ut.one.qc = qc; ut.one.ttot = (int32_t)ttot; ut.one.qtot = (int32_t)qtot;
    ut.tab = nullptr; ut.n = 1;
    return ut;
}

// scratch of the level-synchronous ungapped pipeline for nh hits; rec = nh free 8-byte slots (the strand's unsorted keys)
// plane_mul / plane_mask: the scramble the sorted keys' order follows (UxScratch; 1 / ~0 = none) -- the planes then hold plane_mask + 1 bits
static UxScratch ux_scratch(Workspace &w, unsigned long long *rec, size_t nh, int64_t n_diagonals, uint32_t plane_mul = 1u, uint32_t plane_mask = 0xFFFFFFFFu) {
    UxScratch sc;
    static const bool plain_planes = env_long("MIBLAST_UX_PLAIN_PLANES", 0) != 0;      // (A/B switch: bit planes indexed by the diagonal itself)
    if (plain_planes || plane_mul == 1u || plane_mask > 0x3FFFFFFFu || (int64_t)plane_mask + 1 < n_diagonals) { plane_mul = 1u; plane_mask = 0xFFFFFFFFu; }
    else n_diagonals = (int64_t)plane_mask + 1;
    sc.plane_mul = plane_mul; sc.plane_mask = plane_mask;
    // unfinished hits: 16 slots per block of 256 hits + a shared list of nh / 16 + 4096 (40-byte entries: 5 bytes per hit); the same
    // memory later holds the list of dirty runs (4-byte entries: room for 1.25 per hit)
    const size_t n_blk = (nh + 255) / 256, blk_slots = n_blk * 16, cap = nh / 16 + 4096;
    w.ux_entries.ensure(blk_slots + cap); w.ux_cnt.ensure(4 + 2 * n_blk + 4);
    const size_t plane = ((size_t)((n_diagonals + 31) / 32) + 1 + 3) & ~(size_t)3; w.ux_bits.ensure(2 * plane);
    sc.zero_bits_bytes = 2 * plane * 4; sc.zero_cnt_bytes = up16((4 + 2 * n_blk) * 4);
    sc.rec = rec;
    sc.blk_entries = w.ux_entries.p; sc.blk_cnt = w.ux_cnt.p + 4; sc.n_blk = (unsigned)n_blk;
    sc.entries = w.ux_entries.p + blk_slots; sc.entry_cap = (unsigned)std::min<size_t>(cap, 0x7fffffffu); sc.n_entries = w.ux_cnt.p;
    sc.long_bits = w.ux_bits.p; sc.dirty_bits = w.ux_bits.p + plane;
    sc.dirty_runs = (unsigned *)w.ux_entries.p; sc.dirty_cap = (unsigned)std::min<size_t>((blk_slots + cap) * (sizeof(UxEntry) / sizeof(unsigned)), 0x7fffffffu);
    sc.extent = nullptr; sc.extent_live = 1;
    sc.t_px = sc.q_px = nullptr; sc.t_n = sc.q_n = 0;
    return sc;
}

// seed windows of a strand without N or soft-masked bases (the counter seed_lookups = this x the word variants).  Eight bases at a time
// while all of them are unmasked A, C, G, T: a stretch of x such bases holds max(0, x - 18) windows.
static int64_t valid_seed_windows(const uint8_t *qc, int64_t qtot) {
    auto held = [](int64_t x) -> int64_t { return x >= kSeedSpan ? x - kSeedSpan + 1 : 0; };
    int64_t run = 0, valid = 0, i = 0;
    while (i < qtot) {
        if (i + 8 <= qtot) {
            uint64_t w8; memcpy(&w8, qc + i, 8);
            if ((w8 & 0xFCFCFCFCFCFCFCFCull) == 0) { valid += held(run + 8) - held(run); run += 8; i += 8; continue; }
        }
        const int64_t stop = std::min(qtot, i + 8);
        for (; i < stop; i++) { run = qc[i] < 4 ? run + 1 : 0; valid += run >= kSeedSpan; }
    }
    return valid;
}

// host half of the seed stage of one strand: lookup counter, discovery order, entropy filter, --queryhsplimit/--queryhspbest.
// Touches only the strand's own fields of the job, so it can run beside the device half of the other strand or pair.
static void seed_host(const miblast_params &p, PairJob &job, int strand) {
    const double t_h0 = now_s();
    const SeqSet &Q = *job.Q;
    const int64_t qtot = Q.total;
    const uint8_t *tc_h = job.tc_h;
    const uint8_t *const *qc_h = job.qc_h;
    std::vector<DevHsp> &found = job.found[strand];
    std::vector<miblast_hsp> *strand_hsps = job.strand_hsps;
    PairJob::HostOut &out = job.host_out[strand];
        // number of seed word lookups = valid query windows x variants (counter only)
        {
            // (the '-' strand is the contig-wise mirror image of the '+' strand: same number of valid windows)
            int64_t valid = job.valid_windows;                         // (the shared seed stage counts them once per pair before the strands' halves)
            if (valid < 0) {
                valid = valid_seed_windows(qc_h[strand], qtot);
                if (strand == 0) job.valid_windows = valid;
            }
            out.lookups = valid * (p.transitions ? 1 + kSeedWeight : 1);
            if ((p.strands == 1 && strand == 1) || (p.strands == 2 && strand == 0)) out.lookups = 0;      // --strand: this one is not searched
        }
        // (the level-synchronous kernels leave the candidates of hits that the suppression rule dropped in the list, marked)
        found.erase(std::remove_if(found.begin(), found.end(), [](const DevHsp &d) { return d.score == INT32_MIN; }), found.end());
        out.pre = (int64_t)found.size();
        // order HSPs the way the sequential search discovers them: q ascending, word variant, target descending
        struct Key { int32_t q_end, rank, neg_t; size_t idx; };
        std::vector<Key> order(found.size());
        for (size_t k = 0; k < found.size(); k++)
            order[k] = Key{found[k].seed_q_end, variant_rank(tc_h, qc_h[strand], found[k].seed_t_end, found[k].seed_q_end),
                           -found[k].seed_t_end, k};
        std::sort(order.begin(), order.end(), [](const Key &a, const Key &b) {
            if (a.q_end != b.q_end) return a.q_end < b.q_end;
            if (a.rank != b.rank) return a.rank < b.rank;
            return a.neg_t < b.neg_t;
        });
        // entropy filter in IEEE double on the host, like lastz (hard part H4)
        std::vector<miblast_hsp> &hs = strand_hsps[strand];
        std::vector<int32_t> &anc = job.strand_anchor[strand];
        anc.clear();
        for (const Key &k : order) {
            const DevHsp &d = found[k.idx];
            bool keep = true;
            if (p.entropy) {
                int64_t n = (int64_t)d.cnt[0] + d.cnt[1] + d.cnt[2] + d.cnt[3];
                if (n == 0) keep = false;
                else {
                    double H = 0.0;
                    for (int c = 0; c < 4; c++)
                        if (d.cnt[c] > 0) { double pr = (double)d.cnt[c] / (double)n; H -= pr * std::log(pr); }
                    H /= std::log(4.0);
                    keep = (double)d.score * H >= (double)p.hspthresh;
                }
            }
            if (!keep) continue;
            miblast_hsp h;
            h.t_start = d.t_start; h.q_start = d.q_start; h.len = d.len; h.score = d.score;
            h.seed_t_end = d.seed_t_end; h.seed_q_end = d.seed_q_end;
            for (int c = 0; c < 4; c++) h.cnt[c] = d.cnt[c];
            h.strand = strand;
            h.q_contig = Q.contig_of(d.q_start);
            hs.push_back(h);
            anc.push_back(d.anchor_off);
        }
        // --queryhsplimit=keep,nowarn:N: the search of a query stops at N HSPs and keeps them, i.e. the first N in found
        // order per query contig and strand (repeat-masker option set, cactus_progressive_config.xml:36)
        if (p.queryhsplimit > 0) {
            std::vector<int64_t> seen(Q.starts.size() + 1, 0);
            size_t wr = 0;
            for (size_t k = 0; k < hs.size(); k++)
                if (seen[(size_t)hs[k].q_contig]++ < p.queryhsplimit) { anc[wr] = anc[k]; hs[wr++] = hs[k]; }
            hs.resize(wr); anc.resize(wr);
        }
        // --queryhspbest=N per query contig and strand: N best scores, ties to the earlier found
        if (job.res->best_cut && (p.queryhspbest > 0 || p.queryhsplimit > 0)) {
            // the target is one block of several (mb_multi.cpp): which HSPs the WHOLE target keeps was decided over all blocks.  --queryhsplimit: an
            // HSP stays when it is not behind the last of the whole target's first N in found order (the block's own first N, above, hold them);
            // --queryhspbest: when it is not behind the last one kept (score first; of equal scores the earlier found, or with hspbest_ties
            // the later found)
            const std::vector<HspBestCut> &cuts = *job.res->best_cut;
            size_t wr = 0;
            for (size_t k = 0; k < hs.size(); k++) {
                const HspBestCut &c = cuts[2 * (size_t)hs[k].q_contig + (size_t)strand];
                int32_t rank = -1;
                const int64_t neg_t = -(job.res->best_cut_t_origin + (int64_t)hs[k].seed_t_end);
                auto found_order = [&](int32_t q_end, int32_t c_rank, int64_t c_neg_t) -> int {      // -1: found before the cut's HSP, 0: it is that HSP, 1: after
                    if (hs[k].seed_q_end != q_end) return hs[k].seed_q_end < q_end ? -1 : 1;
                    if (rank < 0) rank = variant_rank(tc_h, qc_h[strand], hs[k].seed_t_end, hs[k].seed_q_end);
                    if (rank != c_rank) return rank < c_rank ? -1 : 1;
                    return neg_t != c_neg_t ? (neg_t < c_neg_t ? -1 : 1) : 0;
                };
                bool keep = true;
                if (c.lim_active && p.queryhsplimit > 0) keep = found_order(c.lim_q_end, c.lim_rank, c.lim_neg_t) <= 0;
                if (keep && c.active && p.queryhspbest > 0) {
                    if (hs[k].score != c.score) keep = hs[k].score > c.score;
                    else { const int cmp = found_order(c.q_end, c.rank, c.neg_t); keep = p.hspbest_ties ? cmp >= 0 : cmp <= 0; }
                }
                if (keep) { anc[wr] = anc[k]; hs[wr++] = hs[k]; }
            }
            hs.resize(wr); anc.resize(wr);
        } else
        if (p.queryhspbest > 0) {
            std::vector<std::vector<size_t>> by_contig(Q.starts.size());
            for (size_t k = 0; k < hs.size(); k++) by_contig[(size_t)hs[k].q_contig].push_back(k);
            std::vector<miblast_hsp> kept;
            std::vector<int32_t> kept_anc;
            for (std::vector<size_t> &idx : by_contig) {
                if ((int64_t)idx.size() > p.queryhspbest) {
                    // (ties at the cut: the earlier found -- SURVEY A.10 --, or with miblast_params.hspbest_ties the later found: A.9 #11)
                    if (p.hspbest_ties) std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return hs[a].score != hs[b].score ? hs[a].score > hs[b].score : a > b; });
                    else
                    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return hs[a].score > hs[b].score; });
                    idx.resize((size_t)p.queryhspbest);
                    std::sort(idx.begin(), idx.end());
                }
                for (size_t k : idx) { kept.push_back(hs[k]); kept_anc.push_back(anc[k]); }
            }
            hs.swap(kept); anc.swap(kept_anc);
        }
        out.kept = (int64_t)hs.size();
        out.seconds = now_s() - t_h0;
}

// folds the host halves into the pair's result (after both strands are done)
static void seed_finish(PairJob &job) {
    miblast_stats &st = job.res->stats;
    for (int strand = 0; strand < 2; strand++) {
        const PairJob::HostOut &o = job.host_out[strand];
        st.seed_lookups += o.lookups; st.hsps_pre_entropy += o.pre; st.hsps += o.kept; st.t_seed += o.seconds;
        job.res->hsps.insert(job.res->hsps.end(), job.strand_hsps[strand].begin(), job.strand_hsps[strand].end());
        std::vector<DevHsp>().swap(job.found[strand]);
    }
}

// the plan k_bin_scan left for a strand's nh keys (mb_seed_bin.h): were they counted, and does the largest bin fit the large LDS sorter?
static bool bin_plan_fits(const uint32_t *plan, unsigned long long nh) {
    return nh > 0 && nh < (1ull << 31) && plan[4] == 0u && plan[3] == (uint32_t)nh && plan[1] > 0u && plan[1] <= (uint32_t)bin_cap_big();
}

// index build + seed search + ungapped extension + HSP filters of one pair (uses the shared seed workspace)
// MIBLAST_DEBUG_KEYS=1 (diagnostics, with the guard runs of round 6): the sorted hit keys of a strand as the ungapped kernels are about to
// read them -- every q_end in (0, qtot], every t_end = diagonal - qtot + q_end in (0, ttot]; anything else is reported and the call refused.
static bool debug_check_keys(const char *where, const unsigned long long *d_keys, int64_t n, int64_t ttot, int64_t qtot, hipStream_t s) {
    static const bool on = env_long("MIBLAST_DEBUG_KEYS", 0) != 0;
    if (!on || n <= 0) return true;
    std::vector<unsigned long long> h((size_t)n);
    MB_HIP(hipStreamSynchronize(s));
    MB_HIP(hipMemcpy(h.data(), d_keys, (size_t)n * 8, hipMemcpyDeviceToHost));
    int64_t bad = 0, first = -1;
    for (int64_t i = 0; i < n; i++) {
        const int64_t q_end = (int64_t)(uint32_t)h[(size_t)i], d = (int64_t)(h[(size_t)i] >> 32), t_end = d - qtot + q_end;
        if (q_end < 1 || q_end > qtot || t_end < 1 || t_end > ttot) { if (first < 0) first = i; bad++; }
    }
    if (bad) {
        const int64_t q_end = (int64_t)(uint32_t)h[(size_t)first], d = (int64_t)(h[(size_t)first] >> 32);
        fprintf(stderr, "[miblast debug] %s: %lld of %lld keys lie outside the pair (ttot %lld, qtot %lld); the first: index %lld key %016llx diagonal %lld q_end %lld t_end %lld\n", where,
                (long long)bad, (long long)n, (long long)ttot, (long long)qtot, (long long)first, h[(size_t)first], (long long)d, (long long)q_end, (long long)(d - qtot + q_end));
    }
    return bad == 0;
}

static int seed_phase(Ctx &ctx, const miblast_params &p, PairJob &job) {
    hipStream_t s = ctx.stream;
    const SeqSet &T = *job.T, &Q = *job.Q;
    Result &res = *job.res;
    if (T.device != ctx.device || Q.device != ctx.device) { set_error("sequence set lives on another device"); return MIBLAST_EINVAL; }
    if (T.total + Q.total + 4 >= (int64_t)0x7fffffff) {
        set_error("target+query longer than 2^31-1 bases: chunk the input (Cactus chunkSize is 30 Mb)");
        return MIBLAST_ELIMIT;
    }
    miblast_stats &st = res.stats;
    memset(&st, 0, sizeof st);
    const double t_begin = now_s();
    job.t_begin = t_begin;
    const int64_t qtot = Q.total, ttot = T.total;

    // ---- seed position table ------------------------------------------------------------------
    Workspace &w = *ctx.ws;
    bool table_built = false;
    job.table = acquire_table(ctx, T, p.step, table_built);            // resident with the target: built by the first pair that meets it
    const SeedTable &tab = *job.table;
    st.t_index = table_built ? now_s() - t_begin : 0.0;
    double tp[8] = {t_begin, now_s(), 0, 0, 0, 0, 0, 0};          // MIBLAST_DEBUG_SPIKE: where a slow seed phase spent its time

    // ---- '-' strand of the query (device + pinned host copy: discovery order, anchors, '='/'X' classification read it) and the packed
    //      form of both strands: resident with the query set, made by the first pair that meets it
    const bool ordered = env_long("MIBLAST_SEED_ORDERED", 1) != 0;     // q-ordered one-pass search (mb_seed_dense.h); 0: k_seed_search + whole-key sort
    const long packed_mode = env_long("MIBLAST_SEED_PACKED", 1);
    const bool packed = ordered && (packed_mode == 2 || (packed_mode == 1 && qtot >= (1 << 16)));
    job.strands = acquire_strands(ctx, Q, packed);
    const SetDerived &qs = *job.strands;
    // level 1 of the ungapped extension from the packed strands (mb_ungapped_ux.h, round 6): the target's packed form stays resident with it as
    // its table does; MIBLAST_UX_PACKED=0: windows from the code bytes as before (A/B switch; 2: also below the size that packs the query)
    const long ux_packed_mode = env_long("MIBLAST_UX_PACKED", 1);
    const bool ux_packed = packed && ux_packed_mode != 0 && (ux_packed_mode == 2 || ttot >= (1 << 16));
    if (ux_packed) job.target_packed = acquire_packed_target(ctx, T);
    auto ux_windows = [&](UxScratch &sc, int strand) {
        if (!ux_packed || !job.target_packed || !job.target_packed->packed_t.ready || !qs.packed[strand].ready) return;
        sc.t_px = job.target_packed->packed_t.px.p; sc.t_n = ttot;
        sc.q_px = qs.packed[strand].px.p; sc.q_n = qtot;
    };
    job.tc_h = T.host(); job.qc_h[0] = Q.host(); job.qc_h[1] = qs.h_rc.p + 1;
    job.qc_d[0] = Q.dev(); job.qc_d[1] = qs.d_rc.p + kDevPad;
    const uint8_t *const *qc_d = job.qc_d;
    // the diagonal of a key is scrambled for the sort: (d * hmul) mod 2^B, B = bits of the diagonal space; k_keys_unhash undoes it
    const int diag_bits = std::max(1, (int)std::ceil(std::log2((double)(ttot + qtot + 2))));
    const bool scramble = env_long("MIBLAST_DIAG_SCRAMBLE", 1) != 0;
    const uint32_t hmask = diag_bits >= 32 ? 0xFFFFFFFFu : ((1u << diag_bits) - 1u);
    const uint32_t hmul = scramble ? 0x9E3779B1u : 1u;
    uint32_t hinv = 1u;
    for (int it = 0; it < 5; it++) hinv *= 2u - hmul * hinv;           // Newton: hmul * hinv = 1 mod 2^32 (hmul odd)
    // grouping by diagonal through bins + LDS instead of the device-wide sort (mb_seed_bin.h); rocprim stays for the strands it does not fit
    const bool binned = ordered && env_long("MIBLAST_SORT_BIN", 1) != 0;
    const int bin_mean = (int)std::max<long>(1, env_long("MIBLAST_BIN_MEAN", 11000));
    const int64_t bsw = bin_state_words();
    // the most keys a plan of bins is made for: it follows the largest strand seen so far (by any lane of the call), NOT the key buffer -- that one
    // grows in steps of its own, and a matrix sized by it was allocated anew (hipFree: every lane waits, 0.5 - 1 s with the runtime polling) in the
    // middle of later steps.  A strand beyond it gets no plan and goes through rocprim once.
    auto bin_room = [&](unsigned long long cap) -> unsigned long long {
        const unsigned long long seen = std::max<unsigned long long>(ctx.hits_hint ? ctx.hits_hint->load(std::memory_order_relaxed) : 0ull, w.last_strand_hits);
        return std::min<unsigned long long>(cap, std::max<unsigned long long>(1ull << 20, seen + seen / 4));
    };

    // ---- seed search + ungapped extension, per strand ----------------------------------------------
    // hits per q batch of a strand.  A large pair's strand is ONE batch whenever it can be (no extent[] then, one sort, one launch of the
    // ungapped kernels): 2^27 hits = 1 GiB of keys per buffer, sized for the 288 GB of an MI355X -- a 30 Mb x 30 Mb chunk pair of the
    // human-mouse kind has ~10^8 hits per strand.  MIBLAST_HIT_CAP (the shared seed stage's limit, and the tests' way of forcing batches)
    // wins when it is set.
    const int64_t hit_cap = getenv("MIBLAST_HIT_CAP") ? env_long("MIBLAST_HIT_CAP", 32l << 20) : env_long("MIBLAST_DENSE_HIT_CAP", 128l << 20);
    const bool one_pass = env_long("MIBLAST_SEED_ONE_PASS", 1) != 0;
    const int sort_bits = 32 + std::max(1, (int)std::ceil(std::log2((double)(ttot + qtot + 2))));
    DevBuf<int32_t> &extent = w.extent;                            // (only a strand whose hits need several q batches has one: extent_get / extent_put)
    DevBuf<uint32_t> &qcnt = w.qcnt, &hit_off = w.hit_off;
    qcnt.ensure((size_t)std::max<int64_t>(1, qtot));
    int64_t n_qblk = (qtot + 2047) / 2048;
    DevBuf<unsigned long long> &qbsum = w.qbsum, &scan_scratch = w.scan_scratch, &keys_a = w.keys_a, &keys_b = w.keys_b;
    qbsum.ensure((size_t)n_qblk + 2); scan_scratch.ensure((size_t)n_qblk + 2);
    DevBuf<char> &sort_temp = w.sort_temp;
    DevBuf<DevHsp> &d_hsps = w.hsps;
    DevBuf<UngappedCounters> &d_ctr = w.ctr;
    d_ctr.ensure(2);
    std::vector<unsigned long long> h_qbsum((size_t)n_qblk + 2);
    std::vector<miblast_hsp> *strand_hsps = job.strand_hsps;
    strand_hsps[0].clear(); strand_hsps[1].clear();

    // ---- both strands in one go.  When the key buffer of an earlier call is likely to hold the hits of both strands (half each),
    // the two seed searches are queued back to back with ONE read-back of their totals, then sort + ungapped extension of both
    // strands back to back with ONE read-back of the counters, then the HSPs: three synchronisations per pair instead of six, and
    // no bubble between the strands' kernels.  A strand whose hits do not fit goes through the per-strand path below.
    bool strand_done[2] = {false, false};
    unsigned long long strand_hits[2] = {0, 0};                  // (for sizing the key buffer of the next call)
    // A lane of a call of several large pairs takes whichever pair comes next: the first time it meets a larger one than before, every buffer
    // below grows -- five device allocations of up to 2 GB, each freeing the old block first (hipFree waits for the device: all lanes stand
    // still) -- and a step now and then took 650 ms instead of 170.  The lanes share the largest strand any of them has met and size for it
    // once, before anything is queued.
    if (ctx.hits_hint) {
        const unsigned long long hint = ctx.hits_hint->load(std::memory_order_relaxed);
        if (hint > 0 && 2 * (hint + hint / 8) <= (unsigned long long)hit_cap) {
            const size_t want = (size_t)(hint + hint / 8);
            keys_a.ensure(2 * want); keys_b.ensure(want); d_hsps.ensure(2 * want);
            w.heads.ensure(2 * want + want / 4 + 64);
            (void)ux_scratch(w, nullptr, want, ttot + qtot + 2, hmul, hmask);
            if (binned) { w.bin_state.ensure(2 * (size_t)bsw); w.bin_matrix.ensure(2 * (size_t)bin_matrix_words_for(bin_room(std::min<unsigned long long>((unsigned long long)keys_a.n, (unsigned long long)hit_cap) / 2), diag_bits, bin_mean)); }
        }
    }
    std::future<void> host0;
    // --strand=plus / minus (miblast_params.strands): the other strand is not searched -- it has no hits, no look-ups, no HSPs
    const bool skip_strand[2] = {p.strands == 2, p.strands == 1};
    if (one_pass && qtot >= kSeedSpan && p.strands == 0 && env_long("MIBLAST_SEED_FUSED", 1) != 0) {
        const unsigned long long capH = std::min<unsigned long long>((unsigned long long)keys_a.n, (unsigned long long)hit_cap) / 2;
        // (a pair like the previous one of this workspace must fit, else the attempt costs two searches for nothing)
        if (capH > 0 && w.last_strand_hits + w.last_strand_hits / 8 <= capH) {
            const double t0 = now_s();
            for (auto &row : w.sev) for (hipEvent_t &e : row) if (!e) MB_HIP(hipEventCreate(&e));
            qbsum.ensure(4); w.pin_u64.ensure(16); w.pin_ctr.ensure(2); d_ctr.ensure(2);
            const int64_t ord_words = seed_ord_state_words(qtot);
            if (ordered) { w.ord_state.ensure(2 * (size_t)ord_words); MB_HIP(hipMemsetAsync(w.ord_state.p, 0, up16(2 * (size_t)ord_words * 8), s)); keys_b.ensure((size_t)capH); }
            else MB_HIP(hipMemsetAsync(qbsum.p, 0, 16, s));
            const unsigned long long bcap = binned ? bin_room(capH) : 0;
            const int64_t bmw = binned ? bin_matrix_words_for(bcap, diag_bits, bin_mean) : 0;
            if (binned) { w.bin_state.ensure(2 * (size_t)bsw); w.bin_matrix.ensure(2 * (size_t)bmw); MB_HIP(hipMemsetAsync(w.bin_state.p, 0, 2 * (size_t)bsw * 4, s)); }
            for (int strand = 0; strand < 2; strand++) {
                MB_HIP(hipEventRecord(w.sev[strand][0], s));
                if (ordered)
                    launch_seed_search_ord(qc_d[strand], packed ? qs.packed[strand].p2.p : nullptr, packed ? qs.packed[strand].pm.p : nullptr, qtot, tab.offsets.p, tab.occ.p,
                                           tab.positions.p, p.transitions, hmul, hmask, keys_a.p + (size_t)strand * capH, keys_b.p, capH, w.ord_state.p + (size_t)strand * (size_t)ord_words, s);
                else
                    launch_seed_search(qc_d[strand], qtot, tab.offsets.p, tab.occ.p, tab.positions.p, p.transitions, keys_a.p + (size_t)strand * capH, capH, qbsum.p + strand, s);
                MB_HIP(hipEventRecord(w.sev[strand][1], s));
                // the plan of the bins (how many, the largest) comes back with the strand's hit count: the device knows that count first
                if (binned) launch_bin_plan(keys_a.p + (size_t)strand * capH, w.ord_state.p + (size_t)strand * (size_t)ord_words, bcap, diag_bits, bin_mean, w.bin_state.p + (size_t)strand * (size_t)bsw,
                                            w.bin_matrix.p + (size_t)strand * (size_t)bmw, s);
            }
            if (ordered) {
                for (int strand = 0; strand < 2; strand++) {
                    MB_HIP(hipMemcpyAsync(w.pin_u64.p + strand, w.ord_state.p + (size_t)strand * (size_t)ord_words, 8, hipMemcpyDeviceToHost, s));
                    if (binned) MB_HIP(hipMemcpyAsync(w.pin_u64.p + 4 + 4 * strand, w.bin_state.p + (size_t)strand * (size_t)bsw, 32, hipMemcpyDeviceToHost, s));
                }
            } else
            MB_HIP(hipMemcpyAsync(w.pin_u64.p, qbsum.p, 16, hipMemcpyDeviceToHost, s));
            tp[2] = now_s();
            MB_HIP(hipStreamSynchronize(s));                                       // (1) hits per strand
            tp[3] = now_s();
            unsigned long long nh[2] = {w.pin_u64.p[0], w.pin_u64.p[1]};
            bool fits[2] = {nh[0] <= capH && nh[0] < (1ull << 31), nh[1] <= capH && nh[1] < (1ull << 31)};
            const unsigned long long nh_max = std::max(fits[0] ? nh[0] : 0ull, fits[1] ? nh[1] : 0ull);
            const size_t hoff[2] = {0, fits[0] ? (size_t)nh[0] : 0};
            if (nh_max) {
                keys_b.ensure((size_t)nh_max);
                d_hsps.ensure((fits[0] ? (size_t)nh[0] : 0) + (fits[1] ? (size_t)nh[1] : 0));
                w.heads.ensure(2 * (size_t)nh_max + (size_t)nh_max / 4 + 64); w.n_heads.ensure(8);
                const size_t tb = sort_keys_temp_bytes((int64_t)nh_max, sort_bits);
                sort_temp.ensure(tb + 16);
                MB_HIP(hipMemsetAsync(d_ctr.p, 0, 2 * sizeof(UngappedCounters), s));
                // every strand's counters and its first kBlind HSPs are copied back right behind its kernels, so that the host half
                // of strand '+' (discovery order, entropy filter) starts while the device is still busy with strand '-'
                constexpr size_t kBlind = 16384;
                size_t blind[2] = {0, 0};
                w.pin_hsps.ensure(2 * kBlind);
                for (int strand = 0; strand < 2; strand++) {
                    if (!fits[strand] || !nh[strand]) continue;
                    MB_HIP(hipEventRecord(w.sev[strand][2], s));
                    const uint32_t *plan = (const uint32_t *)(w.pin_u64.p + 4 + 4 * strand);
                    if (binned && bin_plan_fits(plan, nh[strand])) {
                        // the keys dealt into bins by the top bits of the scrambled diagonal, every bin ordered in LDS (mb_seed_bin.h): the same array
                        launch_bin_group(keys_a.p + (size_t)strand * capH, keys_b.p, (int64_t)nh[strand], diag_bits, (int)plan[0], (int)plan[6], (int)plan[2], w.bin_state.p + (size_t)strand * (size_t)bsw,
                                         w.bin_matrix.p + (size_t)strand * (size_t)bmw, hinv, hmask, s);
                        st.seed_binned++;
                    } else {
                    // (q-ordered keys: a stable sort by the diagonal bits alone)
                    sort_keys(sort_temp.p, sort_keys_temp_bytes((int64_t)nh[strand], sort_bits), keys_a.p + (size_t)strand * capH, keys_b.p, (int64_t)nh[strand], ordered ? 32 : 0, sort_bits, s);
                    if (ordered && hmul != 1u) launch_keys_unhash(keys_b.p, (int64_t)nh[strand], hinv, hmask, s);
                    }
                    MB_HIP(hipEventRecord(w.sev[strand][3], s));
                    MB_HIP(hipEventRecord(w.sev[strand][4], s));
                    // (sized for the larger strand before the first strand's kernels are queued: growing a buffer later would free
                    //  memory that queued kernels still use; the strand's unsorted keys are free after its sort and hold the records)
                    const uint32_t pmul = ordered ? hmul : 1u;                   // (the sorted keys follow the scrambled diagonals)
                    if (strand == 0 || !fits[0] || !nh[0]) (void)ux_scratch(w, nullptr, (size_t)nh_max, ttot + qtot + 2, pmul, hmask);
                    UxScratch uxs = ux_scratch(w, keys_a.p + (size_t)strand * capH, (size_t)nh[strand], ttot + qtot + 2, pmul, hmask);
                    ux_windows(uxs, strand);
                    if (!debug_check_keys("both strands in one go", keys_b.p, (int64_t)nh[strand], ttot, qtot, s)) { set_error("MIBLAST_DEBUG_KEYS: hit keys outside the pair"); return MIBLAST_EHIP; }
                    launch_ungapped(keys_b.p, (int64_t)nh[strand], w.heads.p, w.n_heads.p, one_unit(T.dev(), qc_d[strand], ttot, qtot), ttot + qtot, nullptr, p.xdrop, p.hspthresh,
                                    d_hsps.p + hoff[strand], (int64_t)nh[strand], d_ctr.p + strand, &uxs, true, s);
                    MB_HIP(hipEventRecord(w.sev[strand][5], s));
                    MB_HIP(hipMemcpyAsync(w.pin_ctr.p + strand, d_ctr.p + strand, sizeof(UngappedCounters), hipMemcpyDeviceToHost, s));
                    blind[strand] = std::min<size_t>(kBlind, (size_t)nh[strand]);
                    MB_HIP(hipMemcpyAsync(w.pin_hsps.p + (size_t)strand * kBlind, d_hsps.p + hoff[strand], blind[strand] * sizeof(DevHsp), hipMemcpyDeviceToHost, s));
                    MB_HIP(hipEventRecord(strand == 0 ? ctx.ev0 : ctx.ev1, s));
                }
                for (int strand = 0; strand < 2; strand++) {
                    if (!fits[strand] || !nh[strand]) continue;
                    if (strand == 0) tp[4] = now_s();
                    MB_HIP(hipEventSynchronize(strand == 0 ? ctx.ev0 : ctx.ev1));           // (2), (3): this strand is done (the other may still run)
                    tp[5 + strand] = now_s();
                    const UngappedCounters hc = w.pin_ctr.p[strand];
                    if (hc.hsps > nh[strand]) { set_error("HSP buffer overflow"); return MIBLAST_ELIMIT; }
                    st.hits_extended += (int64_t)hc.extended; st.ungapped_cols += (int64_t)hc.cols;
                    float ms;
                    MB_HIP(hipEventElapsedTime(&ms, w.sev[strand][0], w.sev[strand][1])); st.t_seedfill_ms += ms;
                    MB_HIP(hipEventElapsedTime(&ms, w.sev[strand][2], w.sev[strand][3])); st.t_sort_ms += ms;
                    MB_HIP(hipEventElapsedTime(&ms, w.sev[strand][4], w.sev[strand][5])); st.t_ungapped_kernel_ms += ms; st.ungapped_kernel_launches++;
                    std::vector<DevHsp> &found = job.found[strand];
                    found.assign(w.pin_hsps.p + (size_t)strand * kBlind, w.pin_hsps.p + (size_t)strand * kBlind + std::min<size_t>(blind[strand], (size_t)hc.hsps));
                    if ((size_t)hc.hsps > blind[strand]) {                                   // more HSPs than the blind copy took: the rest now
                        found.resize((size_t)hc.hsps);
                        MB_HIP(hipMemcpy(found.data() + blind[strand], d_hsps.p + hoff[strand] + blind[strand], ((size_t)hc.hsps - blind[strand]) * sizeof(DevHsp), hipMemcpyDeviceToHost));
                    }
                    strand_done[strand] = true; strand_hits[strand] = nh[strand]; st.seed_hits += (int64_t)nh[strand]; st.seed_batches++;
                    if (!job.defer_host && strand == 0) host0 = std::async(std::launch::async, [&p, &job] { seed_host(p, job, 0); });
                }
            } else {
                float ms;
                for (int strand = 0; strand < 2; strand++) { MB_HIP(hipEventElapsedTime(&ms, w.sev[strand][0], w.sev[strand][1])); st.t_seedfill_ms += ms; }
            }
            for (int strand = 0; strand < 2; strand++) if (fits[strand] && !nh[strand]) strand_done[strand] = true;      // a strand without a hit
            st.t_seed += now_s() - t0;
        }
    }
    for (int strand = 0; strand < 2 && qtot >= kSeedSpan; strand++) {
        if (skip_strand[strand]) { job.found[strand].clear(); continue; }
        if (strand_done[strand]) {
            if (!job.defer_host && strand == 0 && !host0.valid()) host0 = std::async(std::launch::async, [&p, &job] { seed_host(p, job, 0); });
            continue;
        }
        const double t0 = now_s();
        int32_t *ext_p = nullptr;                                       // set when the strand turns out to need several batches
        unsigned long long batch_cap = (unsigned long long)hit_cap;     // hits per q batch (two-pass path)
        bool ext_scrambled = false;                                     // extent[] slots follow the keys' scramble (q batches)
        std::vector<DevHsp> found;
        int rc_batch = MIBLAST_OK;
        // sort + ungapped extension of the nh keys in keys_a (one q-ordered batch); collects the HSPs
        auto extend_batch = [&](unsigned long long nh, bool timed_fill, bool q_ordered, bool hashed, const uint32_t *plan = nullptr) -> int {
            if (nh >= (1ull << 31)) { set_error("more than 2^31 seed hits in one batch (unmasked repeat?)"); return MIBLAST_ELIMIT; }
            strand_hits[strand] += nh;
            st.seed_hits += (int64_t)nh;
            st.seed_batches++;
            keys_b.ensure((size_t)nh);
            d_hsps.ensure((size_t)nh);
            w.heads.ensure(2 * (size_t)nh + (size_t)nh / 4 + 64); w.n_heads.ensure(8);       // four short-run lists + the long-run list (launch_ungapped)
            size_t tb = sort_keys_temp_bytes((int64_t)nh, sort_bits);
            sort_temp.ensure(tb + 16);
            MB_HIP(hipEventRecord(ctx.ev1, s));
            if (plan && hashed && bin_plan_fits(plan, nh)) {
                launch_bin_group(keys_a.p, keys_b.p, (int64_t)nh, diag_bits, (int)plan[0], (int)plan[6], (int)plan[2], w.bin_state.p, w.bin_matrix.p, hinv, hmask, s);
                st.seed_binned++;
            } else {
            sort_keys(sort_temp.p, tb, keys_a.p, keys_b.p, (int64_t)nh, q_ordered ? 32 : 0, sort_bits, s);       // (k_seed_fill writes the keys in q order)
            if (hashed && hmul != 1u) launch_keys_unhash(keys_b.p, (int64_t)nh, hinv, hmask, s);
            }
            MB_HIP(hipEventRecord(ctx.ev2, s));
            MB_HIP(hipMemsetAsync(d_ctr.p, 0, up16(sizeof(UngappedCounters)), s));
            MB_HIP(hipEventRecord(ctx.ev3, s));
            UxScratch uxs = ux_scratch(w, keys_a.p, (size_t)nh, ttot + qtot + 2, hashed ? hmul : 1u, hmask);               // (the unsorted keys are free now)
            ux_windows(uxs, strand);
            if (!debug_check_keys("a strand at a time", keys_b.p, (int64_t)nh, ttot, qtot, s)) { set_error("MIBLAST_DEBUG_KEYS: hit keys outside the pair"); return MIBLAST_EHIP; }
            UnitTab ut1 = one_unit(T.dev(), qc_d[strand], ttot, qtot);
            if (ext_p && ext_scrambled && hashed) { ut1.ext_mul = hmul; ut1.ext_mask = hmask; }
            launch_ungapped(keys_b.p, (int64_t)nh, w.heads.p, w.n_heads.p, ut1, ttot + qtot, ext_p, p.xdrop, p.hspthresh, d_hsps.p,
                            (int64_t)d_hsps.n, d_ctr.p, &uxs, found.empty() && strand_hits[strand] == nh, s);     // (extent[] is all zero in the first batch only)
            MB_HIP(hipEventRecord(ctx.ev4, s));
            UngappedCounters hc;
            w.stage.d2h(&hc, d_ctr.p, sizeof hc, s);
            MB_HIP(hipStreamSynchronize(s));
            w.stage.done();
            float ms;
            if (timed_fill) { MB_HIP(hipEventElapsedTime(&ms, ctx.ev0, ctx.ev1)); st.t_seedfill_ms += ms; }
            MB_HIP(hipEventElapsedTime(&ms, ctx.ev1, ctx.ev2)); st.t_sort_ms += ms;
            MB_HIP(hipEventElapsedTime(&ms, ctx.ev3, ctx.ev4)); st.t_ungapped_kernel_ms += ms; st.ungapped_kernel_launches++;
            st.hits_extended += (int64_t)hc.extended;
            st.ungapped_cols += (int64_t)hc.cols;
            if (hc.hsps > d_hsps.n) { set_error("HSP buffer overflow"); return MIBLAST_ELIMIT; }
            size_t base = found.size();
            found.resize(base + (size_t)hc.hsps);
            if (hc.hsps) MB_HIP(hipMemcpy(found.data() + base, d_hsps.p, (size_t)hc.hsps * sizeof(DevHsp), hipMemcpyDeviceToHost));
            return MIBLAST_OK;
        };
        // One pass (count, reserve with one atomic per block, fill) when the key buffer of an earlier call is likely to hold
        // all hits of the strand; otherwise, or if they did not fit, the two-pass path below with exact sizes and q-batches.
        bool one_pass_done = false;
        const unsigned long long cap1 = std::min<unsigned long long>((unsigned long long)keys_a.n, (unsigned long long)hit_cap);
        // (a strand like the last one seeded with this workspace must fit, else the attempt costs a search for nothing: 1.6 ms on an
        //  8 Mb pair)
        if (one_pass && cap1 > 0 && w.last_strand_hits <= cap1) {
            qbsum.ensure(2);
            const int64_t ord_words = seed_ord_state_words(qtot);
            if (ordered) { w.ord_state.ensure((size_t)ord_words); MB_HIP(hipMemsetAsync(w.ord_state.p, 0, up16((size_t)ord_words * 8), s)); keys_b.ensure((size_t)cap1); }
            else MB_HIP(hipMemsetAsync(qbsum.p, 0, 16, s));
            MB_HIP(hipEventRecord(ctx.ev0, s));
            if (ordered)
                launch_seed_search_ord(qc_d[strand], packed ? qs.packed[strand].p2.p : nullptr, packed ? qs.packed[strand].pm.p : nullptr, qtot, tab.offsets.p, tab.occ.p,
                                       tab.positions.p, p.transitions, hmul, hmask, keys_a.p, keys_b.p, cap1, w.ord_state.p, s);
            else
                launch_seed_search(qc_d[strand], qtot, tab.offsets.p, tab.occ.p, tab.positions.p, p.transitions, keys_a.p, cap1, qbsum.p, s);
            unsigned long long total = 0;
            uint32_t plan[8] = {0, 0, 0, 0, 1, 0, 0, 0};
            if (binned) {
                const unsigned long long bcap1 = bin_room(cap1);
                w.bin_state.ensure((size_t)bsw); w.bin_matrix.ensure((size_t)bin_matrix_words_for(bcap1, diag_bits, bin_mean));
                MB_HIP(hipMemsetAsync(w.bin_state.p, 0, (size_t)bsw * 4, s));
                launch_bin_plan(keys_a.p, w.ord_state.p, bcap1, diag_bits, bin_mean, w.bin_state.p, w.bin_matrix.p, s);
                w.pin_u64.ensure(16);
                MB_HIP(hipMemcpyAsync(w.pin_u64.p + 4, w.bin_state.p, 32, hipMemcpyDeviceToHost, s));
            }
            w.stage.d2h(&total, ordered ? w.ord_state.p : qbsum.p, 8, s);
            MB_HIP(hipStreamSynchronize(s));
            w.stage.done();
            if (binned) memcpy(plan, w.pin_u64.p + 4, 32);
            if (total <= cap1) {
                one_pass_done = true;
                if (total) { rc_batch = extend_batch(total, true, ordered, ordered, binned ? plan : nullptr); if (rc_batch != MIBLAST_OK) return rc_batch; }
            }
        }
        if (!one_pass_done) {
        launch_seed_count(qc_d[strand], qtot, tab.offsets.p, tab.occ.p, p.transitions, qcnt.p, s);
        launch_block_sums(qcnt.p, qtot, qbsum.p, s);
        MB_HIP(hipMemcpyAsync(h_qbsum.data(), qbsum.p, (size_t)n_qblk * 8, hipMemcpyDeviceToHost, s));
        MB_HIP(hipStreamSynchronize(s));
        {
            unsigned long long all = 0;
            for (int64_t b = 0; b < n_qblk; b++) all += h_qbsum[(size_t)b];
            // A strand in several q batches (round 6): a batch that fits a plan of bins goes through the bins + LDS as a single-batch strand does
            // (mb_seed_bin.h) instead of four radix passes; MIBLAST_BATCH_BINS=2 also cuts the batches to the size the STAGED scatter takes
            // (2 048 bins of the mean size: ~ 2 x 10^7 keys instead of 2^27) -- see batch_bins below for what that measured.  MIBLAST_HIT_CAP, when
            // set, wins.
            batch_cap = (unsigned long long)hit_cap;
            if (binned && !getenv("MIBLAST_HIT_CAP") && env_long("MIBLAST_BATCH_BINS", 0) == 2)
                batch_cap = std::min<unsigned long long>(batch_cap, (unsigned long long)bin_mean << 11);
            if (all > batch_cap) {                                      // several q batches: the diagonals' extents go from one to the next
                // (the batches' keys are sorted by the scrambled diagonal: the extents lie in the same order -- UnitTab::ext_mul -- so that the
                //  kernels walk the array as they walk the keys; MIBLAST_EXTENT_SCRAMBLE=0: a slot per plain diagonal)
                ext_scrambled = hmul != 1u && diag_bits < 32 && env_long("MIBLAST_EXTENT_SCRAMBLE", 1) != 0;
                const size_t slots = ext_scrambled ? (size_t)hmask + 1 : (size_t)(ttot + qtot + 2);
                extent.ensure(slots + 8);
                MB_HIP(hipMemsetAsync(extent.p, 0, up16(slots * 4), s));
                ext_p = extent.p;
            }
        }
        // MIBLAST_BATCH_BINS: 1 (default) a batch that fits a plan of bins takes the bins; 2 also cuts the batches to the staged scatter's size
        // (measured on the 32 Mb x 32 Mb pair of hm30: sorts 36 -> 15 ms of kernel time per step, the step 235 -> 243 ms -- fourteen times the
        // batches, each with its read-backs; not the default); 0: radix sort as before
        const bool batch_bins = binned && env_long("MIBLAST_BATCH_BINS", 1) != 0;
        int64_t b0 = 0;
        while (b0 < n_qblk) {
            // greedy batch of whole 2048-position blocks with at most batch_cap hits
            int64_t b1 = b0;
            unsigned long long nh = 0;
            while (b1 < n_qblk && (b1 == b0 || nh + h_qbsum[(size_t)b1] <= batch_cap)) nh += h_qbsum[(size_t)b1++];
            const int64_t q0 = b0 * 2048, q1 = std::min(qtot, b1 * 2048);
            b0 = b1;
            if (nh == 0) continue;
            hit_off.ensure((size_t)(q1 - q0));
            keys_a.ensure((size_t)nh);
            MB_HIP(hipEventRecord(ctx.ev0, s));
            launch_scan_u32(qcnt.p + q0, hit_off.p, q1 - q0, scan_scratch.p, s);
            launch_seed_fill(qc_d[strand], q0, q1, qtot, tab.offsets.p, tab.occ.p, tab.positions.p, p.transitions, hit_off.p, keys_a.p, s, hmul, hmask);
            uint32_t plan[8] = {0, 0, 0, 0, 1, 0, 0, 0};
            if (batch_bins && nh <= bin_keys_max()) {
                // the plan of the batch's bins: its key count goes to the device (the kernels read it there), the plan comes back
                w.ord_state.ensure(2); w.bin_state.ensure((size_t)bsw); w.bin_matrix.ensure((size_t)bin_matrix_words_for(nh, diag_bits, bin_mean)); w.pin_u64.ensure(16);
                w.stage.h2d(w.ord_state.p, &nh, 8, s);
                MB_HIP(hipMemsetAsync(w.bin_state.p, 0, (size_t)bsw * 4, s));
                launch_bin_plan(keys_a.p, w.ord_state.p, nh, diag_bits, bin_mean, w.bin_state.p, w.bin_matrix.p, s);
                MB_HIP(hipMemcpyAsync(w.pin_u64.p + 4, w.bin_state.p, 32, hipMemcpyDeviceToHost, s));
                MB_HIP(hipStreamSynchronize(s));
                w.stage.done();
                memcpy(plan, w.pin_u64.p + 4, 32);
            }
            rc_batch = extend_batch(nh, true, env_long("MIBLAST_SORT_DIAG_ONLY", 1) != 0, true, batch_bins ? plan : nullptr);
            if (rc_batch != MIBLAST_OK) return rc_batch;
        }
        }
        job.found[strand].swap(found);
        st.t_seed += now_s() - t0;
        // the host half of strand '+' (ordering, entropy filter) overlaps the device half of strand '-'
        if (!job.defer_host && strand == 0) host0 = std::async(std::launch::async, [&p, &job] { seed_host(p, job, 0); });
    }
    // room for both strands of a pair like this one in the key buffer, so that the next call takes the fused path
    {
        const unsigned long long want = 2 * (std::max(strand_hits[0], strand_hits[1]) + std::max(strand_hits[0], strand_hits[1]) / 4) + 1024;
        if (want <= (unsigned long long)hit_cap && (unsigned long long)keys_a.n < want) keys_a.ensure((size_t)want);
        w.last_strand_hits = std::max(strand_hits[0], strand_hits[1]);
        if (ctx.hits_hint) {
            unsigned long long seen = ctx.hits_hint->load(std::memory_order_relaxed);
            while (w.last_strand_hits > seen && !ctx.hits_hint->compare_exchange_weak(seen, w.last_strand_hits, std::memory_order_relaxed)) {}
        }
    }
    {
        static const long spike_ms = env_long("MIBLAST_DEBUG_SPIKE", 0);
        tp[7] = now_s();
        if (spike_ms > 0 && (tp[7] - tp[0]) * 1e3 > (double)spike_ms)
            fprintf(stderr, "[miblast] slow seed phase %.2f ms (T %lld, Q %lld): index %.2f | to search queued %.2f | wait hits %.2f | to strands queued %.2f | wait '+' %.2f | wait '-' %.2f | rest %.2f\n",
                    (tp[7] - tp[0]) * 1e3, (long long)ttot, (long long)qtot, (tp[1] - tp[0]) * 1e3, (tp[2] - tp[1]) * 1e3, (tp[3] - tp[2]) * 1e3, (tp[4] - tp[3]) * 1e3,
                    (tp[5] - tp[4]) * 1e3, (tp[6] - tp[5]) * 1e3, (tp[7] - tp[6]) * 1e3);
    }
    if (host0.valid()) host0.get();
    if (!job.defer_host) {
        if (qtot >= kSeedSpan) { if (!host0.valid() && p.strands == 2) seed_host(p, job, 0); seed_host(p, job, 1); }
        seed_finish(job);
    }
    return MIBLAST_OK;
}


static void build_units(const miblast_params &p, PairJob &job, int pair, std::vector<Unit> &units);

// ---- the seed stage of ALL pairs of a call in shared launches -----------------------------------------------------------------------
// One sparse seed table per distinct target, one seed search over every (pair, strand) unit, ONE sort of all hit keys by diagonal,
// one launch of the ungapped kernels, two synchronisations per call (hit totals; counters + HSPs) -- instead of ~25 launches and three
// synchronisations per pair.  The pairs of a call are independent jobs (/root/reference/src/cactus/paf/local_alignment.py:395-405):
// a unit's hits, diagonals and counters never meet another unit's (SeedUnit, mb_common.h), so the results are, pair by pair, those
// of separate calls.  handled = false: the call does not fit this path (too many hits for one key buffer, coordinates beyond 32
// bits) and goes pair by pair through seed_phase.
static int seed_phase_batched(Ctx &ctx, const miblast_params &p, std::vector<PairJob *> &jobs, bool &handled) {
    handled = false;
    hipStream_t s = ctx.stream;
    Workspace &w = *ctx.ws;
    const size_t n = jobs.size();
    const double t_begin = now_s();
    // ---- does the call fit?  diagonals and q slots of all units in 31 bits, a bounded number of tables
    int64_t n_diag = 0, q_slots = 0;
    for (size_t k = 0; k < n; k++) {
        const SeqSet &T = *jobs[k]->T, &Q = *jobs[k]->Q;
        if (T.device != ctx.device || Q.device != ctx.device) { set_error("sequence set lives on another device"); return MIBLAST_EINVAL; }
        if (T.total + Q.total + 4 >= (int64_t)0x7fffffff) return MIBLAST_OK;                 // (seed_phase reports it)
        n_diag += 2 * (T.total + Q.total + 2);
        q_slots += 2 * ((Q.total + kBsTile - 1) / kBsTile * kBsTile);
    }
    if (n_diag >= (int64_t)0x7fffffff || q_slots >= (int64_t)0x7fffffff) return MIBLAST_OK;
    std::vector<const SeqSet *> targets;
    std::vector<int> target_of(n);
    for (size_t k = 0; k < n; k++) {
        size_t t = 0;
        while (t < targets.size() && targets[t] != jobs[k]->T) t++;
        if (t == targets.size()) targets.push_back(jobs[k]->T);
        target_of[k] = (int)t;
    }
    if (targets.size() > (size_t)env_long("MIBLAST_BATCH_TARGETS", 64)) return MIBLAST_OK;
    if (p.diag_hash16 && 2 * n > 256) return MIBLAST_OK;                                      // (the class keys of mb_hash16.h hold 8 bits of unit)
    const int64_t hit_cap = env_long("MIBLAST_HIT_CAP", 32l << 20);
    // A call whose CHANCE hits alone (strands x word variants x indexed target positions x query positions / 4^12) exceed one key buffer
    // goes pair by pair: decided here, from the sizes, before a table is built or a query copied (round 4 built the tables of all
    // targets and counted the hits of all units first -- 13 % of a 42-pair human-mouse step spent on an answer the sizes give).
    if (n > 1 && !p.diag_hash16) {
        double expected = 0;
        for (size_t k = 0; k < n; k++)
            expected += 2.0 * (p.transitions ? 1 + kSeedWeight : 1) * ((double)jobs[k]->T->total / std::max(1, p.step)) * (double)jobs[k]->Q->total / (double)kBuckets;
        if (expected > (double)hit_cap) return MIBLAST_OK;
    }

    // ---- tables
    std::vector<BatchTarget> tg(targets.size());
    int64_t slots = 0, blocks = 0;
    for (size_t t = 0; t < targets.size(); t++) {
        const SeqSet &T = *targets[t];
        BatchTarget &g = tg[t];
        memset(&g, 0, sizeof g);
        g.codes = T.dev(); g.n = T.total; g.step = p.step;
        g.first = (p.step - T.origin % p.step) % p.step;                                      // (a block of a larger file keeps the file's --step phase)
        g.n_slots = T.total > g.first ? (T.total - g.first + p.step - 1) / p.step : 0;
        g.slot0 = slots; g.cbase = slots + (int64_t)t; g.blk0 = blocks;
        slots += g.n_slots; blocks += (g.n_slots + 255) / 256;
    }
    const int64_t n_cnt = slots + (int64_t)targets.size();
    if (n_cnt >= (int64_t)0x7fffffff) return MIBLAST_OK;
    std::vector<SeedUnit> units(2 * n);
    {
        int64_t d0 = 0, q0 = 0;
        for (size_t k = 0; k < n; k++)
            for (int strand = 0; strand < 2; strand++) {
                SeedUnit &u = units[2 * k + (size_t)strand];
                memset(&u, 0, sizeof u);
                u.qtot = (int32_t)jobs[k]->Q->total; u.ttot = (int32_t)jobs[k]->T->total;
                u.dbase = (uint32_t)d0; u.index = target_of[k]; u.qpos0 = q0;
                d0 += jobs[k]->T->total + jobs[k]->Q->total + 2;
                q0 += (jobs[k]->Q->total + kBsTile - 1) / kBsTile * kBsTile;
            }
    }
    handled = true;
    for (size_t k = 0; k < n; k++) { memset(&jobs[k]->res->stats, 0, sizeof(miblast_stats)); jobs[k]->t_begin = t_begin; }

    const double t_prep0 = now_s();
    // ---- '-' strands (device + pinned host copy: discovery order, anchors, '='/'X' classification read it): one buffer, one launch and
    //      one copy back for the distinct query sets of the call
    {
        std::vector<const SeqSet *> qsets;
        std::vector<RcItem> rc_items;
        std::vector<size_t> q_of(n);
        long long grid = 0;
        for (size_t k = 0; k < n; k++) {
            const SeqSet *Q = jobs[k]->Q;
            size_t x = 0;
            while (x < qsets.size() && qsets[x] != Q) x++;
            if (x == qsets.size()) {
                qsets.push_back(Q);
                rc_items.push_back(RcItem{Q->dev(), Q->d_starts, Q->d_lens, Q->total, grid, (int)Q->starts.size(), 0});
                grid += (long long)(((size_t)Q->total + 2 * kDevPad + 255) & ~(size_t)255);
            }
            q_of[k] = x;
        }
        w.rc.ensure((size_t)grid + 16);
        w.h_rc.ensure((size_t)grid + 16);
        w.rc_items.ensure(rc_items.size());
        w.stage.h2d(w.rc_items.p, rc_items.data(), rc_items.size() * sizeof(RcItem), s);
        launch_revcomp_sets(w.rc_items.p, (int)rc_items.size(), grid, w.rc.p, s);
        MB_HIP(hipMemcpyAsync(w.h_rc.p, w.rc.p, (size_t)grid, hipMemcpyDeviceToHost, s));
        for (size_t k = 0; k < n; k++) {
            PairJob &job = *jobs[k];
            const SeqSet &T = *job.T, &Q = *job.Q;
            const long long off = rc_items[q_of[k]].grid_off + kDevPad;
            job.tc_h = T.host(); job.qc_h[0] = Q.host(); job.qc_h[1] = w.h_rc.p + off;
            job.qc_d[0] = Q.dev(); job.qc_d[1] = w.rc.p + off;
            job.strand_hsps[0].clear(); job.strand_hsps[1].clear();
            for (int strand = 0; strand < 2; strand++) { units[2 * k + (size_t)strand].tc = T.dev(); units[2 * k + (size_t)strand].qc = job.qc_d[strand]; }
        }
    }

    const double t_prep1 = now_s();
    // ---- seed tables of the distinct targets
    // (bitmaps, bucket counts and the scatter's cursors lie one behind the other: one fill zeroes them)
    const size_t bx_cnt_words = up16((size_t)n_cnt * 4) / 4;
    w.bx_bits.ensure(targets.size() * (size_t)kBxWordsPerTarget + bx_cnt_words + 8); w.bx_dir.ensure(targets.size() * (size_t)kBxWordsPerTarget);
    uint32_t *const bx_cnt = (uint32_t *)(w.bx_bits.p + targets.size() * (size_t)kBxWordsPerTarget), *const bx_cursor = bx_cnt + bx_cnt_words;
    w.bx_bsum.ensure(targets.size() * (size_t)kBxDirBlocks);
    w.bx_words.ensure((size_t)std::max<int64_t>(1, slots)); w.bx_positions.ensure((size_t)std::max<int64_t>(1, slots));
    w.bx_starts.ensure((size_t)n_cnt + 8);
    const int64_t scan_tiles = std::max((n_cnt + kBsTile - 1) / kBsTile, q_slots / kBsTile) + 2;
    w.bx_scan.ensure((size_t)scan_tiles);
    w.bx_targets.ensure(tg.size()); w.bx_units.ensure(units.size());
    w.stage.h2d(w.bx_targets.p, tg.data(), tg.size() * sizeof(BatchTarget), s);
    w.stage.h2d(w.bx_units.p, units.data(), units.size() * sizeof(SeedUnit), s);
    for (hipEvent_t &e : w.sev[0]) if (!e) MB_HIP(hipEventCreate(&e));
    MB_HIP(hipEventRecord(w.sev[0][0], s));
    launch_batch_index(w.bx_targets.p, (int)tg.size(), blocks, n_cnt, w.bx_words.p, w.bx_bits.p, w.bx_dir.p, w.bx_bsum.p, bx_cnt, bx_cursor, w.bx_starts.p, w.bx_scan.p,
                       w.bx_positions.p, s);
    MB_HIP(hipEventRecord(w.sev[0][1], s));

    // ---- hits per query position of every unit, their scan; the tile totals come back
    const int64_t n_tiles = q_slots / kBsTile;
    w.qcnt.ensure((size_t)std::max<int64_t>(1, q_slots)); w.hit_off.ensure((size_t)std::max<int64_t>(1, q_slots));
    w.pin_scan.ensure((size_t)n_tiles + 2);
    launch_batch_seed_count(w.bx_units.p, (int)units.size(), w.bx_targets.p, w.bx_bits.p, w.bx_dir.p, w.bx_starts.p, p.transitions, q_slots, w.qcnt.p, w.hit_off.p,
                            w.bx_scan.p, s);
    MB_HIP(hipEventRecord(w.sev[0][2], s));
    unsigned long long total = 0;
    const double t_prep2 = now_s();
    if (q_slots > 0) {
        MB_HIP(hipMemcpyAsync(w.pin_scan.p, w.bx_scan.p, ((size_t)n_tiles + 1) * 8, hipMemcpyDeviceToHost, s));
        MB_HIP(hipStreamSynchronize(s));                                           // (1) hits of every unit
        total = w.pin_scan.p[n_tiles];
    } else {
        MB_HIP(hipStreamSynchronize(s));
    }
    const double t_index = now_s() - t_begin;
    if (total > (unsigned long long)hit_cap || total >= (1ull << 31)) { handled = false; return MIBLAST_OK; }      // q batches: the per-pair path has them
    std::vector<unsigned long long> unit_hits(units.size(), 0);
    for (size_t u = 0; u < units.size(); u++) {
        const unsigned long long lo = w.pin_scan.p[units[u].qpos0 / kBsTile];
        const unsigned long long hi = u + 1 < units.size() ? w.pin_scan.p[units[u + 1].qpos0 / kBsTile] : total;
        unit_hits[u] = hi - lo;
    }

    // ---- keys, ONE sort by diagonal (k_bs_fill writes a unit's keys in q order, the units in order), ungapped extension of all units
    std::vector<UngappedCounters> hc(units.size());
    size_t n_found = 0;
    float ms_fill = 0, ms_sort = 0, ms_ung = 0, ms_index = 0, ms_count = 0;
    if (total) {
        const size_t nh = (size_t)total;
        w.keys_a.ensure(nh); w.keys_b.ensure(nh);
        w.hsps.ensure(nh);
        w.heads.ensure(2 * nh + nh / 4 + 64); w.n_heads.ensure(8);
        const int sort_bits = 32 + std::max(1, (int)std::ceil(std::log2((double)(n_diag + 2))));
        const size_t tb = sort_keys_temp_bytes((int64_t)nh, sort_bits);
        const size_t pb = p.diag_hash16 ? sort_pairs_temp_bytes((int64_t)nh) : 0;
        w.sort_temp.ensure(std::max(tb, pb) + 16);
        if (p.diag_hash16) { w.h16_ka.ensure(nh); w.h16_kb.ensure(nh); w.h16_va.ensure(nh); w.h16_vb.ensure(nh); }
        // the units' counters and the run-list counters lie one behind the other: one fill.  (No extent[]: the hits of the call are ONE
        // batch, so no diagonal carries an extent from an earlier one -- the kernels take nullptr for "all zero, nothing kept".)
        const size_t ext_bytes = 0, ctr_bytes = up16(units.size() * sizeof(UngappedCounters)), nh_bytes = 32;
        w.extent.ensure((ext_bytes + ctr_bytes + nh_bytes) / 4 + 8);
        UngappedCounters *const d_ctr = (UngappedCounters *)((uint8_t *)w.extent.p + ext_bytes);
        unsigned *const d_n_heads = (unsigned *)((uint8_t *)w.extent.p + ext_bytes + ctr_bytes);
        MB_HIP(hipMemsetAsync(w.extent.p, 0, ext_bytes + ctr_bytes + nh_bytes, s));
        (void)ux_scratch(w, nullptr, nh, n_diag + 2);                               // (sized before anything is queued)
        MB_HIP(hipEventRecord(w.sev[0][3], s));
        launch_batch_seed_fill(w.bx_units.p, (int)units.size(), w.bx_targets.p, w.bx_bits.p, w.bx_dir.p, w.bx_starts.p, w.bx_positions.p, p.transitions, q_slots,
                               w.hit_off.p, w.keys_a.p, s);
        MB_HIP(hipEventRecord(w.sev[0][4], s));
        sort_keys(w.sort_temp.p, tb, w.keys_a.p, w.keys_b.p, (int64_t)nh, env_long("MIBLAST_SORT_DIAG_ONLY", 1) != 0 ? 32 : 0, sort_bits, s);
        MB_HIP(hipEventRecord(w.sev[0][5], s));
        const UxScratch uxs = ux_scratch(w, w.keys_a.p, nh, n_diag + 2);            // (the unsorted keys are free now)
        UnitTab ut;
        ut.one = units[0]; ut.tab = w.bx_units.p; ut.n = (int32_t)units.size(); ut.ext_mul = 0; ut.ext_mask = 0;
        if (p.diag_hash16) {
            // lastz's 16-bit diagonal hash: every hit extended, the rule per hash class afterwards (mb_hash16.h)
            launch_ungapped_hash16(w.keys_b.p, (int64_t)nh, ut, n_diag, p.xdrop, p.hspthresh, w.hsps.p, (int64_t)nh, d_ctr, &uxs, w.h16_ka.p, w.h16_kb.p, w.h16_va.p,
                                   w.h16_vb.p, w.sort_temp.p, pb, nullptr, s);
        } else
        launch_ungapped(w.keys_b.p, (int64_t)nh, w.heads.p, d_n_heads, ut, n_diag, nullptr, p.xdrop, p.hspthresh, w.hsps.p, (int64_t)nh, d_ctr, &uxs, true, s, true);
        MB_HIP(hipEventRecord(ctx.ev0, s));
        constexpr size_t kBlind = 1 << 16;                                          // HSPs copied back before their number is known
        w.pin_ctr.ensure(units.size());
        w.pin_hsps.ensure(kBlind);
        const size_t blind = std::min(kBlind, nh);
        MB_HIP(hipMemcpyAsync(w.pin_ctr.p, d_ctr, units.size() * sizeof(UngappedCounters), hipMemcpyDeviceToHost, s));
        MB_HIP(hipMemcpyAsync(w.pin_hsps.p, w.hsps.p, blind * sizeof(DevHsp), hipMemcpyDeviceToHost, s));
        MB_HIP(hipStreamSynchronize(s));                                           // (2) counters + HSPs
        for (size_t u = 0; u < units.size(); u++) hc[u] = w.pin_ctr.p[u];
        n_found = (size_t)hc[0].hsps;                                              // (the slot counter of the launch)
        if (n_found > nh) { set_error("HSP buffer overflow"); return MIBLAST_ELIMIT; }
        MB_HIP(hipEventElapsedTime(&ms_fill, w.sev[0][3], w.sev[0][4]));
        MB_HIP(hipEventElapsedTime(&ms_sort, w.sev[0][4], w.sev[0][5]));
        MB_HIP(hipEventElapsedTime(&ms_ung, w.sev[0][5], ctx.ev0));
        std::vector<DevHsp> rest;
        if (n_found > blind) {
            rest.resize(n_found - blind);
            MB_HIP(hipMemcpy(rest.data(), w.hsps.p + blind, rest.size() * sizeof(DevHsp), hipMemcpyDeviceToHost));
        }
        for (size_t x = 0; x < n_found; x++) {
            const DevHsp &d = x < blind ? w.pin_hsps.p[x] : rest[x - blind];
            if (d.unit < 0 || (size_t)d.unit >= units.size()) { set_error("internal: HSP of an unknown seed unit"); return MIBLAST_EHIP; }
            jobs[(size_t)d.unit / 2]->found[d.unit & 1].push_back(d);
        }
    }
    MB_HIP(hipEventElapsedTime(&ms_index, w.sev[0][0], w.sev[0][1]));
    MB_HIP(hipEventElapsedTime(&ms_count, w.sev[0][1], w.sev[0][2]));
    const double t_dev = now_s() - t_begin;
    // ---- per-pair counters; launch-level times are booked on the first pair (they add up over the pairs of a call)
    for (size_t k = 0; k < n; k++) {
        miblast_stats &st = jobs[k]->res->stats;
        for (int strand = 0; strand < 2; strand++) {
            const size_t u = 2 * k + (size_t)strand;
            st.seed_hits += (int64_t)unit_hits[u];
            st.hits_extended += (int64_t)hc[u].extended;
            st.ungapped_cols += (int64_t)hc[u].cols;
            if (unit_hits[u]) st.seed_batches++;
        }
    }
    {
        miblast_stats &st = jobs[0]->res->stats;
        st.t_index = t_index; st.t_seed = t_dev - t_index;
        st.t_seedfill_ms = ms_fill + ms_count; st.t_sort_ms = ms_sort; st.t_ungapped_kernel_ms = ms_ung; st.ungapped_kernel_launches = total ? 1 : 0;
        (void)ms_index;
    }
    if (env_long("MIBLAST_DEBUG", 0))
        fprintf(stderr, "[miblast]   host timeline of the index part: set-up %.2f ms, '-' strands queued %.2f, tables + count queued %.2f, wait %.2f\n", (t_prep0 - t_begin) * 1e3, (t_prep1 - t_prep0) * 1e3,
                (t_prep2 - t_prep1) * 1e3, (t_begin + t_index - t_prep2) * 1e3);
    if (env_long("MIBLAST_DEBUG", 0))
        fprintf(stderr, "[miblast] batched seed stage: %zu pairs, %zu targets, %llu hits, %zu HSP candidates; index %.2f ms (kernels %.2f), count %.2f, fill %.2f, sort %.2f, ungapped %.2f; device part %.2f ms\n",
                n, targets.size(), total, n_found, t_index * 1e3, ms_index, ms_count, ms_fill, ms_sort, ms_ung, t_dev * 1e3);
    // ---- host halves: discovery order, entropy filter, HSP limits, anchors -- pair by pair on the worker threads
    const double t_h0 = now_s();
    parallel_for(n, [&](size_t k) { if (jobs[k]->Q->total >= kSeedSpan) jobs[k]->valid_windows = valid_seed_windows(jobs[k]->qc_h[0], jobs[k]->Q->total); });
    const double t_h1 = now_s();
    parallel_for(2 * n, [&](size_t ks) {                                  // (a strand's half touches only the strand's own fields of the job)
        PairJob &job = *jobs[ks / 2];
        if (job.Q->total >= kSeedSpan) seed_host(p, job, (int)(ks & 1));
    });
    const double t_h2 = now_s();
    parallel_for(n, [&](size_t k) {
        PairJob &job = *jobs[k];
        seed_finish(job);
        build_units(p, job, (int)k, job.units);
    });
    if (env_long("MIBLAST_DEBUG", 0))
        fprintf(stderr, "[miblast] host halves of the seed stage: window count %.2f ms, strands %.2f ms, anchors %.2f ms\n", (t_h1 - t_h0) * 1e3, (t_h2 - t_h1) * 1e3, (now_s() - t_h2) * 1e3);
    return MIBLAST_OK;
}

// the runs of at least min_len N bases (code 4, soft-masked or not) of codes[0, n), as [start, end) pairs in order: chunks on the
// worker threads, runs that cross a chunk border stitched afterwards
static std::vector<std::pair<int32_t, int32_t>> n_runs_of(const uint8_t *codes, int64_t n, int32_t min_len) {
    const size_t kChunk = 1 << 18, parts = (size_t)((n + (int64_t)kChunk - 1) / (int64_t)kChunk);
    std::vector<std::vector<std::pair<int32_t, int32_t>>> found(parts);
    parallel_for(parts, [&](size_t c) {
        const int64_t lo = (int64_t)c * (int64_t)kChunk, hi = std::min<int64_t>(n, lo + (int64_t)kChunk);
        int64_t x = lo;
        while (x < hi) {
            // (8 bases at a time while none is an N: bit 2 set and bit 7 clear -- a separator is not an N)
            while (x + 8 <= hi) {
                uint64_t w; memcpy(&w, codes + x, 8);
                if ((w & ~(w >> 5) & 0x0404040404040404ull) != 0) break;
                x += 8;
            }
            while (x < hi && (codes[x] & 7u) != 4u) x++;
            if (x >= hi) break;
            const int64_t s = x;
            while (x < hi && (codes[x] & 7u) == 4u) x++;
            found[c].push_back({(int32_t)s, (int32_t)x});
        }
    });
    std::vector<std::pair<int32_t, int32_t>> runs;
    for (size_t c = 0; c < parts; c++)
        for (const auto &r : found[c]) {
            if (!runs.empty() && runs.back().second == r.first) runs.back().second = r.second;      // the same run on both sides of a border
            else runs.push_back(r);
        }
    runs.erase(std::remove_if(runs.begin(), runs.end(), [&](const std::pair<int32_t, int32_t> &r) { return r.second - r.first < min_len; }), runs.end());
    return runs;
}

// The N runs of a resident set are found once, when the set is made resident (upload_seqset; a set the device cuts out of another one --
// seqset_unaligned -- at its first call), and kept for as long as the set lives: the genomes of a phase take part in call after call,
// and scanning target and both strands of the query was a third of a pair's host half.  All runs, per set, keyed by the set's host
// image AND its length (a block view of a parsed file shares its parent's pointer -- block 0 -- but not its length); the '-' strand's
// runs are the '+' strand's mirrored contig by contig.
namespace {
struct NRunKey {
    const uint8_t *host; int64_t total;
    bool operator==(const NRunKey &o) const { return host == o.host && total == o.total; }
};
struct NRunKeyHash { size_t operator()(const NRunKey &k) const { return std::hash<const void *>()(k.host) ^ (std::hash<int64_t>()(k.total) * 0x9E3779B97F4A7C15ull); } };
struct NRunCache {
    std::mutex mu;
    std::unordered_map<NRunKey, std::shared_ptr<const std::vector<std::pair<int32_t, int32_t>>>, NRunKeyHash> runs;
};
NRunCache &n_run_cache() { static NRunCache *c = new NRunCache(); return *c; }
}  // namespace

static std::shared_ptr<const std::vector<std::pair<int32_t, int32_t>>> n_runs_cached(const SeqSet &S) {
    NRunCache &c = n_run_cache();
    {
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.runs.find(NRunKey{S.host(), S.total});
        if (it != c.runs.end()) return it->second;
    }
    auto made = std::make_shared<const std::vector<std::pair<int32_t, int32_t>>>(n_runs_of(S.host(), S.total, 1));
    std::lock_guard<std::mutex> lk(c.mu);
    return c.runs.emplace(NRunKey{S.host(), S.total}, made).first->second;
}
void note_n_runs(const SeqSet &S) { if (S.total > 0) (void)n_runs_cached(S); }
void forget_n_runs(const SeqSet &S) {
    NRunCache &c = n_run_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    c.runs.erase(NRunKey{S.host(), S.total});
}
// runs of at least min_len bases; mirrored = in the coordinates of the contig-wise reverse complement
static std::vector<std::pair<int32_t, int32_t>> n_runs_for(const SeqSet &S, int32_t min_len, bool mirrored) {
    const auto all = n_runs_cached(S);
    std::vector<std::pair<int32_t, int32_t>> out;
    for (const auto &r : *all) {
        if (r.second - r.first < min_len || r.first < 0 || (int64_t)r.second > S.total) continue;      // (runs lie inside [0, total) by construction)
        if (!mirrored) { out.push_back(r); continue; }
        const int cg = S.contig_of(r.first);
        const int32_t c0 = (int32_t)S.starts[(size_t)cg], c1 = c0 + (int32_t)S.lens[(size_t)cg];
        out.push_back({c0 + (c1 - r.second), c0 + (c1 - r.first)});
    }
    if (mirrored) std::sort(out.begin(), out.end());
    return out;
}

// anchors of one pair, one unit per (query contig, strand), sorted by (-score, t, q)  
static void build_units(const miblast_params &p, PairJob &job, int pair, std::vector<Unit> &units) {
    const SeqSet &Q = *job.Q;
    miblast_stats &st = job.res->stats;
    const uint8_t *tc_h = job.tc_h;
    const uint8_t *const *qc_h = job.qc_h;
    std::vector<miblast_hsp> *strand_hsps = job.strand_hsps;
    const size_t first_unit = units.size();
    double t_bu[4] = {0, 0, 0, 0};
    if (p.gapped) {
        for (int strand = 0; strand < 2; strand++) {
            const double t_b0 = now_s();
            std::vector<std::vector<Anchor>> per((size_t)Q.starts.size());
            const uint8_t *qc = qc_h[strand];
            // anchor = middle of the best-scoring 31-column window (first on ties); SURVEY A.6
            const std::vector<miblast_hsp> &hs = strand_hsps[strand];
            std::vector<int> offs(hs.size());
            const size_t kChunk = 256;
            // (the device computed the offsets: k_hsp_anchor.  MIBLAST_HOST_ANCHORS=1 scans on the host instead, MIBLAST_CHECK_ANCHORS=1
            //  does both and fails on a difference)
            const bool host_anchors = env_long("MIBLAST_HOST_ANCHORS", 0) != 0, check_anchors = env_long("MIBLAST_CHECK_ANCHORS", 0) != 0;
            const std::vector<int32_t> &dev_offs = job.strand_anchor[strand];
            const bool have_dev = dev_offs.size() == hs.size();
            if (have_dev && !host_anchors) for (size_t x = 0; x < hs.size(); x++) offs[x] = dev_offs[x];
            if (!have_dev || host_anchors || check_anchors) {
            std::atomic<long> n_bad{0};
            parallel_for((hs.size() + kChunk - 1) / kChunk, [&](size_t c) {
                for (size_t x = c * kChunk; x < std::min(hs.size(), (c + 1) * kChunk); x++) {
                    const miblast_hsp &h = hs[x];
                    int off;
                    if (h.len <= 31) off = h.len / 2;
                    else {
                        // every column is scored once: the scores of the window live in a 32-entry ring
                        const uint8_t *tp = tc_h + h.t_start, *qp = qc + h.q_start;
                        int8_t ring[32];
                        int sum = 0;
                        for (int k = 0; k < 31; k++) { ring[k] = (int8_t)host_score(tp[k], qp[k]); sum += ring[k]; }
                        int bestsum = sum, bestc = 0;
                        for (int cc = 1; cc + 31 <= h.len; cc++) {
                            const int in = host_score(tp[cc + 30], qp[cc + 30]);
                            sum += in - ring[(cc - 1) & 31];
                            ring[(cc + 30) & 31] = (int8_t)in;
                            if (sum > bestsum) { bestsum = sum; bestc = cc; }
                        }
                        off = bestc + 15;
                    }
                    if (have_dev && !host_anchors && off != offs[x]) n_bad++;
                    offs[x] = off;
                }
            });
            if (n_bad.load()) { set_error("k_hsp_anchor disagrees with the host scan on " + std::to_string(n_bad.load()) + " HSPs"); job.anchor_mismatch = true; }
            }
            const double t_b1 = now_s();
            for (size_t x = 0; x < hs.size(); x++)
                per[(size_t)hs[x].q_contig].push_back(Anchor{hs[x].t_start + offs[x], hs[x].q_start + offs[x], hs[x].score});
            for (size_t qc_i = 0; qc_i < per.size(); qc_i++) {
                if (per[qc_i].empty()) continue;
                Unit u;
                u.pair = pair; u.strand = strand; u.q_contig = (int)qc_i;
                u.anchors.swap(per[qc_i]);
                st.anchors += (int64_t)u.anchors.size();
                units.push_back(std::move(u));
            }
            t_bu[0] += t_b1 - t_b0; t_bu[1] += now_s() - t_b1;
        }
        const double t_b2 = now_s();
        const int32_t n_run = (int32_t)std::max(8l, env_long("MIBLAST_GROUP_NRUN", p.ydrop / 100));
        std::vector<std::pair<int32_t, int32_t>> n_t, n_q[2];
        bool have_nq[2] = {false, false};
        if (units.size() > first_unit) n_t = n_runs_for(*job.T, n_run, false);
        for (size_t x = first_unit; x < units.size(); x++) {
            Unit &u = units[x];
            if (!have_nq[u.strand]) { n_q[u.strand] = n_runs_for(Q, n_run, u.strand == 1); have_nq[u.strand] = true; }
            parallel_sort(u.anchors.begin(), u.anchors.end(), [](const Anchor &a, const Anchor &b) {
                if (a.score != b.score) return a.score > b.score;
                if (a.t != b.t) return a.t < b.t;
                return a.q < b.q;
            });
            // group gap: a few dozen times the unit's mean anchor spacing (sparser seeding -- larger --step, soft-masked chunks -- must
            // not fragment one alignment into a head per kilobase), within [4096, 65536]
            long group_gap = env_long("MIBLAST_GROUP_GAP", 0);
            if (group_gap <= 0 && !u.anchors.empty()) {
                int32_t q_min = u.anchors[0].q, q_max = q_min;
                for (const Anchor &a : u.anchors) { q_min = std::min(q_min, a.q); q_max = std::max(q_max, a.q); }
                group_gap = std::min(65536l, std::max(4096l, 48l * (long)(q_max - q_min) / (long)u.anchors.size()));
            }
            u.index_anchors((int32_t)std::max(1l, group_gap), (int32_t)env_long("MIBLAST_GROUP_TOL", 64), &n_t, &n_q[u.strand], n_run);
        }
        t_bu[2] = now_s() - t_b2;
        if (env_long("MIBLAST_DEBUG", 0) > 1) fprintf(stderr, "[miblast]   build_units: window scan %.2f ms, distribute %.2f ms, sorts %.2f ms\n", t_bu[0] * 1e3, t_bu[1] * 1e3, t_bu[2] * 1e3);
    }
}

// ---- the gapped stage, in units (round 5; until then gapped_phase was one function of 1 200 lines) -------------------------------------
// gapped_phase (below) runs the rounds; a round is
//   gapped_commit_and_nominate   commit what can be committed in anchor order, pick the next speculative batch of anchors
//   (gapped_phase itself)        the one-sided DPs of the batch as chains of pieces: relay planting, the DP launches with their hand-over
//                                checks, continuations, the wide sides -- what is left behind is a RoundDp
//   gapped_finish_round          scores and boxes of the batch, the traceback in two phases, the merge into '=' / 'X' / 'I' / 'D' runs
// and the trace arena is taken and grown by acquire_trace_arena / grow_trace_arena.
namespace {
struct Pending { size_t unit, anchor; };          // an anchor nominated for a speculative pair of one-sided DPs
struct Piece {
    int unit; int32_t ot, oq, dir;      // origin (concatenated coordinates) and direction
    int32_t row_lo, min_row, stop_row;
    int target;                         // relay point the stop row is aimed at (index into relay_pts, -1: none)
    int ckpt;                           // which entry snapshot of that relay the hand-over is checked against (0: after relay_w rows, 1: 2x, 2: 4x)
    int init_piece;                     // continuation: the piece whose exit snapshot it starts from
    int vjob;                           // index into vres of the hand-over check made after it ran (-1: none)
    int cont;                           // the piece that continues this one after a rejected hand-over (-1: none)
};
struct SideRun {
    DpProb base;
    int unit = 0;
    std::vector<int> cur;               // pieces of the current run (one origin), in row order
    size_t accounted = 0;               // how many of them are folded into the result
    std::vector<int> chain;             // validated pieces, head first
    std::vector<int32_t> chain_floor;   // per chain entry: the rows of that piece up to this one belong to the piece before it (head: -1)
    long long c_off = 0;                // score of the current run's origin in the head's scores
    long long acc_cells = 0, acc_rows = 0, entry_cells = 0, entry_rows = 0;
    int gbest = -1, gbi = 0, gbj = 0, best_piece = -1;
    bool done = false, wide = false;
};
// what the DPs of a round leave behind for its traceback: the sides' validated chains and the pieces they are made of
struct RoundDp {
    std::vector<SideRun> sides;
    std::vector<Piece> pieces;
    std::vector<DpProb> probs;
    std::vector<DpOut> outs;
};
struct NominateCfg { size_t batch_max; long shadow_q0, shadow_d, spec_target, relay_s0_env; bool chain_heads, walls; };
}  // namespace

namespace {
struct NextPt { int32_t ok, t, q, set_tail; };                 // set_tail >= 0: a virtual tail relay, that many lattice lines past the last anchor
struct RelayLattice { long relay_s, relay_w, relay_tol, relay_gap, relay_tail_rows; };     // spacing, warm-up rows, diagonal tolerance, bridging distance (lattice steps), tail
}  // namespace
// the relay after the point (t, q) of a unit, walking in direction dir: in the first q-bucket of width relay_s at least
// min_dq rows away, the best-scoring anchor (= smallest index, preferably near the start of the bucket) whose
// diagonal lies within relay_tol of (t - q).  A bucket without such an anchor gets a VIRTUAL relay on the line to
// the next anchor further down (any cell near the path works as an origin -- a fresh DP locks onto the path and
// its state converges all the same; k_verify decides) so that stretches without seeds (soft-masked repeats)
// do not turn into one long piece.  No anchor within relay_gap buckets: the chain ends.
// Depends on the unit's anchors and the point only, so chains started from different heads merge.
static NextPt relay_next_point(const Unit &u, const RelayLattice &lat, const DpProb &b, int32_t t, int32_t q, int32_t min_dq, int from_tail) {
    const long relay_s = lat.relay_s, relay_w = lat.relay_w, relay_tol = lat.relay_tol, relay_gap = lat.relay_gap, relay_tail_rows = lat.relay_tail_rows;
    const int32_t dirn = b.dir;
    const long s_from = (long)dirn * q + min_dq;                 // first admissible position in walking order, s = dir * q
    const long line = (s_from >= 0 ? (s_from + relay_s - 1) / relay_s : -((-s_from) / relay_s)) * relay_s;   // next lattice line (ceil)
    auto in_bounds = [&](int32_t ct, int32_t cq) -> bool {
        const int32_t dr = (cq - b.q0) * dirn, dc = (ct - b.t0) * dirn;
        return dr > 0 && dc > 0 && dc < b.na - 64 && dr < b.nb - (int32_t)relay_w - 64;
    };
    // anchors with s in [s_lo, s_hi), nearest to the lattice line first
    auto scan = [&](long s_lo, long s_hi, long want) -> long {
        const long q_lo = dirn > 0 ? s_lo : -(s_hi - 1), q_hi = dirn > 0 ? s_hi : -s_lo + 1;      // [q_lo, q_hi)
        auto it = std::lower_bound(u.by_q.begin(), u.by_q.end(), q_lo, [&](uint32_t x, long qq) { return (long)u.anchors[x].q < qq; });
        long best = -1, best_d = 0;
        for (; it != u.by_q.end() && (long)u.anchors[*it].q < q_hi; ++it) {
            const Anchor &c = u.anchors[*it];
            if (!in_bounds(c.t, c.q)) continue;
            if (std::labs((long)(c.t - c.q) - (long)(t - q)) > relay_tol) continue;
            const long d = std::labs((long)dirn * c.q - want);
            if (best < 0 || d < best_d || (d == best_d && (long)*it < best)) { best = (long)*it; best_d = d; }
        }
        return best;
    };
    const long near = scan(std::max(s_from, line - relay_s / 4), line + relay_s / 4, line);
    if (near >= 0) return NextPt{1, u.anchors[(size_t)near].t, u.anchors[(size_t)near].q, -1};
    // no anchor at this lattice line: bridge towards the next anchor further down, if there is one
    const long far = scan(line + relay_s / 4, line + relay_gap * relay_s, line);
    if (far < 0) {
        // past the last anchor an alignment may still run on for a while (soft-masked sequence has no seeds): a few
        // more virtual relays straight down the diagonal keep that tail from becoming one long piece
        if ((long)(from_tail + 1) * relay_s > relay_tail_rows) return NextPt{0, 0, 0, -1};
        const int32_t vq = (int32_t)(dirn * line), vt = (int32_t)((long)t + (long)(vq - q));
        if ((long)(vq - q) * dirn <= 0 || !in_bounds(vt, vq)) return NextPt{0, 0, 0, -1};
        return NextPt{1, vt, vq, from_tail + 1};
    }
    const Anchor &c = u.anchors[(size_t)far];
    const int32_t vq = (int32_t)(dirn * line);
    const long span = (long)(c.q - q) * dirn, step = (long)(vq - q) * dirn;
    const long ddiag = (long)(c.t - c.q) - (long)(t - q);
    const int32_t vt = (int32_t)((long)t + (long)(vq - q) + (span > 0 ? ddiag * step / span : 0));
    if (step <= 0 || !in_bounds(vt, vq)) return NextPt{1, c.t, c.q, -1};
    return NextPt{1, vt, vq, -1};
}

// commit + nomination of a round: `pend` = the anchors whose DPs the round runs (empty: the stage is done), shadow_q = the thinning
// neighbourhood that was used (diagnostics)
static void gapped_commit_and_nominate(std::vector<PairJob *> &jobs, std::vector<Unit> &units, const NominateCfg &cfg, int round, std::vector<Pending> &pend,
                                       long &shadow_q) {
    const size_t batch_max = cfg.batch_max;
    const long shadow_q0 = cfg.shadow_q0, shadow_d = cfg.shadow_d, spec_target = cfg.spec_target, relay_s0_env = cfg.relay_s0_env;
    const bool chain_heads = cfg.chain_heads, walls = cfg.walls;
    auto PS = [&](const Unit &u) -> miblast_stats & { return jobs[(size_t)u.pair]->res->stats; };
    // commit what can be committed, then nominate the next speculative batch of every unit
    // commit what can be committed
    for (size_t ui = 0; ui < units.size(); ui++) {
        Unit &u = units[ui];
        while (u.next < u.anchors.size()) {
            const Anchor &a = u.anchors[u.next];
            if (u.cov[u.next]) { PS(u).anchors_skipped++; u.cache.erase(u.next); u.next++; continue; }
            auto it = u.cache.find(u.next);
            if (it == u.cache.end()) break;
            Cached &c = it->second;
            if (walls && c.epoch != u.kept.size()) { u.cache.erase(it); break; }     // ran against fewer walls than the unit has now: evaluate it again
            if (c.accepted && !c.traced) { u.cache.erase(it); break; }    // predicted covered, but is not: evaluate it again
            PS(u).dp_sides += 2; PS(u).dp_cells += c.cells; PS(u).dp_rows += c.rows;
            if (c.accepted) {
                miblast_aln A;
                memset(&A, 0, sizeof A);
                A.strand = u.strand; A.q_contig = u.q_contig; A.t_contig = jobs[(size_t)u.pair]->T->contig_of(a.t);
                A.t_lo = c.t_lo; A.t_hi = c.t_hi; A.q_lo = c.q_lo; A.q_hi = c.q_hi;
                A.score = c.score; A.dmin = c.dmin; A.dmax = c.dmax; A.anchor_t = a.t; A.anchor_q = a.q;
                u.kept_anchor_score.push_back(a.score);
                A.ops_off = (int64_t)u.unit_ops.size(); A.n_ops = (int64_t)c.ops.size();
                u.unit_ops.insert(u.unit_ops.end(), c.ops.begin(), c.ops.end());
                u.kept.push_back(A);
                u.mark(u.cov, A.t_lo, A.t_hi, A.q_lo, A.q_hi, A.dmin, A.dmax);
            }
            u.cache.erase(it);
            u.next++;
        }
        if (walls)                                                       // what ran against fewer walls is stale: drop it now, so that it is nominated again
            for (auto it = u.cache.begin(); it != u.cache.end();) it = it->second.epoch != u.kept.size() ? u.cache.erase(it) : std::next(it);
    }
    // Nomination of the next speculative batch is a pure scheduling heuristic: results never depend on it because
    // anchors are committed strictly in order above.  The first unresolved anchor of every unit is always nominated
    // (progress); the others are thinned: skip what an uncommitted accepted result would cover, and keep at most one
    // new anchor per (diagonal band, query neighbourhood).  The neighbourhood is the smallest of a 4x ladder that
    // keeps the batch within `spec_target` anchors, so an idle GPU is filled with probes in the first round (a
    // single one-sided DP is a row-sequential chain: rounds cost latency, parallel probes cost almost nothing).
    parallel_for(units.size(), [&](size_t ui) {
        Unit &u = units[ui];
        std::fill(u.tent.begin(), u.tent.end(), 0);
        for (const auto &kv : u.cache) {
            const Cached &c = kv.second;
            if (c.accepted) u.mark(u.tent, c.t_lo, c.t_hi, c.q_lo, c.q_hi, c.dmin, c.dmax);
        }
    });
    pend.clear();
    shadow_q = shadow_q0;
    // (the units are independent: every unit's candidates are picked on the worker threads and strung together in unit order)
    auto gather = [&](std::vector<std::vector<Pending>> &per, std::vector<Pending> &out) {
        size_t total = 0;
        for (const auto &v : per) total += v.size();
        out.clear(); out.reserve(total);
        for (const auto &v : per) out.insert(out.end(), v.begin(), v.end());
    };
    // First round: one head per colinear group of anchors (Unit::index_anchors) -- the best anchor of the group that is still
    // open; its relay chain covers the rest of the group.  Whatever is left uncovered after that round (groups that bridge a
    // stretch the extension does not survive) goes through the spatial thinning below, many at a time.
    const long heads_rounds = env_long("MIBLAST_HEAD_ROUNDS", 1);      // 0: groups give heads in the first round only (rounds 1.. thin spatially, as before round 5)
    const bool by_groups = chain_heads && relay_s0_env != 0 && (round == 0 || heads_rounds != 0);
    if (by_groups) {
        std::vector<std::vector<Pending>> per(units.size());
        // (round 5: a head per STRETCH of a group -- head_span rows of q, at most head_max stretches per group.  A group is what ONE alignment is
        //  likely to run through, but the head's extension often ends inside it -- a diverged stretch the y-drop does not survive -- and what
        //  lies behind used to wait for a round of its own: on the phase's 6-pair call 20 anchors, 429 pieces and 2 ms of the critical
        //  path.  Heads on the same alignment share their relay pieces (the lattice does not depend on the head), so a further head costs its
        //  own first piece and nothing else; what it finds that an earlier head has covered is dropped before the traceback.  0: one per group.)
        // In the FIRST round a group gives one head (MIBLAST_HEAD_SPAN0 = 0: stretch heads everywhere cost the 6-pair call of the mammals phase
        // 50 % more cells for nothing, and a 30 Mb chunk pair 20 % of its step); a group that still has open anchors after that IS broken into
        // several alignments, and from the second round on every stretch of it gives a head -- its fragments are found side by side instead
        // of eight duplicates on the nearest one (the spatial thinning of the rounds before round 5: MIBLAST_HEAD_ROUNDS=0); the phase's
        // step 18.0-18.1 -> 17.1-17.5 ms.  (MIBLAST_HEAD_SPAN0 with MIBLAST_HEAD_FEW = 16 groups took the evolverPrimates stand-in from 16.8 to
        // 12.6 ms -- but for the reason the wider bridging distance of the relay lattice now removes at no cost in cells: MIBLAST_RELAY_GAP.)
        long head_span = round == 0 ? env_long("MIBLAST_HEAD_SPAN0", 0) : env_long("MIBLAST_HEAD_SPAN", 16384);
        const long head_max = std::max(1l, env_long("MIBLAST_HEAD_MAX", 8));
        if (round == 0 && head_span > 0) {
            // (stretch heads in the first round only for a call of FEW groups: its launches are as long as their longest piece whatever they hold)
            size_t groups = 0;
            for (const Unit &u : units) groups += u.n_comp;
            if ((long)groups > env_long("MIBLAST_HEAD_FEW", 16)) head_span = 0;
        }
        parallel_for(units.size(), [&](size_t ui) {
            Unit &u = units[ui];
            // a group's q range -> the width of its stretches
            std::vector<int32_t> q_lo(u.n_comp, 0x7fffffff), q_hi(u.n_comp, -0x7fffffff - 1);
            if (head_span > 0)
                for (size_t k = 0; k < u.anchors.size(); k++) { q_lo[u.comp[k]] = std::min(q_lo[u.comp[k]], u.anchors[k].q); q_hi[u.comp[k]] = std::max(q_hi[u.comp[k]], u.anchors[k].q); }
            std::vector<uint8_t> taken((size_t)u.n_comp * (size_t)head_max, 0);
            size_t n_taken = 0;
            for (size_t k = u.next; k < u.anchors.size() && n_taken < batch_max; k++) {
                if (u.cov[k] || u.cache.count(k)) continue;
                const uint32_t c = u.comp[k];
                size_t slot = (size_t)c * (size_t)head_max;
                if (head_span > 0) {
                    const long width = std::max(head_span, ((long)q_hi[c] - (long)q_lo[c]) / head_max + 1);
                    slot += (size_t)std::min<long>(head_max - 1, ((long)u.anchors[k].q - (long)q_lo[c]) / width);
                }
                if (k != u.next && (u.tent[k] || taken[slot])) continue;
                taken[slot] = 1;
                n_taken++;
                per[ui].push_back(Pending{ui, k});
            }
        });
        gather(per, pend);
    }
    for (int level = 0; level < 6 && !by_groups; level++) {
        std::vector<Pending> cand;
        const long sq = std::max(64l, shadow_q0 >> (2 * level));
        std::vector<std::vector<Pending>> per(units.size());
        parallel_for(units.size(), [&](size_t ui) {
            Unit &u = units[ui];
            // taken anchors are bucketed on a (diagonal band, query neighbourhood) grid: the shadow test looks at 3x3 cells
            std::unordered_map<long long, std::vector<Anchor>> grid;
            size_t n_taken = 0;
            auto cell = [&](const Anchor &a, long dd, long dq) -> long long {
                const long gd = ((long)(a.t - a.q) + (1l << 31)) / (shadow_d + 1) + dd, gq = (long)a.q / (sq + 1) + dq;
                return (long long)gd * (1ll << 32) + gq;
            };
            for (size_t k = u.next; k < u.anchors.size() && n_taken < batch_max; k++) {
                if (u.cov[k] || u.cache.count(k)) continue;
                const Anchor &a = u.anchors[k];
                if (k != u.next) {
                    if (u.tent[k]) continue;
                    bool shadowed = false;
                    for (long dd = -1; dd <= 1 && !shadowed; dd++)
                        for (long dq = -1; dq <= 1 && !shadowed; dq++) {
                            auto it = grid.find(cell(a, dd, dq));
                            if (it == grid.end()) continue;
                            for (const Anchor &b : it->second)
                                if (std::labs((long)(a.t - a.q) - (long)(b.t - b.q)) <= shadow_d && std::labs((long)a.q - (long)b.q) <= sq) { shadowed = true; break; }
                        }
                    if (shadowed) continue;
                }
                grid[cell(a, 0, 0)].push_back(a);
                n_taken++;
                per[ui].push_back(Pending{ui, k});
            }
        });
        gather(per, cand);
        if (level > 0 && (long)cand.size() > spec_target) break;      // keep the previous (coarser) level
        pend.swap(cand);
        shadow_q = sq;
        if ((long)pend.size() >= spec_target / 2) break;              // full enough
    }
    if (round > 0 && env_long("MIBLAST_DEBUG", 0) > 2) {
        // why a later round: every nominee with the alignments its unit has committed so far (an anchor of a group whose head's alignment
        // did not reach it, or reached it on another diagonal)
        for (const Pending &pd : pend) {
            const Unit &u = units[pd.unit];
            const Anchor &a = u.anchors[pd.anchor];
            fprintf(stderr, "[miblast]   round %d nominee: unit %zu anchor %zu of %zu (group %u) t %d q %d diagonal %d score %d;", round, pd.unit, pd.anchor, u.anchors.size(), u.comp[pd.anchor], a.t, a.q,
                    a.t - a.q, a.score);
            for (const miblast_aln &A : u.kept)
                if (a.q >= A.q_lo - 20000 && a.q < A.q_hi + 20000)
                    fprintf(stderr, " [kept t %d..%d q %d..%d band %d..%d anchor (%d,%d)%s]", A.t_lo, A.t_hi, A.q_lo, A.q_hi, A.dmin, A.dmax, A.anchor_t, A.anchor_q,
                            a.t >= A.t_lo && a.t < A.t_hi && a.q >= A.q_lo && a.q < A.q_hi ? " IN BOX" : "");
            fprintf(stderr, "\n");
        }
    }
}

// What a stage USED of its arena teaches the factor of its context (round 6; before, a stage that had outgrown its arena taught the size that
// finally held it -- four times the one before -- up to 16 x: 16 GiB arenas for stages whose trace is 1 GiB, 0.7 s of hipMalloc each whenever
// more of them ran side by side than the pool held).  The factor follows 1.5 x the largest used / estimated ratio met, the class of the largest
// arena asked for is remembered (reserve_arenas).
static void arena_learn(Ctx &ctx, size_t raw_estimate, unsigned long long used) {
    if (!raw_estimate || !used) return;
    const unsigned need = (unsigned)std::min<double>(64.0 * 256.0, std::ceil(1.5 * (double)used * 256.0 / (double)raw_estimate));
    std::atomic<unsigned> &sc = arena_scale_of(ctx);
    unsigned cur = sc.load();
    while (need > cur && !sc.compare_exchange_weak(cur, need)) {}
}
static std::atomic<size_t> &arena_class_of(Ctx &ctx);     // (the largest arena a stage of the context has asked for: Workspace::arena_class)

// Before the lanes of a pipelined call start: enough arenas of the context's largest class in the pool for the stages that may run side by side
// (at most `stages`, at most MIBLAST_ARENA_RESERVE_MB -- 32 GiB -- of them in all), so that no lane has to make one in the middle of the call.
static void reserve_arenas(Ctx &ctx, size_t stages) {
    if (getenv("MIBLAST_ARENA_MB")) return;
    const size_t cls = arena_class_of(ctx).load(std::memory_order_relaxed);
    if (!cls || !stages) return;
    const size_t budget = (size_t)std::max(0l, env_long("MIBLAST_ARENA_RESERVE_MB", 32l << 10)) << 20;
    const size_t n_want = std::min(stages, std::max<size_t>(1, budget / cls));
    size_t have = arena_pool().count_free(ctx.device, cls);
    for (; have < n_want; have++) {
        uint8_t *p = nullptr;
        count_device_alloc(); note_device_alloc("trace arena (reserved at the start of a call)", cls);
        if (hipMalloc((void **)&p, cls) != hipSuccess) { (void)hipGetLastError(); break; }
        arena_pool().give(ctx.device, p, cls);
    }
}

// the trace arena of a stage: borrowed from the pool (or made) at a size estimated from the anchors' HSPs
static int acquire_trace_arena(Ctx &ctx, const miblast_params &p, std::vector<PairJob *> &jobs, const std::vector<size_t> *members, size_t n_members,
                               size_t &arena_raw_estimate) {
    Workspace &g = *ctx.ws;
    // estimate: 0.65 B per evaluated cell (codes + row records), ~200 columns per row, rows ~ the anchors' HSP columns, twice for
    // speculation and block granularity; MIBLAST_ARENA_MB fixes the size (tests of the grow-and-retry path)
    size_t want = 256ull << 20;
    for (size_t jk = 0; jk < n_members; jk++) {
        const PairJob *j = jobs[members ? (*members)[jk] : jk];
        for (const miblast_hsp &h : j->res->hsps) want += (size_t)h.len * 260u;
    }
    // (260 B per row is a window of ~200 columns, Cactus's --ydrop=3000 .. 9400; a wider window -- (Y - O) / E columns and a
    //  quarter again -- takes that much more: --ydrop=20000 four times)
    {
        const long win = (p.ydrop > p.gap_open ? (p.ydrop - p.gap_open) / std::max(1, p.gap_extend) : 0) * 5 / 4 + 32;
        if (win > 512) want = (size_t)((double)want * (double)win / 400.0);
    }
    arena_raw_estimate = want;
    // (a stage that had to grow its arena teaches the estimate: the largest ratio of what was needed to what was estimated so far)
    want = (size_t)((double)want * (double)arena_scale_of(ctx).load() / 256.0);
    { size_t cls = (size_t)1 << 30; while (cls < want) cls <<= 1; want = cls; }      // (size classes: 1 GiB, 2 GiB, ... -- the pool's arenas are reused, not multiplied)
    if (getenv("MIBLAST_ARENA_MB")) want = (size_t)env_long("MIBLAST_ARENA_MB", 4096) << 20;
    else { std::atomic<size_t> &cm = arena_class_of(ctx); size_t cur = cm.load(); while (want > cur && !cm.compare_exchange_weak(cur, want)) {} }
    // (MIBLAST_ARENA_MB: an arena of exactly that size, whatever the pool holds -- the tests' way into the grow-and-retry path)
    if (getenv("MIBLAST_ARENA_MB") || !arena_pool().take(ctx.device, want, g.arena.p, g.arena.n) || g.arena.n < want) {
        arena_pool().give(ctx.device, g.arena.p, g.arena.n);        // (too small a one: it stays in the pool for a lighter stage)
        g.arena.p = nullptr; g.arena.n = 0;
        // (several lanes size their arenas at the same time: the look at the free memory and the allocation are one step, and an
        //  allocation that does not fit after all -- another context of the process took the room -- gives the pool's idle arenas back
        //  to the runtime and tries again with half, down to 256 MiB; the stage grows its arena later if the trace needs more)
        static std::mutex arena_alloc_mutex;
        std::lock_guard<std::mutex> lk(arena_alloc_mutex);
        size_t free_b = 0, total_b = 0;
        MB_HIP(hipMemGetInfo(&free_b, &total_b));
        want = std::min<size_t>(want, free_b > ((size_t)4 << 30) ? free_b - ((size_t)2 << 30) : free_b / 2);
        count_device_alloc(); note_device_alloc("trace arena", want);
        while (hipMalloc((void **)&g.arena.p, want) != hipSuccess) {
            (void)hipGetLastError();
            g.arena.p = nullptr;
            arena_pool().trim(ctx.device);
            if (want <= ((size_t)256 << 20)) { set_error("trace arena does not fit in device memory"); return MIBLAST_ELIMIT; }
            want = std::max<size_t>((size_t)256 << 20, want / 2);
        }
        g.arena.n = want;
    }
    g.arena_next.ensure(1);
    return MIBLAST_OK;
}

// the round's trace did not fit: a larger arena (the round is repeated)
static int grow_trace_arena(Ctx &ctx, size_t arena_raw_estimate) {
    Workspace &g = *ctx.ws;
    size_t free_b = 0, total_b = 0;
    MB_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t room = free_b + g.arena.n > (2ull << 30) ? free_b + g.arena.n - (2ull << 30) : 0;   // leave 2 GiB for the rest
    size_t bigger = std::min(g.arena.n * 4, room);      // few retries: every retry repeats the round
    if (bigger <= g.arena.n) { set_error("trace arena does not fit in device memory"); return MIBLAST_ELIMIT; }
    // (the one that was too small goes back to the pool -- or, when the larger one needs the room, to the runtime; a free one of the
    //  size wanted is taken if there is one)
    const size_t old_n = g.arena.n;
    if (bigger + (2ull << 30) > free_b) { count_device_alloc(); note_device_alloc("(release) trace arena before growing", g.arena.n); (void)hipFree(g.arena.p); arena_pool().trim(ctx.device); }
    else arena_pool().give(ctx.device, g.arena.p, g.arena.n);
    g.arena.p = nullptr; g.arena.n = 0;
    if (!arena_pool().take(ctx.device, bigger, g.arena.p, g.arena.n) || g.arena.n < bigger) {
        arena_pool().give(ctx.device, g.arena.p, g.arena.n);
        g.arena.p = nullptr; g.arena.n = 0;
        count_device_alloc(); note_device_alloc("trace arena (grown)", bigger);
        if (hipMalloc((void **)&g.arena.p, bigger) != hipSuccess) {
            (void)hipGetLastError();
            g.arena.p = nullptr;
            arena_pool().trim(ctx.device);               // (what other stages left free in the meantime)
            MB_HIP(hipMemGetInfo(&free_b, &total_b));
            bigger = std::min<size_t>(bigger, free_b > ((size_t)2 << 30) ? free_b - ((size_t)2 << 30) : (size_t)0);
            if (bigger <= old_n || hipMalloc((void **)&g.arena.p, bigger) != hipSuccess) {
                (void)hipGetLastError();
                g.arena.p = nullptr;
                set_error("trace arena does not fit in device memory");
                return MIBLAST_ELIMIT;
            }
        }
        g.arena.n = bigger;
    }
    (void)arena_raw_estimate;                                  // (the factor learns from what the stage USED, at its end: arena_learn)
    return MIBLAST_OK;
}

// results of a round's DPs: score and box of every nominated anchor into its unit's cache, then the traceback of what reaches
// --gappedthresh, in two phases (see below), and the merge of the walkers' runs into the alignment's ops
static int gapped_finish_round(Ctx &ctx, const miblast_params &p, std::vector<PairJob *> &jobs, std::vector<Unit> &units, const std::vector<Pending> &pend,
                               const RoundDp &rd, miblast_stats &st, bool debug) {
    Workspace &g = *ctx.ws;
    hipStream_t s = ctx.stream;
    const std::vector<SideRun> &sides = rd.sides;
    const std::vector<Piece> &pieces = rd.pieces;
    const std::vector<DpProb> &probs = rd.probs;
    const std::vector<DpOut> &outs = rd.outs;
    // ---- results of the round; traceback of the anchors reaching --gappedthresh -----------------------------
    // Speculation produces duplicates: anchors whose DP found an alignment that an earlier anchor of the unit will have
    // committed by the time their turn comes (they are then skipped as covered).  Tracing and merging those is wasted, so
    // the traceback runs in two phases: first the results that lie in no earlier accepted box of their unit (nothing can
    // cover them: they will be committed), then -- their diagonal bands now known -- the rest minus what they cover.
    std::vector<Cached *> cres(pend.size(), nullptr);
    for (size_t k = 0; k < pend.size(); k++) {
        Unit &u = units[pend[k].unit];
        Cached c;
        const SideRun &R = sides[2 * k], &L = sides[2 * k + 1];
        const Anchor &a = u.anchors[pend[k].anchor];
        c.score = R.gbest + L.gbest;
        c.cells = R.acc_cells + L.acc_cells; c.rows = R.acc_rows + L.acc_rows;
        c.t_lo = a.t - L.gbj; c.t_hi = a.t + R.gbj; c.q_lo = a.q - L.gbi; c.q_hi = a.q + R.gbi;
        c.accepted = c.score >= p.gappedthresh;
        c.traced = false;
        c.epoch = u.kept.size();
        c.dmin = 0x7fffffff; c.dmax = -0x7fffffff - 1;          // set by the merge; until then covers nothing
        cres[k] = &u.cache.emplace(pend[k].anchor, std::move(c)).first->second;      // (references into an unordered_map stay valid)
    }
    auto trace = [&](const std::vector<size_t> &acc) -> int {
    std::vector<TbSide> tbs;
    std::vector<TbWalk> tbw;
    uint64_t ooff = 0, roff = 0, soff = 0;          // run slots, row records (3 x u32 each), segments
    for (size_t k : acc) {
            for (int side = 0; side < 2; side++) {
                const SideRun &sd = sides[2 * k + (size_t)side];
                TbSide ts;
                memset(&ts, 0, sizeof ts);
                ts.first_walk = (int32_t)tbw.size();
                // one walker per piece of the chain, from the piece that holds the best cell back to the head
                size_t at = 0;
                while (at < sd.chain.size() && sd.chain[at] != sd.best_piece) at++;
                uint64_t side_slots = 0;
                for (size_t x = at + 1; x-- > 0;) {
                    const int pc = sd.chain[x];
                    const Piece &pp = pieces[(size_t)pc];
                    TbWalk w;
                    memset(&w, 0, sizeof w);
                    w.row_off = probs[(size_t)pc].row_off; w.row_lo = pp.row_lo; w.floor = x > 0 ? sd.chain_floor[x] : -1;
                    if (x == at) { w.si = sd.gbi - (pp.oq - sd.base.q0) * pp.dir; w.sj = sd.gbj - (pp.ot - sd.base.t0) * pp.dir; }
                    else { w.si = pp.stop_row; w.sj = outs[(size_t)pc].exit_j; }      // guess: the best cell of the piece's last row
                    if (x > 0) {
                        const Piece &pv = pieces[(size_t)sd.chain[x - 1]];
                        w.dr = (pp.oq - pv.oq) * pp.dir; w.dc = (pp.ot - pv.ot) * pp.dir;
                    }
                    const uint64_t rows = (uint64_t)(w.si - w.floor);
                    const uint64_t slots = 2 * rows + 2 * (uint64_t)kLdsRowCap + 8;      // a run per step at worst
                    w.ops_off = ooff; ooff += slots; side_slots += slots;
                    w.rec_off = roff; roff += 3 * rows;
                    tbw.push_back(w);
                }
                ts.n_walks = (int32_t)tbw.size() - ts.first_walk;
                ts.jops_off = ooff; ooff += side_slots;                  // the join walk can at worst repeat every walk
                for (size_t x = (size_t)ts.first_walk; x < tbw.size(); x++) {          // ... each piece's in its own share
                    const unsigned long long jo = x == (size_t)ts.first_walk ? ~0ull : ts.jops_off + (tbw[x].ops_off - tbw[(size_t)ts.first_walk].ops_off);
                    memcpy(tbw[x].pad, &jo, 8);
                }
                ts.seg_off = soff; soff += 2 * (uint64_t)ts.n_walks + 1;
                tbs.push_back(ts);
            }
                }
    const double t_tb0 = now_s();
    std::vector<unsigned long long> coff;           // first packed run of every side (+ total)
    if (!acc.empty()) {
        // walks, sides and segments lie one behind the other: walks + sides go up in one copy, sides + segments come back in one
        const size_t at_sides = Stager::behind(tbw.size() * sizeof(TbWalk)), at_segs = at_sides + Stager::behind(tbs.size() * sizeof(TbSide));
        const size_t at_joins = at_segs + Stager::behind(((size_t)soff + 1) * sizeof(TbSeg));
        g.tb_blk.ensure(at_joins + tbw.size() * sizeof(TbJoin) + 256);
        TbWalk *const d_walks = (TbWalk *)g.tb_blk.p;
        TbSide *const d_sides = (TbSide *)(g.tb_blk.p + at_sides);
        TbSeg *const d_segs = (TbSeg *)(g.tb_blk.p + at_segs);
        TbJoin *const d_joins = (TbJoin *)(g.tb_blk.p + at_joins);
        g.ops.ensure((size_t)ooff + 64); g.recs.ensure((size_t)roff + 64);
        g.stage.h2d2(g.tb_blk.p, tbw.data(), tbw.size() * sizeof(TbWalk), tbs.data(), tbs.size() * sizeof(TbSide), s);
        launch_trace_walk(d_walks, (int)tbw.size(), g.arena.p, (unsigned long long)g.arena.n, g.rowdir.p, g.ops.p, g.recs.p, s);
        // the join walks of all pieces at once, from predicted entries (MIBLAST_TRACE_PREJOIN=0: none, the sides walk themselves;
        // 2: every other prediction is made wrong on purpose -- both for the tests)
        const long prejoin = env_long("MIBLAST_TRACE_PREJOIN", 1);
        if (prejoin) launch_trace_prejoin(d_walks, (int)tbw.size(), d_joins, g.arena.p, (unsigned long long)g.arena.n, g.rowdir.p, g.ops.p, g.recs.p, prejoin == 2, s);
        launch_trace_join(d_sides, (int)tbs.size(), d_walks, d_segs, g.arena.p, (unsigned long long)g.arena.n, g.rowdir.p,
                          g.ops.p, g.recs.p, prejoin ? d_joins : nullptr, s);
        std::vector<TbSeg> segs((size_t)soff + 1);
        g.stage.d2h2(tbs.data(), tbs.size() * sizeof(TbSide), segs.data(), (size_t)soff * sizeof(TbSeg), d_sides, s);
        MB_HIP(hipStreamSynchronize(s));
        g.stage.done();
        // only the run slots actually used travel: the segments are packed on the device in walk order, then one copy
        std::vector<TbSeg> flat;
        std::vector<unsigned long long> dst;
        coff.assign(tbs.size() + 1, 0);
        unsigned long long ctot = 0;
        for (size_t x = 0; x < tbs.size(); x++) {
            coff[x] = ctot;
            for (int q = 0; q < tbs[x].n_segs; q++) {
                const TbSeg &sg = segs[(size_t)tbs[x].seg_off + (size_t)q];
                flat.push_back(sg); dst.push_back(ctot); ctot += (unsigned long long)sg.n_runs;
            }
        }
        coff[tbs.size()] = ctot;
        const size_t at_dst = Stager::behind(flat.size() * sizeof(TbSeg));
        g.tb_blk.ensure(at_dst + (dst.size() + 1) * sizeof(unsigned long long) + 256);
        g.ops_packed.ensure((size_t)ctot + 64); g.hops.ensure((size_t)ctot + 64);
        g.stage.h2d2(g.tb_blk.p, flat.data(), flat.size() * sizeof(TbSeg), dst.data(), dst.size() * sizeof(unsigned long long), s);
        launch_pack_segs((const TbSeg *)g.tb_blk.p, (const unsigned long long *)(g.tb_blk.p + at_dst), (int)flat.size(), g.ops.p, g.ops_packed.p, s);
        if (ctot) MB_HIP(hipMemcpyAsync(g.hops.p, g.ops_packed.p, (size_t)ctot * 4, hipMemcpyDeviceToHost, s));
        MB_HIP(hipStreamSynchronize(s));
        g.stage.done();
        const uint32_t *hops = g.hops.p;
        st.t_traceback_ms += (now_s() - t_tb0) * 1e3;
        if (debug) fprintf(stderr, "[miblast]   traceback kernel + copies: %.2f ms (%zu sides, %llu run slots)\n", (now_s() - t_tb0) * 1e3, tbs.size(), (unsigned long long)ooff);
        if (debug) fprintf(stderr, "[miblast]   %zu walkers, %zu segments, %llu runs\n", tbw.size(), flat.size(), ctot);
        const double t_mg0 = now_s();
        // merge the two sides into a forward run-length '=XID' string (left walk-back order is already
        // forward, the right one is reversed), split aligned pairs into '=' / 'X', track the diagonal band
        std::vector<Cached *> cptr(acc.size());
        for (size_t x = 0; x < acc.size(); x++) cptr[x] = &units[pend[acc[x]].unit].cache[pend[acc[x]].anchor];   // no map mutation inside the workers
        // A long alignment is merged in chunks of runs on several threads: the (t, q) position of every chunk start is
        // a cheap prefix over the run lengths; the chunks' run-length strings are stitched afterwards.
        struct MergeTask { size_t x, r0, r1; int64_t tt, qq; std::vector<uint32_t> ops; int32_t dmin, dmax; };
        std::vector<MergeTask> tasks;
        std::vector<std::pair<size_t, size_t>> task_range(acc.size());
        std::vector<std::pair<int64_t, int64_t>> reached(acc.size());
        const size_t kRunsPerTask = 2048;
        auto run_at = [&](size_t x, size_t run) -> uint32_t {
            const uint32_t *Rops = hops + coff[2 * x], *Lops = hops + coff[2 * x + 1];
            const size_t nR = (size_t)(coff[2 * x + 1] - coff[2 * x]), nL = (size_t)(coff[2 * x + 2] - coff[2 * x + 1]);
            return run < nL ? Lops[run] : Rops[nR - 1 - (run - nL)];
        };
        // (the prefix walk of every alignment on the worker threads, the task lists strung together afterwards)
        std::vector<std::vector<MergeTask>> per(acc.size());
        parallel_for(acc.size(), [&](size_t x) {
            const Cached &c = *cptr[x];
            const size_t nruns = (size_t)(coff[2 * x + 2] - coff[2 * x]);
            int64_t tt = c.t_lo, qq = c.q_lo;
            for (size_t r0 = 0; r0 < nruns || r0 == 0; r0 += kRunsPerTask) {
                const size_t r1 = std::min(nruns, r0 + kRunsPerTask);
                per[x].push_back(MergeTask{x, r0, r1, tt, qq, {}, 0x7fffffff, -0x7fffffff - 1});
                for (size_t r = r0; r < r1; r++) {
                    const uint32_t e = run_at(x, r), o = e & 3u, len = e >> 2;
                    if (o == 0) { tt += len; qq += len; } else if (o == 2) qq += len; else tt += len;
                }
                if (r1 >= nruns) break;
            }
            reached[x] = {tt, qq};
        });
        for (size_t x = 0; x < acc.size(); x++) {
            task_range[x].first = tasks.size();
            for (MergeTask &t : per[x]) tasks.push_back(std::move(t));
            task_range[x].second = tasks.size();
        }
        const double t_mg1 = now_s();
        parallel_for(tasks.size(), [&](size_t ti) {
            MergeTask &t = tasks[ti];
            const Unit &u = units[pend[acc[t.x]].unit];
            const uint8_t *tc_h = jobs[(size_t)u.pair]->tc_h;
            const uint8_t *qc = jobs[(size_t)u.pair]->qc_h[u.strand];
            int64_t tt = t.tt, qq = t.qq;
            uint32_t cur_op = 0, cur_len = 0;
            std::vector<uint32_t> out;                          // thread-local until the end: no false sharing on the task array
            out.reserve(4 * (t.r1 - t.r0) + 16);
            int32_t dmin = 0x7fffffff, dmax = -0x7fffffff - 1;
            auto push = [&](uint32_t op, uint32_t len) {
                if (cur_len && op == cur_op) cur_len += len;
                else { if (cur_len) out.push_back((cur_len << 2) | cur_op); cur_op = op; cur_len = len; }
            };
            for (size_t run = t.r0; run < t.r1; run++) {
                const uint32_t e = run_at(t.x, run);
                const uint32_t o = e & 3u, len = e >> 2;
                if (len == 0) continue;                         // a splice that fell on a run boundary
                if (o == 0) {
                    const int32_t d = (int32_t)(tt - qq);
                    dmin = std::min(dmin, d); dmax = std::max(dmax, d);
                    const uint8_t *tp = tc_h + tt, *qp = qc + qq;
                    // '=' iff both bases are the same of A, C, G, T.  Eight columns per step: the matching columns of the
                    // eight as a bit mask, then whole runs of equal bits at a time (a run of '=' is dozens of columns long)
                    uint32_t m = 0;
                    for (; m + 8 <= len; m += 8) {
                        unsigned bits = match_bits8(tp + m, qp + m), left = 8;
                        while (left) {
                            const unsigned one = bits & 1u;
                            const unsigned n = std::min(left, (unsigned)__builtin_ctz((one ? ~bits : bits) | 0x100u));
                            push(one ? 0u : 1u, n);
                            bits >>= n; left -= n;
                        }
                    }
                    for (; m < len; m++) {
                        const unsigned a = tp[m] & 7u, b = qp[m] & 7u;
                        push((a < 4u && a == b) ? 0u : 1u, 1);
                    }
                    tt += len; qq += len;
                } else if (o == 2) { push(2, len); qq += len; }
                else { push(3, len); tt += len; }
            }
            if (cur_len) out.push_back((cur_len << 2) | cur_op);
            t.ops.swap(out); t.dmin = dmin; t.dmax = dmax;
        });
        const double t_mg2 = now_s();
        if (debug) fprintf(stderr, "[miblast]   merge: prefix %.2f ms, %zu tasks %.2f ms (hw threads %u)\n", (t_mg1 - t_mg0) * 1e3, tasks.size(), (t_mg2 - t_mg1) * 1e3, std::thread::hardware_concurrency());
        std::atomic<int> bad{0};
        parallel_for(acc.size(), [&](size_t x) {
            Cached &c = *cptr[x];
            size_t total = 0;
            for (size_t ti = task_range[x].first; ti < task_range[x].second; ti++) total += tasks[ti].ops.size();
            c.ops.reserve(total);
            int32_t dmin = 0x7fffffff, dmax = -0x7fffffff - 1;
            for (size_t ti = task_range[x].first; ti < task_range[x].second; ti++) {
                const MergeTask &t = tasks[ti];
                dmin = std::min(dmin, t.dmin); dmax = std::max(dmax, t.dmax);
                size_t from = 0;
                if (!c.ops.empty() && !t.ops.empty() && ((c.ops.back() ^ t.ops[0]) & 3u) == 0) { c.ops.back() += t.ops[0] & ~3u; from = 1; }   // same op across the seam
                c.ops.insert(c.ops.end(), t.ops.begin() + (long)from, t.ops.end());
            }
            c.dmin = dmin; c.dmax = dmax;
            if (reached[x].first != c.t_hi || reached[x].second != c.q_hi) {
                const size_t k = acc[x];
                if (!bad++ && debug) {
                    const SideRun &R = sides[2 * k], &L = sides[2 * k + 1];
                    fprintf(stderr, "[miblast] span error: anchor %zu box t %d..%d q %d..%d reached t %lld q %lld; R: best %d at (%d,%d) chain %zu; L: best %d at (%d,%d) chain %zu\n",
                            k, c.t_lo, c.t_hi, c.q_lo, c.q_hi, (long long)reached[x].first, (long long)reached[x].second, R.gbest, R.gbi, R.gbj, R.chain.size(),
                            L.gbest, L.gbi, L.gbj, L.chain.size());
                }
            }
        });
        if (bad) { set_error("internal: traceback does not span the alignment box"); return (int)MIBLAST_EHIP; }
        for (Cached *c : cptr) c->traced = true;
        st.t_merge_ms += (now_s() - t_mg0) * 1e3;
        if (debug) fprintf(stderr, "[miblast]   host merge: %.2f ms\n", (now_s() - t_mg0) * 1e3);
    }
    return (int)MIBLAST_OK;
    };
    {
        // commit order inside a unit = anchor index; group this round's accepted results by unit
        std::unordered_map<size_t, std::vector<size_t>> by_unit;
        for (size_t k = 0; k < pend.size(); k++) if (cres[k]->accepted) by_unit[pend[k].unit].push_back(k);
        std::vector<size_t> first, later;
        struct Box { size_t anchor; const Cached *c; };
        std::unordered_map<size_t, std::vector<Box>> boxes;     // per unit: accepted results, old (traced in earlier rounds) and new
        for (auto &kv : by_unit) {
            Unit &u = units[kv.first];
            std::vector<size_t> &ks = kv.second;
            std::sort(ks.begin(), ks.end(), [&](size_t x, size_t y) { return pend[x].anchor < pend[y].anchor; });
            std::vector<Box> &bx = boxes[kv.first];
            for (const auto &e : u.cache) if (e.second.accepted) bx.push_back(Box{e.first, &e.second});
            for (size_t k : ks) {
                const Anchor &a = u.anchors[pend[k].anchor];
                bool inside = false;
                for (const Box &b : bx)
                    if (b.anchor < pend[k].anchor && a.t >= b.c->t_lo && a.t < b.c->t_hi && a.q >= b.c->q_lo && a.q < b.c->q_hi) { inside = true; break; }
                (inside ? later : first).push_back(k);
            }
        }
        std::sort(first.begin(), first.end());
        int rc = trace(first);
        if (rc != MIBLAST_OK) return rc;
        std::vector<size_t> second;
        for (size_t k : later) {
            const Unit &u = units[pend[k].unit];
            const Anchor &a = u.anchors[pend[k].anchor];
            const int32_t d = a.t - a.q;
            bool covered = false;
            for (const Box &b : boxes[pend[k].unit])
                if (b.c->traced && b.anchor < pend[k].anchor && a.t >= b.c->t_lo && a.t < b.c->t_hi && a.q >= b.c->q_lo && a.q < b.c->q_hi &&
                    d >= b.c->dmin && d <= b.c->dmax) { covered = true; break; }
            if (!covered) second.push_back(k);
        }
        std::sort(second.begin(), second.end());
        rc = trace(second);
        if (rc != MIBLAST_OK) return rc;
    }
    return MIBLAST_OK;
}

// score-ordered gapped extension of every unit of every pair of the batch: all speculative one-sided DPs of a round
// share one k_ydrop launch, so several chunk pairs fill the GPU together
// `members`: the pairs whose units are in `units` (nullptr: all pairs of `jobs`) -- a large call runs the gapped stages of two
// groups of its pairs side by side, each on a stream and workspace of its own (align_pairs)
// `in_flight`: how many gapped stages of this size share the GPU at a time (0: all groups of the call, jobs / members) -- the relay regime follows it
static int gapped_phase(Ctx &ctx, const miblast_params &p, std::vector<PairJob *> &jobs, std::vector<Unit> &units, const std::vector<size_t> *members = nullptr, size_t in_flight = 0) {
    const size_t batch_max = (size_t)env_long("MIBLAST_GAPPED_BATCH_MAX", 4096);
    const long shadow_q0 = env_long("MIBLAST_SHADOW_Q", 1 << 16);        // spatial thinning of speculative anchors
    // anchors per round the thinning aims at: a lone pair is probed generously (an idle GPU, rounds cost latency); in a batch every
    // pair brings its own heads and further heads on the same alignment only duplicate relay pieces
    const size_t n_members = members ? members->size() : jobs.size();
    const long spec_target = env_long("MIBLAST_SPEC_TARGET", jobs.size() > 1 ? 6 : 24) * (long)std::max<size_t>(1, n_members);
    const long shadow_d = env_long("MIBLAST_SHADOW_D", 2 * (p.ydrop / std::max(1, p.gap_extend)) + 64);
    const unsigned kBlk = 64u << 10, kBlkWide = 4u << 20;
    // relay hand-over (see the DP section below): first stop, relay spacing, warm-up rows, diagonal tolerance, relays per side
    const long relay_s0_env = env_long("MIBLAST_RELAY_S0", -1), relay_s_env = env_long("MIBLAST_RELAY_S", 0);
    long relay_s0 = relay_s0_env >= 0 ? relay_s0_env : 256;
    long relay_s = std::max(256l, relay_s_env > 0 ? relay_s_env : 1280l);
    const long relay_w_env = env_long("MIBLAST_RELAY_W", 0), relay_tol = env_long("MIBLAST_RELAY_TOL", 512);
    long relay_w = std::max(64l, relay_w_env > 0 ? relay_w_env : 192l);
    // (relay_gap: lattice steps a chain bridges without an anchor -- virtual relays on the line to the next anchor.  8 until round 5; 16: the
    //  evolverPrimates stand-in -- option set "one": every second position, no transitions, half the sequence soft-masked -- has stretches of
    //  5 000 - 10 000 rows without an anchor inside its alignments; a chain that ended there left the rest of the side to a planting of
    //  its own after the launch, three times per round: 16.8 -> 10.6 ms per phase, the cells evaluated unchanged)
    const long relay_max = env_long("MIBLAST_RELAY_MAX", 4096), relay_gap = std::max(1l, env_long("MIBLAST_RELAY_GAP", 16)),
               relay_tail_rows = env_long("MIBLAST_RELAY_TAIL_ROWS", 4096);     // how far past the last anchor virtual relays are planted
    // No piece of a relayed side runs unbounded: where a chain ends (no anchor ahead within the bridging distance -- a long
    // soft-masked stretch has no seeds) the piece stops after a few lattice steps and, if the extension is still alive, the side
    // looks for relays again from there (make_cont).  Without this one alignment running through a 10 kb seedless stretch of a
    // 30 Mb chunk pair kept a single wave busy for a million rows (1 s).
    const long relay_end_steps = std::max(1l, env_long("MIBLAST_RELAY_END_STEPS", 4));
    const long relay_force_reject = env_long("MIBLAST_RELAY_FORCE_REJECT", 0);   // test knob: reject every n-th hand-over
    const bool chain_heads = env_long("MIBLAST_CHAIN_HEADS", 1) != 0;            // first round: one speculative head per colinear group of anchors
    const bool relay_ckpt = env_long("MIBLAST_RELAY_CKPT", 1) != 0;             // retry a rejected hand-over at the relay's later entry snapshots
    // the hand-over inside the launch (mb_ydrop2.h): a piece checks its hand-over itself and, rejected, goes on in the same wave.  0: off (every
    // rejected hand-over is a continuation piece in a launch of its own, as before round 5); MIBLAST_RELAY_INLINE_ROWS: how far past its
    // planned stop row a piece may go on its own (default: the aimed relay's last entry snapshot and three relays further)
    const bool relay_inline = env_long("MIBLAST_RELAY_INLINE", 1) != 0;
    const long relay_inline_rows_env = env_long("MIBLAST_RELAY_INLINE_ROWS", -1);
    const long relay_inline_force = env_long("MIBLAST_RELAY_INLINE_FORCE_REJECT", 0);   // test knob (mb_ydrop2.h, force_mod)
    // DP kernel of the pieces: the typical window is (Y-O)/E columns to the right of the path and about a quarter of that to
    // the left; windows that outgrow the lanes make the piece overflow and it is rerun with the next wider kernel
    const long win_typ = (p.ydrop > p.gap_open ? (p.ydrop - p.gap_open) / std::max(1, p.gap_extend) : 0) * 5 / 4 + 32;
    const long dp_kernel_env = env_long("MIBLAST_DP_KERNEL", 0);         // 4 / 8: columns per lane of the one-wave kernel, 100: 4-wave LDS kernel
    const bool debug = env_long("MIBLAST_DEBUG", 0) != 0;
    // walls (miblast_params.walls): base pairs on the path of an earlier alignment of the unit are dead cells of
    // later DPs.  The DPs of a round run against the alignments their unit has COMMITTED by then (the 4-wave kernel's WALLS variant);
    // a cached result is only good while its unit commits nothing new -- the commit loop drops what went stale and the anchors are
    // evaluated again, against the new walls.  Relays, continuations and the traceback work as they are: every piece of a round sees
    // the same walls, so equal states still evolve identically.
    const bool walls = p.walls != 0;
    Workspace &g = *ctx.ws;
    hipStream_t s = ctx.stream;
    miblast_stats st;                        // launch-level counters shared by all pairs of the batch
    memset(&st, 0, sizeof st);
    const double t_g0 = now_s();
    size_t arena_raw_estimate = 0;
    unsigned long long arena_used = 0;                                    // the most a round of this stage wrote into its arena
    struct ArenaLoan {                                                   // back to the pool however the stage ends
        Workspace &g; int device;
        ~ArenaLoan() { arena_pool().give(device, g.arena.p, g.arena.n); g.arena.p = nullptr; g.arena.n = 0; }
    } arena_loan{g, ctx.device};
    if (!units.empty()) {
        const int rc_arena = acquire_trace_arena(ctx, p, jobs, members, n_members, arena_raw_estimate);
        if (rc_arena != MIBLAST_OK) return rc_arena;
    }
    const NominateCfg nominate_cfg{batch_max, shadow_q0, shadow_d, spec_target, relay_s0_env, chain_heads, walls};
    for (int round = 0;; round++) {
        double tm[6] = {0, 0, 0, 0, 0, 0};                 // debug: where a round's host time goes
        double tm_t = now_s();
        auto lap = [&](int k) { const double n = now_s(); tm[k] += n - tm_t; tm_t = n; };
        std::vector<Pending> pend;
        long shadow_q = shadow_q0;
        gapped_commit_and_nominate(jobs, units, nominate_cfg, round, pend, shadow_q);
        if (pend.empty()) break;
        st.gapped_rounds++;
        lap(0);

        // ---- the one-sided DPs of the round, trace stored in the arena -----------------------------------
        // A side (anchor, direction) is evaluated as a chain of PIECES (k_ydrop problems).  A one-sided DP is a
        // row-sequential chain, so a 250 000-row alignment would keep one workgroup busy for a quarter of a second
        // while the rest of the GPU idles.  Instead the first piece of a side stops after `relay_s0` rows; a side that
        // is still alive there is continued from its exit snapshot AND relayed: fresh DPs are started at downstream
        // anchors of the same unit -- the ungapped HSPs a long alignment passes through -- and each stops `relay_w`
        // rows after the origin of the next one.  All of them run concurrently.  A hand-over is accepted only if the
        // state of the upstream piece after the hand-over row equals the relay's state after the same row (k_verify:
        // same window, every value that can still matter equal up to one constant).  The recurrence is invariant
        // under adding a constant, so from that row on both compute the same rows and the relay's rows simply ARE the
        // rows of the sequential DP.  A rejected hand-over costs nothing but time: the upstream piece is continued
        // from its snapshot to the relay after that.  Relays sit on a lattice of the unit's anchors that does not
        // depend on the side (next_relay), so every side running through the same alignment shares them.  Results
        // (score, end cell, trace, cell and row counts) never depend on where relays start or whether they are accepted.
        const int nsides = (int)pend.size() * 2;
        // few sides (one chunk pair): short pieces, the longest one sets the time.  Many sides (batched pairs): the GPU is full
        // anyway, longer pieces waste less on warm-up overlap.
        // Regimes (measured on the bench workloads, scripts/gpu_r02_sched2.sh, gpu_r02_s2y.sh): a lone pair (few sides) -- 640-row pieces; a
        // batch of a few pairs (the evolver phase: hundreds of sides) -- 768-row pieces (round 3: 512 -> 768 takes the speculation factor of the phase
        // from 1.51 to 1.39 -- a relay's warm-up rows are evaluated twice -- and ~0.8 ms off a step), still planted together with the heads: short
        // pieces balance the launch and a rejected hand-over costs one short retry; thousands of sides -- the GPU is full anyway,
        // long pieces waste less on warm-up and relays are only spent on sides that survive relay_s0 rows (16 x 1 Mb pairs, ~600
        // sides: 66 ms per call against 75 ms with the short pieces).
        // (a group of a split call shares the GPU with the other group: the regime follows the sides of the whole call)
        const long nsides_call = in_flight ? (long)nsides * (long)in_flight : (long)nsides * (long)jobs.size() / (long)std::max<size_t>(1, n_members);
        const bool crowd = nsides_call > env_long("MIBLAST_CROWD_SIDES", 400);
        // (a handful of sides -- a trimmed outgroup call of the phase: every launch runs at lone-wave speed and most hand-overs are
        //  retried; 448-row pieces: 19 -> 17 DP launches and 10.2 -> 9.4 ms of DP kernel time per phase; 320 and 256 need more launches)
        const long relay_s_tiny = env_long("MIBLAST_RELAY_S_TINY", 448);
        if (relay_s_env <= 0) relay_s = crowd ? env_long("MIBLAST_RELAY_S_CROWD", 2048) : nsides_call > 96 ? env_long("MIBLAST_RELAY_S_MID", 768) : nsides_call > 16 ? env_long("MIBLAST_RELAY_S_FEW", 640) : relay_s_tiny;
        if (relay_s0_env < 0) relay_s0 = crowd ? 256 : 64;
        // (a handful of sides: until round 5 384 warm-up rows -- most hand-overs of such a call are rejected after 128, and a retry was a
        //  launch of its own.  With the hand-over inside the launch a rejected piece goes on to the relay's snapshots after 256 and 512 rows
        //  by itself, and the launch is as long as its longest piece: 128 again -- the trimmed levels' launches 1.05 -> ~0.6 ms, the
        //  phase 17.3-17.8 -> 17.0 ms, the evolverPrimates stand-in 10.4 -> 9.8)
        if (relay_w_env <= 0) relay_w = crowd ? env_long("MIBLAST_RELAY_W_CROWD", 192) : nsides_call > 16 ? env_long("MIBLAST_RELAY_W_MID", 128) : env_long("MIBLAST_RELAY_W_TINY", 128);
        const long plant_env = env_long("MIBLAST_RELAY_PLANT_AT_ONCE", 1);            // 0: never, 1: unless thousands of sides are in flight, 2: always
        const bool plant_at_once = plant_env == 2 || (!crowd && plant_env != 0);
        // one wave per piece; 4 columns per lane when the GPU is saturated and the typical window fits 256 columns (fewest
        // instructions per row; the few pieces that outgrow the lanes are rerun), else 8 columns per lane
        const int dp_kernel = walls ? kDpLds : dp_kernel_env ? (int)dp_kernel_env : win_typ > 448 ? kDpLds : win_typ <= 224 ? kDpWave2x4 : kDpWave8;
        // rows a piece with a stop row may go on past it inside its launch (k_ydrop2 only); its row-chunk directory is reserved that far
        const long inline_rows = dp_kernel == kDpWave2x4 && relay_inline && relay_s0 > 0 ? (relay_inline_rows_env >= 0 ? relay_inline_rows_env : 4 * relay_w + 3 * relay_s) : 0;
        auto reach_of = [&](int32_t stop_row, int32_t nb) -> int32_t { return stop_row > 0 ? (int32_t)std::min<long>((long)nb, (long)stop_row + inline_rows) : nb; };
        // walls of the round: the gap-free runs of every unit's committed alignments, in the strand's coordinates
        std::vector<int32_t> wall_segs, wall_alns;                          // WallSeg = 3 x int32; run range per alignment = 2 x int32
        std::vector<std::pair<int32_t, int32_t>> unit_walls(units.size(), {0, 0});      // alignments [first, last) of a unit
        if (walls) {
            for (size_t ui = 0; ui < units.size(); ui++) {
                const Unit &u = units[ui];
                unit_walls[ui].first = (int32_t)(wall_alns.size() / 2);
                for (const miblast_aln &A : u.kept) {
                    const int32_t s0 = (int32_t)(wall_segs.size() / 3);
                    int32_t tt = A.t_lo, qq = A.q_lo;
                    for (int64_t k = 0; k < A.n_ops; k++) {
                        const uint32_t e = u.unit_ops[(size_t)(A.ops_off + k)], o = e & 3u, len = e >> 2;
                        if (o <= 1u) {                                       // '=' or 'X': aligned pairs
                            const size_t nseg = wall_segs.size() / 3;
                            if ((int32_t)nseg > s0 && wall_segs[3 * (nseg - 1)] + wall_segs[3 * (nseg - 1) + 2] == qq && wall_segs[3 * (nseg - 1) + 1] + wall_segs[3 * (nseg - 1) + 2] == tt)
                                wall_segs[3 * (nseg - 1) + 2] += (int32_t)len;             // '=' and 'X' runs of one gap-free stretch
                            else { wall_segs.push_back(qq); wall_segs.push_back(tt); wall_segs.push_back((int32_t)len); }
                            tt += (int32_t)len; qq += (int32_t)len;
                        } else if (o == 2u) qq += (int32_t)len;
                        else tt += (int32_t)len;
                    }
                    wall_alns.push_back(s0); wall_alns.push_back((int32_t)(wall_segs.size() / 3));
                }
                unit_walls[ui].second = (int32_t)(wall_alns.size() / 2);
            }
            g.wall_segs.ensure(wall_segs.size() + 3); g.wall_alns.ensure(wall_alns.size() + 2);
            g.stage.h2d(g.wall_segs.p, wall_segs.data(), wall_segs.size() * 4, s);
            g.stage.h2d(g.wall_alns.p, wall_alns.data(), wall_alns.size() * 4, s);
        }
        RoundDp rd;
        std::vector<SideRun> &sides = rd.sides;
        std::vector<Piece> &pieces = rd.pieces;
        // per piece the alignments of its unit (uploaded with the pieces of a launch)
        auto upload_wall_refs = [&](size_t first, size_t last) -> const int32_t * {
            if (!walls) return nullptr;
            std::vector<int32_t> ref(2 * (last - first));
            for (size_t x = first; x < last; x++) { ref[2 * (x - first)] = unit_walls[(size_t)pieces[x].unit].first; ref[2 * (x - first) + 1] = unit_walls[(size_t)pieces[x].unit].second; }
            g.wall_ref.ensure_keep(2 * last + 2);
            g.stage.h2d(g.wall_ref.p + 2 * first, ref.data(), ref.size() * 4, s);
            return g.wall_ref.p + 2 * first;
        };
        std::vector<DpProb> &probs = rd.probs;
        std::vector<DpOut> &outs = rd.outs;
        std::vector<VerifyJob> vjobs;
        std::vector<VerifyOut> vres;
        struct RelayPt { int unit; int32_t t, q, dir; int piece; int tail; };   // origin of a relay; piece = its fresh DP (-1: not queued yet); tail = virtual relays since the last anchor
        std::vector<RelayPt> relay_pts;
        // relay points are looked up per (unit, direction) -- chains never leave theirs --: group 2 * unit + (dir > 0), key (t, q),
        // value = index into relay_pts minus the group's base (the parallel planting below numbers a group's points locally)
        std::vector<std::unordered_map<unsigned long long, int>> relay_id(2 * units.size());
        std::vector<int> relay_base(2 * units.size(), 0);
        bool arena_full = false;
        long n_verify_ok = 0, n_verify_bad = 0, n_subrounds = 0;
        auto relay_key = [](int32_t t, int32_t q) -> unsigned long long { return ((unsigned long long)(uint32_t)t << 32) | (uint32_t)q; };
        auto relay_at = [&](int unit, int32_t dir, int32_t t, int32_t q) -> int {
            const size_t gi = 2 * (size_t)unit + (dir > 0 ? 1 : 0);
            auto ins = relay_id[gi].emplace(relay_key(t, q), (int)relay_pts.size() - relay_base[gi]);
            if (!ins.second) return relay_base[gi] + ins.first->second;
            relay_pts.push_back(RelayPt{unit, t, q, dir, -1, 0});
            return (int)relay_pts.size() - 1;
        };
        const RelayLattice lattice{relay_s, relay_w, relay_tol, relay_gap, relay_tail_rows};
        auto next_point = [&](int unit, const DpProb &b, int32_t t, int32_t q, int32_t min_dq, int from_tail) -> NextPt { return relay_next_point(units[(size_t)unit], lattice, b, t, q, min_dq, from_tail); };
        // The lattice step is a pure function of the unit's anchors and the point, and it is where planting spends its time (two
        // scans of the anchor index per step): the chains of all heads are therefore walked ahead of the planting, one task per
        // (unit, direction) on the worker threads, into chain_memo; the serial planting below then finds its steps there.
        std::vector<std::unordered_map<unsigned long long, NextPt>> chain_memo;
        auto memo_key = [](int32_t t, int32_t q) -> unsigned long long { return ((unsigned long long)(uint32_t)t << 32) | (uint32_t)q; };
        auto next_relay = [&](int unit, const DpProb &b, int32_t t, int32_t q, int32_t min_dq, int from_tail) -> int {
            NextPt np{0, 0, 0, -1};
            bool have = false;
            if (min_dq == (int32_t)(relay_s / 2) && !chain_memo.empty()) {
                const auto &m = chain_memo[2 * (size_t)unit + (b.dir > 0 ? 1 : 0)];
                const auto it = m.find(memo_key(t, q));
                if (it != m.end()) { np = it->second; have = true; }
            }
            if (!have) np = next_point(unit, b, t, q, min_dq, from_tail);
            if (!np.ok) return -1;
            const int id = relay_at(unit, b.dir, np.t, np.q);
            if (np.set_tail >= 0) relay_pts[(size_t)id].tail = np.set_tail;
            return id;
        };
        // base problem of every side (anchor, direction, room to the contig ends)
        std::vector<DpProb> side_base((size_t)nsides);
        for (size_t k = 0; k < pend.size(); k++) {
            const Unit &u = units[pend[k].unit];
            const Anchor &a = u.anchors[pend[k].anchor];
            const SeqSet &T = *jobs[(size_t)u.pair]->T, &Q = *jobs[(size_t)u.pair]->Q;
            const int tcg = T.contig_of(a.t);
            const int64_t tlo = T.starts[(size_t)tcg], thi = tlo + T.lens[(size_t)tcg];
            const int64_t qlo = Q.starts[(size_t)u.q_contig], qhi = qlo + Q.lens[(size_t)u.q_contig];
            for (int sdn = 0; sdn < 2; sdn++) {
                DpProb b;
                memset(&b, 0, sizeof b);
                b.t0 = a.t; b.q0 = a.q; b.strand = u.strand; b.pad0 = u.pair; b.init_snap = -1; b.snap_idx = -1;
                if (sdn == 0) { b.dir = +1; b.na = (int32_t)(thi - a.t); b.nb = (int32_t)(qhi - a.q); }
                else { b.dir = -1; b.na = (int32_t)(a.t - tlo); b.nb = (int32_t)(a.q - qlo); }
                side_base[2 * k + (size_t)sdn] = b;
            }
        }
        // MIBLAST_PLANT_THREADS: 1 (default) the chains of every (unit, direction) are planted on the worker threads, group by group,
        // and strung together afterwards; 2: only the lattice steps are walked ahead in parallel (chain_memo), planting is serial; 0: serial
        const long plant_threads = env_long("MIBLAST_PLANT_THREADS", 1);
        const bool plant_parallel = relay_s0 > 0 && plant_at_once && plant_threads == 1;
        if (relay_s0 > 0 && plant_at_once && plant_threads == 2) {
            chain_memo.assign(2 * units.size(), {});
            std::vector<std::vector<size_t>> group(2 * units.size());     // sides of a (unit, direction), in planting order
            for (size_t si = 0; si < (size_t)nsides; si++) group[2 * pend[si / 2].unit + (side_base[si].dir > 0 ? 1 : 0)].push_back(si);
            std::vector<size_t> busy;
            for (size_t gi = 0; gi < group.size(); gi++) if (!group[gi].empty()) busy.push_back(gi);
            parallel_for(busy.size(), [&](size_t bi) {
                const size_t gi = busy[bi];
                auto &memo = chain_memo[gi];
                std::unordered_map<unsigned long long, int> tails;       // tail count of the points met so far (first writer wins, as relay_at does)
                for (size_t si : group[gi]) {
                    const DpProb &b = side_base[si];
                    int32_t t = b.t0, q = b.q0;
                    int tail = 0;
                    for (long n = 0; n <= relay_max; n++) {
                        const unsigned long long key = memo_key(t, q);
                        if (memo.count(key)) break;                      // the rest of the chain is known
                        const NextPt np = next_point((int)(gi / 2), b, t, q, (int32_t)(relay_s / 2), tail);
                        memo.emplace(key, np);
                        if (!np.ok) break;
                        const unsigned long long nk = memo_key(np.t, np.q);
                        if (np.set_tail >= 0) tails[nk] = np.set_tail; else tails.emplace(nk, 0);
                        t = np.t; q = np.q; tail = tails[nk];
                    }
                }
            });
        }
        while (true) {                                   // retried with a larger arena if the trace does not fit
            sides.assign((size_t)nsides, SideRun());
            pieces.clear(); probs.clear(); outs.clear(); vjobs.clear(); vres.clear(); relay_pts.clear();
            for (auto &m : relay_id) m.clear();
            std::fill(relay_base.begin(), relay_base.end(), 0);
            pieces.reserve(4096); probs.reserve(4096); vjobs.reserve(4096); relay_pts.reserve(4096);
            arena_full = false;
            uint64_t dir_entries = 0;
            static std::atomic<int> stamp_counter{1};
            const int stamp = stamp_counter.fetch_add(1) + 1;               // of this round's snapshots (a new one for every attempt: the piece numbers start over)
            MB_HIP(hipMemsetAsync(g.arena_next.p, 0, 8, s));
            // entry snapshots of a relay: after relay_w rows, and again after 2 and 4 times that -- a hand-over rejected at the first
            // (the relay's state had not converged yet) is retried at the next with a SHORT continuation of the upstream piece
            // instead of running it all the way to the relay after
            auto add_piece = [&](int unit, const DpProb &base, int32_t ot, int32_t oq, int32_t row_lo, int32_t min_row, int32_t stop_row,
                                 int32_t snap_row, int init_piece, int target, int ckpt = 0) -> int {
                const int id = (int)pieces.size();
                DpProb pr = base;
                const int32_t dr = (oq - base.q0) * base.dir, dc = (ot - base.t0) * base.dir;
                pr.t0 = ot; pr.q0 = oq; pr.na = base.na - dc; pr.nb = base.nb - dr;
                pr.row_lo = row_lo; pr.stop_row = stop_row; pr.snap_row = snap_row;
                pr.snap_row2 = pr.snap_row3 = 0;
                if (snap_row > 0 && relay_ckpt) {                        // a relay: later entry snapshots while it is still running
                    if (stop_row == 0 || 2 * snap_row < stop_row) pr.snap_row2 = 2 * snap_row;
                    if (stop_row == 0 || 4 * snap_row < stop_row) pr.snap_row3 = 4 * snap_row;
                }
                pr.init_snap = init_piece >= 0 ? kSnapSlots * init_piece + 1 : -1; pr.snap_idx = kSnapSlots * id;
                pr.row_off = dir_entries;
                const int64_t last = reach_of(stop_row, pr.nb);
                pr.cap_row = stop_row > 0 && inline_rows > 0 ? (int32_t)last : 0; pr.ck0 = ckpt;
                dir_entries += (uint64_t)((last - row_lo) / 4096) + 2;
                probs.push_back(pr);
                pieces.push_back(Piece{unit, ot, oq, base.dir, row_lo, min_row, stop_row, target, ckpt, init_piece, -1, -1});
                if (target >= 0) {
                    // checked right after the launch: this piece's exit state against the aimed relay's entry state
                    const RelayPt ta = relay_pts[(size_t)target];
                    pieces.back().vjob = (int)vjobs.size();
                    vjobs.push_back(VerifyJob{kSnapSlots * id + 1, -1, (ta.t - ot) * base.dir, (ta.q - oq) * base.dir});   // nslot set at launch
                }
                return id;
            };
            // fresh pieces of the relay chain that starts at anchor `a` (created once per unit and direction)
            auto plant_chain = [&](int unit, const DpProb &base, int a) {
                for (long n = 0; a >= 0 && n < relay_max; n++) {
                    if (relay_pts[(size_t)a].piece >= 0) return;                 // the rest of the chain exists already
                    const RelayPt c = relay_pts[(size_t)a];                      // (copy: next_relay may grow the table)
                    int nx = next_relay(unit, base, c.t, c.q, (int32_t)(relay_s / 2), c.tail);
                    int32_t stop = nx >= 0 ? (relay_pts[(size_t)nx].q - c.q) * base.dir + (int32_t)relay_w : (int32_t)(relay_end_steps * relay_s + relay_w);
                    if (nx >= 0 && n + 1 == relay_max) { nx = -1; stop = (int32_t)(relay_s + relay_w); }   // chain cut: whoever gets here plants the rest
                    const int id = add_piece(unit, base, c.t, c.q, 0, (int32_t)relay_w, stop, (int32_t)relay_w, -1, nx);
                    relay_pts[(size_t)a].piece = id;
                    a = nx;
                }
            };
            if (plant_parallel) {
                // ---- every (unit, direction) plants its heads and their relay chains into tables of its own (the same steps as the
                //      serial code below, with local numbers); the tables are then strung together group by group.  Which number a
                //      piece gets is scheduling only.
                struct Plant {
                    std::vector<Piece> pieces; std::vector<DpProb> probs; std::vector<VerifyJob> vjobs; std::vector<RelayPt> pts;
                    std::unordered_map<unsigned long long, int> id;
                    uint64_t dir_entries = 0;
                    std::vector<std::pair<size_t, int>> heads;              // (side, its head piece)
                };
                std::vector<std::vector<size_t>> group(2 * units.size());     // sides of a (unit, direction), in planting order
                for (size_t si = 0; si < (size_t)nsides; si++) group[2 * pend[si / 2].unit + (side_base[si].dir > 0 ? 1 : 0)].push_back(si);
                std::vector<size_t> busy;
                for (size_t gi = 0; gi < group.size(); gi++) if (!group[gi].empty()) busy.push_back(gi);
                std::vector<Plant> plants(busy.size());
                parallel_for(busy.size(), [&](size_t bi) {
                    const size_t gi = busy[bi];
                    const int unit = (int)(gi / 2);
                    Plant &pl = plants[bi];
                    auto l_relay_at = [&](int32_t dir, int32_t t, int32_t q) -> int {
                        auto ins = pl.id.emplace(relay_key(t, q), (int)pl.pts.size());
                        if (ins.second) pl.pts.push_back(RelayPt{unit, t, q, dir, -1, 0});
                        return ins.first->second;
                    };
                    auto l_next_relay = [&](const DpProb &b, int32_t t, int32_t q, int32_t min_dq, int from_tail) -> int {
                        const NextPt np = next_point(unit, b, t, q, min_dq, from_tail);
                        if (!np.ok) return -1;
                        const int id = l_relay_at(b.dir, np.t, np.q);
                        if (np.set_tail >= 0) pl.pts[(size_t)id].tail = np.set_tail;
                        return id;
                    };
                    auto l_add_piece = [&](const DpProb &base, int32_t ot, int32_t oq, int32_t stop_row, int32_t snap_row, int target) -> int {
                        const int id = (int)pl.pieces.size();
                        DpProb pr = base;
                        const int32_t dr = (oq - base.q0) * base.dir, dc = (ot - base.t0) * base.dir;
                        pr.t0 = ot; pr.q0 = oq; pr.na = base.na - dc; pr.nb = base.nb - dr;
                        pr.row_lo = 0; pr.stop_row = stop_row; pr.snap_row = snap_row;
                        pr.snap_row2 = pr.snap_row3 = 0;
                        if (snap_row > 0 && relay_ckpt) {
                            if (stop_row == 0 || 2 * snap_row < stop_row) pr.snap_row2 = 2 * snap_row;
                            if (stop_row == 0 || 4 * snap_row < stop_row) pr.snap_row3 = 4 * snap_row;
                        }
                        pr.init_snap = -1; pr.snap_idx = id;                  // (local number: kSnapSlots x the global one when the tables are strung together)
                        pr.row_off = pl.dir_entries;
                        const int64_t last = reach_of(stop_row, pr.nb);
                        pr.cap_row = stop_row > 0 && inline_rows > 0 ? (int32_t)last : 0; pr.ck0 = 0;
                        pl.dir_entries += (uint64_t)(last / 4096) + 2;
                        pl.probs.push_back(pr);
                        pl.pieces.push_back(Piece{unit, ot, oq, base.dir, 0, snap_row > 0 ? (int32_t)relay_w : -1, stop_row, target, 0, -1, -1, -1});
                        if (target >= 0) {
                            const RelayPt ta = pl.pts[(size_t)target];
                            pl.pieces.back().vjob = (int)pl.vjobs.size();
                            pl.vjobs.push_back(VerifyJob{id, -1, (ta.t - ot) * base.dir, (ta.q - oq) * base.dir});      // eslot: local piece for now
                        }
                        return id;
                    };
                    for (size_t si : group[gi]) {
                        const DpProb b = side_base[si];
                        int aim = l_next_relay(b, b.t0, b.q0, (int32_t)(relay_s / 2), 0);
                        int32_t stop = (int32_t)(relay_end_steps * relay_s);
                        if (aim >= 0) {
                            int a = aim;
                            for (long n = 0; a >= 0 && n < relay_max; n++) {            // plant_chain
                                if (pl.pts[(size_t)a].piece >= 0) break;
                                const RelayPt c = pl.pts[(size_t)a];
                                int nx = l_next_relay(b, c.t, c.q, (int32_t)(relay_s / 2), c.tail);
                                int32_t cstop = nx >= 0 ? (pl.pts[(size_t)nx].q - c.q) * b.dir + (int32_t)relay_w : (int32_t)(relay_end_steps * relay_s + relay_w);
                                if (nx >= 0 && n + 1 == relay_max) { nx = -1; cstop = (int32_t)(relay_s + relay_w); }
                                const int id = l_add_piece(b, c.t, c.q, cstop, (int32_t)relay_w, nx);
                                pl.pts[(size_t)a].piece = id;
                                a = nx;
                            }
                            stop = (pl.pts[(size_t)aim].q - b.q0) * b.dir + (int32_t)relay_w;
                        }
                        pl.heads.emplace_back(si, l_add_piece(b, b.t0, b.q0, stop, 0, aim));
                    }
                });
                for (size_t bi = 0; bi < busy.size(); bi++) {
                    Plant &pl = plants[bi];
                    const size_t gi = busy[bi];
                    const int pb = (int)pieces.size(), rb = (int)relay_pts.size(), vb = (int)vjobs.size();
                    relay_base[gi] = rb;
                    relay_id[gi].swap(pl.id);
                    for (RelayPt &r : pl.pts) { if (r.piece >= 0) r.piece += pb; relay_pts.push_back(r); }
                    for (size_t x = 0; x < pl.pieces.size(); x++) {
                        Piece pc = pl.pieces[x];
                        DpProb pr = pl.probs[x];
                        if (pc.target >= 0) pc.target += rb;
                        if (pc.vjob >= 0) pc.vjob += vb;
                        pr.snap_idx = kSnapSlots * (pb + (int)x);
                        pr.row_off += dir_entries;
                        pieces.push_back(pc); probs.push_back(pr);
                    }
                    for (VerifyJob v : pl.vjobs) { v.eslot = kSnapSlots * (pb + v.eslot) + 1; vjobs.push_back(v); }
                    dir_entries += pl.dir_entries;
                    for (const auto &h : pl.heads) {
                        SideRun &sd = sides[h.first];
                        sd.base = side_base[h.first]; sd.unit = (int)pend[h.first / 2].unit;
                        const int id = pb + h.second;
                        sd.cur.push_back(id); sd.chain.push_back(id); sd.chain_floor.push_back(-1);
                    }
                }
            }
            for (size_t k = 0; k < pend.size() && !plant_parallel; k++) {
                for (int sdn = 0; sdn < 2; sdn++) {
                    const DpProb b = side_base[2 * k + (size_t)sdn];
                    SideRun &sd = sides[2 * k + (size_t)sdn];
                    sd.base = b; sd.unit = (int)pend[k].unit;
                    // Few sides in flight (one chunk pair): the relay chain of every head is planted at once and the head is aimed
                    // at its first relay -- one launch less on the critical path.  Many sides: most alignments are short, so a side
                    // first has to survive relay_s0 rows before relays are spent on it.
                    int aim = -1;
                    int32_t stop = (int32_t)relay_s0;
                    if (relay_s0 > 0 && plant_at_once) {
                        aim = next_relay(sd.unit, b, b.t0, b.q0, (int32_t)(relay_s / 2), 0);
                        if (aim >= 0) { plant_chain(sd.unit, b, aim); stop = (relay_pts[(size_t)aim].q - b.q0) * b.dir + (int32_t)relay_w; }
                        else stop = (int32_t)(relay_end_steps * relay_s);
                        if (aim < 0 && env_long("MIBLAST_DEBUG", 0) > 2) {
                            const Unit &uu = units[(size_t)sd.unit];
                            long near_cnt = 0;
                            for (const Anchor &c : uu.anchors) if (std::labs((long)c.q - b.q0) < 8 * relay_s && std::labs((long)(c.t - c.q) - (long)(b.t0 - b.q0)) <= relay_tol) near_cnt++;
                            fprintf(stderr, "[miblast]   head without a relay: unit %d (%zu anchors, %u groups) at (%d,%d) dir %d na %d nb %d, %ld anchors within 8 S on the diagonal band\n",
                                    sd.unit, uu.anchors.size(), uu.n_comp, b.t0, b.q0, b.dir, b.na, b.nb, near_cnt);
                        }
                    }
                    const int id = add_piece(sd.unit, b, b.t0, b.q0, 0, -1, stop, 0, -1, aim);
                    sd.cur.push_back(id); sd.chain.push_back(id); sd.chain_floor.push_back(-1);
                }
            }
            size_t launched = 0, vlaunched = 0;           // pieces [0, launched) have run, checks [0, vlaunched) are made
            lap(1);
            while (launched < pieces.size() && !arena_full) {
                n_subrounds++;
                const size_t n_new = pieces.size() - launched, v_new = vjobs.size() - vlaunched;
                g.rowdir.ensure_keep((size_t)dir_entries + 1);
                g.snaps.ensure_keep(pieces.size() * kSnapSlots * kSnapBytes);
                outs.resize(pieces.size()); vres.resize(vjobs.size());
                for (size_t x = launched; x < pieces.size(); x++) {
                    probs[x].aim1 = 0; probs[x].vjob1 = 0;
                    if (pieces[x].vjob >= 0) {
                        const int rp = relay_pts[(size_t)pieces[x].target].piece;
                        vjobs[(size_t)pieces[x].vjob].nslot = kSnapSlots * rp + (pieces[x].ckpt == 0 ? 0 : pieces[x].ckpt + 1);
                        probs[x].aim1 = rp + 1; probs[x].vjob1 = pieces[x].vjob - (int)vlaunched + 1;       // (the piece may look at the hand-over itself: mb_ydrop2.h)
                    }
                }
                // The launch's pieces go behind the round's table on the device (a piece that checks its own hand-over reads the records of
                // the relays down its chain, whichever launch they were queued for), the hand-over checks behind them: one copy; both kinds of
                // results come back in one copy.
                const size_t up_v = Stager::behind(n_new * sizeof(DpProb)), down_v = Stager::behind(n_new * sizeof(DpOut));
                g.round_tab.ensure_keep(launched * sizeof(DpProb) + up_v + v_new * sizeof(VerifyJob) + 256); g.dp_down.ensure(down_v + v_new * sizeof(VerifyOut) + 256);
                DpProb *const d_tab = (DpProb *)g.round_tab.p;
                DpProb *const d_probs = d_tab + launched;
                VerifyJob *const d_vjobs = (VerifyJob *)(g.round_tab.p + launched * sizeof(DpProb) + up_v);
                DpOut *const d_outs = (DpOut *)g.dp_down.p;
                VerifyOut *const d_vres = (VerifyOut *)(g.dp_down.p + down_v);
                g.stage.h2d2(d_probs, probs.data() + launched, n_new * sizeof(DpProb), vjobs.data() + vlaunched, v_new * sizeof(VerifyJob), s);
                // valid = 0 in every header of the new pieces' slots (the headers only: the slots are 16 KiB apart); k_ydrop2 does it itself
                if (dp_kernel != kDpWave2x4) MB_HIP(hipMemset2DAsync(g.snaps.p + launched * kSnapSlots * kSnapBytes, kSnapBytes, 0, sizeof(SnapHdr), n_new * kSnapSlots, s));
                // DP launch, hand-over checks and the copies of both results: one synchronisation.  (Checks made on pieces that
                // turn out to need a rerun are simply made again.)
                run_ydrop_timed(ctx, st, dp_kernel, d_tab, d_outs, (int)n_new, g.pair_ptrs.p, p, kBlk, true, probs.data() + launched,
                                upload_wall_refs(launched, pieces.size()), (int)launched, d_vjobs, stamp, (int)relay_inline_force);
                launch_verify(d_vjobs, d_vres, (int)v_new, g.snaps.p, p.ydrop, p.gap_extend, s);
                g.stage.d2h2(outs.data() + launched, n_new * sizeof(DpOut), vres.data() + vlaunched, v_new * sizeof(VerifyOut), g.dp_down.p, s);
                unsigned long long arena_now = 0;                       // what the round's trace has taken of the arena so far (the same copy-back)
                g.stage.d2h(&arena_now, g.arena_next.p, 8, s);
                MB_HIP(hipStreamSynchronize(s));
                g.stage.done();
                arena_used = std::max(arena_used, arena_now);
                collect_dp_time(ctx, st);
                lap(2);
                if (debug) fprintf(stderr, "[miblast]   first pass (kernel %d): dp kernel total %.2f ms\n", dp_kernel, st.t_dp_kernel_ms);
                if (dp_kernel != kDpLds) {
                    // pieces whose window outgrew the lanes of the one-wave kernel: once more with the LDS ring (same snapshots)
                    std::vector<size_t> again;
                    for (size_t x = launched; x < pieces.size(); x++) if (outs[x].overflow == 1) again.push_back(x);
                    if (!again.empty()) {
                        std::vector<DpProb> sub(again.size());
                        for (size_t y = 0; y < again.size(); y++) sub[y] = probs[again[y]];
                        g.probs.ensure(again.size()); g.outs.ensure(again.size());
                        g.stage.h2d(g.probs.p, sub.data(), sub.size() * sizeof(DpProb), s);
                        run_ydrop_timed(ctx, st, kDpLds, g.probs.p, g.outs.p, (int)again.size(), g.pair_ptrs.p, p, kBlk);
                        std::vector<DpOut> so(again.size());
                        g.stage.d2h(so.data(), g.outs.p, so.size() * sizeof(DpOut), s);
                        MB_HIP(hipStreamSynchronize(s));
                        g.stage.done();
                        // only k_ydrop2 writes the fin_* fields: a rerun piece made no check of its own and ended with what it was queued with
                        for (size_t y = 0; y < again.size(); y++) {
                            const size_t x = again[y];
                            so[y].fin_stop = probs[x].stop_row; so[y].fin_aim1 = probs[x].aim1; so[y].fin_ck = probs[x].ck0; so[y].fin_checks = 0;
                            outs[x] = so[y];
                        }
                        st.dp_reruns += (int64_t)again.size();
                        if (debug) fprintf(stderr, "[miblast]   %zu of %zu pieces outgrew the one-wave kernel and were rerun (dp kernel total %.2f ms)\n", again.size(), n_new, st.t_dp_kernel_ms);
                        launch_verify(d_vjobs, d_vres, (int)v_new, g.snaps.p, p.ydrop, p.gap_extend, s);
                        g.stage.d2h(vres.data() + vlaunched, d_vres, v_new * sizeof(VerifyOut), s);
                        MB_HIP(hipStreamSynchronize(s));
                        g.stage.done();
                    }
                }
                for (size_t x = launched; x < pieces.size(); x++) arena_full |= outs[x].overflow == 3;
                for (size_t x = launched; x < pieces.size(); x++)
                    if (outs[x].overflow == 4) {
                        set_error("walls: a unit holds more than 1024 earlier alignments: beyond what the walls mode of the MI355X path covers");
                        return MIBLAST_ELIMIT;
                    }
                if (arena_full) break;
                if (dp_kernel == kDpWave2x4) {
                    // pieces that looked at their hand-over themselves and went on: the stop row, the aimed relay and the entry snapshot they
                    // ended with are the piece's from here on (k_verify has judged THAT hand-over: the piece rewrote its check)
                    std::vector<int> pt_of;
                    for (size_t x = launched; x < pieces.size(); x++) {
                        const DpOut &o = outs[x];
                        if (o.overflow != 0) continue;
                        st.relay_inline_checks += o.fin_checks;
                        if (pieces[x].stop_row <= 0 || o.fin_stop == pieces[x].stop_row) continue;
                        st.relay_inline_continued++;
                        if (pt_of.empty()) {
                            pt_of.assign(pieces.size(), -1);
                            for (size_t r = 0; r < relay_pts.size(); r++) if (relay_pts[r].piece >= 0) pt_of[(size_t)relay_pts[r].piece] = (int)r;
                        }
                        pieces[x].stop_row = probs[x].stop_row = o.fin_stop;
                        pieces[x].ckpt = o.fin_ck;
                        if (o.fin_aim1 > 0 && pt_of[(size_t)o.fin_aim1 - 1] >= 0) pieces[x].target = pt_of[(size_t)o.fin_aim1 - 1];
                        if (pieces[x].vjob >= 0) {
                            const RelayPt &ta = relay_pts[(size_t)pieces[x].target];
                            vjobs[(size_t)pieces[x].vjob] = VerifyJob{kSnapSlots * (int)x + 1, kSnapSlots * ta.piece + (o.fin_ck == 0 ? 0 : o.fin_ck + 1),
                                                                     (ta.t - pieces[x].ot) * pieces[x].dir, (ta.q - pieces[x].oq) * pieces[x].dir};
                        }
                    }
                }
                for (size_t x = launched; x < pieces.size(); x++) {
                    st.dp_sides_run++;
                    const DpOut *o0 = pieces[x].init_piece >= 0 ? &outs[(size_t)pieces[x].init_piece] : nullptr;
                    st.dp_cells_run += outs[x].cells - (o0 ? o0->cells : 0);
                    st.dp_rows_run += outs[x].rows - (o0 ? o0->rows : 0);
                }
                if (debug) {
                    int maxrows = 0; long long clk = 0;
                    for (size_t x = launched; x < pieces.size(); x++) {
                        const int r = outs[x].rows - (pieces[x].init_piece >= 0 ? outs[(size_t)pieces[x].init_piece].rows : 0);
                        if (r > maxrows) { maxrows = r; clk = outs[x].clocks; }
                        if (2 * r > 3 * (relay_s + relay_w) && relay_s0 > 0 && env_long("MIBLAST_DEBUG", 0) > 1)
                            fprintf(stderr, "[miblast]   long piece %zu: origin (%d,%d) dir %d row_lo %d stop_row %d target %d rows %d stopped %d na %d nb %d\n", x, pieces[x].ot, pieces[x].oq,
                                    pieces[x].dir, pieces[x].row_lo, pieces[x].stop_row, pieces[x].target, r, outs[x].stopped, probs[x].na, probs[x].nb);
                    }
                    fprintf(stderr, "[miblast] round %d.%ld: %zu pieces, %zu checks, max rows %d (%lld shader clocks = %.0f per row), dp kernel total %.2f ms so far, shadow_q %ld\n",
                            round, n_subrounds, n_new, v_new, maxrows, clk, (double)clk / std::max(1, maxrows), st.t_dp_kernel_ms, shadow_q);
                    {   // the pieces that set the launch's time: the three that ran the most rows, and how many rows the others ran
                        std::vector<std::pair<int, size_t>> rr;
                        for (size_t x = launched; x < pieces.size(); x++) rr.emplace_back(outs[x].rows - (pieces[x].init_piece >= 0 ? outs[(size_t)pieces[x].init_piece].rows : 0), x);
                        std::sort(rr.begin(), rr.end(), [](const std::pair<int, size_t> &a, const std::pair<int, size_t> &b) { return a.first > b.first; });
                        for (size_t y = 0; y < std::min<size_t>(3, rr.size()); y++) {
                            const size_t x = rr[y].second;
                            fprintf(stderr, "[miblast]     longest %zu: piece %zu rows %d (row_lo %d planned stop %d, ended at stop %d aim %d ck %d, %d checks inside, stopped %d, snap_row %d) nb %d\n", y, x, rr[y].first,
                                    probs[x].row_lo, probs[x].stop_row, outs[x].fin_stop, outs[x].fin_aim1 - 1, outs[x].fin_ck, outs[x].fin_checks, outs[x].stopped, probs[x].snap_row, probs[x].nb);
                        }
                        if (rr.size() > 8) fprintf(stderr, "[miblast]     rows: median %d, 90th percentile %d, 99th %d\n", rr[rr.size() / 2].first, rr[rr.size() / 10].first, rr[rr.size() / 100].first);
                    }
                }
                const size_t first_new = launched;
                launched = pieces.size(); vlaunched = vjobs.size();
                auto accepted = [&](int x) -> bool {
                    const Piece &px = pieces[(size_t)x];
                    if (px.vjob < 0 || !vres[(size_t)px.vjob].ok) return false;
                    return !(relay_force_reject > 0 && x % relay_force_reject == 0);
                };
                // the continuation of a stopped piece whose hand-over was rejected (or that stopped without an aim): same
                // origin, starts from the exit snapshot, aimed at the next relay whose entry row is still ahead
                auto make_cont = [&](int x) -> int {
                    if (pieces[(size_t)x].cont >= 0) return pieces[(size_t)x].cont;
                    const Piece cp = pieces[(size_t)x];                          // (copies: the vectors grow below)
                    const DpProb cb = probs[(size_t)x];
                    int aim = cp.target;
                    auto entry_row = [&](int r) -> int32_t { return (relay_pts[(size_t)r].q - cp.oq) * cp.dir + (int32_t)relay_w; };   // in this piece's rows
                    if (aim >= 0 && relay_ckpt && cp.ckpt < 2) {
                        // the same relay once more, at its next entry snapshot (if it has one): a short continuation
                        const DpProb &rp = probs[(size_t)relay_pts[(size_t)aim].piece];
                        const int32_t snap_next = cp.ckpt == 0 ? rp.snap_row2 : rp.snap_row3;
                        const int32_t stop_next = (relay_pts[(size_t)aim].q - cp.oq) * cp.dir + snap_next;
                        if (snap_next > 0 && stop_next > cp.stop_row) {
                            const int id = add_piece(cp.unit, cb, cp.ot, cp.oq, cp.stop_row, cp.stop_row, stop_next, 0, x, aim, cp.ckpt + 1);
                            pieces[(size_t)x].cont = id;
                            return id;
                        }
                    }
                    if (aim >= 0) aim = pieces[(size_t)relay_pts[(size_t)aim].piece].target;      // the relay after the rejected one
                    else if (relay_s0 > 0) {
                        // a stop without an aim (first stop of a side, end of a capped chain): the lattice relay beyond the best cell
                        const DpOut &o = outs[(size_t)x];
                        aim = next_relay(cp.unit, cb, cp.ot + cp.dir * o.bj, cp.oq + cp.dir * o.bi, 0, 0);
                        while (aim >= 0 && entry_row(aim) <= cp.stop_row + 64) {
                            const RelayPt r = relay_pts[(size_t)aim];
                            aim = next_relay(cp.unit, cb, r.t, r.q, (int32_t)(relay_s / 2), r.tail);
                        }
                        if (aim >= 0) plant_chain(cp.unit, cb, aim);
                    }
                    while (aim >= 0 && entry_row(aim) <= cp.stop_row + 64) aim = pieces[(size_t)relay_pts[(size_t)aim].piece].target;
                    const int32_t stop = aim >= 0 ? entry_row(aim) : relay_s0 > 0 ? cp.stop_row + (int32_t)(relay_end_steps * relay_s) : 0;
                    const int id = add_piece(cp.unit, cb, cp.ot, cp.oq, cp.stop_row, cp.stop_row, stop, 0, x, aim);
                    pieces[(size_t)x].cont = id;
                    return id;
                };
                // rejected hand-overs are continued at once, whether or not a side has reached them yet: a side then never
                // waits more than one launch per rejection on its path
                for (size_t x = first_new; x < launched; x++)
                    if (outs[x].stopped && outs[x].overflow == 0 && pieces[x].target >= 0) {
                        if (accepted((int)x)) n_verify_ok++;
                        else { n_verify_bad++; make_cont((int)x); }
                    }
                // ---- advance every side along its chain
                for (int si = 0; si < nsides; si++) {
                    SideRun &sd = sides[(size_t)si];
                    while (!sd.done && !sd.wide) {
                        const int tp = sd.cur.back();
                        if (tp >= (int)launched) break;                          // queued, not run yet
                        const Piece cp = pieces[(size_t)tp];
                        const int32_t dr = (cp.oq - sd.base.q0) * sd.base.dir, dc = (cp.ot - sd.base.t0) * sd.base.dir;
                        for (; sd.accounted < sd.cur.size(); sd.accounted++) {   // fold the run's finished pieces into the side's result
                            const int pc = sd.cur[sd.accounted];
                            const DpOut &o = outs[(size_t)pc];
                            if (o.overflow == 1) { sd.wide = true; break; }
                            if ((long long)o.best + sd.c_off > sd.gbest) {
                                sd.gbest = (int)((long long)o.best + sd.c_off); sd.gbi = o.bi + dr; sd.gbj = o.bj + dc; sd.best_piece = pc;
                            }
                        }
                        if (sd.wide) break;
                        const DpOut o = outs[(size_t)tp];
                        if (!o.stopped) {                                        // natural end of the DP
                            sd.acc_cells += o.cells - sd.entry_cells; sd.acc_rows += o.rows - sd.entry_rows;
                            sd.done = true;
                            break;
                        }
                        if (cp.target >= 0 && accepted(tp)) {
                            // the relay's rows are the rows of this DP from its entry row on
                            const VerifyOut &v = vres[(size_t)cp.vjob];
                            const int np0 = relay_pts[(size_t)cp.target].piece;
                            sd.acc_cells += o.cells - sd.entry_cells; sd.acc_rows += o.rows - sd.entry_rows;
                            sd.entry_cells = v.n_cells; sd.entry_rows = v.n_rows;
                            sd.c_off += v.c;
                            sd.cur.assign(1, np0); sd.accounted = 0;
                            sd.chain.push_back(np0); sd.chain_floor.push_back(pieces[(size_t)np0].min_row * (cp.ckpt == 0 ? 1 : cp.ckpt == 1 ? 2 : 4));
                            continue;
                        }
                        const int id = make_cont(tp);
                        sd.cur.push_back(id); sd.chain.push_back(id); sd.chain_floor.push_back(pieces[(size_t)id].min_row);
                    }
                }
            }
            if (!arena_full) {
                // rows wider than the LDS ring: those sides are evaluated in one piece with the C/D ring in HBM
                std::vector<int> wide;
                for (int si = 0; si < nsides; si++) if (sides[(size_t)si].wide) wide.push_back(si);
                for (size_t w0 = 0; w0 < wide.size() && !arena_full; w0 += 32) {
                    const size_t w1 = std::min(wide.size(), w0 + 32);
                    const size_t first = pieces.size();
                    std::vector<int> owner;
                    for (size_t k = w0; k < w1; k++) {
                        SideRun &sd = sides[(size_t)wide[k]];
                        sd.chain.clear(); sd.chain_floor.clear(); sd.cur.clear(); sd.c_off = 0; sd.acc_cells = sd.acc_rows = sd.entry_cells = sd.entry_rows = 0; sd.gbest = -1;
                        const int id = add_piece(sd.unit, sd.base, sd.base.t0, sd.base.q0, 0, -1, 0, 0, -1, -1);
                        probs[(size_t)id].snap_idx = -1;
                        sd.chain.push_back(id); sd.chain_floor.push_back(-1); sd.cur.push_back(id);
                        owner.push_back(wide[k]);
                    }
                    const size_t n_new = pieces.size() - first;
                    g.probs.ensure_keep(pieces.size()); g.outs.ensure_keep(pieces.size()); g.rowdir.ensure_keep((size_t)dir_entries + 1);
                    outs.resize(pieces.size());
                    g.grows.ensure(n_new * 2 * (size_t)kGlobalRowCap);
                    g.stage.h2d(g.probs.p + first, probs.data() + first, n_new * sizeof(DpProb), s);
                    run_ydrop_timed(ctx, st, kDpHbm, g.probs.p + first, g.outs.p + first, (int)n_new, g.pair_ptrs.p, p, kBlkWide, false, nullptr, upload_wall_refs(first, pieces.size()));
                    g.stage.d2h(outs.data() + first, g.outs.p + first, n_new * sizeof(DpOut), s);
                    MB_HIP(hipStreamSynchronize(s));
                    g.stage.done();
                    for (size_t x = first; x < pieces.size(); x++) {
                        const DpOut &o = outs[x];
                        if (o.overflow == 1) { set_error("DP row wider than 2^20 columns"); return MIBLAST_ELIMIT; }
                        if (o.overflow == 4) { set_error("walls: a unit holds more than 1024 earlier alignments: beyond what the walls mode of the MI355X path covers"); return MIBLAST_ELIMIT; }
                        if (o.overflow == 3) { arena_full = true; continue; }
                        SideRun &sd = sides[(size_t)owner[x - first]];
                        sd.gbest = o.best; sd.gbi = o.bi; sd.gbj = o.bj; sd.best_piece = (int)x;
                        sd.acc_cells = o.cells; sd.acc_rows = o.rows; sd.done = true;
                        st.dp_sides_run++; st.dp_cells_run += o.cells; st.dp_rows_run += o.rows;
                    }
                }
            }
            if (!arena_full) break;
            const int rc_grow = grow_trace_arena(ctx, arena_raw_estimate);
            if (rc_grow != MIBLAST_OK) return rc_grow;
        }
        lap(3);
        st.relay_accepted += n_verify_ok; st.relay_rejected += n_verify_bad;
        if (debug) fprintf(stderr, "[miblast] round %d: %d sides in %zu pieces, %ld launches, hand-overs %ld accepted / %ld rejected (regime: %ld sides in flight%s, S0 %ld S %ld W %ld, %s)\n",
                           round, nsides, pieces.size(), n_subrounds, n_verify_ok, n_verify_bad, nsides_call, crowd ? " = crowd" : "", relay_s0, relay_s, relay_w, plant_at_once ? "chains planted with the heads" : "relays after the first stop");

        const int rc_fin = gapped_finish_round(ctx, p, jobs, units, pend, rd, st, debug);
        if (rc_fin != MIBLAST_OK) return rc_fin;
        lap(4);
        if (debug) fprintf(stderr, "[miblast]   round %d host timeline: commit+nominate %.2f ms, plant %.2f, launches+sync %.2f, advance+continuations %.2f, results+traceback %.2f\n",
                           round, tm[0] * 1e3, tm[1] * 1e3, tm[2] * 1e3, tm[3] * 1e3, tm[4] * 1e3);
    }
    arena_learn(ctx, arena_raw_estimate, arena_used);
    st.t_gapped = now_s() - t_g0;
    for (size_t jk = 0; jk < n_members; jk++) {      // launch-level figures are shared by the pairs that were in flight together
        PairJob *j = jobs[members ? (*members)[jk] : jk];
        miblast_stats &d = j->res->stats;
        d.t_gapped = st.t_gapped; d.gapped_rounds = st.gapped_rounds; d.dp_sides_run = st.dp_sides_run; d.dp_cells_run = st.dp_cells_run;
        d.dp_rows_run = st.dp_rows_run; d.t_dp_kernel_ms = st.t_dp_kernel_ms; d.dp_kernel_launches = st.dp_kernel_launches;
        d.relay_accepted = st.relay_accepted; d.relay_rejected = st.relay_rejected; d.relay_inline_checks = st.relay_inline_checks; d.relay_inline_continued = st.relay_inline_continued; d.dp_reruns = st.dp_reruns; d.t_traceback_ms = st.t_traceback_ms; d.t_merge_ms = st.t_merge_ms;
    }
    return MIBLAST_OK;

}

// PAF (or the general HSP format) of a call's pairs, in the reference's output order.  Three passes over the pairs, so that the
// cigar text of ALL pairs' alignments is formatted in one parallel region (a pair's own region would run inline inside the pairs'):
// collect (alignments and ops in output order, the list of text chunks), format the chunks, lay the lines out.
struct CigarTask { size_t aln; int64_t k0, k1; std::string text; int64_t nmatch, alen; size_t at; };
struct OutputJob {
    std::vector<CigarTask> ctasks;
    std::vector<size_t> cfirst;
    double t_out0 = 0;
};

static void output_collect(PairJob &job, int pair, std::vector<Unit> &units, OutputJob &oj) {
    const SeqSet &T = *job.T, &Q = *job.Q;
    Result &res = *job.res;
    miblast_stats &st = res.stats;
    // ---- output: per query contig, '+' then '-', commit order ------------------------------
    for (int qc_i = 0; qc_i < (int)Q.starts.size(); qc_i++)
        for (int strand = 0; strand < 2; strand++)
            for (Unit &u : units) {
                if (u.pair != pair || u.q_contig != qc_i || u.strand != strand) continue;
                res.aln_anchor_score.insert(res.aln_anchor_score.end(), u.kept_anchor_score.begin(), u.kept_anchor_score.end());
                for (miblast_aln A : u.kept) {
                    int64_t off = (int64_t)res.ops.size();
                    res.ops.insert(res.ops.end(), u.unit_ops.begin() + A.ops_off, u.unit_ops.begin() + A.ops_off + A.n_ops);
                    A.ops_off = off;
                    res.alns.push_back(A);
                }
            }
    st.alignments = (int64_t)res.alns.size();
    oj.t_out0 = now_s();
    {
        size_t reserve = 0;
        for (const miblast_aln &A : res.alns) reserve += 256 + Q.names[(size_t)A.q_contig].size() + T.names[(size_t)A.t_contig].size() + (size_t)A.n_ops * 9;
        res.paf.reserve(reserve);
    }
    // the cigar text of long alignments is formatted in chunks on several threads
    std::vector<CigarTask> &ctasks = oj.ctasks;
    std::vector<size_t> &cfirst = oj.cfirst;
    ctasks.clear();
    cfirst.assign(res.alns.size() + 1, 0);
    const int64_t kOpsPerTask = env_long("MIBLAST_OUTPUT_CHUNK", 4096);
    for (size_t x = 0; x < res.alns.size(); x++) {
        cfirst[x] = ctasks.size();
        for (int64_t k0 = 0; k0 < res.alns[x].n_ops; k0 += kOpsPerTask) ctasks.push_back(CigarTask{x, k0, std::min(res.alns[x].n_ops, k0 + kOpsPerTask), {}, 0, 0, 0});
    }
    cfirst[res.alns.size()] = ctasks.size();
}

static void output_cigar_chunk(const Result &res, CigarTask &t) {
    // (nearly every run length has one or two digits: those come from a table, straight into a buffer sized for the worst case)
    static const char two[] = "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
                              "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
    const miblast_aln &A = res.alns[t.aln];
    std::unique_ptr<char[]> room(new char[(size_t)(t.k1 - t.k0) * 11 + 1]);    // (30 bits of length: at most 10 digits and the op; not zeroed)
    char *const w0 = room.get();
    char *w = w0;
    const uint32_t *ops = res.ops.data() + A.ops_off;
    for (int64_t k = t.k0; k < t.k1; k++) {
        const uint32_t o = ops[k];
        uint32_t u = o >> 2;
        t.alen += u;
        if ((o & 3u) == 0) t.nmatch += u;
        if (u < 10) *w++ = (char)('0' + u);
        else if (u < 100) { *w++ = two[2 * u]; *w++ = two[2 * u + 1]; }
        else {
            char buf[12]; int n = 0;
            do { buf[n++] = (char)('0' + u % 10); u /= 10; } while (u);
            while (n) *w++ = buf[--n];
        }
        *w++ = "=XID"[o & 3u];
    }
    t.text.assign(w0, (size_t)(w - w0));
}

static void output_layout(const miblast_params &p, PairJob &job, OutputJob &oj) {
    const SeqSet &T = *job.T, &Q = *job.Q;
    Result &res = *job.res;
    miblast_stats &st = res.stats;
    std::vector<CigarTask> &ctasks = oj.ctasks;
    const std::vector<size_t> &cfirst = oj.cfirst;
    auto put_num = [&](long long v) {                   // decimal formatting without snprintf (hundreds of thousands of cigar ops)
        char buf[24]; int n = 0;
        unsigned long long u = v < 0 ? (unsigned long long)(-v) : (unsigned long long)v;
        do { buf[n++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) buf[n++] = '-';
        while (n) res.paf.push_back(buf[--n]);
    };
    // line heads first (short), then every piece of text is copied to its place
    std::vector<std::string> heads(res.alns.size());
    size_t total = res.paf.size();
    const size_t paf0 = total;
    for (size_t ai = 0; ai < res.alns.size(); ai++) {
        const miblast_aln &A = res.alns[ai];
        int64_t qst = Q.starts[(size_t)A.q_contig], qlen = Q.lens[(size_t)A.q_contig];
        int64_t tst = T.starts[(size_t)A.t_contig], tlen = T.lens[(size_t)A.t_contig];
        int64_t qs = A.q_lo - qst, qe = A.q_hi - qst;
        if (A.strand) { int64_t s2 = qlen - qe, e2 = qlen - qs; qs = s2; qe = e2; }
        int64_t nmatch = 0, alen = 0;
        for (size_t ti = cfirst[ai]; ti < cfirst[ai + 1]; ti++) { nmatch += ctasks[ti].nmatch; alen += ctasks[ti].alen; }
        // qname qlen qstart qend strand tname tlen tstart tend nmatch alnlen 255 AS:i:<score> cg:Z:<cigar>  
        std::string &h = heads[ai];
        auto num = [&](long long v) { h += std::to_string(v); };
        h += Q.names[(size_t)A.q_contig]; h.push_back('\t');
        num(qlen); h.push_back('\t'); num(qs); h.push_back('\t'); num(qe); h.push_back('\t');
        h.push_back(A.strand ? '-' : '+'); h.push_back('\t');
        h += T.names[(size_t)A.t_contig]; h.push_back('\t');
        num(tlen); h.push_back('\t'); num(A.t_lo - tst); h.push_back('\t'); num(A.t_hi - tst); h.push_back('\t');
        num(nmatch); h.push_back('\t'); num(alen);
        h += "\t255\tAS:i:"; num(A.score); h += "\tcg:Z:";
        total += h.size();
        for (size_t ti = cfirst[ai]; ti < cfirst[ai + 1]; ti++) { ctasks[ti].at = total; total += ctasks[ti].text.size(); }
        total += 1;                                     // '\n'
    }
    res.paf.resize(total);
    {
        size_t at = paf0;
        res.line_off.clear();
        for (size_t ai = 0; ai < res.alns.size(); ai++) {
            res.line_off.push_back(at);
            memcpy(&res.paf[at], heads[ai].data(), heads[ai].size());
            at += heads[ai].size();
            for (size_t ti = cfirst[ai]; ti < cfirst[ai + 1]; ti++) at += ctasks[ti].text.size();
            res.paf[at++] = '\n';
        }
        res.line_off.push_back(at);
    }
    for (const CigarTask &t : ctasks) if (!t.text.empty()) memcpy(&res.paf[t.at], t.text.data(), t.text.size());
    if (p.format == 1) {
        // --format=general:name1,zstart1,end1,name2,zstart2+,end2+ (cactus_lastzRepeatMask.py:104): one line per HSP
        res.paf += "#name1\tzstart1\tend1\tname2\tzstart2+\tend2+\n";
        std::vector<size_t> order(res.hsps.size());
        for (size_t k = 0; k < order.size(); k++) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {      // per query contig, '+' then '-', found order
            if (res.hsps[a].q_contig != res.hsps[b].q_contig) return res.hsps[a].q_contig < res.hsps[b].q_contig;
            return res.hsps[a].strand < res.hsps[b].strand;
        });
        for (size_t k : order) {
            const miblast_hsp &h = res.hsps[k];
            const int qc_i = h.q_contig, strand = h.strand;
            const int tcg = T.contig_of(h.t_start);
            const int64_t qst = Q.starts[(size_t)qc_i], qlen = Q.lens[(size_t)qc_i];
            int64_t qs = h.q_start - qst, qe = qs + h.len;
            if (strand) { int64_t s2 = qlen - qe, e2 = qlen - qs; qs = s2; qe = e2; }
            res.paf += T.names[(size_t)tcg]; res.paf.push_back('\t');
            put_num(h.t_start - T.starts[(size_t)tcg]); res.paf.push_back('\t');
            put_num(h.t_start - T.starts[(size_t)tcg] + h.len); res.paf.push_back('\t');
            res.paf += Q.names[(size_t)qc_i]; res.paf.push_back('\t');
            put_num(qs); res.paf.push_back('\t'); put_num(qe); res.paf.push_back('\n');
        }
    }
    if (p.markend) res.paf += "# lastz end-of-file\n";
    if (env_long("MIBLAST_DEBUG", 0)) fprintf(stderr, "[miblast] PAF formatting: %.2f ms; index %.2f ms, seed %.2f ms, gapped %.2f ms\n", (now_s() - oj.t_out0) * 1e3, st.t_index * 1e3, st.t_seed * 1e3, st.t_gapped * 1e3);
    st.t_total = now_s() - job.t_begin;
}

// A lane's seed-stage buffers sized for the largest strand the call's context has met, at the lane's START -- before it takes a pair: a lane
// that the others leave no pair in one step (they are taken from a queue) would otherwise make these allocations in the middle of a later
// step, and a device allocation while other lanes keep the device busy stalls every lane (seed_phase).
static void presize_lane(Ctx &lc, int64_t max_diags, int64_t max_q) {
    if (!lc.hits_hint) return;
    Workspace &w = *lc.ws;
    w.presize_gapped();                                    // the gapped stage's tables at the largest size any lane of the context has needed
    const unsigned long long hint = lc.hits_hint->load(std::memory_order_relaxed);
    const int64_t hit_cap = getenv("MIBLAST_HIT_CAP") ? env_long("MIBLAST_HIT_CAP", 32l << 20) : env_long("MIBLAST_DENSE_HIT_CAP", 128l << 20);
    if (hint == 0) return;
    if (2 * (hint + hint / 8) > (unsigned long long)hit_cap) {
        // strands in q batches (a 30 Mb x 30 Mb pair under the default option set: 4 x 10^8 hits per strand): every batch's buffers at the size of
        // the largest batch there can be, ONCE -- the batches of a strand differ in size, and a lane that met a slightly larger one in a later step
        // made its key, HSP and head buffers anew in the middle of that step (hm30: a step of 928 ms among steps of 228)
        const size_t want = (size_t)hit_cap;
        w.keys_a.ensure(want); w.keys_b.ensure(want); w.hsps.ensure(want);
        w.heads.ensure(2 * want + want / 4 + 64); w.n_heads.ensure(8);
        const int diag_bits = std::min(30, std::max(1, (int)std::ceil(std::log2((double)(max_diags + 2)))));
        (void)ux_scratch(w, nullptr, want, (int64_t)1 << diag_bits);
        w.sort_temp.ensure(sort_keys_temp_bytes((int64_t)want, 32 + diag_bits) + 16);
        w.extent.ensure(((size_t)1 << diag_bits) + 8);
        // ... and what the two-pass search keeps per query position and per batch (the lane that had the call's small pairs so far met the large one
        // in a later step and made these anew: hm30, 13 allocations in the first timed steps of most runs)
        const int64_t nq = std::max<int64_t>(1, max_q);
        w.qcnt.ensure((size_t)nq); w.hit_off.ensure((size_t)nq);
        w.qbsum.ensure((size_t)((nq + 2047) / 2048) + 2); w.scan_scratch.ensure((size_t)((nq + 2047) / 2048) + 2);
        if (env_long("MIBLAST_SEED_ORDERED", 1) != 0 && env_long("MIBLAST_SORT_BIN", 1) != 0) {
            const int bin_mean = (int)std::max<long>(1, env_long("MIBLAST_BIN_MEAN", 11000));
            w.bin_state.ensure(2 * (size_t)bin_state_words()); w.bin_matrix.ensure(2 * (size_t)bin_matrix_words_for(bin_keys_max(), diag_bits, bin_mean));
        }
        return;
    }
    const size_t want = (size_t)(hint + hint / 8);
    w.keys_a.ensure(2 * want); w.keys_b.ensure(want); w.hsps.ensure(2 * want);
    w.heads.ensure(2 * want + want / 4 + 64); w.n_heads.ensure(8);
    const int diag_bits = std::min(30, std::max(1, (int)std::ceil(std::log2((double)(max_diags + 2)))));
    (void)ux_scratch(w, nullptr, want, (int64_t)1 << diag_bits);
    if (env_long("MIBLAST_SEED_ORDERED", 1) != 0 && env_long("MIBLAST_SORT_BIN", 1) != 0) {
        const int bin_mean = (int)std::max<long>(1, env_long("MIBLAST_BIN_MEAN", 11000));
        const unsigned long long seen = std::max<unsigned long long>(hint, w.last_strand_hits);
        const unsigned long long room = std::min<unsigned long long>(std::min<unsigned long long>((unsigned long long)w.keys_a.n, (unsigned long long)hit_cap) / 2,
                                                                     std::max<unsigned long long>(1ull << 20, seen + seen / 4));
        w.bin_state.ensure(2 * (size_t)bin_state_words()); w.bin_matrix.ensure(2 * (size_t)bin_matrix_words_for(room, diag_bits, bin_mean));
    }
}

static int align_pairs_impl(Ctx &ctx, const SeqSet *const *Ts, const SeqSet *const *Qs, size_t n, const miblast_params &pin, Result **results) {
    const double t_call0 = now_s();
    MB_HIP(hipSetDevice(ctx.device));
    ctx.ws->share_marks(*ctx.ws);                          // (the context's own tables take part in its lanes' high-water marks)
    if (!ctx.arena_scale) { ctx.arena_scale = &ctx.ws->arena_scale; ctx.arena_class = &ctx.ws->arena_class; }
    for (Ctx *l : ctx.ws->lanes) { l->arena_scale = ctx.arena_scale; l->arena_class = ctx.arena_class; }
    Pool::Hot keep_workers_awake;
    ctx.ws->stage.abort();
    for (Ctx *lane : ctx.ws->lanes) lane->ws->stage.abort();
    DpSpans spans;
    if (!ctx.ws->ev_base) MB_HIP(hipEventCreate(&ctx.ws->ev_base));
    spans.base = ctx.ws->ev_base;
    MB_HIP(hipEventRecord(spans.base, ctx.stream));
    struct SpanScope {                                                  // the contexts of the call report to `spans` while it lives
        Ctx &c;
        SpanScope(Ctx &cx, DpSpans *sp) : c(cx) { c.spans = sp; for (Ctx *l : c.ws->lanes) l->spans = sp; }
        ~SpanScope() { c.spans = nullptr; for (Ctx *l : c.ws->lanes) l->spans = nullptr; }
    } span_scope(ctx, &spans);
    miblast_params p = pin;
    if (p.gappedthresh < 0) p.gappedthresh = p.hspthresh;
    if (p.step < 1) p.step = 1;
    if (p.query_softmask || p.step_origin) {
        set_error("query_softmask / step_origin (SURVEY A.9 #2, #5) are switches of the CPU oracle only: the MI355X path implements the A.10 reading of both");
        return MIBLAST_EINVAL;
    }
    // (A.9 #9: every ungapped kernel stops a walk at run < best - xdrop; with integers run <= best - xdrop is the same test one lower)
    if (p.xdrop_le) p.xdrop -= 1;
    std::vector<std::unique_ptr<PairJob>> store;
    std::vector<PairJob *> jobs;
    std::vector<Unit> units;
    for (size_t k = 0; k < n; k++) {
        store.emplace_back(new PairJob());
        PairJob &j = *store.back();
        j.T = Ts[k]; j.Q = Qs[k]; j.res = results[k];
        j.defer_host = n > 1;
        jobs.push_back(&j);
    }
    // Several LARGE pairs in one call (chunk pairs of a genome pair: each one's seed stage fills the GPU, a pair with homology fills it
    // with its DP pieces too): a pair's whole job -- seed stage, gapped stage -- runs on ONE lane, a few lanes side by side, so that one
    // pair's latency-bound stretches (hand-over checks, tail launches, traceback, the host halves) lie under the other pairs' kernels
    // instead of under nothing: the gapped stage no longer waits for the seed stage of the LAST pair.  MIBLAST_PAIR_PIPELINE=0: seed
    // stages of all pairs first (twelve lanes), then the gapped stages of two groups of pairs (round 3's order).
    // (large: 2 Mb x 2 Mb and more on average -- the sixteen 1 Mb pairs of a batched call fill the GPU only together, in shared DP launches)
    double cells_avg = 0;
    for (size_t k = 0; k < n; k++) cells_avg += (double)Ts[k]->total * (double)Qs[k]->total / (double)n;
    const long pipe_env = env_long("MIBLAST_PAIR_PIPELINE", 1);           // 0 never, 1 large pairs, 2 always (tests)
    const bool pipeline = n > 1 && !pin.walls && !pin.diag_hash16 && (pipe_env == 2 || (pipe_env == 1 && cells_avg >= 4e12));
    const size_t n_lanes = n > 1 ? (size_t)std::min<long>((long)n, std::max(1l, pipeline ? env_long("MIBLAST_PIPELINE_LANES", 6) : env_long("MIBLAST_SEED_LANES", 12))) : 1;
    bool pipelined = false;                                              // set when the lanes below have run the gapped stages as well
    std::vector<char> group_leader;                                      // pipelined: first pair of every group (launch-level figures are counted once per group)
    std::vector<double> lane_gapped;                                     // ... and the gapped stages' wall time per lane
    // MIBLAST_SEED_BATCHED: 1 (default) the seed stages of a call of several pairs share their launches (seed_phase_batched), and so does
    // a single pair that will fit one key buffer with room to spare (half the launches of the pair-by-pair path; chance hits expected
    // from the sizes: 2 strands x word variants x |T| x |Q| / 4^12 -- an 8 Mb pair would count its 10^8 hits only to be sent back);
    // 2: every single pair goes that way (tests); 0: never -- pair by pair on the lanes below
    if (p.diag_hash16 && p.strands) { set_error("--strand=plus / minus together with diag=hash16 is not provided"); return MIBLAST_EINVAL; }
    // (diag=hash16 lives in the shared seed stage only; --strand=plus / minus in the pair-by-pair one: it is the work unit of large chunk pairs)
    const long batched_mode = p.diag_hash16 ? 2 : p.strands ? 0 : env_long("MIBLAST_SEED_BATCHED", 1);
    bool small_single = false;
    if (n == 1) {
        const double expected = 2.0 * (p.transitions ? 1 + kSeedWeight : 1) * (double)Ts[0]->total * (double)Qs[0]->total / (double)kBuckets;
        small_single = expected <= (double)env_long("MIBLAST_HIT_CAP", 32l << 20) / 8.0;
    }
    bool batched_done = false;
    if (n >= 1 && (batched_mode >= 2 || (batched_mode == 1 && (n > 1 || small_single)))) {
        int rc = seed_phase_batched(ctx, p, jobs, batched_done);
        guard::check_all("seed stage (batched)");
        if (rc != MIBLAST_OK) return rc;
        if (!batched_done) for (PairJob *j : jobs) { j->found[0].clear(); j->found[1].clear(); j->units.clear(); }
    }
    if (p.diag_hash16 && !batched_done) {
        set_error("diag=hash16: the call does not fit one seed batch (more hits than MIBLAST_HIT_CAP, more than 128 pairs, or coordinates beyond 31 bits)");
        return MIBLAST_ELIMIT;
    }
    if (batched_done) {
    } else if (n_lanes > 1) {
        // Batched call: the pairs are dealt to a few lanes, each a host thread with its own stream and seed-stage buffers.  A
        // lane runs the device half of a pair's seed stage, then the host half (discovery order, entropy filter, anchors)
        // while the other lanes keep the device busy.  A pair's result does not depend on its lane.
        Workspace &w = *ctx.ws;
        while (w.lanes.size() < n_lanes) { w.lanes.push_back(lane_create(ctx.device, ctx.priority)); w.lanes.back()->ws->share_marks(w); w.lanes.back()->arena_scale = ctx.arena_scale; w.lanes.back()->arena_class = ctx.arena_class; }
        for (Ctx *l : w.lanes) { l->spans = ctx.spans; l->hits_hint = &w.hits_hint; l->ws->share_marks(w); }
        int64_t max_diags = 0;
        for (size_t k = 0; k < n; k++) max_diags = std::max<int64_t>(max_diags, Ts[k]->total + Qs[k]->total);
        int64_t max_q = 0;
        for (size_t k = 0; k < n; k++) max_q = std::max<int64_t>(max_q, Qs[k]->total);
        std::vector<int> lane_rc(n_lanes, MIBLAST_OK);
        std::vector<std::string> lane_err(n_lanes);
        std::vector<std::future<void>> lane_threads;
        // (pipelined: a lane takes the next pair when it is done with one -- which pairs carry the long gapped stages is not known
        //  beforehand, and three of them dealt to one lane is the whole step's time; else the pairs are dealt round robin)
        std::atomic<size_t> next_pair{0};
        // (pipelined, many pairs: a lane takes a GROUP of consecutive pairs -- their seed stages one after the other, then ONE gapped stage
        //  for the group: its DP launches are as long as their longest piece whatever they hold, so the pairs of a group share them instead
        //  of each paying for its own.  Many small pairs only (four and more per lane): 42 pairs on 6 lanes go in groups of 7 -- 185 -> 39 DP
        //  launches per step of the human-mouse stand-in, the DP's busy time 125 -> 39 ms; nine 30 Mb pairs stay single: a group of two would
        //  put two of the three heavy diagonal pairs on one lane.  MIBLAST_PIPELINE_GROUP fixes the size)
        const size_t group = !pipeline ? 1 : (size_t)std::max(1l, env_long("MIBLAST_PIPELINE_GROUP", n >= 4 * n_lanes ? (long)std::min<size_t>(8, (n + n_lanes - 1) / n_lanes) : 1l));
        const size_t n_groups = (n + group - 1) / group;
        reserve_arenas(ctx, std::min(n_lanes, n_groups));          // (nothing of the call is queued yet: the allocations stall nobody)
        group_leader.assign(n, 0); lane_gapped.assign(n_lanes, 0.0);
        for (size_t lane = 0; lane < n_lanes; lane++)
            lane_threads.push_back(std::async(std::launch::async, [&, lane] {
                try {
                    MB_HIP(hipSetDevice(ctx.device));
                    Ctx &lc = *w.lanes[lane];
                    presize_lane(lc, max_diags, max_q);
                    for (size_t k0 = pipeline ? next_pair.fetch_add(group) : lane; k0 < n; k0 = pipeline ? next_pair.fetch_add(group) : k0 + n_lanes) {
                        std::vector<size_t> mem;
                        for (size_t k = k0; k < std::min(n, k0 + group); k++) {
                            PairJob &j = *jobs[k];
                            int rc = seed_phase(lc, p, j);
                            guard::check_all("seed stage (lane)");
                            if (rc != MIBLAST_OK) { lane_rc[lane] = rc; lane_err[lane] = last_error_text(); return; }
                            seed_host(p, j, 0); seed_host(p, j, 1); seed_finish(j);
                            build_units(p, j, (int)k, j.units);
                            if (j.anchor_mismatch) { lane_rc[lane] = MIBLAST_EHIP; lane_err[lane] = "MIBLAST_CHECK_ANCHORS: k_hsp_anchor disagrees with the host scan"; return; }
                            mem.push_back(k);
                        }
                        if (!pipeline) continue;
                        // the group's gapped stage, right away, on this lane's stream and workspace (DpProb.pad0 = a pair's index in the call)
                        std::vector<PairPtrs> pp(n);
                        memset(pp.data(), 0, n * sizeof(PairPtrs));
                        std::vector<Unit> gu;
                        for (size_t k : mem) {
                            PairJob &j = *jobs[k];
                            pp[k].tc = j.T->dev(); pp[k].qf = j.qc_d[0]; pp[k].qr = j.qc_d[1];
                            for (Unit &u : j.units) gu.push_back(std::move(u));
                            j.units.clear();
                        }
                        lc.ws->pair_ptrs.ensure(n);
                        lc.ws->stage.h2d(lc.ws->pair_ptrs.p, pp.data(), n * sizeof(PairPtrs), lc.stream);
                        const int rc = gapped_phase(lc, p, jobs, gu, &mem, std::min(n_lanes, n_groups));
                        guard::check_all("gapped stage (lane)");
                        for (Unit &u : gu) jobs[(size_t)u.pair]->units.push_back(std::move(u));          // (back to their pairs, in order)
                        group_leader[mem[0]] = 1; lane_gapped[lane] += jobs[mem[0]]->res->stats.t_gapped;
                        if (rc != MIBLAST_OK) { lane_rc[lane] = rc; lane_err[lane] = last_error_text(); return; }
                    }
                } catch (const HipFailure &e) {
                    lane_rc[lane] = MIBLAST_EHIP;
                    lane_err[lane] = std::string("HIP call did not succeed: ") + e.what + " -> " + hipGetErrorString(e.code);
                } catch (const std::exception &e) {
                    lane_rc[lane] = MIBLAST_EHIP;
                    lane_err[lane] = std::string("internal: ") + e.what();
                }
            }));
        for (auto &f : lane_threads) f.get();
        for (size_t lane = 0; lane < n_lanes; lane++)
            if (lane_rc[lane] != MIBLAST_OK) { set_error(lane_err[lane]); return lane_rc[lane]; }
        pipelined = pipeline;
    } else {
    // Seed stages run back to back on the device; in a batched call the host half of every pair (discovery order, entropy
    // filter, anchors) runs on worker threads meanwhile.
    std::vector<std::future<void>> host_tasks;
    size_t waited = 0;
    for (size_t k = 0; k < n; k++) {
        PairJob &j = *jobs[k];
        const double t_a = now_s();
        int rc = seed_phase(ctx, p, j);
        guard::check_all("seed stage");
        if (rc != MIBLAST_OK) { for (auto &f : host_tasks) if (f.valid()) f.wait(); return rc; }
        const double t_b = now_s();
        if (j.defer_host) {
            host_tasks.push_back(std::async(std::launch::async, [&p, &j, k] {
                seed_host(p, j, 0); seed_host(p, j, 1); seed_finish(j);
                build_units(p, j, (int)k, j.units);
            }));
            while (host_tasks.size() - waited > 24) host_tasks[waited++].get();
        } else {
            build_units(p, j, (int)k, j.units);
            if (env_long("MIBLAST_DEBUG", 0)) fprintf(stderr, "[miblast] seed phase %.2f ms, build_units %.2f ms\n", (t_b - t_a) * 1e3, (now_s() - t_b) * 1e3);
        }
    }
    for (; waited < host_tasks.size(); waited++) host_tasks[waited].get();
    }
    const double t_call1 = now_s();
    for (size_t k = 0; k < n; k++)
        if (jobs[k]->anchor_mismatch) { set_error("MIBLAST_CHECK_ANCHORS: k_hsp_anchor disagrees with the host scan"); return MIBLAST_EHIP; }
    for (size_t k = 0; k < n; k++) {
        for (Unit &u : jobs[k]->units) units.push_back(std::move(u));
        jobs[k]->units.clear();
    }
    // device table of the pairs' sequence pointers (k_ydrop picks its pair through DpProb.pad0)
    if (!units.empty()) {
        std::vector<PairPtrs> pp(n);
        for (size_t k = 0; k < n; k++) { pp[k].tc = jobs[k]->T->dev(); pp[k].qf = jobs[k]->qc_d[0]; pp[k].qr = jobs[k]->qc_d[1]; }
        ctx.ws->pair_ptrs.ensure(n);
        ctx.ws->stage.h2d(ctx.ws->pair_ptrs.p, pp.data(), n * sizeof(PairPtrs), ctx.stream);          // (in stream order before the first DP launch)
    }
    // A call of several pairs with enough work: the gapped stages of two groups of its pairs run side by side, each on a stream and
    // workspace of its own -- the host work of one group (relay planting, commits, traceback merge) overlaps the kernels of the
    // other, and their latency-bound tail launches share the GPU.  Pairs are independent jobs: results do not depend on the split.
    const size_t gapped_lanes = (size_t)std::min<long>(8, std::max(1l, env_long("MIBLAST_GAPPED_LANES", 2)));
    size_t total_anchors = 0;
    for (const Unit &u : units) total_anchors += u.anchors.size();
    int rc = MIBLAST_OK;
    if (pipelined) {
        // the lanes have run every pair's gapped stage already; launch-level figures of the call = all pairs together (wall time: the call's)
        miblast_stats sum;
        memset(&sum, 0, sizeof sum);
        for (size_t k = 0; k < n; k++) {
            if (!group_leader[k]) continue;
            const miblast_stats &a = jobs[k]->res->stats;
            sum.gapped_rounds = std::max(sum.gapped_rounds, a.gapped_rounds);
            sum.dp_sides_run += a.dp_sides_run; sum.dp_cells_run += a.dp_cells_run; sum.dp_rows_run += a.dp_rows_run;
            sum.t_dp_kernel_ms += a.t_dp_kernel_ms; sum.dp_kernel_launches += a.dp_kernel_launches;
            sum.relay_accepted += a.relay_accepted; sum.relay_rejected += a.relay_rejected; sum.relay_inline_checks += a.relay_inline_checks; sum.relay_inline_continued += a.relay_inline_continued; sum.dp_reruns += a.dp_reruns;
            sum.t_traceback_ms += a.t_traceback_ms; sum.t_merge_ms += a.t_merge_ms;
        }
        // (the call's gapped wall time: the lanes run side by side -- the longest lane's, as the grouped branch below takes the longest group's)
        for (double t : lane_gapped) sum.t_gapped = std::max(sum.t_gapped, t);
        for (PairJob *j : jobs) {
            miblast_stats &d = j->res->stats;
            d.t_gapped = sum.t_gapped; d.gapped_rounds = sum.gapped_rounds; d.dp_sides_run = sum.dp_sides_run; d.dp_cells_run = sum.dp_cells_run;
            d.dp_rows_run = sum.dp_rows_run; d.t_dp_kernel_ms = sum.t_dp_kernel_ms; d.dp_kernel_launches = sum.dp_kernel_launches;
            d.relay_accepted = sum.relay_accepted; d.relay_rejected = sum.relay_rejected; d.relay_inline_checks = sum.relay_inline_checks; d.relay_inline_continued = sum.relay_inline_continued; d.dp_reruns = sum.dp_reruns;
            d.t_traceback_ms = sum.t_traceback_ms; d.t_merge_ms = sum.t_merge_ms;
        }
    } else
    if (gapped_lanes > 1 && !p.walls && n >= 4 && total_anchors >= (size_t)env_long("MIBLAST_GAPPED_LANES_MIN_ANCHORS", 4096)) {
        // pairs dealt to the groups heaviest first (anchors as the weight)
        const size_t L = std::min(gapped_lanes, n / 2);
        std::vector<size_t> weight(n, 0), order(n);
        for (const Unit &u : units) weight[(size_t)u.pair] += u.anchors.size();
        for (size_t k = 0; k < n; k++) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weight[a] > weight[b]; });
        std::vector<std::vector<size_t>> members(L);
        std::vector<size_t> load(L, 0);
        std::vector<size_t> group_of(n, 0);
        for (size_t k : order) {
            size_t gsel = 0;
            for (size_t x = 1; x < L; x++) if (load[x] < load[gsel]) gsel = x;
            members[gsel].push_back(k); load[gsel] += weight[k] + 1; group_of[k] = gsel;
        }
        for (auto &m : members) std::sort(m.begin(), m.end());
        std::vector<std::vector<Unit>> gunits(L);
        for (Unit &u : units) gunits[group_of[(size_t)u.pair]].push_back(std::move(u));
        units.clear();
        Workspace &w0 = *ctx.ws;
        while (w0.lanes.size() + 1 < L) { w0.lanes.push_back(lane_create(ctx.device, ctx.priority)); w0.lanes.back()->ws->share_marks(w0); w0.lanes.back()->arena_scale = ctx.arena_scale; w0.lanes.back()->arena_class = ctx.arena_class; }
        for (Ctx *l : w0.lanes) l->spans = ctx.spans;
        std::vector<PairPtrs> pp(n);
        for (size_t k = 0; k < n; k++) { pp[k].tc = jobs[k]->T->dev(); pp[k].qf = jobs[k]->qc_d[0]; pp[k].qr = jobs[k]->qc_d[1]; }
        std::vector<int> lane_rc(L, MIBLAST_OK);
        std::vector<std::string> lane_err(L);
        std::vector<std::future<void>> others;
        for (size_t x = 1; x < L; x++) {
            Ctx &lane = *w0.lanes[x - 1];
            lane.ws->stage.abort();
            lane.ws->pair_ptrs.ensure(n);
            lane.ws->stage.h2d(lane.ws->pair_ptrs.p, pp.data(), n * sizeof(PairPtrs), lane.stream);
            others.push_back(std::async(std::launch::async, [&, x] {
                try {
                    MB_HIP(hipSetDevice(ctx.device));
                    lane_rc[x] = gapped_phase(*w0.lanes[x - 1], p, jobs, gunits[x], &members[x]);
                    guard::check_all("gapped stage (part)");
                    if (lane_rc[x] != MIBLAST_OK) lane_err[x] = last_error_text();
                } catch (const HipFailure &e) {
                    lane_rc[x] = MIBLAST_EHIP;
                    lane_err[x] = std::string("HIP call did not succeed: ") + e.what + " -> " + hipGetErrorString(e.code);
                } catch (const std::exception &e) {
                    lane_rc[x] = MIBLAST_EHIP;
                    lane_err[x] = std::string("internal: ") + e.what();
                }
            }));
        }
        try {
            rc = gapped_phase(ctx, p, jobs, gunits[0], &members[0]);
            guard::check_all("gapped stage (first part)");
        } catch (...) {
            for (auto &f : others) f.get();
            throw;
        }
        for (auto &f : others) f.get();
        for (size_t x = 1; x < L && rc == MIBLAST_OK; x++) if (lane_rc[x] != MIBLAST_OK) { set_error(lane_err[x]); rc = lane_rc[x]; }
        for (auto &gu : gunits) for (Unit &u : gu) units.push_back(std::move(u));
        if (rc == MIBLAST_OK) {
            // launch-level figures of the call = all groups together (wall time: the longest one)
            miblast_stats sum;
            memset(&sum, 0, sizeof sum);
            for (size_t x = 0; x < L; x++) {
                const miblast_stats &a = jobs[members[x][0]]->res->stats;
                sum.t_gapped = std::max(sum.t_gapped, a.t_gapped); sum.gapped_rounds = std::max(sum.gapped_rounds, a.gapped_rounds);
                sum.dp_sides_run += a.dp_sides_run; sum.dp_cells_run += a.dp_cells_run; sum.dp_rows_run += a.dp_rows_run;
                sum.t_dp_kernel_ms += a.t_dp_kernel_ms; sum.dp_kernel_launches += a.dp_kernel_launches;
                sum.relay_accepted += a.relay_accepted; sum.relay_rejected += a.relay_rejected; sum.relay_inline_checks += a.relay_inline_checks; sum.relay_inline_continued += a.relay_inline_continued; sum.dp_reruns += a.dp_reruns;
                sum.t_traceback_ms += a.t_traceback_ms; sum.t_merge_ms += a.t_merge_ms;
            }
            for (PairJob *j : jobs) {
                miblast_stats &d = j->res->stats;
                d.t_gapped = sum.t_gapped; d.gapped_rounds = sum.gapped_rounds; d.dp_sides_run = sum.dp_sides_run; d.dp_cells_run = sum.dp_cells_run;
                d.dp_rows_run = sum.dp_rows_run; d.t_dp_kernel_ms = sum.t_dp_kernel_ms; d.dp_kernel_launches = sum.dp_kernel_launches;
                d.relay_accepted = sum.relay_accepted; d.relay_rejected = sum.relay_rejected; d.relay_inline_checks = sum.relay_inline_checks; d.relay_inline_continued = sum.relay_inline_continued; d.dp_reruns = sum.dp_reruns;
                d.t_traceback_ms = sum.t_traceback_ms; d.t_merge_ms = sum.t_merge_ms;
            }
        }
    } else {
        rc = gapped_phase(ctx, p, jobs, units);
        guard::check_all("gapped stage");
    }
    if (rc != MIBLAST_OK) return rc;
    {
        for (Ctx *l : ctx.ws->lanes) l->spans = &spans;                  // (lanes created during the call)
        const double busy = spans.busy_ms();
        for (PairJob *j : jobs) j->res->stats.t_dp_busy_ms = busy;
    }
    const double t_o = now_s();
    {
        std::vector<OutputJob> ojs(n);
        const double t_o0 = now_s();
        parallel_for(n, [&](size_t k) { output_collect(*jobs[k], (int)k, units, ojs[k]); });
        const double t_o1 = now_s();
        std::vector<std::pair<size_t, size_t>> chunks;                   // (pair, chunk of its cigar text)
        for (size_t k = 0; k < n; k++) for (size_t ti = 0; ti < ojs[k].ctasks.size(); ti++) chunks.emplace_back(k, ti);
        parallel_for(chunks.size(), [&](size_t x) { output_cigar_chunk(*jobs[chunks[x].first]->res, ojs[chunks[x].first].ctasks[chunks[x].second]); });
        const double t_o2 = now_s();
        parallel_for(n, [&](size_t k) { output_layout(p, *jobs[k], ojs[k]); });
        if (env_long("MIBLAST_DEBUG", 0))
            fprintf(stderr, "[miblast] output: collect %.2f ms, %zu chunks of cigar text %.2f ms, layout %.2f ms\n", (t_o1 - t_o0) * 1e3, chunks.size(), (t_o2 - t_o1) * 1e3, (now_s() - t_o2) * 1e3);
    }
    if (env_long("MIBLAST_DEBUG", 0))
        fprintf(stderr, "[miblast] call of %zu pairs: seed stages %.2f ms, gapped stage %.2f ms, output %.2f ms, all %.2f ms\n", n, (t_call1 - t_call0) * 1e3, (t_o - t_call1) * 1e3,
                (now_s() - t_o) * 1e3, (now_s() - t_call0) * 1e3);
    return MIBLAST_OK;
}

int align_pairs(Ctx &ctx, const SeqSet *const *Ts, const SeqSet *const *Qs, size_t n, const miblast_params &pin, Result **results) {
    const int rc = align_pairs_impl(ctx, Ts, Qs, n, pin, results);
    guard::check_all("end of a call");                     // (MIBLAST_DEBUG_GUARD; nothing otherwise)
    return rc;
}

int align(Ctx &ctx, const SeqSet &T, const SeqSet &Q, const miblast_params &pin, Result &res) {
    const SeqSet *tp = &T, *qp = &Q;
    Result *rp = &res;
    return align_pairs(ctx, &tp, &qp, 1, pin, &rp);
}

}  // namespace mb

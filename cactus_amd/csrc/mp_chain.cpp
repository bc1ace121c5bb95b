// mp_chain.cpp -- host side and C ABI (include/mipaf.h) of the chaining stage: PAF records in and out, the orchestration
// of the sorts and of k_chain_dp / k_tile / k_trim (mp_kernels.hip), and the O(n) bookkeeping between them (group
// boundaries, peeling chains off in score order, splicing trimmed op lists).  Replaces the paffy sub-commands of
// /root/reference/src/cactus/paf/local_alignment.py:607-727; rules in DESIGN.md section 11.  No CPU path for the
// three compute steps: without a device context they cannot be called at all.
#include "mp_common.h"
#include "mb_pipeline.h"
#include "mb_guard.h"

#include "../../include/mipaf.h"

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>
#include <unistd.h>
#include <vector>

namespace mb {

struct PafRec {
    uint32_t qn = 0, tn = 0;              // name ids
    int64_t ql = 0, qs = 0, qe = 0, tl = 0, ts = 0, te = 0, nm = 0, nb = 0, mq = 0;
    uint8_t same = 1, has_as = 0, has_cg = 0;
    char tp = 0;
    int64_t as = 0, tile = -1, cn = -1, s1 = -1;
    uint64_t ops_off = 0;                 // the record's ops: PafSet::ops[ops_off .. ops_off + n_ops)
    uint32_t n_ops = 0;
};

struct PafSet {
    std::vector<std::string> names;
    std::unordered_map<std::string, uint32_t> name_id;
    std::vector<PafRec> recs;
    std::vector<uint32_t> ops;
    uint32_t intern(const char *s, size_t n) {
        std::string key(s, n);
        auto it = name_id.find(key);
        if (it != name_id.end()) return it->second;
        uint32_t id = (uint32_t)names.size();
        names.push_back(key);
        name_id.emplace(std::move(key), id);
        return id;
    }
};

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

long env_long_mp(const char *name, long dflt) {
    const char *v = getenv(name);
    return v && *v ? atol(v) : dflt;
}

// Device buffers of the chaining stage are kept by the context between calls: a job makes ~60 temporaries and hipMalloc /
// hipFree (which synchronises) cost more than its kernels.  get() hands out the smallest free buffer that fits without
// wasting more than 4x, else allocates; buffers return on destruction of their Dev<>.  One job at a time per context.
struct DevCache {
    struct Slot { void *p; size_t cap; bool used; };
    std::vector<Slot> slots;
    size_t total = 0;
    void *get(size_t bytes, size_t &cap) {
        if (mb::guard::on()) {                               // MIBLAST_DEBUG_GUARD: exact size, canary behind it, nothing reused
            void *p = nullptr;
            MB_HIP(mb::guard::alloc(&p, bytes, "DevCache::get"));
            slots.push_back(Slot{p, bytes, true});
            total += bytes; cap = bytes;
            return p;
        }
        bytes = std::max<size_t>(bytes, 256);
        int best = -1;
        for (size_t i = 0; i < slots.size(); i++)
            if (!slots[i].used && slots[i].cap >= bytes && slots[i].cap <= 4 * bytes + (1u << 20) && (best < 0 || slots[i].cap < slots[(size_t)best].cap)) best = (int)i;
        if (best >= 0) { slots[(size_t)best].used = true; cap = slots[(size_t)best].cap; return slots[(size_t)best].p; }
        const size_t want = ((bytes + bytes / 4 + 0xffff) >> 16) << 16;
        void *p = nullptr;
        mb::count_device_alloc();
        MB_HIP(hipMalloc(&p, want));
        slots.push_back(Slot{p, want, true});
        total += want;
        cap = want;
        return p;
    }
    void put(void *p) {
        if (mb::guard::on()) {
            for (size_t i = 0; i < slots.size(); i++) if (slots[i].p == p) { total -= slots[i].cap; slots.erase(slots.begin() + (long)i); break; }
            mb::guard::free(p, "DevCache::put");
            return;
        }
        for (Slot &s : slots) if (s.p == p) { s.used = false; return; }
    }
    void trim(size_t keep_bytes) {                            // after a job: give back what is idle beyond the budget
        for (size_t i = slots.size(); i-- > 0 && total > keep_bytes;)
            if (!slots[i].used) { mb::count_device_alloc(); (void)hipFree(slots[i].p); total -= slots[i].cap; slots.erase(slots.begin() + (long)i); }
    }
    ~DevCache() { for (Slot &s : slots) { if (mb::guard::on()) mb::guard::free(s.p, "~DevCache"); else (void)hipFree(s.p); } }
};
thread_local DevCache *g_cache = nullptr;
struct UseCache {                          // the calling thread's Dev<> objects draw from this context's cache
    DevCache *prev;
    explicit UseCache(Ctx &ctx) : prev(g_cache) {
        if (!ctx.chain_cache) ctx.chain_cache = new DevCache();
        g_cache = (DevCache *)ctx.chain_cache;
    }
    ~UseCache() {
        mb::guard::check_all("end of a chaining-stage call");
        if (g_cache) g_cache->trim((size_t)std::max(0l, env_long_mp("MIPAF_CACHE_MB", 8192)) << 20);
        g_cache = prev;
    }
};

template <typename T>
struct Dev {                               // device array with the lifetime of one call
    T *p = nullptr;
    size_t n = 0;
    DevCache *from = nullptr;
    Dev() = default;
    explicit Dev(size_t count) { alloc(count); }
    Dev(const Dev &) = delete;
    Dev &operator=(const Dev &) = delete;
    ~Dev() {
        if (!p) return;
        if (from) from->put(p); else if (mb::guard::on()) mb::guard::free(p, "~Dev"); else (void)hipFree(p);
    }
    void alloc(size_t count) {
        n = count;
        const size_t bytes = std::max<size_t>(1, count) * sizeof(T);
        if (g_cache) { size_t cap; p = (T *)g_cache->get(bytes, cap); from = g_cache; }
        else if (mb::guard::on()) MB_HIP(mb::guard::alloc((void **)&p, bytes, "Dev::alloc"));
        else MB_HIP(hipMalloc((void **)&p, bytes));
    }
    void upload(const std::vector<T> &v, hipStream_t s) {
        if (!p) alloc(v.size());
        if (!v.empty()) MB_HIP(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(std::vector<T> &v, hipStream_t s) const {
        v.resize(n);
        if (n) MB_HIP(hipMemcpyAsync(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost, s));
    }
};

struct EventTimer {                        // HIP-event time of a stretch of the stream
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    explicit EventTimer(hipStream_t st) : s(st) {
        MB_HIP(hipEventCreate(&a));
        MB_HIP(hipEventCreate(&b));
        MB_HIP(hipEventRecord(a, s));
    }
    ~EventTimer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    double stop_ms() {
        MB_HIP(hipEventRecord(b, s));
        MB_HIP(hipEventSynchronize(b));
        float ms = 0;
        MB_HIP(hipEventElapsedTime(&ms, a, b));
        return ms;
    }
};

struct PhaseLog {                          // MIPAF_DEBUG=1: host wall time per phase of a call, on stderr
    const char *what;
    bool on;
    double t0, last;
    explicit PhaseLog(const char *w) : what(w), on(getenv("MIPAF_DEBUG") != nullptr), t0(now_s()), last(t0) {}
    void mark(const char *phase) {
        if (!on) return;
        const double t = now_s();
        fprintf(stderr, "[mipaf] %s: %-28s %7.2f ms\n", what, phase, (t - last) * 1e3);
        last = t;
    }
    ~PhaseLog() { if (on) fprintf(stderr, "[mipaf] %s: total %.2f ms\n", what, (now_s() - t0) * 1e3); }
};

int bits_for(unsigned long long mx) {
    int b = 1;
    while (b < 64 && (mx >> b)) b++;
    return b;
}

// ---- text -> records -----------------------------------------------------------------------------------------------
bool parse_i64(const char *b, const char *e, int64_t &v) {
    if (b == e) return false;
    auto r = std::from_chars(b, e, v);
    return r.ec == std::errc() && r.ptr == e;
}

int parse_line(PafSet &set, const char *b, const char *e, std::string &err) {
    const char *col[13];
    const char *end[13];
    int n = 0;
    const char *p = b;
    const char *tags = nullptr;
    while (n < 12) {
        const char *t = (const char *)memchr(p, '\t', (size_t)(e - p));
        col[n] = p;
        end[n] = t ? t : e;
        n++;
        if (!t) { p = e; break; }
        p = t + 1;
        if (n == 12) tags = p;
    }
    auto bad = [&](const char *what) {
        err = what;
        return MIBLAST_EINVAL;
    };
    if (n < 12) return bad("fewer than 12 columns");
    PafRec r;
    r.qn = set.intern(col[0], (size_t)(end[0] - col[0]));
    r.tn = set.intern(col[5], (size_t)(end[5] - col[5]));
    int64_t *num[] = {&r.ql, &r.qs, &r.qe, nullptr, nullptr, &r.tl, &r.ts, &r.te, &r.nm, &r.nb, &r.mq};
    for (int k = 1; k < 12; k++) {
        if (k == 4 || k == 5) continue;
        if (!parse_i64(col[k], end[k], *num[k - 1])) return bad("a numeric column does not parse");
    }
    if (end[4] - col[4] != 1 || (col[4][0] != '+' && col[4][0] != '-')) return bad("strand is neither + nor -");
    r.same = col[4][0] == '+';
    for (p = tags; p && p < e;) {
        const char *t = (const char *)memchr(p, '\t', (size_t)(e - p));
        const char *te = t ? t : e;
        if (te - p >= 5) {
            const char *val = p + 5;
            if (!memcmp(p, "tp:A:", 5)) r.tp = val < te ? *val : 0;
            else if (!memcmp(p, "AS:i:", 5)) { if (!parse_i64(val, te, r.as)) return bad("AS:i: does not parse"); r.has_as = 1; }
            else if (!memcmp(p, "tl:i:", 5)) { if (!parse_i64(val, te, r.tile)) return bad("tl:i: does not parse"); }
            else if (!memcmp(p, "cn:i:", 5)) { if (!parse_i64(val, te, r.cn)) return bad("cn:i: does not parse"); }
            else if (!memcmp(p, "s1:i:", 5)) { if (!parse_i64(val, te, r.s1)) return bad("s1:i: does not parse"); }
            else if (!memcmp(p, "cg:Z:", 5)) {
                r.has_cg = 1;
                r.ops_off = set.ops.size();
                // (an op is two characters at least: room for the most there can be, written through a pointer, trimmed afterwards)
                const size_t room = (size_t)(te - val) / 2 + 1;
                if (set.ops.capacity() - set.ops.size() < room) set.ops.reserve(std::max(2 * set.ops.capacity(), set.ops.size() + room));
                set.ops.resize(r.ops_off + room);
                uint32_t *w = set.ops.data() + r.ops_off;
                static const struct OpCode { uint8_t of[256]; OpCode() { memset(of, 99, sizeof of); of[(int)'='] = kOpEq; of[(int)'X'] = kOpX; of[(int)'M'] = kOpM; of[(int)'I'] = kOpI; of[(int)'D'] = kOpD; } } op_code;
                const char *why = nullptr;
                for (const char *c = val; c < te;) {
                    unsigned d = (unsigned)(*c - '0');
                    if (d > 9u) { why = "cigar: a length is expected"; break; }
                    uint64_t len = d;
                    c++;
                    while (c < te && (d = (unsigned)(*c - '0')) <= 9u) { len = len * 10 + d; c++; if (len >= (1u << 28)) break; }
                    if (len >= (1u << 28)) { why = "cigar: op longer than 2^28"; break; }
                    if (c == te) { why = "cigar: op letter missing"; break; }
                    const uint32_t code = op_code.of[(unsigned char)*c];
                    if (code == 99u || len == 0) { why = "cigar: unknown op or zero length"; break; }
                    c++;
                    *w++ = (uint32_t)(len << 3) | code;
                }
                set.ops.resize((size_t)(w - set.ops.data()));
                if (why) return bad(why);
                r.n_ops = (uint32_t)(set.ops.size() - r.ops_off);
            }
        }
        p = t ? t + 1 : e;
    }
    set.recs.push_back(r);
    return MIBLAST_OK;
}

// parses text[0, len) into `set`; on a malformed line returns its 1-based number (within this text) in bad_line
int parse_chunk(PafSet &set, const char *text, size_t len, size_t &bad_line, std::string &err) {
    size_t line_no = 0;
    for (const char *p = text, *end = text + len; p < end;) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *e = nl ? nl : end;
        const char *le = e;
        while (le > p && (le[-1] == '\r' || le[-1] == '\n')) le--;
        line_no++;
        if (le > p) {
            int rc = parse_line(set, p, le, err);
            if (rc != MIBLAST_OK) { bad_line = line_no; return rc; }
        }
        p = nl ? nl + 1 : end;
    }
    return MIBLAST_OK;
}

// Large inputs are cut at line ends into one chunk per worker thread, parsed into private sets and merged in order (names
// re-interned, op offsets shifted): the result is the same as one pass over the text.
int parse_text(PafSet &set, const char *text, size_t len) {
    const size_t kMinChunk = 1u << 19;
    size_t parts = std::min<size_t>((size_t)std::max(1, host_threads()), len / kMinChunk);
    if (parts <= 1) {
        size_t bad = 0;
        std::string err;
        int rc = parse_chunk(set, text, len, bad, err);
        if (rc != MIBLAST_OK) set_error("PAF line " + std::to_string(bad) + ": " + err);
        return rc;
    }
    std::vector<size_t> cut(parts + 1, len);
    cut[0] = 0;
    for (size_t k = 1; k < parts; k++) {
        size_t at = std::max(cut[k - 1], len * k / parts);
        const char *nl = at < len ? (const char *)memchr(text + at, '\n', len - at) : nullptr;
        cut[k] = nl ? (size_t)(nl - text) + 1 : len;
    }
    std::vector<PafSet> sub(parts);
    std::vector<int> rc(parts, MIBLAST_OK);
    std::vector<size_t> bad(parts, 0);
    std::vector<std::string> err(parts);
    host_parallel_for(parts, [&](size_t k) { rc[k] = parse_chunk(sub[k], text + cut[k], cut[k + 1] - cut[k], bad[k], err[k]); });
    for (size_t k = 0; k < parts; k++)
        if (rc[k] != MIBLAST_OK) {
            size_t before = 0;
            for (const char *p = text; (p = (const char *)memchr(p, '\n', (size_t)(text + cut[k] - p))) != nullptr; p++) before++;
            set_error("PAF line " + std::to_string(before + bad[k]) + ": " + err[k]);
            return rc[k];
        }
    size_t n_recs = set.recs.size(), n_ops = set.ops.size();
    for (const PafSet &p : sub) { n_recs += p.recs.size(); n_ops += p.ops.size(); }
    set.recs.reserve(n_recs);
    set.ops.reserve(n_ops);
    for (PafSet &p : sub) {
        std::vector<uint32_t> remap(p.names.size());
        for (size_t i = 0; i < p.names.size(); i++) remap[i] = set.intern(p.names[i].data(), p.names[i].size());
        const uint64_t shift = set.ops.size();
        set.ops.insert(set.ops.end(), p.ops.begin(), p.ops.end());
        for (PafRec r : p.recs) {
            r.qn = remap[r.qn]; r.tn = remap[r.tn];
            if (r.has_cg) r.ops_off += shift;
            set.recs.push_back(r);
        }
    }
    return MIBLAST_OK;
}

// ---- records -> text -----------------------------------------------------------------------------------------------
void put_i64(std::string &out, int64_t v) {
    char buf[24];
    auto r = std::to_chars(buf, buf + sizeof buf, v);
    out.append(buf, (size_t)(r.ptr - buf));
}

void format_rec(const PafSet &set, const PafRec &r, std::string &out) {
    static const char kOpCh[5] = {'=', 'X', 'M', 'I', 'D'};
    out += set.names[r.qn]; out += '\t';
    put_i64(out, r.ql); out += '\t'; put_i64(out, r.qs); out += '\t'; put_i64(out, r.qe); out += '\t';
    out += r.same ? '+' : '-'; out += '\t';
    out += set.names[r.tn]; out += '\t';
    put_i64(out, r.tl); out += '\t'; put_i64(out, r.ts); out += '\t'; put_i64(out, r.te); out += '\t';
    put_i64(out, r.nm); out += '\t'; put_i64(out, r.nb); out += '\t'; put_i64(out, r.mq);
    if (r.tp || r.tile != -1) { out += "\ttp:A:"; out += r.tp ? r.tp : (r.tile > 1 ? 'S' : 'P'); }
    if (r.has_as) { out += "\tAS:i:"; put_i64(out, r.as); }
    if (r.tile != -1) { out += "\ttl:i:"; put_i64(out, r.tile); }
    if (r.cn != -1) { out += "\tcn:i:"; put_i64(out, r.cn); }
    if (r.s1 != -1) { out += "\ts1:i:"; put_i64(out, r.s1); }
    if (r.has_cg) {
        out += "\tcg:Z:";
        // (the cigars of whole-chunk alignments run to hundreds of thousands of ops, nearly all of one or two digits: those come
        //  from a table, a few thousand characters at a time into the string)
        static const char two[] = "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
                                  "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
        char buf[4096];
        size_t at = 0;
        const uint32_t *ops = set.ops.data() + r.ops_off;
        for (uint32_t k = 0; k < r.n_ops; k++) {
            if (at + 16 > sizeof buf) { out.append(buf, at); at = 0; }
            const uint32_t o = ops[k];
            uint32_t u = o >> 3;
            if (u < 10) buf[at++] = (char)('0' + u);
            else if (u < 100) { buf[at++] = two[2 * u]; buf[at++] = two[2 * u + 1]; }
            else {
                char d[12]; int n = 0;
                do { d[n++] = (char)('0' + u % 10); u /= 10; } while (u);
                while (n) buf[at++] = d[--n];
            }
            buf[at++] = kOpCh[o & 7u];
        }
        out.append(buf, at);
    }
    out += '\n';
}

std::string format_set(const PafSet &set) {
    const size_t n = set.recs.size();
    const size_t parts = std::min<size_t>((size_t)std::max(1, host_threads()), std::max<size_t>(1, n / 2048));
    std::vector<std::string> piece(parts);
    host_parallel_for(parts, [&](size_t k) {
        const size_t a = n * k / parts, b = n * (k + 1) / parts;
        size_t ops = 0;
        for (size_t i = a; i < b; i++) ops += set.recs[i].n_ops;
        piece[k].reserve((b - a) * 112 + ops * 5);
        for (size_t i = a; i < b; i++) format_rec(set, set.recs[i], piece[k]);
    });
    if (parts == 1) return std::move(piece[0]);
    std::vector<size_t> at(parts + 1, 0);
    for (size_t k = 0; k < parts; k++) at[k + 1] = at[k] + piece[k].size();
    std::string out(at[parts], '\0');
    host_parallel_for(parts, [&](size_t k) { memcpy(&out[at[k]], piece[k].data(), piece[k].size()); });
    return out;
}

int write_all(int fd, const char *p, size_t n) {
    while (n) {
        ssize_t w = write(fd, p, n);
        if (w < 0) { set_error("write failed"); return MIBLAST_EIO; }
        p += w; n -= (size_t)w;
    }
    return MIBLAST_OK;
}

// a cigar must walk exactly the intervals of its record (the contract caf asserts later)
int check_cigars(const PafSet &set) {
    const size_t n = set.recs.size();
    const size_t parts = std::min<size_t>((size_t)std::max(1, host_threads()), std::max<size_t>(1, n / 4096));
    std::vector<size_t> first_bad(parts, SIZE_MAX);
    host_parallel_for(parts, [&](size_t k) {
        for (size_t i = n * k / parts, e = n * (k + 1) / parts; i < e; i++) {
            const PafRec &r = set.recs[i];
            bool ok = r.qs >= 0 && r.qs <= r.qe && r.qe <= r.ql && r.ts >= 0 && r.ts <= r.te && r.te <= r.tl;
            if (ok && r.has_cg) {
                int64_t q = 0, t = 0;
                for (uint32_t o = 0; o < r.n_ops; o++) {
                    const uint32_t op = set.ops[r.ops_off + o];
                    if ((op & 7u) != kOpD) q += op >> 3;
                    if ((op & 7u) != kOpI) t += op >> 3;
                }
                ok = q == r.qe - r.qs && t == r.te - r.ts;
            }
            if (!ok) { first_bad[k] = i; return; }
        }
    });
    for (size_t k = 0; k < parts; k++)
        if (first_bad[k] != SIZE_MAX) {
            set_error("PAF record " + std::to_string(first_bad[k] + 1) + ": coordinates and cigar do not agree");
            return MIBLAST_EINVAL;
        }
    return MIBLAST_OK;
}

int64_t record_score(const PafRec &r) { return r.has_as ? r.as : r.nm; }

// ---- paffy chain ---------------------------------------------------------------------------------------------------
void chain(Ctx &ctx, PafSet &set, const mipaf_chain_params &cp, mipaf_stats &st) {
    const size_t n = set.recs.size();
    st.records = (int64_t)n;
    if (n == 0) return;
    if (n >= (1ull << 31)) throw std::length_error("more than 2^31 PAF records in one chaining job");
    MB_HIP(hipSetDevice(ctx.device));
    hipStream_t s = ctx.stream;
    UseCache use_cache(ctx);
    PhaseLog log("chain");
    // R-C1: names compare as byte strings; a group = (query, target, strand), numbered in that order
    std::vector<uint32_t> by_name(set.names.size()), name_rank(set.names.size());
    for (uint32_t i = 0; i < by_name.size(); i++) by_name[i] = i;
    std::sort(by_name.begin(), by_name.end(), [&](uint32_t a, uint32_t b) { return strcmp(set.names[a].c_str(), set.names[b].c_str()) < 0; });
    for (uint32_t i = 0; i < by_name.size(); i++) name_rank[by_name[i]] = i;
    std::vector<unsigned long long> gkey(n);
    for (size_t i = 0; i < n; i++) {
        const PafRec &r = set.recs[i];
        gkey[i] = ((unsigned long long)name_rank[r.qn] << 33) | ((unsigned long long)name_rank[r.tn] << 1) | (unsigned long long)r.same;
    }
    {   // k_chain_dp packs (value, lane) into 64 bits: chain scores must stay below 2^57
        long double sum = 0;
        for (const PafRec &r : set.recs) sum += (long double)std::max<int64_t>(0, record_score(r));
        if (sum >= 7.2e16L) throw std::length_error("chain scores could exceed 2^56: outside the range of the chain DP");
    }
    std::vector<unsigned long long> uniq(gkey);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    const size_t ng = uniq.size();
    st.groups = (int64_t)ng;
    std::vector<unsigned long long> k_grp(n), k_qs(n), k_ts(n);
    std::vector<ChainRec> crec(n);
    std::vector<uint32_t> gcount(ng + 1, 0);
    std::vector<int64_t> glmax(ng, 0);
    unsigned long long max_qs = 0, max_ts = 0;
    for (size_t i = 0; i < n; i++) {
        const PafRec &r = set.recs[i];
        const size_t g = (size_t)(std::lower_bound(uniq.begin(), uniq.end(), gkey[i]) - uniq.begin());
        k_grp[i] = g; k_qs[i] = (unsigned long long)r.qs; k_ts[i] = (unsigned long long)r.ts;
        max_qs = std::max(max_qs, k_qs[i]); max_ts = std::max(max_ts, k_ts[i]);
        gcount[g + 1]++;
        glmax[g] = std::max(glmax[g], r.qe - r.qs);
        // R-C2 (host arithmetic, the same expression as the oracle's)
        const int64_t tq = (int64_t)((double)(r.qe - r.qs) * cp.trim_fraction / 2.0);
        const int64_t tt = (int64_t)((double)(r.te - r.ts) * cp.trim_fraction / 2.0);
        crec[i] = ChainRec{r.qs, r.qs + tq, r.qe - tq, r.ts + tt, r.te - tt, record_score(r), (int32_t)r.same, 0};
    }
    for (size_t g = 0; g < ng; g++) gcount[g + 1] += gcount[g];

    log.mark("keys, groups, boxes (host)");
    Dev<unsigned long long> d_key(n), d_key2(n), d_ts, d_qs, d_grp;
    Dev<uint32_t> d_pa(n), d_pb(n), d_gstart;
    Dev<ChainRec> d_rec, d_sorted(n);
    Dev<int64_t> d_lmax;
    Dev<long long> d_cs(n), d_tqe(n), d_tend(n);
    Dev<int32_t> d_pred(n);
    size_t temp_bytes = 0;
    for (int bits : {bits_for(max_ts), bits_for(max_qs), bits_for(ng), 64}) temp_bytes = std::max(temp_bytes, sort_pairs_temp_bytes((int64_t)n, bits));
    Dev<uint8_t> d_temp(temp_bytes);
    d_rec.upload(crec, s);
    d_gstart.upload(gcount, s);
    d_lmax.upload(glmax, s);
    d_ts.upload(k_ts, s);
    d_qs.upload(k_qs, s);
    d_grp.upload(k_grp, s);

    log.mark("alloc + upload");
    EventTimer t_sort(s);
    // LSD over the sort keys: target start, then query start, then group; ties keep the input order (the sorts are stable)
    launch_iota(d_pa.p, (int64_t)n, s);
    sort_pairs(d_temp.p, temp_bytes, d_ts.p, d_key2.p, d_pa.p, d_pb.p, (int64_t)n, bits_for(max_ts), s);
    launch_gather_u64(d_qs.p, d_pb.p, d_key.p, (int64_t)n, s);
    sort_pairs(d_temp.p, temp_bytes, d_key.p, d_key2.p, d_pb.p, d_pa.p, (int64_t)n, bits_for(max_qs), s);
    launch_gather_u64(d_grp.p, d_pa.p, d_key.p, (int64_t)n, s);
    sort_pairs(d_temp.p, temp_bytes, d_key.p, d_key2.p, d_pa.p, d_pb.p, (int64_t)n, bits_for(ng), s);
    launch_gather_chain(d_rec.p, d_pb.p, d_sorted.p, d_tqe.p, d_tend.p, (int64_t)n, s);
    st.t_sort_ms += t_sort.stop_ms();

    EventTimer t_dp(s);
    launch_chain_dp(d_sorted.p, d_tqe.p, d_tend.p, d_gstart.p, d_lmax.p, (int)ng, cp.max_gap_length, cp.gap_open, cp.gap_extend, d_cs.p, d_pred.p,
                    (int)env_long_mp("MIPAF_CHAIN_THREADS", 1024), s);
    st.t_chain_dp_ms += t_dp.stop_ms();

    // R-C6 order: chain score descending, R-C1 position on ties
    EventTimer t_sort2(s);
    launch_desc_keys(d_cs.p, d_key.p, (int64_t)n, s);
    launch_iota(d_pa.p, (int64_t)n, s);
    Dev<uint32_t> d_rank(n);
    sort_pairs(d_temp.p, temp_bytes, d_key.p, d_key2.p, d_pa.p, d_rank.p, (int64_t)n, 64, s);
    st.t_sort_ms += t_sort2.stop_ms();

    std::vector<uint32_t> order, rank;
    std::vector<long long> cs;
    std::vector<int32_t> pred;
    d_pb.download(order, s);                                // order[k] = input index of the k-th record in R-C1 order
    d_rank.download(rank, s);
    d_cs.download(cs, s);
    d_pred.download(pred, s);
    MB_HIP(hipStreamSynchronize(s));

    log.mark("sorts + dp + download");
    // peel the chains off (R-C6): sequential by nature, O(n)
    std::vector<int64_t> cn(n, -1), s1(n, -1);              // by R-C1 position
    int64_t next_id = 0;
    for (size_t k = 0; k < n; k++) {
        uint32_t p = rank[k];
        if (cn[p] != -1) continue;
        const int64_t id = next_id++, score = cs[p];
        for (;;) {
            cn[p] = id; s1[p] = score;
            if (pred[p] < 0 || cn[(size_t)pred[p]] != -1) break;
            p = (uint32_t)pred[p];
        }
    }
    // R-C7: chains by number, members in R-C1 order = a counting sort of the R-C1 positions by chain number
    std::vector<size_t> at((size_t)next_id + 1, 0);
    for (size_t p = 0; p < n; p++) at[(size_t)cn[p] + 1]++;
    for (size_t c = 0; c < (size_t)next_id; c++) at[c + 1] += at[c];
    std::vector<PafRec> out(n);
    for (size_t p = 0; p < n; p++) {
        PafRec r = set.recs[order[p]];
        r.cn = cn[p]; r.s1 = s1[p];
        out[at[(size_t)cn[p]]++] = r;
    }
    set.recs.swap(out);
    log.mark("peel + reorder (host)");
    // pairs (j before i) inside the groups: the upper bound of the candidates the DP inspects (56 B of ChainRec + 8 B of cs each)
    st.chain_pairs = 0;
    for (size_t g = 0; g < ng; g++) { const int64_t m = gcount[g + 1] - gcount[g]; st.chain_pairs += m * (m - 1) / 2; }
}

// ---- paffy tile ----------------------------------------------------------------------------------------------------
// Sort-based levelling (mp_kernels.hip "Tiling without the walk").  rank[k] = input index of the k-th record in R-T1 order;
// seq_off[q] = first global base of query sequence q.  Returns false, leaving `level` alone, when the number of pieces
// would exceed `max_pieces` (deep pile-ups): the caller then takes the counter walk.
bool tile_by_sorting(Ctx &ctx, const PafSet &set, const std::vector<uint32_t> &rank, const std::vector<uint32_t> &qid_of_name,
                     const std::vector<uint64_t> &seq_off, uint64_t max_pieces, std::vector<int32_t> &level_by_rank, mipaf_stats &st) {
    const size_t n = rank.size();
    hipStream_t s = ctx.stream;
    PhaseLog log("tile/sorting");
    // maximal runs of aligned query bases (= X M; a D does not move along the query, an I ends the run)
    std::unique_ptr<unsigned long long[]> bounds;
    std::unique_ptr<uint32_t[]> rrank;
    size_t nr = 0;
    {
        const size_t parts = std::min<size_t>((size_t)std::max(1, host_threads()), std::max<size_t>(1, n / 1024));
        std::vector<std::vector<unsigned long long>> prs(parts), pre(parts);
        std::vector<std::vector<uint32_t>> prk(parts);
        host_parallel_for(parts, [&](size_t part) {
            std::vector<unsigned long long> &lrs = prs[part], &lre = pre[part];
            std::vector<uint32_t> &lrk = prk[part];
            for (size_t k = n * part / parts, k_end = n * (part + 1) / parts; k < k_end; k++) {
                const PafRec &r = set.recs[rank[k]];
                if (!r.has_cg) continue;
                const uint64_t base = seq_off[qid_of_name[r.qn]];
                int64_t q = r.same ? r.qs : r.qe, open_at = -1;
                auto close = [&](int64_t at) {
                    if (open_at < 0) return;
                    const uint64_t a = base + (uint64_t)std::min(open_at, at), b = base + (uint64_t)std::max(open_at, at);
                    if (b > a) { lrs.push_back(a); lre.push_back(b); lrk.push_back((uint32_t)k); }
                    open_at = -1;
                };
                for (uint32_t o = 0; o < r.n_ops; o++) {
                    const uint32_t op = set.ops[r.ops_off + o], code = op & 7u;
                    const int64_t len = op >> 3;
                    if (code == kOpD) continue;
                    if (code == kOpI) { close(q); q += r.same ? len : -len; continue; }
                    if (open_at < 0) open_at = q;
                    q += r.same ? len : -len;
                }
                close(q);
            }
        });
        std::vector<size_t> at(parts + 1, 0);
        for (size_t p = 0; p < parts; p++) at[p + 1] = at[p] + prs[p].size();
        nr = at[parts];
        bounds.reset(new unsigned long long[std::max<size_t>(1, 2 * nr)]);      // starts, then ends: no zero fill, no second copy
        rrank.reset(new uint32_t[std::max<size_t>(1, nr)]);
        host_parallel_for(parts, [&](size_t p) {
            std::copy(prs[p].begin(), prs[p].end(), bounds.get() + at[p]);
            std::copy(pre[p].begin(), pre[p].end(), bounds.get() + nr + at[p]);
            std::copy(prk[p].begin(), prk[p].end(), rrank.get() + at[p]);
        });
    }
    level_by_rank.assign(n, 1);
    if (nr == 0) return true;
    const size_t nb = 2 * nr;
    const int coord_bits = bits_for(seq_off.back());
    log.mark("runs (host)");

    EventTimer t(s);
    Dev<unsigned long long> d_b(nb), d_bs(nb), d_flag(nb), d_pos(nb), d_cnt(nr), d_off(nr);
    Dev<uint32_t> d_lo(nr), d_rrank(nr);
    MB_HIP(hipMemcpyAsync(d_b.p, bounds.get(), nb * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    MB_HIP(hipMemcpyAsync(d_rrank.p, rrank.get(), nr * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    size_t temp_bytes = std::max(sort_keys64_temp_bytes((int64_t)nb, coord_bits), scan_u64_temp_bytes((int64_t)nb));
    Dev<uint8_t> d_temp(temp_bytes);
    log.mark("alloc + upload");
    sort_keys64(d_temp.p, temp_bytes, d_b.p, d_bs.p, (int64_t)nb, coord_bits, s);
    launch_tile_heads(d_bs.p, (int64_t)nb, d_flag.p, s);
    scan_u64(d_temp.p, temp_bytes, d_flag.p, d_pos.p, (int64_t)nb, false, s);
    unsigned long long last_pos = 0, last_flag = 0;
    MB_HIP(hipMemcpyAsync(&last_pos, d_pos.p + (nb - 1), sizeof last_pos, hipMemcpyDeviceToHost, s));
    MB_HIP(hipMemcpyAsync(&last_flag, d_flag.p + (nb - 1), sizeof last_flag, hipMemcpyDeviceToHost, s));
    MB_HIP(hipStreamSynchronize(s));
    const size_t nu = (size_t)(last_pos + last_flag);
    Dev<unsigned long long> d_u(nu);
    launch_tile_unique(d_bs.p, d_flag.p, d_pos.p, (int64_t)nb, d_u.p, s);
    launch_tile_span(d_b.p, d_b.p + nr, (int64_t)nr, d_u.p, (int64_t)nu, d_lo.p, d_cnt.p, s);
    scan_u64(d_temp.p, temp_bytes, d_cnt.p, d_off.p, (int64_t)nr, false, s);
    unsigned long long last_off = 0, last_cnt = 0;
    MB_HIP(hipMemcpyAsync(&last_off, d_off.p + (nr - 1), sizeof last_off, hipMemcpyDeviceToHost, s));
    MB_HIP(hipMemcpyAsync(&last_cnt, d_cnt.p + (nr - 1), sizeof last_cnt, hipMemcpyDeviceToHost, s));
    MB_HIP(hipStreamSynchronize(s));
    const uint64_t np = last_off + last_cnt;
    log.mark("intervals + spans");
    if (np > max_pieces) { st.t_tile_ms += t.stop_ms(); return false; }
    Dev<unsigned long long> d_key(np), d_key_s(np), d_key2(np), d_w64(np), d_wsum(np);
    Dev<uint32_t> d_w(np), d_w_s(np);
    const int piece_bits = 32 + bits_for(nu);
    const int key2_bits = 15 + bits_for(n);
    size_t temp2 = std::max({sort_keys64_temp_bytes((int64_t)np, piece_bits), sort_pairs_temp_bytes((int64_t)np, key2_bits), scan_u64_temp_bytes((int64_t)np)});
    Dev<uint8_t> d_temp2(temp2);
    launch_tile_expand(d_lo.p, d_cnt.p, d_off.p, d_rrank.p, (int64_t)nr, d_key.p, s);
    sort_keys64(d_temp2.p, temp2, d_key.p, d_key_s.p, (int64_t)np, piece_bits, s);
    launch_tile_cover(d_key_s.p, (int64_t)np, d_u.p, d_key2.p, d_w.p, s);
    sort_pairs(d_temp2.p, temp2, d_key2.p, d_key.p, d_w.p, d_w_s.p, (int64_t)np, key2_bits, s);      // d_key now holds the sorted key2
    launch_widen(d_w_s.p, d_w64.p, (int64_t)np, s);
    scan_u64(d_temp2.p, temp2, d_w64.p, d_wsum.p, (int64_t)np, true, s);
    Dev<int32_t> d_level(n);
    launch_tile_median(d_key.p, d_wsum.p, (int64_t)np, (int64_t)n, d_level.p, s);
    st.t_tile_ms += t.stop_ms();
    d_level.download(level_by_rank, s);
    MB_HIP(hipStreamSynchronize(s));
    log.mark("pieces, sorts, median");
    st.ops = (int64_t)np;                                    // pieces: the unit of work of this path
    return true;
}

// The literal counter walk (k_tile): records grouped by query sequence, R-T1 order inside a sequence; level by input index.
void tile_by_walking(Ctx &ctx, const PafSet &set, const std::vector<uint32_t> &rank, const std::vector<uint32_t> &qid_of_name,
                     const std::vector<uint64_t> &seq_off, int hist_bins, std::vector<int32_t> &level, mipaf_stats &st) {
    const size_t n = rank.size(), nq = seq_off.size() - 1;
    hipStream_t s = ctx.stream;
    std::vector<uint32_t> qstart(nq + 1, 0);
    for (size_t k = 0; k < n; k++) qstart[qid_of_name[set.recs[rank[k]].qn] + 1]++;
    for (size_t q = 0; q < nq; q++) qstart[q + 1] += qstart[q];
    // per-op query offsets (op order) and the records as the kernel reads them, by a stable counting sort on the sequence
    std::vector<uint32_t> at(qstart.begin(), qstart.end() - 1), qoff(set.ops.size());
    std::vector<TileRec> trec(n);
    for (size_t k = 0; k < n; k++) {
        const PafRec &r = set.recs[rank[k]];
        uint32_t q = 0;
        if (r.has_cg)
            for (uint32_t o = 0; o < r.n_ops; o++) {
                qoff[r.ops_off + o] = q;
                if ((set.ops[r.ops_off + o] & 7u) != kOpD) q += set.ops[r.ops_off + o] >> 3;
            }
        trec[at[qid_of_name[r.qn]]++] = TileRec{r.ops_off, r.has_cg ? r.n_ops : 0u, (int32_t)r.same, r.qs, r.qe, rank[k], 0u};
    }
    st.ops = (int64_t)set.ops.size();
    Dev<TileRec> d_rec;
    Dev<uint32_t> d_qstart, d_ops, d_qoff;
    Dev<uint64_t> d_cnt_off;
    Dev<uint16_t> d_cnt((size_t)seq_off[nq]);
    Dev<int32_t> d_level(n);
    d_rec.upload(trec, s);
    d_qstart.upload(qstart, s);
    d_cnt_off.upload(seq_off, s);
    d_ops.upload(set.ops, s);
    d_qoff.upload(qoff, s);
    MB_HIP(hipMemsetAsync(d_cnt.p, 0, std::max<size_t>(1, (size_t)seq_off[nq]) * sizeof(uint16_t), s));
    EventTimer t_tile(s);
    launch_tile(d_rec.p, d_qstart.p, d_cnt_off.p, (int)nq, d_cnt.p, d_ops.p, d_qoff.p, hist_bins, d_level.p, s);
    st.t_tile_ms += t_tile.stop_ms();
    d_level.download(level, s);
    MB_HIP(hipStreamSynchronize(s));
}

void tile(Ctx &ctx, PafSet &set, int hist_bins, mipaf_stats &st) {
    const size_t n = set.recs.size();
    st.records = (int64_t)n;
    if (n == 0) return;
    if (n >= (1ull << 31)) throw std::length_error("more than 2^31 PAF records in one tiling job");
    UseCache use_cache(ctx);
    PhaseLog log("tile");
    const bool force_walk = hist_bins > 0;
    hist_bins = force_walk ? std::min(8192, std::max(2, hist_bins)) : 4096;
    MB_HIP(hipSetDevice(ctx.device));
    hipStream_t s = ctx.stream;
    // R-T1 order on the device; query sequences numbered by first appearance, laid end to end in one base coordinate
    std::vector<unsigned long long> key(n);
    std::vector<uint32_t> qid_of_name(set.names.size(), UINT32_MAX);
    std::vector<int64_t> qlen;
    for (size_t i = 0; i < n; i++) {
        const PafRec &r = set.recs[i];
        const int64_t k = r.s1 != -1 ? r.s1 : record_score(r);
        key[i] = ~((unsigned long long)k ^ 0x8000000000000000ull);
        if (qid_of_name[r.qn] == UINT32_MAX) { qid_of_name[r.qn] = (uint32_t)qlen.size(); qlen.push_back(r.ql); }
        qlen[qid_of_name[r.qn]] = std::max(qlen[qid_of_name[r.qn]], r.ql);
    }
    const size_t nq = qlen.size();
    st.query_sequences = (int64_t)nq;
    std::vector<uint64_t> seq_off(nq + 1, 0);
    for (size_t q = 0; q < nq; q++) seq_off[q + 1] = seq_off[q] + (uint64_t)std::max<int64_t>(qlen[q], 0);
    std::vector<uint32_t> rank;
    {
        Dev<unsigned long long> d_key, d_key2(n);
        Dev<uint32_t> d_pa(n), d_rank(n);
        const size_t temp_bytes = sort_pairs_temp_bytes((int64_t)n, 64);
        Dev<uint8_t> d_temp(temp_bytes);
        d_key.upload(key, s);
        EventTimer t_sort(s);
        launch_iota(d_pa.p, (int64_t)n, s);
        sort_pairs(d_temp.p, temp_bytes, d_key.p, d_key2.p, d_pa.p, d_rank.p, (int64_t)n, 64, s);
        st.t_sort_ms += t_sort.stop_ms();
        d_rank.download(rank, s);
        MB_HIP(hipStreamSynchronize(s));
    }
    log.mark("keys + order");

    std::vector<int32_t> level;                              // by input index
    bool done = false;
    if (!force_walk) {
        // pieces beyond this (default 2^28 ~ 7 GiB of keys and weights) mean a pile-up: take the bounded-memory walk instead
        const uint64_t max_pieces = (uint64_t)std::max(1l, env_long_mp("MIPAF_TILE_MAX_PIECES", 1l << 28));
        std::vector<int32_t> by_rank;
        if (tile_by_sorting(ctx, set, rank, qid_of_name, seq_off, max_pieces, by_rank, st)) {
            level.assign(n, 1);
            for (size_t k = 0; k < n; k++) level[rank[k]] = by_rank[k];
            done = true;
        }
    }
    if (!done) tile_by_walking(ctx, set, rank, qid_of_name, seq_off, hist_bins, level, st);
    log.mark("levels");
    std::vector<PafRec> out(n);
    for (size_t k = 0; k < n; k++) {                         // R-T5
        PafRec r = set.recs[rank[k]];
        r.tile = level[rank[k]];
        r.tp = r.tile == 1 ? 'P' : 'S';
        out[k] = r;
    }
    set.recs.swap(out);
}

// ---- paffy trim ----------------------------------------------------------------------------------------------------
bool parse_fraction(const char *s, long long &num, long long &den) {
    num = 0; den = 1;
    int digits = 0;
    bool dot = false;
    if (!s) return false;
    for (; *s; s++) {
        if (*s == '.' && !dot) { dot = true; continue; }
        if (*s < '0' || *s > '9' || ++digits > 6) return false;
        num = num * 10 + (*s - '0');
        if (dot) den *= 10;
    }
    return digits > 0 && num <= den;
}

void trim(Ctx &ctx, PafSet &set, long long num, long long den, mipaf_stats &st) {
    const size_t n = set.recs.size();
    st.records = (int64_t)n;
    std::vector<uint32_t> which;                             // the records with a cigar, by ascending position in the op arena
    std::vector<uint8_t> drop(n, 0);
    for (size_t i = 0; i < n; i++) {
        if (set.recs[i].has_cg && set.recs[i].n_ops) which.push_back((uint32_t)i);
        else if (set.recs[i].has_cg) drop[i] = 1;             // an empty cigar has no column to keep (R-R3)
    }
    std::sort(which.begin(), which.end(), [&](uint32_t a, uint32_t b) { return set.recs[a].ops_off < set.recs[b].ops_off; });
    UseCache use_cache(ctx);
    PhaseLog log("trim");
    std::vector<TrimOut> res;
    if (!which.empty()) {
        MB_HIP(hipSetDevice(ctx.device));
        hipStream_t s = ctx.stream;
        const size_t nr = which.size(), n_ops = set.ops.size();
        std::vector<unsigned long long> rec_start(nr);
        std::vector<uint32_t> rec_n(nr);
        for (size_t k = 0; k < nr; k++) { rec_start[k] = set.recs[which[k]].ops_off; rec_n[k] = set.recs[which[k]].n_ops; }
        Dev<uint32_t> d_ops, d_rec_n;
        Dev<unsigned long long> d_rec_start, d_pc(n_ops + 1), d_pm(n_ops + 1), d_pq(n_ops + 1), d_pt(n_ops + 1), d_pre(nr), d_suf(nr);
        Dev<TrimOut> d_out(nr);
        const size_t temp_bytes = scan_u64_temp_bytes((int64_t)n_ops + 1);
        Dev<uint8_t> d_temp(temp_bytes);
        d_ops.upload(set.ops, s);
        d_rec_start.upload(rec_start, s);
        d_rec_n.upload(rec_n, s);
        st.ops = (int64_t)n_ops;
        EventTimer t(s);
        MB_HIP(hipMemsetAsync(d_pre.p, 0, nr * sizeof(unsigned long long), s));
        MB_HIP(hipMemsetAsync(d_suf.p, 0, nr * sizeof(unsigned long long), s));
        launch_trim_values(d_ops.p, (int64_t)n_ops, d_pc.p, d_pm.p, d_pq.p, d_pt.p, s);
        for (unsigned long long *p : {d_pc.p, d_pm.p, d_pq.p, d_pt.p}) scan_u64(d_temp.p, temp_bytes, p, p, (int64_t)n_ops + 1, false, s);
        launch_trim_ops(d_ops.p, (int64_t)n_ops, d_rec_start.p, d_rec_n.p, (int64_t)nr, d_pc.p, d_pm.p, num, den, d_pre.p, d_suf.p, s);
        launch_trim_finish(d_ops.p, d_rec_start.p, d_rec_n.p, (int64_t)nr, d_pc.p, d_pm.p, d_pq.p, d_pt.p, d_pre.p, d_suf.p, d_out.p, s);
        st.t_trim_ms += t.stop_ms();
        d_out.download(res, s);
        MB_HIP(hipStreamSynchronize(s));
    }
    log.mark("upload + kernels + download");
    for (size_t k = 0; k < res.size(); k++) {
        PafRec &r = set.recs[which[k]];
        const TrimOut &t = res[k];
        if (t.pre + t.suf >= t.cols) { drop[which[k]] = 1; continue; }      // R-R3
        r.nm = t.nm; r.nb = t.nb;                                           // R-R3: columns 10/11 always follow the cigar, trimmed or not
        if (t.pre == 0 && t.suf == 0) continue;
        uint32_t *ops = set.ops.data() + r.ops_off;
        const uint32_t code_f = ops[t.first_op] & 7u, code_l = ops[t.last_op] & 7u;
        if (t.first_op == t.last_op) {
            const uint32_t len = ops[t.first_op] >> 3;
            ops[t.first_op] = ((t.first_len + t.last_len - len) << 3) | code_f;
        } else {
            ops[t.first_op] = (t.first_len << 3) | code_f;
            ops[t.last_op] = (t.last_len << 3) | code_l;
        }
        r.ops_off += t.first_op;
        r.n_ops = t.last_op - t.first_op + 1;
        r.ts += t.ta; r.te -= t.tb;
        if (r.same) { r.qs += t.qa; r.qe -= t.qb; } else { r.qe -= t.qa; r.qs += t.qb; }
        r.nm = t.nm; r.nb = t.nb;
    }
    size_t w = 0;
    for (size_t i = 0; i < n; i++)
        if (!drop[i]) set.recs[w++] = set.recs[i];
    set.recs.resize(w);
}

void invert(PafSet &set) {
    for (PafRec &r : set.recs) {
        std::swap(r.qn, r.tn); std::swap(r.ql, r.tl); std::swap(r.qs, r.ts); std::swap(r.qe, r.te);
        if (!r.has_cg) continue;
        uint32_t *ops = set.ops.data() + r.ops_off;
        if (!r.same) std::reverse(ops, ops + r.n_ops);
        for (uint32_t k = 0; k < r.n_ops; k++) {
            const uint32_t c = ops[k] & 7u;
            if (c == kOpI || c == kOpD) ops[k] = (ops[k] & ~7u) | (c == kOpI ? kOpD : kOpI);
        }
    }
}

void filter(PafSet &set, int64_t max_tile, int64_t min_chain, bool inv) {
    size_t w = 0;
    for (size_t i = 0; i < set.recs.size(); i++) {
        const PafRec &r = set.recs[i];
        const bool ok = (max_tile < 0 || r.tile <= max_tile) && (min_chain < 0 || r.s1 >= min_chain);
        if (ok != inv) set.recs[w++] = r;
    }
    set.recs.resize(w);
}

template <typename F>
int guarded(F &&f) {
    try {
        return f();
    } catch (const HipFailure &e) {
        char buf[512];
        snprintf(buf, sizeof buf, "HIP call did not succeed: %s -> %s (%s:%d)", e.what, hipGetErrorString(e.code), e.file, e.line);
        set_error(buf);
        return e.code == hipErrorNoDevice || e.code == hipErrorInvalidDevice ? MIBLAST_ENODEV : MIBLAST_EHIP;
    } catch (const std::bad_alloc &) {
        set_error("out of host memory");
        return MIBLAST_ELIMIT;
    } catch (const std::length_error &e) {
        set_error(e.what());
        return MIBLAST_ELIMIT;
    } catch (const std::exception &e) {
        set_error(std::string("internal: ") + e.what());
        return MIBLAST_EHIP;
    }
}

int need(const void *ctx, const void *s, const char *fn) {
    if (!s) { set_error(std::string(fn) + ": null set"); return MIBLAST_EINVAL; }
    if (!ctx) { set_error(std::string(fn) + ": no device context (this build has no CPU path)"); return MIBLAST_ENODEV; }
    return MIBLAST_OK;
}

}  // namespace

void chain_cache_destroy(void *cache) { delete (DevCache *)cache; }

}  // namespace mb

struct mipaf_set { mb::PafSet s; };

extern "C" {

int mipaf_set_from_mem(const char *text, size_t len, mipaf_set **out) {
    return mb::guarded([&] {
        if (!out || (!text && len)) { mb::set_error("mipaf_set_from_mem: null argument"); return (int)MIBLAST_EINVAL; }
        mipaf_set *s = new mipaf_set();
        mb::HostHot keep_workers_awake;
        int rc = mb::parse_text(s->s, text, len);
        if (rc != MIBLAST_OK) { delete s; return rc; }
        *out = s;
        return (int)MIBLAST_OK;
    });
}

int mipaf_set_from_file(const char *path, mipaf_set **out) {
    return mb::guarded([&] {
        if (!path || !out) { mb::set_error("mipaf_set_from_file: null argument"); return (int)MIBLAST_EINVAL; }
        std::string text;
        if (!strcmp(path, "-") || !strcmp(path, "/dev/stdin")) {
            char buf[1 << 16];
            ssize_t got;
            while ((got = read(0, buf, sizeof buf)) > 0) text.append(buf, (size_t)got);
        } else {
            std::ifstream f(path, std::ios::binary);
            if (!f) { mb::set_error(std::string("cannot open ") + path); return (int)MIBLAST_EIO; }
            text.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        }
        return mipaf_set_from_mem(text.data(), text.size(), out);
    });
}

void mipaf_set_free(mipaf_set *s) { delete s; }
int64_t mipaf_set_size(const mipaf_set *s) { return s ? (int64_t)s->s.recs.size() : 0; }

int mipaf_set_text(const mipaf_set *s, char **text, size_t *len) {
    return mb::guarded([&] {
        if (!s || !text || !len) { mb::set_error("mipaf_set_text: null argument"); return (int)MIBLAST_EINVAL; }
        mb::HostHot keep_workers_awake;
        std::string t = mb::format_set(s->s);
        char *buf = (char *)malloc(t.size() + 1);
        if (!buf) throw std::bad_alloc();
        memcpy(buf, t.data(), t.size());
        buf[t.size()] = 0;
        *text = buf; *len = t.size();
        return (int)MIBLAST_OK;
    });
}

int mipaf_set_write(const mipaf_set *s, int fd) {
    return mb::guarded([&] {
        if (!s) { mb::set_error("mipaf_set_write: null set"); return (int)MIBLAST_EINVAL; }
        std::string t = mb::format_set(s->s);
        return mb::write_all(fd, t.data(), t.size());
    });
}

int mipaf_invert(mipaf_set *s) {
    return mb::guarded([&] {
        if (!s) { mb::set_error("mipaf_invert: null set"); return (int)MIBLAST_EINVAL; }
        mb::invert(s->s);
        return (int)MIBLAST_OK;
    });
}

static bool dechunk_name(std::string &name, std::string &len, long long &start) {
    const size_t b = name.rfind('|');
    if (b == std::string::npos || b == 0) return false;
    const size_t a = name.rfind('|', b - 1);
    if (a == std::string::npos) return false;
    char *end = nullptr;
    start = strtoll(name.c_str() + b + 1, &end, 10);
    if (!end || *end) return false;
    len = name.substr(a + 1, b - a - 1);
    if (len.empty() || len.find_first_not_of("0123456789") != std::string::npos) return false;
    name.resize(a);
    return true;
}

int mipaf_dechunk_text(const char *paf, size_t len, int32_t query_only, char **out_text, size_t *out_len) {
    if ((!paf && len) || !out_text || !out_len) return MIBLAST_EINVAL;
    *out_text = nullptr; *out_len = 0;
    return mb::guarded([&]() -> int {                            // (whole-genome text: bad_alloc / length_error must not cross the C ABI)
    std::string out;
    out.reserve(len + 64);
    size_t line_no = 0;
    for (size_t pos = 0; pos < len;) {
        const char *nl = (const char *)memchr(paf + pos, '\n', len - pos);
        size_t end = nl ? (size_t)(nl - paf) : len;
        const size_t next = end + 1;
        while (end > pos && paf[end - 1] == '\r') end--;
        line_no++;
        if (end == pos) { pos = next; continue; }
        std::string col[9];
        size_t p = pos;
        int n = 0;
        bool rest = false;
        for (; n < 9; n++) {
            const char *t = (const char *)memchr(paf + p, '\t', end - p);
            if (!t) { col[n++].assign(paf + p, end - p); p = end; break; }
            col[n].assign(paf + p, (size_t)(t - (paf + p)));
            p = (size_t)(t - paf) + 1;
            rest = true;
        }
        if (n < 9) { mb::set_error("dechunk: PAF line " + std::to_string(line_no) + " has fewer than 9 columns"); return MIBLAST_EINVAL; }
        rest = rest && p <= end && n == 9 && paf[p - 1] == '\t';
        for (int side = 0; side < (query_only ? 1 : 2); side++) {
            const int c = side == 0 ? 0 : 5;
            std::string slen;
            long long start = 0;
            if (!dechunk_name(col[c], slen, start)) { mb::set_error("dechunk: PAF line " + std::to_string(line_no) + ": name is not NAME|LENGTH|START"); return MIBLAST_EINVAL; }
            col[c + 1] = slen;
            col[c + 2] = std::to_string(atoll(col[c + 2].c_str()) + start);
            col[c + 3] = std::to_string(atoll(col[c + 3].c_str()) + start);
        }
        for (int k = 0; k < 9; k++) { if (k) out += '\t'; out += col[k]; }
        if (rest) { out += '\t'; out.append(paf + p, end - p); }
        out += '\n';
        pos = next;
    }
    char *buf = (char *)malloc(out.size() + 1);
    if (!buf) { mb::set_error("out of host memory"); return MIBLAST_ELIMIT; }
    memcpy(buf, out.data(), out.size());
    buf[out.size()] = 0;
    *out_text = buf; *out_len = out.size();
    return MIBLAST_OK;
    });
}

int mipaf_unaligned_fasta(const char *paf, size_t paf_len, const char *fasta, size_t fasta_len, int64_t min_size, int64_t flank,
                          char **out_text, size_t *out_len) {
    if ((!paf && paf_len) || (!fasta && fasta_len) || !out_text || !out_len) return MIBLAST_EINVAL;
    *out_text = nullptr; *out_len = 0;
    return mb::guarded([&]() -> int {
    // ---- records of the FASTA text: name = first word of the header, body = the lines up to the next '>' at a line start
    struct Rec { std::string name; size_t body, body_end; int64_t len; std::vector<std::pair<int64_t, int64_t>> spans; };
    std::vector<Rec> recs;
    for (size_t pos = 0; pos < fasta_len;) {
        if (fasta[pos] != '>') {                                   // (text before the first header is ignored, as the parsers do)
            const char *nl = (const char *)memchr(fasta + pos, '\n', fasta_len - pos);
            pos = nl ? (size_t)(nl - fasta) + 1 : fasta_len;
            continue;
        }
        const char *nl = (const char *)memchr(fasta + pos, '\n', fasta_len - pos);
        const size_t hend = nl ? (size_t)(nl - fasta) : fasta_len;
        size_t a = pos + 1;
        while (a < hend && (fasta[a] == ' ' || fasta[a] == '\t')) a++;
        size_t b = a;
        while (b < hend && fasta[b] != ' ' && fasta[b] != '\t' && fasta[b] != '\r') b++;
        Rec r;
        r.name.assign(fasta + a, b - a);
        r.body = nl ? hend + 1 : fasta_len;
        size_t q = r.body;
        int64_t n = 0;
        while (q < fasta_len && fasta[q] != '>') {                   // one line at a time
            const char *e = (const char *)memchr(fasta + q, '\n', fasta_len - q);
            size_t le = e ? (size_t)(e - fasta) : fasta_len;
            const size_t next = e ? le + 1 : fasta_len;
            while (le > q && fasta[le - 1] == '\r') le--;
            n += (int64_t)(le - q);
            q = next;
        }
        r.body_end = q; r.len = n;
        recs.push_back(std::move(r));
        pos = q;
    }
    std::unordered_map<std::string, size_t> by_name;
    for (size_t k = 0; k < recs.size(); k++)
        if (!by_name.emplace(recs[k].name, k).second) {            // (Cactus's inputs have unique headers: checkUniqueHeaders.py; a repeat would lose spans silently)
            mb::set_error("to_bed: sequence name " + recs[k].name + " occurs twice in the FASTA file");
            return MIBLAST_EINVAL;
        }
    // ---- query intervals of the alignments
    size_t line_no = 0;
    for (size_t pos = 0; pos < paf_len;) {
        const char *nl = (const char *)memchr(paf + pos, '\n', paf_len - pos);
        const size_t end = nl ? (size_t)(nl - paf) : paf_len;
        line_no++;
        size_t p = pos;
        pos = end + 1;
        bool blank = true;
        for (size_t x = p; x < end && blank; x++) blank = paf[x] == ' ' || paf[x] == '\t' || paf[x] == '\r';
        if (blank) continue;
        const char *t[4];
        bool ok = true;
        size_t c = p;
        for (int k = 0; k < 4 && ok; k++) {
            t[k] = (const char *)memchr(paf + c, '\t', end - c);
            if (!t[k]) ok = false; else c = (size_t)(t[k] - paf) + 1;
        }
        if (!ok) { mb::set_error("to_bed: PAF line " + std::to_string(line_no) + " has fewer than 5 columns"); return MIBLAST_EINVAL; }
        const std::string name(paf + p, (size_t)(t[0] - (paf + p)));
        const int64_t s0 = strtoll(t[1] + 1, nullptr, 10), e0 = strtoll(t[2] + 1, nullptr, 10);
        auto it = by_name.find(name);
        if (it == by_name.end()) { mb::set_error("to_bed: PAF line " + std::to_string(line_no) + ": query " + name + " is not in the FASTA file"); return MIBLAST_EINVAL; }
        if (e0 > s0) recs[it->second].spans.emplace_back(s0, e0);
    }
    // ---- uncovered stretches >= min_size, widened by flank and merged, cut out with 60 columns per line
    std::string out;
    std::string seq;
    for (Rec &r : recs) {
        std::sort(r.spans.begin(), r.spans.end());
        std::vector<std::pair<int64_t, int64_t>> cut;
        auto add = [&](int64_t s, int64_t e) {
            s = std::max<int64_t>(0, s - flank); e = std::min<int64_t>(r.len, e + flank);
            if (!cut.empty() && s <= cut.back().second) cut.back().second = std::max(cut.back().second, e);
            else cut.emplace_back(s, e);
        };
        int64_t at = 0;
        for (const auto &sp : r.spans) {
            if (sp.first - at >= min_size && sp.first > at) add(at, sp.first);
            at = std::max(at, sp.second);
        }
        if (r.len - at >= min_size && r.len > at) add(at, r.len);
        if (cut.empty()) continue;
        seq.clear();
        seq.reserve((size_t)r.len);
        for (size_t q = r.body; q < r.body_end;) {
            const char *e = (const char *)memchr(fasta + q, '\n', r.body_end - q);
            size_t le = e ? (size_t)(e - fasta) : r.body_end;
            const size_t next = e ? le + 1 : r.body_end;
            while (le > q && fasta[le - 1] == '\r') le--;
            seq.append(fasta + q, le - q);
            q = next;
        }
        for (const auto &iv : cut) {
            out += '>'; out += r.name; out += '|'; out += std::to_string(r.len); out += '|'; out += std::to_string(iv.first); out += '\n';
            for (int64_t x = iv.first; x < iv.second; x += 60) {
                out.append(seq, (size_t)x, (size_t)std::min<int64_t>(60, iv.second - x));
                out += '\n';
            }
        }
    }
    char *buf = (char *)malloc(out.size() + 1);
    if (!buf) { mb::set_error("out of host memory"); return MIBLAST_ELIMIT; }
    memcpy(buf, out.data(), out.size());
    buf[out.size()] = 0;
    *out_text = buf; *out_len = out.size();
    return MIBLAST_OK;
    });
}

void mipaf_chain_params_default(mipaf_chain_params *p) {
    p->max_gap_length = 1000000; p->gap_open = 5000; p->gap_extend = 1; p->trim_fraction = 1.0;
}

int mipaf_chain(miblast_ctx *ctx, mipaf_set *s, const mipaf_chain_params *p, mipaf_stats *stats) {
    return mb::guarded([&] {
        int rc = mb::need(ctx, s, "mipaf_chain");
        if (rc != MIBLAST_OK) return rc;
        mipaf_chain_params cp;
        mipaf_chain_params_default(&cp);
        if (p) cp = *p;
        if (cp.max_gap_length < 0 || cp.trim_fraction < 0.0 || cp.trim_fraction > 1.0) { mb::set_error("mipaf_chain: parameter out of range"); return (int)MIBLAST_EINVAL; }
        mipaf_stats st{};
        const double t0 = mb::now_s();
        mb::chain(ctx->c, s->s, cp, st);
        st.t_total_s = mb::now_s() - t0;
        if (stats) *stats = st;
        return (int)MIBLAST_OK;
    });
}

int mipaf_tile(miblast_ctx *ctx, mipaf_set *s, int32_t hist_bins, mipaf_stats *stats) {
    return mb::guarded([&] {
        int rc = mb::need(ctx, s, "mipaf_tile");
        if (rc != MIBLAST_OK) return rc;
        rc = mb::check_cigars(s->s);
        if (rc != MIBLAST_OK) return rc;
        mipaf_stats st{};
        const double t0 = mb::now_s();
        mb::tile(ctx->c, s->s, hist_bins, st);
        st.t_total_s = mb::now_s() - t0;
        if (stats) *stats = st;
        return (int)MIBLAST_OK;
    });
}

int mipaf_trim(miblast_ctx *ctx, mipaf_set *s, const char *trim_identity, mipaf_stats *stats) {
    return mb::guarded([&] {
        int rc = mb::need(ctx, s, "mipaf_trim");
        if (rc != MIBLAST_OK) return rc;
        long long num, den;
        if (!mb::parse_fraction(trim_identity, num, den)) { mb::set_error("mipaf_trim: --trimIdentity must be a decimal in [0, 1] with at most 6 digits"); return (int)MIBLAST_EINVAL; }
        rc = mb::check_cigars(s->s);
        if (rc != MIBLAST_OK) return rc;
        mipaf_stats st{};
        const double t0 = mb::now_s();
        mb::trim(ctx->c, s->s, num, den, st);
        st.t_total_s = mb::now_s() - t0;
        if (stats) *stats = st;
        return (int)MIBLAST_OK;
    });
}

int mipaf_filter(mipaf_set *s, int64_t max_tile_level, int64_t min_chain_score, int32_t invert) {
    return mb::guarded([&] {
        if (!s) { mb::set_error("mipaf_filter: null set"); return (int)MIBLAST_EINVAL; }
        mb::filter(s->s, max_tile_level, min_chain_score, invert != 0);
        return (int)MIBLAST_OK;
    });
}

int mipaf_split_by_query(const mipaf_set *s, const char *prefix, int64_t min_length, int32_t *n_parts) {
    return mb::guarded([&] {
        if (!s || !prefix) { mb::set_error("mipaf_split_by_query: null argument"); return (int)MIBLAST_EINVAL; }
        const mb::PafSet &set = s->s;
        std::vector<int32_t> part_of(set.names.size(), -1);
        std::vector<std::string> parts;
        int32_t part = 0;
        int64_t acc = 0;
        for (const mb::PafRec &r : set.recs) {                // R-S1
            if (part_of[r.qn] < 0) {
                part_of[r.qn] = part;
                if ((size_t)part == parts.size()) parts.emplace_back();
                acc += r.ql;
                if (acc >= min_length) { part++; acc = 0; }
            }
            mb::format_rec(set, r, parts[(size_t)part_of[r.qn]]);
        }
        for (size_t k = 0; k < parts.size(); k++) {
            const std::string path = std::string(prefix) + std::to_string(k) + ".paf";
            FILE *f = fopen(path.c_str(), "wb");
            if (!f || fwrite(parts[k].data(), 1, parts[k].size(), f) != parts[k].size()) { if (f) fclose(f); mb::set_error("cannot write " + path); return (int)MIBLAST_EIO; }
            fclose(f);
        }
        if (n_parts) *n_parts = (int32_t)parts.size();
        return (int)MIBLAST_OK;
    });
}

int mipaf_chain_tile_trim_filter(miblast_ctx *ctx, mipaf_set *s, const mipaf_chain_params *p, const char *trim_identity,
                                 int64_t min_primary_chain_score, int32_t output_secondary, mipaf_stats *stats) {
    return mb::guarded([&] {
        int rc = mb::need(ctx, s, "mipaf_chain_tile_trim_filter");
        if (rc != MIBLAST_OK) return rc;
        mipaf_chain_params cp;
        mipaf_chain_params_default(&cp);
        if (p) cp = *p;
        long long num, den;
        if (!mb::parse_fraction(trim_identity, num, den)) { mb::set_error("mipaf_chain_tile_trim_filter: bad trim identity"); return (int)MIBLAST_EINVAL; }
        rc = mb::check_cigars(s->s);
        if (rc != MIBLAST_OK) return rc;
        mipaf_stats st{};
        const double t0 = mb::now_s();
        mb::HostHot keep_workers_awake;
        mb::PafSet &set = s->s;
        mb::chain(ctx->c, set, cp, st);                       // local_alignment.py:684-690 (and :694-699)
        mb::tile(ctx->c, set, 0, st);
        mb::trim(ctx->c, set, num, den, st);
        mb::filter(set, 1, -1, false);
        std::vector<mb::PafRec> secondary;
        if (output_secondary)                                 // :702-704 applies `filter --maxTileLevel 1 --invert` to filter.paf, i.e. to
            for (const mb::PafRec &r : set.recs)              // what the line above already kept: as in the reference, nothing passes
                if (!(r.tile <= 1)) secondary.push_back(r);
        mb::chain(ctx->c, set, cp, st);
        if (output_secondary) {                               // :710-723
            std::vector<mb::PafRec> out(secondary);
            for (const mb::PafRec &r : set.recs) if (r.s1 >= min_primary_chain_score) out.push_back(r);
            for (mb::PafRec r : set.recs)
                if (!(r.s1 >= min_primary_chain_score)) {     // sed 's/tp:A:P/tp:A:S/' | sed 's/tl:i:1/tl:i:2/'
                    if (r.tp == 'P') r.tp = 'S';
                    if (r.tile == 1) r.tile = 2;
                    out.push_back(r);
                }
            set.recs.swap(out);
        } else {
            mb::filter(set, -1, min_primary_chain_score, false);
        }
        st.records = (int64_t)set.recs.size();
        st.t_total_s = mb::now_s() - t0;
        if (stats) *stats = st;
        return (int)MIBLAST_OK;
    });
}

}  // extern "C"

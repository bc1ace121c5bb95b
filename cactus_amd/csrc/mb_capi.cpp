// mb_capi.cpp -- the C ABI declared in include/miblast.h (extern "C", plain pointers, no exceptions
// escape).  There is deliberately no CPU path: without a gfx950 device every compute entry point
// returns MIBLAST_ENODEV.
#include "mb_pipeline.h"
#include "mb_guard.h"

#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <fstream>
#include <new>
#include <unistd.h>

namespace mb {
static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
const std::string &last_error_text() { return g_last_error; }
}  // namespace mb


namespace {

template <typename F>
int guarded(F &&f) {
    try {
        return f();
    } catch (const mb::HipFailure &e) {
        char buf[512];
        snprintf(buf, sizeof buf, "HIP call did not succeed: %s -> %s (%s:%d)", e.what, hipGetErrorString(e.code), e.file, e.line);
        mb::set_error(buf);
        return e.code == hipErrorNoDevice || e.code == hipErrorInvalidDevice ? MIBLAST_ENODEV : MIBLAST_EHIP;
    } catch (const std::bad_alloc &) {
        mb::set_error("out of host memory");
        return MIBLAST_ELIMIT;
    } catch (const std::exception &e) {
        mb::set_error(std::string("internal: ") + e.what());
        return MIBLAST_EHIP;
    }
}

bool parse_int(const char *s, long &v) {
    if (!s || !*s) return false;
    char *end = nullptr;
    v = strtol(s, &end, 10);
    return end && *end == 0;
}

}  // namespace

extern "C" {

// The HIP runtime deals a process's streams to 4 hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and two streams on one queue
// run their kernels one after the other: a context's stream, its lanes' and the streams of the other contexts of a job want queues of
// their own (16 x 1 Mb pairs in one call: 31 ms side by side, 34 ms one after the other).  Set when the library is loaded -- before the
// runtime starts, which reads it at its first call -- unless the caller has said otherwise.
// HSA_DISABLE_COREDUMP_ON_EXCEPTION: a device fault makes the ROCm runtime write a GPU core dump before it reports anything; under a
// piped kernel.core_pattern inside a container its helper cannot be started and the dumping process dies of SIGPIPE with nothing but
// "GPU coredump: execvp failed" on stderr (GPUTEST_r05) -- and a Toil job has no use for a dump of a 288 GB device.  Without the dump the
// runtime's own report (faulting address, reason) reaches stderr and the exit is the abort cactus_call expects.  Caller's setting wins.
// HSA_ENABLE_INTERRUPT=0 (polling) is NOT set by the library on its own any more: miblast_frontend_runtime_defaults() below, called by the
// front ends, or MIBLAST_POLL=1 in the environment of a process that loads the library.
namespace {
bool poll_rule(int threads) {
    const char *poll = getenv("MIBLAST_POLL");
    if (poll && *poll == '0') return false;
    if (threads <= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        threads = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : 0;
    }
    if (threads < 24 && !(poll && *poll == '1')) return false;
    setenv("HSA_ENABLE_INTERRUPT", "0", 0);
    const char *v = getenv("HSA_ENABLE_INTERRUPT");
    return v && *v == '0';
}
struct RuntimeDefaults {
    RuntimeDefaults() {
        setenv("GPU_MAX_HW_QUEUES", "16", 0);
        setenv("HSA_DISABLE_COREDUMP_ON_EXCEPTION", "1", 0);
        const char *poll = getenv("MIBLAST_POLL");
        if (poll && *poll == '1') (void)poll_rule(1 << 20);
    }
} g_runtime_defaults;
}  // namespace

int miblast_frontend_runtime_defaults(int threads) { return poll_rule(threads) ? 1 : 0; }

size_t miblast_params_size(void) { return sizeof(miblast_params); }

void miblast_params_default(miblast_params *p) {
    p->step = 1; p->transitions = 1; p->xdrop = 910; p->ydrop = 9400; p->hspthresh = 3000; p->gappedthresh = -1;
    p->gap_open = 400; p->gap_extend = 30; p->entropy = 1; p->queryhspbest = 0; p->ambiguous_n = 1; p->gapped = 1;
    p->format = 0; p->markend = 0; p->queryhsplimit = 0; p->diag_hash16 = 0; p->walls = 0; p->strands = 0;
    p->query_softmask = 0; p->step_origin = 0; p->xdrop_le = 0; p->hspbest_ties = 0;
}

int miblast_params_from_argv(int argc, char **argv, miblast_params *p, const char *files[2], int *num_gpu, int *num_threads) {
    miblast_params_default(p);
    int nf = 0, ng = 1, nt = 1;
    bool querydepth_seen = false;
    files[0] = files[1] = nullptr;
    auto bad = [&](const char *a, const char *why) {
        mb::set_error(std::string(why) + ": " + a);
        return MIBLAST_EINVAL;
    };
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-') {
            if (nf >= 2) return bad(a, "more than two sequence files");
            files[nf++] = a;
            continue;
        }
        const char *eq = strchr(a, '=');
        std::string key = eq ? std::string(a, (size_t)(eq - a)) : std::string(a);
        const char *val = eq ? eq + 1 : nullptr;
        long v = 0;
        auto need_int = [&](int32_t &dst, long lo) -> int {
            if (!parse_int(val, v) || v < lo) return bad(a, "bad value for option");
            dst = (int32_t)v;
            return 0;
        };
        int rc = 0;
        if (key == "--step") rc = need_int(p->step, 1);
        else if (key == "--ydrop") rc = need_int(p->ydrop, 0);
        else if (key == "--xdrop") rc = need_int(p->xdrop, 0);
        else if (key == "--hspthresh") rc = need_int(p->hspthresh, 0);
        else if (key == "--gappedthresh") rc = need_int(p->gappedthresh, 0);
        else if (key == "--queryhspbest") rc = need_int(p->queryhspbest, 0);
        else if (key == "--notransition" && !val) p->transitions = 0;
        else if (key == "--transition" && !val) p->transitions = 1;
        else if (key == "--noentropy" && !val) p->entropy = 0;
        else if (key == "--entropy" && !val) p->entropy = 1;
        else if ((key == "--ungapped" || key == "--nogapped") && !val) p->gapped = 0;
        else if (key == "--gapped" && !val) p->gapped = 1;
        else if (key == "--ambiguous") {
            // Cactus always passes iupac,100,100 (cactus_progressive_config.xml:131-145)
            if (!val || (strcmp(val, "iupac,100,100") && strcmp(val, "iupac") && strcmp(val, "n,100,100") && strcmp(val, "n")))
                return bad(a, "unsupported --ambiguous form");
            p->ambiguous_n = 1;
        } else if (key == "--format") {
            // blast: paf:wfmash (local_alignment.py:68); repeat masker: the six-column general form (cactus_lastzRepeatMask.py:104)
            if (val && !strcmp(val, "paf:wfmash")) p->format = 0;
            else if (val && !strcmp(val, "general:name1,zstart1,end1,name2,zstart2+,end2+")) p->format = 1;
            else return bad(a, "unsupported --format");
        } else if (key == "--markend" && !val) p->markend = 1;
        else if (key == "--queryhsplimit") {
            // only the form Cactus passes: keep,nowarn:N (cactus_progressive_config.xml:36 lastzOpts)
            if (!val || strncmp(val, "keep,nowarn:", 12) || !parse_int(val + 12, v) || v < 1) return bad(a, "unsupported --queryhsplimit form");
            p->queryhsplimit = (int32_t)v;
        } else if (key == "--querydepth") {
            // "has no effect when --ungapped is passed" (cactus_lastzRepeatMask.py:100); accepted for that case only
            if (!val || strncmp(val, "keep,nowarn:", 12) || !parse_int(val + 12, v) || v < 1) return bad(a, "unsupported --querydepth form");
            querydepth_seen = true;
        } else if (key == "--miblast-diag") {                  // the oracle's comparison switches: one argv serves both implementations
            if (!val || (strcmp(val, "exact") && strcmp(val, "hash16"))) return bad(a, "unsupported --miblast-diag form");
            p->diag_hash16 = !strcmp(val, "hash16");
        } else if (key == "--miblast-walls" && !val) p->walls = 1;
        else if (key == "--miblast-xdrop") {
            if (!val || (strcmp(val, "lt") && strcmp(val, "le"))) return bad(a, "unsupported --miblast-xdrop form");
            p->xdrop_le = !strcmp(val, "le");
        } else if (key == "--miblast-hspbest-ties") {
            if (!val || (strcmp(val, "earlier") && strcmp(val, "later"))) return bad(a, "unsupported --miblast-hspbest-ties form");
            p->hspbest_ties = !strcmp(val, "later");
        }
        else if (key == "--strand") {                        // lastz's --strand=both|plus|minus
            if (!val) return bad(a, "--strand needs both, plus or minus");
            if (!strcmp(val, "both")) p->strands = 0; else if (!strcmp(val, "plus")) p->strands = 1; else if (!strcmp(val, "minus")) p->strands = 2;
            else return bad(a, "--strand needs both, plus or minus");
        }
        else if (key == "--num_gpu") {                       // run_kegalign form: "--num_gpu N" (local_alignment.py:58)
            if (val) { if (!parse_int(val, v) || v < 1) return bad(a, "bad value for option"); }
            else { if (i + 1 >= argc || !parse_int(argv[++i], v) || v < 1) return bad(a, "bad value for option"); }
            ng = (int)v;
        } else if (key == "--num_threads") {
            if (val) { if (!parse_int(val, v) || v < 0) return bad(a, "bad value for option"); }
            else { if (i + 1 >= argc || !parse_int(argv[++i], v) || v < 0) return bad(a, "bad value for option"); }
            nt = (int)v;
        } else return bad(a, "unknown option");
        if (rc) return rc;
    }
    if (nf != 2) { mb::set_error("expected a target and a query sequence file"); return MIBLAST_EINVAL; }
    if (p->format == 1 && p->gapped) { mb::set_error("--format=general:... is implemented for --ungapped only"); return MIBLAST_EINVAL; }
    if (querydepth_seen && p->gapped) { mb::set_error("--querydepth is only accepted together with --ungapped"); return MIBLAST_EINVAL; }
    if (num_gpu) *num_gpu = ng;
    if (num_threads) *num_threads = nt;
    return MIBLAST_OK;
}

int miblast_set_host_threads(int n) {
    return guarded([&] {
        int got = mb::set_host_threads(n);
        if (got < 0) { mb::set_error("miblast_set_host_threads: an alignment call is running"); return (int)MIBLAST_EINVAL; }
        return got;
    });
}

// $MIBLAST_DEVICE_MAP = "0,0,1": logical device k is HIP ordinal map[k] (several logical devices may share a GPU)
// An entry that does not parse or names a device that is not there makes the whole map invalid (*bad): dropping it would renumber
// the logical devices behind the caller's back.  Without any device the map is empty, valid or not (there is no CPU path to map to).
static std::vector<int> device_map(bool *bad = nullptr) {
    std::vector<int> m;
    int n = 0;
    if (bad) *bad = false;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    const char *e = getenv("MIBLAST_DEVICE_MAP");
    if (e && *e) {
        if (n <= 0) return m;
        for (const char *p = e; *p;) {
            char *end = nullptr;
            long v = strtol(p, &end, 10);
            if (end == p || (*end && *end != ',') || v < 0 || v >= n) {
                if (bad) *bad = true;
                mb::set_error(std::string("MIBLAST_DEVICE_MAP=") + e + ": every entry must be a HIP device ordinal below " + std::to_string(n));
                m.clear();
                return m;
            }
            m.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
        }
        return m;
    }
    for (int d = 0; d < n; d++) m.push_back(d);
    return m;
}

int miblast_device_count(void) {
    bool bad = false;
    const std::vector<int> m = device_map(&bad);
    return bad ? (int)MIBLAST_EINVAL : (int)m.size();
}

int miblast_ctx_create(int device, miblast_ctx **out) {
    if (!out) return MIBLAST_EINVAL;
    *out = nullptr;
    return guarded([&]() -> int {
        bool bad_map = false;
        const std::vector<int> map = device_map(&bad_map);
        if (bad_map) return MIBLAST_ENODEV;                         // (message set by device_map)
        if (map.empty()) {
            mb::set_error("no HIP device visible: libmiblast has no CPU path");
            return MIBLAST_ENODEV;
        }
        if (device < 0 || device >= (int)map.size()) { mb::set_error("device ordinal out of range"); return MIBLAST_ENODEV; }
        device = map[(size_t)device];
        hipDeviceProp_t prop;
        MB_HIP(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !getenv("MIBLAST_ALLOW_ANY_ARCH")) {
            mb::set_error(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
            return MIBLAST_ENODEV;
        }
        MB_HIP(hipSetDevice(device));
        miblast_ctx *c = new miblast_ctx();
        c->c.device = device;
        c->c.ws = mb::workspace_create();
        MB_HIP(hipStreamCreateWithFlags(&c->c.stream, hipStreamNonBlocking));
        MB_HIP(hipEventCreate(&c->c.ev0)); MB_HIP(hipEventCreate(&c->c.ev1)); MB_HIP(hipEventCreate(&c->c.ev2));
        MB_HIP(hipEventCreate(&c->c.ev3)); MB_HIP(hipEventCreate(&c->c.ev4));
        mb::ctx_pair_streams(c->c);
        *out = c;
        return MIBLAST_OK;
    });
}

int miblast_ctx_set_priority(miblast_ctx *ctx, int level) {
    if (!ctx) return MIBLAST_EINVAL;
    return guarded([&]() -> int { return mb::ctx_set_priority(ctx->c, level); });
}

void miblast_ctx_destroy(miblast_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->c.device);
    if (c->c.ev0) (void)hipEventDestroy(c->c.ev0);
    if (c->c.ev1) (void)hipEventDestroy(c->c.ev1);
    if (c->c.ev2) (void)hipEventDestroy(c->c.ev2);
    if (c->c.ev3) (void)hipEventDestroy(c->c.ev3);
    if (c->c.ev4) (void)hipEventDestroy(c->c.ev4);
    if (c->c.stream) (void)hipStreamDestroy(c->c.stream);
    mb::workspace_destroy(c->c.ws);
    mb::chain_cache_destroy(c->c.chain_cache);
    delete c;
}

int miblast_seqset_from_fasta_mem(miblast_ctx *ctx, const char *buf, size_t len, miblast_seqset **out) {
    if (!ctx || !out || (!buf && len)) return MIBLAST_EINVAL;
    *out = nullptr;
    return guarded([&]() -> int {
        miblast_seqset *s = new miblast_seqset();
        mb::parse_fasta(buf, len, s->s);
        try { mb::upload_seqset(s->s, ctx->c.device); } catch (...) { mb::release_seqset(s->s); delete s; throw; }
        *out = s;
        return MIBLAST_OK;
    });
}

// a FASTA file as lastz is handed it: trailing [actions] on the name are accepted (local_alignment.py:60-62)
static int read_fasta_file(const char *path, std::string &data) {
    std::string file(path);
    size_t br = file.find('[');
    bool unmask = false;
    if (br != std::string::npos) {
        unmask = file.find("unmask", br) != std::string::npos;     // [unmask] / [multiple,unmask] (cactus_lastzRepeatMask.py:88)
        file.resize(br);
    }
    FILE *f = fopen(file.c_str(), "rb");
    if (!f) { mb::set_error("cannot open " + file); return MIBLAST_EIO; }
    data.clear();
    if (fseeko(f, 0, SEEK_END) == 0) { const off_t sz = ftello(f); if (sz > 0) data.reserve((size_t)sz); }
    rewind(f);
    std::vector<char> buf(1 << 22);
    for (size_t n; (n = fread(buf.data(), 1, buf.size(), f)) > 0;) data.append(buf.data(), n);
    const bool bad = ferror(f) != 0;
    fclose(f);
    if (bad) { mb::set_error("cannot read " + file); return MIBLAST_EIO; }
    if (unmask) {                                                    // lastz [unmask]: soft-masking is removed on load
        bool header = false;
        for (char &c : data) {
            if (c == '>') header = true;
            else if (c == '\n') header = false;
            else if (!header && c >= 'a' && c <= 'z') c = (char)(c - 32);
        }
    }
    return MIBLAST_OK;
}

static int write_all(int fd, const char *p, size_t len) {
    size_t done = 0;
    while (done < len) {
        ssize_t w = write(fd, p + done, len - done);
        if (w <= 0) { mb::set_error("cannot write PAF output"); return MIBLAST_EIO; }
        done += (size_t)w;
    }
    return MIBLAST_OK;
}

int miblast_seqset_from_fasta_file(miblast_ctx *ctx, const char *path, miblast_seqset **out) {
    if (!ctx || !out || !path) return MIBLAST_EINVAL;
    *out = nullptr;
    std::string data;
    return guarded([&]() -> int {
        int rc = read_fasta_file(path, data);
        return rc != MIBLAST_OK ? rc : miblast_seqset_from_fasta_mem(ctx, data.data(), data.size(), out);
    });
}

void miblast_drop_derived(void) {
    try { mb::drop_derived(); } catch (...) {}
}

int miblast_seqsets_unaligned(miblast_ctx *ctx, size_t n, const miblast_seqset *const *queries, const char *const *pafs, const size_t *paf_lens,
                              int64_t min_size, int64_t flank, miblast_seqset **out) {
    if (!ctx || !out || (n && (!queries || !pafs || !paf_lens))) return MIBLAST_EINVAL;
    for (size_t k = 0; k < n; k++) { out[k] = nullptr; if (!queries[k] || (!pafs[k] && paf_lens[k])) return MIBLAST_EINVAL; }
    return guarded([&]() -> int {
        std::vector<const mb::SeqSet *> qs(n);
        std::vector<std::unique_ptr<miblast_seqset>> made(n);          // (owned here until handed out: nothing leaks when an allocation throws half way)
        std::vector<mb::SeqSet *> outs(n);
        std::unique_ptr<bool[]> none(new bool[n + 1]);
        for (size_t k = 0; k < n; k++) {
            if (queries[k]->s.device != ctx->c.device) { mb::set_error("a query set lives on another device than the context"); return MIBLAST_EINVAL; }
            qs[k] = &queries[k]->s; made[k].reset(new miblast_seqset()); outs[k] = &made[k]->s;
        }
        const int rc = mb::seqset_unaligned(ctx->c, n, qs.data(), pafs, paf_lens, min_size, flank, outs.data(), none.get());
        for (size_t k = 0; k < n; k++) {
            if (rc == MIBLAST_OK && !none[k]) out[k] = made[k].release();
            else if (made[k]->s.d_buf) mb::release_seqset(made[k]->s);
        }
        return rc;
    });
}

int miblast_seqset_fasta(const miblast_seqset *s, char **text, size_t *len) {
    if (!s || !text || !len) return MIBLAST_EINVAL;
    *text = nullptr; *len = 0;
    return guarded([&]() -> int {
        static const char kLetter[16] = {'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N', 'a', 'c', 'g', 't', 'n', 'n', 'n', 'n'};
        const mb::SeqSet &q = s->s;
        std::string out;
        out.reserve((size_t)q.total + (size_t)q.total / 60 + 64 * q.names.size() + 16);
        for (size_t k = 0; k < q.names.size(); k++) {
            out += '>'; out += q.names[k]; out += '\n';
            const uint8_t *c = q.host() + q.starts[k];
            for (int64_t x = 0; x < q.lens[k]; x += 60) {
                const int64_t w = std::min<int64_t>(60, q.lens[k] - x);
                for (int64_t y = 0; y < w; y++) out += kLetter[c[x + y] & 15u];
                out += '\n';
            }
        }
        char *buf = (char *)malloc(out.size() + 1);
        if (!buf) { mb::set_error("out of host memory"); return MIBLAST_ELIMIT; }
        memcpy(buf, out.data(), out.size());
        buf[out.size()] = 0;
        *text = buf; *len = out.size();
        return MIBLAST_OK;
    });
}

void miblast_seqset_free(miblast_seqset *s) {
    if (!s) return;
    if (s->s.device >= 0) (void)hipSetDevice(s->s.device);
    mb::release_seqset(s->s);
    delete s;
}

int32_t miblast_seqset_n_contigs(const miblast_seqset *s) { return s ? (int32_t)s->s.names.size() : 0; }
int64_t miblast_seqset_total(const miblast_seqset *s) { return s ? s->s.total : 0; }
const char *miblast_seqset_name(const miblast_seqset *s, int32_t i) {
    return (s && i >= 0 && (size_t)i < s->s.names.size()) ? s->s.names[(size_t)i].c_str() : nullptr;
}
int64_t miblast_seqset_start(const miblast_seqset *s, int32_t i) {
    return (s && i >= 0 && (size_t)i < s->s.starts.size()) ? s->s.starts[(size_t)i] : -1;
}
int64_t miblast_seqset_len(const miblast_seqset *s, int32_t i) {
    return (s && i >= 0 && (size_t)i < s->s.lens.size()) ? s->s.lens[(size_t)i] : -1;
}

int miblast_align(miblast_ctx *ctx, const miblast_seqset *target, const miblast_seqset *query, const miblast_params *p,
                  miblast_result **out) {
    if (!ctx || !target || !query || !p || !out) return MIBLAST_EINVAL;
    *out = nullptr;
    return guarded([&]() -> int {
        miblast_result *r = new miblast_result();
        int rc;
        try { rc = mb::align(ctx->c, target->s, query->s, *p, r->r); } catch (...) { delete r; throw; }
        if (rc != MIBLAST_OK) { delete r; return rc; }
        *out = r;
        return MIBLAST_OK;
    });
}

int miblast_align_pairs(miblast_ctx *ctx, const miblast_seqset *const *targets, const miblast_seqset *const *queries, size_t n_pairs,
                        const miblast_params *p, miblast_result **results) {
    if (!ctx || !targets || !queries || !p || !results || n_pairs == 0) return MIBLAST_EINVAL;
    for (size_t k = 0; k < n_pairs; k++) { results[k] = nullptr; if (!targets[k] || !queries[k]) return MIBLAST_EINVAL; }
    return guarded([&]() -> int {
        std::vector<const mb::SeqSet *> ts(n_pairs), qs(n_pairs);
        std::vector<mb::Result *> rs(n_pairs);
        std::vector<miblast_result *> owned(n_pairs, nullptr);
        auto cleanup = [&]() { for (miblast_result *r : owned) delete r; };
        int rc;
        try {
            for (size_t k = 0; k < n_pairs; k++) { owned[k] = new miblast_result(); ts[k] = &targets[k]->s; qs[k] = &queries[k]->s; rs[k] = &owned[k]->r; }
            rc = mb::align_pairs(ctx->c, ts.data(), qs.data(), n_pairs, *p, rs.data());
        } catch (...) { cleanup(); throw; }
        if (rc != MIBLAST_OK) { cleanup(); return rc; }
        for (size_t k = 0; k < n_pairs; k++) results[k] = owned[k];
        return MIBLAST_OK;
    });
}

void miblast_result_free(miblast_result *r) { delete r; }

const char *miblast_result_paf(const miblast_result *r, size_t *len) {
    if (len) *len = r ? r->r.paf.size() : 0;
    return r ? r->r.paf.c_str() : nullptr;
}
const miblast_stats *miblast_result_stats(const miblast_result *r) { return r ? &r->r.stats : nullptr; }
const miblast_hsp *miblast_result_hsps(const miblast_result *r, int64_t *n) {
    if (n) *n = r ? (int64_t)r->r.hsps.size() : 0;
    return r ? r->r.hsps.data() : nullptr;
}
const miblast_aln *miblast_result_alns(const miblast_result *r, int64_t *n) {
    if (n) *n = r ? (int64_t)r->r.alns.size() : 0;
    return r ? r->r.alns.data() : nullptr;
}
const uint32_t *miblast_result_ops(const miblast_result *r, int64_t *n) {
    if (n) *n = r ? (int64_t)r->r.ops.size() : 0;
    return r ? r->r.ops.data() : nullptr;
}

// both files parsed on the host, then the blocked path over the given contexts (mb_multi.cpp)
static int align_files_over(const std::vector<mb::Ctx *> &ctxs, const char *target_fa, const char *query_fa, const miblast_params *p, int out_fd,
                            miblast_stats *stats) {
    return guarded([&]() -> int {
        mb::SeqSet T, Q;
        {
            std::string data;
            int rc = read_fasta_file(target_fa, data);
            if (rc != MIBLAST_OK) return rc;
            mb::parse_fasta(data.data(), data.size(), T);
            rc = read_fasta_file(query_fa, data);
            if (rc != MIBLAST_OK) return rc;
            mb::parse_fasta(data.data(), data.size(), Q);
        }
        const mb::SeqSet *tp = &T, *qp = &Q;
        std::string paf;
        int rc = mb::align_blocked(ctxs, &tp, &qp, 1, *p, paf, stats);
        return rc != MIBLAST_OK ? rc : write_all(out_fd, paf.data(), paf.size());
    });
}

int miblast_align_files(miblast_ctx *ctx, const char *target_fa, const char *query_fa, const miblast_params *p, int out_fd,
                        miblast_stats *stats) {
    if (!ctx || !target_fa || !query_fa || !p) return MIBLAST_EINVAL;
    return align_files_over(std::vector<mb::Ctx *>{&ctx->c}, target_fa, query_fa, p, out_fd, stats);
}

int miblast_multi_create(int num_gpu, miblast_multi **out) {
    if (!out) return MIBLAST_EINVAL;
    *out = nullptr;
    const int n = miblast_device_count();
    if (n < 0) return MIBLAST_ENODEV;                               // invalid $MIBLAST_DEVICE_MAP (message set)
    if (n == 0) { mb::set_error("no HIP device visible: libmiblast has no CPU path"); return MIBLAST_ENODEV; }
    if (num_gpu < 1 || num_gpu > n) { mb::set_error("num_gpu out of range: " + std::to_string(num_gpu) + " asked, " + std::to_string(n) + " visible"); return MIBLAST_ENODEV; }
    miblast_multi *m = new (std::nothrow) miblast_multi();
    if (!m) return MIBLAST_ELIMIT;
    for (int d = 0; d < num_gpu; d++) {
        miblast_ctx *c = nullptr;
        const int rc = miblast_ctx_create(d, &c);
        if (rc != MIBLAST_OK) { miblast_multi_destroy(m); return rc; }
        m->ctxs.push_back(c);
    }
    *out = m;
    return MIBLAST_OK;
}

void miblast_multi_destroy(miblast_multi *m) {
    if (!m) return;
    for (miblast_ctx *c : m->ctxs) miblast_ctx_destroy(c);
    delete m;
}

int miblast_multi_num_gpu(const miblast_multi *m) { return m ? (int)m->ctxs.size() : 0; }

static std::vector<mb::Ctx *> ctx_list(miblast_multi *m) {
    std::vector<mb::Ctx *> v;
    for (miblast_ctx *c : m->ctxs) v.push_back(&c->c);
    return v;
}

int miblast_multi_align_files(miblast_multi *m, const char *target_fa, const char *query_fa, const miblast_params *p, int out_fd,
                              miblast_stats *stats) {
    if (!m || m->ctxs.empty() || !target_fa || !query_fa || !p) return MIBLAST_EINVAL;
    return align_files_over(ctx_list(m), target_fa, query_fa, p, out_fd, stats);
}

int miblast_multi_align_fasta_pairs(miblast_multi *m, const miblast_fasta_pair *pairs, size_t n_pairs, const miblast_params *p,
                                    char **paf, size_t *paf_len, miblast_stats *stats) {
    if (!m || m->ctxs.empty() || !pairs || n_pairs == 0 || !p || !paf || !paf_len) return MIBLAST_EINVAL;
    *paf = nullptr; *paf_len = 0;
    return guarded([&]() -> int {
        std::vector<mb::SeqSet> T(n_pairs), Q(n_pairs);
        std::vector<const mb::SeqSet *> tp(n_pairs), qp(n_pairs);
        for (size_t k = 0; k < n_pairs; k++) {
            if ((!pairs[k].target && pairs[k].target_len) || (!pairs[k].query && pairs[k].query_len)) return (int)MIBLAST_EINVAL;
            mb::parse_fasta(pairs[k].target, pairs[k].target_len, T[k]);
            mb::parse_fasta(pairs[k].query, pairs[k].query_len, Q[k]);
            tp[k] = &T[k]; qp[k] = &Q[k];
        }
        std::string text;
        const int rc = mb::align_blocked(ctx_list(m), tp.data(), qp.data(), n_pairs, *p, text, stats);
        if (rc != MIBLAST_OK) return rc;
        char *buf = (char *)malloc(text.size() + 1);
        if (!buf) { mb::set_error("out of host memory"); return (int)MIBLAST_ELIMIT; }
        memcpy(buf, text.data(), text.size());
        buf[text.size()] = 0;
        *paf = buf; *paf_len = text.size();
        return MIBLAST_OK;
    });
}

int miblast_build_index(miblast_ctx *ctx, const miblast_seqset *target, int32_t step, uint32_t **offsets, uint32_t **positions) {
    if (!ctx || !target || !offsets || !positions || step < 1) return MIBLAST_EINVAL;
    return guarded([&]() -> int { return mb::export_index(ctx->c, target->s, step, offsets, positions); });
}

void miblast_free(void *p) { free(p); }

const char *miblast_last_error(void) { return mb::g_last_error.c_str(); }
long long miblast_debug_device_allocs(void) { return mb::device_alloc_calls().load(); }

const char *miblast_version(void) { return "miblast 0.1 (gfx950)"; }

}  // extern "C"

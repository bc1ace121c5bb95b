// mp_text.cpp -- host-side text steps of the blast phase that the reference runs as `paffy` / `faffy` sub-processes and that are not
// part of the chaining stage: the second half of outgroup trimming,
//     paffy to_bed --binary --excludeUnaligned --includeInverted | faffy extract --skipMissing --minSize 1 --flank F | paffy upconvert
// (trim_unaligned_sequences, /root/reference/src/cactus/paf/local_alignment.py:861-904), the general forms of `paffy to_bed` and
// `faffy extract` (:191-216, :476-489) and `faffy chunk` (:378-387).  C ABI: include/mipaf.h; front ends: bin/paffy, bin/faffy.
// Same rules as the Python cores of cactus_amd/paf/chunking.py (intervals sorted and merged -- a whole-genome PAF has a handful of records
// per sequence, no per-base array is needed); oracle/paffy_text_oracle.c restates them with per-base counters.  paffy itself is an absent
// submodule of the reference: PARITY UNPINNED, like every text step.
#include "mb_common.h"
#include "../../include/mipaf.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

// a decimal column of a (ptr, len) text buffer: digits from b up to e at most -- never past the caller's buffer, whatever follows it
// (the C ABI takes buffers that need not be NUL-terminated); an empty or non-numeric column reads as 0 as strtoll's would
static inline long long field_i64(const char *b, const char *e) {
    bool neg = false;
    if (b < e && (*b == '-' || *b == '+')) { neg = *b == '-'; b++; }
    long long v = 0;
    for (; b < e && *b >= '0' && *b <= '9'; b++) v = v * 10 + (*b - '0');
    return neg ? -v : v;
}

namespace {

struct FaRec { std::string name; size_t body = 0, body_end = 0; int64_t len = 0; };

// records of FASTA text: name = first word of the header, body = the lines up to the next '>' at a line start
void fasta_records(const char *fasta, size_t fasta_len, std::vector<FaRec> &recs) {
    for (size_t pos = 0; pos < fasta_len;) {
        const char *nl = (const char *)memchr(fasta + pos, '\n', fasta_len - pos);
        if (fasta[pos] != '>') { pos = nl ? (size_t)(nl - fasta) + 1 : fasta_len; continue; }      // (text before the first header is ignored)
        const size_t hend = nl ? (size_t)(nl - fasta) : fasta_len;
        size_t a = pos + 1;
        while (a < hend && (fasta[a] == ' ' || fasta[a] == '\t')) a++;
        size_t b = a;
        while (b < hend && fasta[b] != ' ' && fasta[b] != '\t' && fasta[b] != '\r') b++;
        FaRec r;
        r.name.assign(fasta + a, b - a);
        r.body = nl ? hend + 1 : fasta_len;
        size_t q = r.body;
        int64_t n = 0;
        while (q < fasta_len && fasta[q] != '>') {
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
}

// bases [s, e) of a record, appended to `out` with `width` columns per line
void append_bases(const char *fasta, const FaRec &r, int64_t s, int64_t e, int width, std::string &out) {
    int64_t at = 0;                    // bases before the line in hand
    int col = 0;
    for (size_t q = r.body; q < r.body_end && at < e;) {
        const char *nl = (const char *)memchr(fasta + q, '\n', r.body_end - q);
        size_t le = nl ? (size_t)(nl - fasta) : r.body_end;
        const size_t next = nl ? le + 1 : r.body_end;
        while (le > q && fasta[le - 1] == '\r') le--;
        const int64_t n = (int64_t)(le - q);
        const int64_t lo = std::max(s, at), hi = std::min(e, at + n);
        for (int64_t x = lo; x < hi;) {
            const int64_t take = std::min<int64_t>(hi - x, width - col);
            out.append(fasta + q + (x - at), (size_t)take);
            x += take; col += (int)take;
            if (col == width) { out += '\n'; col = 0; }
        }
        at += n;
        q = next;
    }
    if (col) out += '\n';
}

struct Span { int64_t s, e; };
void merge_sorted(std::vector<Span> &v) {
    std::sort(v.begin(), v.end(), [](const Span &a, const Span &b) { return a.s != b.s ? a.s < b.s : a.e < b.e; });
    size_t w = 0;
    for (const Span &x : v) {
        if (w && x.s <= v[w - 1].e) v[w - 1].e = std::max(v[w - 1].e, x.e);
        else v[w++] = x;
    }
    v.resize(w);
}

// the first nine columns of a PAF line [p, end): pointers to the tabs behind columns 1..9 (false: fewer columns)
bool paf_tabs(const char *paf, size_t p, size_t end, const char *t[9]) {
    size_t c = p;
    for (int k = 0; k < 9; k++) {
        t[k] = (const char *)memchr(paf + c, '\t', end - c);
        if (!t[k]) { if (k == 8) { t[k] = paf + end; return true; } return false; }
        c = (size_t)(t[k] - paf) + 1;
    }
    return true;
}

template <typename F>
int for_paf_lines(const char *paf, size_t paf_len, const char *who, F &&f) {
    size_t line_no = 0;
    for (size_t pos = 0; pos < paf_len;) {
        const char *nl = (const char *)memchr(paf + pos, '\n', paf_len - pos);
        size_t end = nl ? (size_t)(nl - paf) : paf_len;
        const size_t p = pos;
        pos = end + 1;
        line_no++;
        while (end > p && paf[end - 1] == '\r') end--;
        bool blank = true;
        for (size_t x = p; x < end && blank; x++) blank = paf[x] == ' ' || paf[x] == '\t';
        if (blank) continue;
        const char *t[9];
        if (!paf_tabs(paf, p, end, t)) { mb::set_error(std::string(who) + ": PAF line " + std::to_string(line_no) + " has fewer than 9 columns"); return MIBLAST_EINVAL; }
        const int rc = f(p, end, t);
        if (rc != MIBLAST_OK) return rc;
    }
    return MIBLAST_OK;
}

struct SeqCover { std::string name; int64_t len = -1; std::vector<Span> spans; };

// NAME|SEQLEN|START -> (NAME, SEQLEN, START); false when the name does not end in two numeric fields
bool split_sub_name(const std::string &full, std::string &base, int64_t &seq_len, int64_t &start) {
    const size_t b = full.rfind('|');
    if (b == std::string::npos || b == 0) return false;
    const size_t a = full.rfind('|', b - 1);
    if (a == std::string::npos) return false;
    char *end = nullptr;
    start = strtoll(full.c_str() + b + 1, &end, 10);
    if (end == full.c_str() + b + 1 || *end) return false;
    seq_len = strtoll(full.c_str() + a + 1, &end, 10);
    if (end == full.c_str() + a + 1 || end != full.c_str() + b) return false;
    base = full.substr(0, a);
    return true;
}

int give(const std::string &out, char **text, size_t *len) {
    char *buf = (char *)malloc(out.size() + 1);
    if (!buf) { mb::set_error("out of host memory"); return MIBLAST_ELIMIT; }
    memcpy(buf, out.data(), out.size());
    buf[out.size()] = 0;
    *text = buf; *len = out.size();
    return MIBLAST_OK;
}

template <typename F>
int guarded(F &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        mb::set_error("out of host memory");
        return MIBLAST_ELIMIT;
    } catch (const std::exception &e) {
        mb::set_error(std::string("internal: ") + e.what());
        return MIBLAST_EHIP;
    }
}

}  // namespace

extern "C" {

int mipaf_to_bed_text(const char *paf, size_t paf_len, const char *fasta, size_t fasta_len, int32_t exclude_aligned, int32_t exclude_unaligned,
                      int32_t include_inverted, int64_t min_size, char **out_text, size_t *out_len) {
    if ((!paf && paf_len) || (!fasta && fasta_len) || !out_text || !out_len) return MIBLAST_EINVAL;
    *out_text = nullptr; *out_len = 0;
    if (!exclude_aligned && !exclude_unaligned) { mb::set_error("to_bed: one of --excludeAligned / --excludeUnaligned is needed with --binary (the forms Cactus uses)"); return MIBLAST_EINVAL; }
    return guarded([&]() -> int {
        std::vector<SeqCover> seqs;
        std::unordered_map<std::string, size_t> by_name;
        const bool have_fasta = fasta && fasta_len;
        if (have_fasta) {                                               // --queryFastaFile: its sequences, in its order, with their lengths
            std::vector<FaRec> recs;
            fasta_records(fasta, fasta_len, recs);
            for (const FaRec &r : recs) {
                if (!by_name.emplace(r.name, seqs.size()).second) { mb::set_error("to_bed: sequence name " + r.name + " occurs twice in the FASTA file"); return MIBLAST_EINVAL; }
                SeqCover c; c.name = r.name; c.len = r.len;
                seqs.push_back(std::move(c));
            }
        }
        auto add = [&](const char *name, size_t name_len, int64_t len, int64_t s, int64_t e, bool query) -> int {
            const std::string key(name, name_len);
            auto it = by_name.find(key);
            if (it == by_name.end()) {
                if (have_fasta) {
                    if (!query) return MIBLAST_OK;                      // (a target of an inverted record that the FASTA file does not hold)
                    mb::set_error("to_bed: query " + key + " is not in the FASTA file");
                    return MIBLAST_EINVAL;
                }
                it = by_name.emplace(key, seqs.size()).first;           // without a FASTA file: in order of first appearance, length from the PAF
                SeqCover c; c.name = key; c.len = len;
                seqs.push_back(std::move(c));
            }
            if (e > s) seqs[it->second].spans.push_back(Span{s, e});
            return MIBLAST_OK;
        };
        int rc = for_paf_lines(paf, paf_len, "to_bed", [&](size_t p, size_t, const char *t[9]) -> int {
            int r = add(paf + p, (size_t)(t[0] - (paf + p)), field_i64(t[0] + 1, t[1]), field_i64(t[1] + 1, t[2]), field_i64(t[2] + 1, t[3]), true);
            if (r != MIBLAST_OK || !include_inverted) return r;
            return add(t[4] + 1, (size_t)(t[5] - (t[4] + 1)), field_i64(t[5] + 1, t[6]), field_i64(t[6] + 1, t[7]), field_i64(t[7] + 1, t[8]), false);
        });
        if (rc != MIBLAST_OK) return rc;
        std::string out;
        char buf[64];
        for (SeqCover &c : seqs) {
            merge_sorted(c.spans);
            auto put = [&](int64_t s, int64_t e) {
                if (e <= s || e - s < min_size) return;
                out += c.name;
                snprintf(buf, sizeof buf, "\t%lld\t%lld\n", (long long)s, (long long)e);
                out += buf;
            };
            if (exclude_unaligned) { for (const Span &x : c.spans) put(x.s, x.e); continue; }
            int64_t at = 0;
            for (const Span &x : c.spans) { put(at, x.s); at = std::max(at, x.e); }
            put(at, c.len);
        }
        return give(out, out_text, out_len);
    });
}

int mipaf_fasta_extract_text(const char *bed, size_t bed_len, const char *fasta, size_t fasta_len, int64_t flank, int64_t min_size, int32_t skip_missing,
                             char **out_text, size_t *out_len) {
    if ((!bed && bed_len) || (!fasta && fasta_len) || !out_text || !out_len) return MIBLAST_EINVAL;
    *out_text = nullptr; *out_len = 0;
    return guarded([&]() -> int {
        std::vector<FaRec> recs;
        fasta_records(fasta, fasta_len, recs);
        std::unordered_map<std::string, size_t> by_name;
        for (size_t k = 0; k < recs.size(); k++) by_name.emplace(recs[k].name, k);
        std::vector<std::vector<Span>> want(recs.size());
        size_t line_no = 0;
        for (size_t pos = 0; pos < bed_len;) {
            const char *nl = (const char *)memchr(bed + pos, '\n', bed_len - pos);
            const size_t end = nl ? (size_t)(nl - bed) : bed_len, p = pos;
            pos = end + 1;
            line_no++;
            if (end == p) continue;
            const char *t0 = (const char *)memchr(bed + p, '\t', end - p);
            const char *t1 = t0 ? (const char *)memchr(t0 + 1, '\t', (size_t)(bed + end - (t0 + 1))) : nullptr;
            if (!t1) { mb::set_error("extract: BED line " + std::to_string(line_no) + " has fewer than 3 columns"); return MIBLAST_EINVAL; }
            const std::string name(bed + p, (size_t)(t0 - (bed + p)));
            const int64_t s = field_i64(t0 + 1, t1), e = field_i64(t1 + 1, bed + end);
            auto it = by_name.find(name);
            if (it == by_name.end()) {
                if (skip_missing) continue;
                mb::set_error("extract: sequence " + name + " of the BED file is not in the FASTA file (--skipMissing passes such lines over)");
                return MIBLAST_EINVAL;
            }
            if (e - s < std::max<int64_t>(1, min_size)) continue;
            const FaRec &r = recs[it->second];
            want[it->second].push_back(Span{std::max<int64_t>(0, s - flank), std::min<int64_t>(r.len, e + flank)});
        }
        std::string out;
        char buf[64];
        for (size_t k = 0; k < recs.size(); k++) {
            merge_sorted(want[k]);
            for (const Span &x : want[k]) {
                if (x.e <= x.s) continue;
                out += '>'; out += recs[k].name;
                snprintf(buf, sizeof buf, "|%lld|%lld\n", (long long)recs[k].len, (long long)x.s);
                out += buf;
                append_bases(fasta, recs[k], x.s, x.e, 60, out);
            }
        }
        return give(out, out_text, out_len);
    });
}

int mipaf_upconvert_text(const char *paf, size_t paf_len, const char *const *fastas, const size_t *fasta_lens, size_t n_fastas, char **out_text, size_t *out_len) {
    if ((!paf && paf_len) || (n_fastas && (!fastas || !fasta_lens)) || !out_text || !out_len) return MIBLAST_EINVAL;
    *out_text = nullptr; *out_len = 0;
    return guarded([&]() -> int {
        struct Sub { int64_t s, e; std::string full; };
        std::unordered_map<std::string, std::vector<Sub>> by;
        for (size_t f = 0; f < n_fastas; f++) {
            std::vector<FaRec> recs;
            fasta_records(fastas[f], fasta_lens[f], recs);
            for (const FaRec &r : recs) {
                std::string base; int64_t seq_len = 0, start = 0;
                if (!split_sub_name(r.name, base, seq_len, start)) { mb::set_error("upconvert: record " + r.name + " is not named NAME|SEQLEN|START"); return MIBLAST_EINVAL; }
                by[base].push_back(Sub{start, start + r.len, r.name});
            }
        }
        for (auto &kv : by) std::sort(kv.second.begin(), kv.second.end(), [](const Sub &a, const Sub &b) { return a.s < b.s; });
        auto find = [&](const std::string &name, int64_t s, int64_t e, const Sub *&hit) -> int {
            hit = nullptr;
            auto it = by.find(name);
            if (it == by.end()) return MIBLAST_OK;                      // a sequence none of whose records is in the files keeps its coordinates
            const std::vector<Sub> &v = it->second;
            size_t lo = 0, hi = v.size();
            while (lo < hi) { const size_t mid = (lo + hi) / 2; if (v[mid].s <= s) lo = mid + 1; else hi = mid; }
            if (lo == 0 || !(v[lo - 1].s <= s && e <= v[lo - 1].e)) {
                mb::set_error("upconvert: " + name + ":" + std::to_string(s) + "-" + std::to_string(e) + " lies in no extracted record");
                return MIBLAST_EINVAL;
            }
            hit = &v[lo - 1];
            return MIBLAST_OK;
        };
        std::string out;
        out.reserve(paf_len + paf_len / 8);
        char buf[96];
        int rc = for_paf_lines(paf, paf_len, "upconvert", [&](size_t p, size_t end, const char *t[9]) -> int {
            const std::string qn(paf + p, (size_t)(t[0] - (paf + p))), tn(t[4] + 1, (size_t)(t[5] - (t[4] + 1)));
            const int64_t qs = field_i64(t[1] + 1, t[2]), qe = field_i64(t[2] + 1, t[3]), ts = field_i64(t[6] + 1, t[7]), te = field_i64(t[7] + 1, t[8]);
            const Sub *q = nullptr, *tg = nullptr;
            int r = find(qn, qs, qe, q);
            if (r == MIBLAST_OK) r = find(tn, ts, te, tg);
            if (r != MIBLAST_OK) return r;
            if (q) { out += q->full; snprintf(buf, sizeof buf, "\t%lld\t%lld\t%lld", (long long)(q->e - q->s), (long long)(qs - q->s), (long long)(qe - q->s)); out += buf; }
            else out.append(paf + p, (size_t)(t[3] - (paf + p)));       // columns 1-4 as they are
            out.append(t[3], (size_t)(t[4] - t[3]) + 1);                 // "\t<strand>\t"
            if (tg) { out += tg->full; snprintf(buf, sizeof buf, "\t%lld\t%lld\t%lld", (long long)(tg->e - tg->s), (long long)(ts - tg->s), (long long)(te - tg->s)); out += buf; }
            else out.append(t[4] + 1, (size_t)(t[8] - (t[4] + 1)));     // columns 6-9 as they are
            out.append(t[8], (size_t)(paf + end - t[8]));               // the rest of the line from the tab behind column 9
            out += '\n';
            return MIBLAST_OK;
        });
        if (rc != MIBLAST_OK) return rc;
        return give(out, out_text, out_len);
    });
}

int mipaf_fasta_chunk_files(const char *fasta, size_t fasta_len, const char *out_dir, int64_t chunk_size, int64_t overlap, int32_t *n_files) {
    if ((!fasta && fasta_len) || !out_dir || chunk_size <= 0 || overlap < 0) return MIBLAST_EINVAL;
    if (n_files) *n_files = 0;
    return guarded([&]() -> int {
        std::vector<FaRec> recs;
        fasta_records(fasta, fasta_len, recs);
        FILE *fh = nullptr;
        int64_t remaining = 0;
        int idx = 0;
        std::string text;
        char buf[64];
        auto flush = [&]() -> bool {
            if (!fh) return true;
            const bool ok = fwrite(text.data(), 1, text.size(), fh) == text.size();
            text.clear();
            return (fclose(fh) == 0) && ok;
        };
        for (const FaRec &r : recs)
            for (int64_t start = 0; start < r.len; start += chunk_size) {
                const int64_t end = std::min(r.len, start + chunk_size + overlap);
                if (!fh || remaining <= 0) {
                    if (!flush()) { mb::set_error("chunk: cannot write a chunk file"); return MIBLAST_EIO; }
                    const std::string path = std::string(out_dir) + "/chunk_" + std::to_string(idx++) + ".fa";
                    fh = fopen(path.c_str(), "wb");
                    if (!fh) { mb::set_error("chunk: cannot create " + path); return MIBLAST_EIO; }
                    remaining = chunk_size;
                }
                text += '>'; text += r.name;
                snprintf(buf, sizeof buf, "|%lld|%lld\n", (long long)r.len, (long long)start);
                text += buf;
                append_bases(fasta, r, start, end, 100, text);
                remaining -= end - start;
            }
        if (!flush()) { mb::set_error("chunk: cannot write a chunk file"); return MIBLAST_EIO; }
        if (n_files) *n_files = idx;
        return MIBLAST_OK;
    });
}

}  // extern "C"

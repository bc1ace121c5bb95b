// mp_paffy_main.cpp -- `paffy` front end for the sub-commands of the chaining stage, so that
// chain_alignments / chain_tile_trim_filter_one_contig can stay as they are:
//   * argv of /root/reference/src/cactus/paf/local_alignment.py:624 (invert), :638-642 (split_file), :672-681 (chain, tile,
//     trim, filter), :696-715 (filter --inputFile / --invert); input from --inputFile or stdin, PAF on stdout;
//   * `dechunk` (:352, :515), `to_bed` (:191-204, :476-480, :878-880) and `upconvert` (:899-900) are text-only and done here
//     (mp_text.cpp); any OTHER sub-command (view, add_mismatches ...) is
//     handed to the next `paffy` on PATH, so that putting <repo>/bin first on PATH does not hide the real tool;
//   * unknown option, or a foreign sub-command with no other paffy on PATH -> exit 2 with a message; no GPU for chain / tile /
//     trim -> exit 3, nothing on stdout.
// Everything is done by libmiblast.so through include/mipaf.h.
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#include <vector>

#include "../../include/mipaf.h"

static int fail(int code, const std::string &msg) {
    fprintf(stderr, "paffy (mipaf): %s\n", msg.c_str());
    return code;
}

// `paffy dechunk -i X [--query]`: NAME|SEQLEN|CHUNKSTART -> NAME, coordinates shifted by CHUNKSTART, length restored
// (Appendix B of SURVEY.md; every other column and tag is passed through untouched): mipaf_dechunk_text on the whole input
static int dechunk(const char *input, bool query_only) {
    FILE *in = input ? fopen(input, "r") : stdin;
    if (!in) return fail(1, std::string("cannot open ") + input);
    std::string text;
    char buf[1 << 16];
    for (size_t n; (n = fread(buf, 1, sizeof buf, in)) > 0;) text.append(buf, n);
    if (input) fclose(in);
    char *out = nullptr;
    size_t out_len = 0;
    if (mipaf_dechunk_text(text.data(), text.size(), query_only ? 1 : 0, &out, &out_len) != MIBLAST_OK) return fail(1, miblast_last_error());
    fwrite(out, 1, out_len, stdout);
    miblast_free(out);
    return 0;
}

static bool slurp(const char *path, std::string &text) {
    FILE *in = (!path || !strcmp(path, "-")) ? stdin : fopen(path, "rb");
    if (!in) return false;
    char buf[1 << 16];
    for (size_t n; (n = fread(buf, 1, sizeof buf, in)) > 0;) text.append(buf, n);
    if (in != stdin) fclose(in);
    return true;
}

// `paffy to_bed --binary {--excludeAligned|--excludeUnaligned} [--includeInverted] [--minSize N] [-i paf] [--queryFastaFile fa]`
// (local_alignment.py:191-204, :476-480, :878-880) and `paffy upconvert -i paf trimmed.fa ...` (:899-900): text in, text out
static int text_command(const std::string &cmd, int argc, char **argv) {
    const char *input = nullptr, *fasta = nullptr, *output = nullptr;
    bool ex_al = false, ex_un = false, inverted = false, binary = false;
    long long min_size = 0;
    std::vector<const char *> files;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--binary") { binary = true; continue; }
        if (a == "--excludeAligned") { ex_al = true; continue; }
        if (a == "--excludeUnaligned") { ex_un = true; continue; }
        if (a == "--includeInverted") { inverted = true; continue; }
        if (a[0] != '-' || a == "-") { files.push_back(argv[i]); continue; }
        if (i + 1 >= argc) return fail(2, "option " + a + " needs a value");
        const char *v = argv[++i];
        if (a == "-i" || a == "--inputFile") input = v;
        else if (a == "--queryFastaFile") fasta = v;
        else if (a == "--minSize") min_size = atoll(v);
        else if (a == "--outputFile" || a == "-o") output = v;
        else if (a == "--logLevel" || a == "-l") ;
        else return fail(2, "unknown option " + a);
    }
    std::string paf;
    if (!slurp(input, paf)) return fail(1, std::string("cannot open ") + (input ? input : "stdin"));
    char *out = nullptr;
    size_t out_len = 0;
    int rc;
    if (cmd == "to_bed") {
        if (!binary) return fail(2, "to_bed: only the --binary form is implemented (the one Cactus passes)");
        if (!files.empty()) return fail(2, "to_bed: unexpected argument");
        std::string fa;
        if (fasta && !slurp(fasta, fa)) return fail(1, std::string("cannot open ") + fasta);
        rc = mipaf_to_bed_text(paf.data(), paf.size(), fasta ? fa.data() : nullptr, fasta ? fa.size() : 0, ex_al, ex_un, inverted, min_size, &out, &out_len);
    } else {
        std::vector<std::string> texts(files.size());
        std::vector<const char *> ptrs;
        std::vector<size_t> lens;
        for (size_t k = 0; k < files.size(); k++) {
            if (!slurp(files[k], texts[k])) return fail(1, std::string("cannot open ") + files[k]);
            ptrs.push_back(texts[k].data()); lens.push_back(texts[k].size());
        }
        rc = mipaf_upconvert_text(paf.data(), paf.size(), ptrs.data(), lens.data(), ptrs.size(), &out, &out_len);
    }
    if (rc != MIBLAST_OK) return fail(1, miblast_last_error());
    FILE *f = output ? fopen(output, "wb") : stdout;
    if (!f) { miblast_free(out); return fail(1, std::string("cannot create ") + output); }
    fwrite(out, 1, out_len, f);
    if (output) fclose(f);
    miblast_free(out);
    return 0;
}

// a sub-command this front end does not provide: run the next paffy on PATH in our place
static int delegate(char **argv, bool required = true) {
    char self[PATH_MAX] = {0}, other[PATH_MAX];
    if (!realpath("/proc/self/exe", self)) self[0] = 0;
    const char *path = getenv("PATH");
    for (const char *p = path; p && *p;) {
        const char *e = strchr(p, ':');
        const std::string dir(p, e ? (size_t)(e - p) : strlen(p));
        p = e ? e + 1 : nullptr;
        if (dir.empty()) continue;
        const std::string cand = dir + "/paffy";
        if (access(cand.c_str(), X_OK) != 0 || !realpath(cand.c_str(), other) || !strcmp(other, self)) continue;
        // which implementation ran must be on record: one line per process (MIPAF_QUIET=1 silences it)
        if (const char *q = getenv("MIPAF_QUIET"); !(q && *q && strcmp(q, "0") != 0))
            fprintf(stderr, "paffy (mipaf): `%s` handed to %s%s\n", argv[1], other, required ? "" : " (MIPAF_NATIVE=1 runs the MI355X implementation instead)");
        execv(cand.c_str(), argv);
    }
    if (!required) return -1;
    return fail(2, std::string("sub-command ") + argv[1] + " is not provided by this front end (invert, chain, tile, trim, filter, split_file, dechunk, to_bed, upconvert) "
                "and no other paffy is on PATH");
}

int main(int argc, char **argv) {
    if (argc < 2) return fail(2, "usage: paffy <invert|chain|tile|trim|filter|split_file|dechunk> [options]");
    const std::string cmd = argv[1];
    if (cmd != "invert" && cmd != "chain" && cmd != "tile" && cmd != "trim" && cmd != "filter" && cmd != "split_file" && cmd != "dechunk" && cmd != "to_bed" && cmd != "upconvert")
        return delegate(argv);
    // The chaining rules of this front end are restated from paffy's description (PARITY UNPINNED: the
    // reference's paffy submodule is empty).  A real paffy further down PATH is therefore never shadowed unless asked for
    // with MIPAF_NATIVE=1; without one there is nothing to shadow and the native path runs.
    {
        const char *native = getenv("MIPAF_NATIVE");
        if (!(native && *native && strcmp(native, "0") != 0)) delegate(argv, false);      // returns only if no other paffy exists
    }
    if (cmd == "to_bed" || cmd == "upconvert") return text_command(cmd, argc, argv);
    (void)miblast_frontend_runtime_defaults(0);                    // (before the first device call: include/miblast.h)
    const char *input = nullptr, *output = nullptr, *prefix = "split_", *trim_identity = nullptr;
    mipaf_chain_params cp;
    mipaf_chain_params_default(&cp);
    cp.max_gap_length = 50000;                              // stand-alone default; Cactus always passes its own values
    long long max_tile = -1, min_chain = -1, min_length = 0;
    int invert = 0, hist_bins = 0;
    bool query_flag = false;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : nullptr; };
        const char *v = nullptr;
        if (a == "--invert") { invert = 1; continue; }
        if (a == "--query") { query_flag = true; continue; }
        if (!(v = val())) return fail(2, "option " + a + " needs a value");
        if (a == "--inputFile" || a == "-i") input = v;
        else if (a == "--outputFile" || a == "-o") output = v;
        else if (a == "--logLevel" || a == "-l") ;
        else if (a == "--maxGapLength") cp.max_gap_length = atoll(v);
        else if (a == "--chainGapOpen") cp.gap_open = atoll(v);
        else if (a == "--chainGapExtend") cp.gap_extend = atoll(v);
        else if (a == "--trimFraction") cp.trim_fraction = atof(v);
        else if (a == "--trimIdentity") trim_identity = v;
        else if (a == "--maxTileLevel") max_tile = atoll(v);
        else if (a == "--minChainScore") min_chain = atoll(v);
        else if (a == "--prefix") prefix = v;
        else if (a == "--minLength") min_length = atoll(v);
        else if (a == "--mipaf-hist-bins") hist_bins = atoi(v);   // private, never passed by Cactus: size of k_tile's LDS histogram
        else return fail(2, "unknown option " + a);
    }
    if (cmd == "dechunk") return dechunk(input, query_flag);
    const bool needs_gpu = cmd == "chain" || cmd == "tile" || cmd == "trim";
    if (cmd == "trim" && !trim_identity) return fail(2, "trim: only --trimIdentity is implemented (the option Cactus passes)");
    miblast_ctx *ctx = nullptr;
    if (needs_gpu) {
        if (miblast_device_count() <= 0) return fail(3, "no MI355X visible; this build has no CPU path");
        if (miblast_ctx_create(0, &ctx) != MIBLAST_OK) return fail(3, miblast_last_error());
    }
    mipaf_set *set = nullptr;
    int rc = mipaf_set_from_file(input ? input : "-", &set);
    if (rc == MIBLAST_OK) {
        if (cmd == "invert") rc = mipaf_invert(set);
        else if (cmd == "chain") rc = mipaf_chain(ctx, set, &cp, nullptr);
        else if (cmd == "tile") rc = mipaf_tile(ctx, set, hist_bins, nullptr);
        else if (cmd == "trim") rc = mipaf_trim(ctx, set, trim_identity, nullptr);
        else if (cmd == "filter") rc = mipaf_filter(set, max_tile, min_chain, invert);
        else rc = mipaf_split_by_query(set, prefix, min_length, nullptr);
    }
    if (rc == MIBLAST_OK && cmd != "split_file") {
        int fd = 1;
        FILE *f = nullptr;
        if (output) { f = fopen(output, "wb"); if (!f) rc = MIBLAST_EIO; else fd = fileno(f); }
        if (rc == MIBLAST_OK) rc = mipaf_set_write(set, fd);
        if (f) fclose(f);
    }
    const std::string err = rc == MIBLAST_OK ? "" : miblast_last_error();
    mipaf_set_free(set);
    if (ctx) miblast_ctx_destroy(ctx);
    return rc == MIBLAST_OK ? 0 : fail(1, err.empty() ? "cannot write output" : err);
}

// mp_paffy_main.cpp -- `paffy` front end for the sub-commands of the chaining stage, so that
// chain_alignments / chain_tile_trim_filter_one_contig can stay as they are:
//   * argv of /root/reference/src/cactus/paf/local_alignment.py:624 (invert), :638-642 (split_file), :672-681 (chain, tile,
//     trim, filter), :696-715 (filter --inputFile / --invert); input from --inputFile or stdin, PAF on stdout;
//   * unknown sub-command or option -> exit 2 with a message; no GPU for chain / tile / trim -> exit 3, nothing on stdout.
// Everything is done by libmiblast.so through include/mipaf.h.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/mipaf.h"

static int fail(int code, const std::string &msg) {
    fprintf(stderr, "paffy (mipaf): %s\n", msg.c_str());
    return code;
}

int main(int argc, char **argv) {
    if (argc < 2) return fail(2, "usage: paffy <invert|chain|tile|trim|filter|split_file> [options]");
    const std::string cmd = argv[1];
    const char *input = nullptr, *output = nullptr, *prefix = "split_", *trim_identity = nullptr;
    mipaf_chain_params cp;
    mipaf_chain_params_default(&cp);
    cp.max_gap_length = 50000;                              // stand-alone default; Cactus always passes its own values
    long long max_tile = -1, min_chain = -1, min_length = 0;
    int invert = 0, hist_bins = 0;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : nullptr; };
        const char *v = nullptr;
        if (a == "--invert") { invert = 1; continue; }
        if (a == "--query") continue;
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
    const bool needs_gpu = cmd == "chain" || cmd == "tile" || cmd == "trim";
    if (!needs_gpu && cmd != "invert" && cmd != "filter" && cmd != "split_file") return fail(2, "unknown sub-command " + cmd);
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

// mp_faffy_main.cpp -- `faffy` front end for the two sub-commands the blast phase runs, so that make_chunked_alignments and the
// trimming steps can stay as they are under CACTUS_BINARIES_MODE=local:
//   faffy chunk -c chunkSize -o overlapSize --dir D genome.fa                  /root/reference/src/cactus/paf/local_alignment.py:380-385
//   faffy extract -i bed genome.fa [--flank F] [--minSize N] [--skipMissing]   :208-216, :485-488, :890-893   (FASTA on stdout)
// Any other sub-command is handed to the next `faffy` on PATH (exit 2 when there is none).  Text handling on the host: libmiblast's
// mp_text.cpp through include/mipaf.h.  faffy is part of the absent paffy submodule: the record naming NAME|SEQLEN|START and the
// packing rule are the ones of SURVEY Appendix B / DESIGN.md section 3 (PARITY UNPINNED).
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>

#include "../../include/mipaf.h"

static int fail(int code, const std::string &msg) {
    fprintf(stderr, "faffy (mipaf): %s\n", msg.c_str());
    return code;
}

static bool slurp(const char *path, std::string &text) {
    FILE *in = (!path || !strcmp(path, "-")) ? stdin : fopen(path, "rb");
    if (!in) return false;
    char buf[1 << 16];
    for (size_t n; (n = fread(buf, 1, sizeof buf, in)) > 0;) text.append(buf, n);
    if (in != stdin) fclose(in);
    return true;
}

static int delegate(char **argv) {
    char self[PATH_MAX] = {0}, other[PATH_MAX];
    if (!realpath("/proc/self/exe", self)) self[0] = 0;
    const char *path = getenv("PATH");
    for (const char *p = path; p && *p;) {
        const char *e = strchr(p, ':');
        const std::string dir(p, e ? (size_t)(e - p) : strlen(p));
        p = e ? e + 1 : nullptr;
        if (dir.empty()) continue;
        const std::string cand = dir + "/faffy";
        if (access(cand.c_str(), X_OK) != 0 || !realpath(cand.c_str(), other) || !strcmp(other, self)) continue;
        execv(cand.c_str(), argv);
    }
    return fail(2, std::string("sub-command ") + argv[1] + " is not provided by this front end (chunk, extract) and no other faffy is on PATH");
}

int main(int argc, char **argv) {
    if (argc < 2) return fail(2, "usage: faffy <chunk|extract> [options] genome.fa");
    const std::string cmd = argv[1];
    if (cmd != "chunk" && cmd != "extract") return delegate(argv);
    const char *bed = nullptr, *dir = ".", *fasta = nullptr, *output = nullptr;
    long long chunk = 1000000, overlap = 0, flank = 0, min_size = 1;
    bool skip_missing = false;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--skipMissing") { skip_missing = true; continue; }
        if (a[0] != '-' || a == "-") { if (fasta) return fail(2, "more than one FASTA file"); fasta = argv[i]; continue; }
        if (i + 1 >= argc) return fail(2, "option " + a + " needs a value");
        const char *v = argv[++i];
        if (a == "-c" || a == "--chunkSize") chunk = atoll(v);
        else if (a == "-o" || a == "--overlapSize") overlap = atoll(v);
        else if (a == "--dir" || a == "-d") dir = v;
        else if (a == "-i" || a == "--bedFile") bed = v;
        else if (a == "--flank" || a == "-f") flank = atoll(v);
        else if (a == "--minSize" || a == "-m") min_size = atoll(v);
        else if (a == "--outputFile") output = v;
        else if (a == "--logLevel" || a == "-l") ;
        else return fail(2, "unknown option " + a);
    }
    std::string fa;
    if (!slurp(fasta, fa)) return fail(1, std::string("cannot open ") + (fasta ? fasta : "stdin"));
    if (cmd == "chunk") {
        if (chunk <= 0 || overlap < 0) return fail(2, "chunk: -c must be positive and -o non-negative");
        int32_t n = 0;
        if (mipaf_fasta_chunk_files(fa.data(), fa.size(), dir, chunk, overlap, &n) != MIBLAST_OK) return fail(1, miblast_last_error());
        return 0;
    }
    if (!bed) return fail(2, "extract: -i BED is needed");
    std::string bd;
    if (!slurp(bed, bd)) return fail(1, std::string("cannot open ") + bed);
    char *out = nullptr;
    size_t out_len = 0;
    if (mipaf_fasta_extract_text(bd.data(), bd.size(), fa.data(), fa.size(), flank, min_size, skip_missing ? 1 : 0, &out, &out_len) != MIBLAST_OK) return fail(1, miblast_last_error());
    FILE *f = output ? fopen(output, "wb") : stdout;
    if (!f) { miblast_free(out); return fail(1, std::string("cannot create ") + output); }
    fwrite(out, 1, out_len, f);
    if (output) fclose(f);
    miblast_free(out);
    return 0;
}

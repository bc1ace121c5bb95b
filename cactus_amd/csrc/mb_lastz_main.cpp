// mb_lastz_main.cpp -- bin/lastz and bin/run_kegalign: the executables the UNMODIFIED Toil job
// run_lastz (/root/reference/src/cactus/paf/local_alignment.py:29-97) finds on PATH under
// CACTUS_BINARIES_MODE=local (/root/reference/src/cactus/shared/common.py:793-795).
//
// Contract kept from the reference:
//   * argv grammar of local_alignment.py:60-68 (lastz) and :54-58 (run_kegalign, adds --num_gpu/--num_threads);
//   * PAF on stdout, nothing on stdout otherwise; empty result = zero bytes (never the "no alignment"
//     text that :85-94 has to patch for KegAlign);
//   * exit code != 0 on any problem (common.py:962-988); on success stderr stays EMPTY, because the
//     GPU branch greps it for terminate/error/fail/assert/signal/abort/... (local_alignment.py:75-83).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/miblast.h"

int main(int argc, char **argv) {
    miblast_params p;
    const char *files[2];
    int num_gpu = 1, num_threads = 1;
    bool show_stats = false;
    // private extension, never passed by Cactus: --miblast-stats prints counters as JSON on stderr
    int kept = 1;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--miblast-stats")) show_stats = true;
        else argv[kept++] = argv[i];
    }
    argc = kept;
    if (miblast_params_from_argv(argc, argv, &p, files, &num_gpu, &num_threads) != MIBLAST_OK) {
        fprintf(stderr, "lastz (miblast): %s\n", miblast_last_error());
        return 2;
    }
    // private, never set by Cactus: MIBLAST_PARSE_ONLY=1 stops after the command line has been accepted and both files could be opened
    // (tests drive the reference's unmodified job functions against this front end on a box without a GPU: tests/refjobs.py)
    if (const char *po = getenv("MIBLAST_PARSE_ONLY"); po && *po && strcmp(po, "0") != 0) {
        for (int k = 0; k < 2; k++) {
            std::string path(files[k]);
            const size_t br = path.find('[');                          // file.fa[multiple][nameparse=darkspace]
            if (br != std::string::npos) path.resize(br);
            FILE *f = fopen(path.c_str(), "rb");
            if (!f) { fprintf(stderr, "lastz (miblast): cannot open %s\n", path.c_str()); return 1; }
            fclose(f);
        }
        return 0;
    }
    bool threads_given = false;
    for (int i = 1; i < argc; i++) threads_given |= !strcmp(argv[i], "--num_threads");
    if (threads_given) miblast_set_host_threads(num_threads);      // run_kegalign form: the cores the job owns
    (void)miblast_frontend_runtime_defaults(threads_given ? num_threads : 0);      // (before the first device call: polling waits when the job owns the cores for them)
    int ndev = miblast_device_count();
    if (ndev <= 0) { fprintf(stderr, "lastz (miblast): no MI355X visible; this build has no CPU path\n"); return 3; }
    if (num_gpu > ndev) { fprintf(stderr, "lastz (miblast): --num_gpu %d but only %d visible\n", num_gpu, ndev); return 3; }
    // one context per GPU of the job; block pairs of the two files are dealt to them (include/miblast.h, miblast_multi)
    miblast_multi *ctx = nullptr;
    int rc = miblast_multi_create(num_gpu, &ctx);
    miblast_stats st;
    memset(&st, 0, sizeof st);
    if (rc == MIBLAST_OK) rc = miblast_multi_align_files(ctx, files[0], files[1], &p, 1 /* stdout */, &st);
    if (rc != MIBLAST_OK) { fprintf(stderr, "lastz (miblast): %s\n", miblast_last_error()); miblast_multi_destroy(ctx); return 1; }
    if (show_stats)
        fprintf(stderr,
                "{\"seed_lookups\":%lld,\"seed_hits\":%lld,\"hits_extended\":%lld,\"ungapped_cols\":%lld,\"hsps\":%lld,"
                "\"anchors\":%lld,\"anchors_skipped\":%lld,\"dp_sides\":%lld,\"dp_cells\":%lld,\"dp_rows\":%lld,\"alignments\":%lld,"
                "\"dp_cells_run\":%lld,\"gapped_rounds\":%lld,\"t_index\":%.6f,\"t_seed\":%.6f,\"t_gapped\":%.6f,\"t_total\":%.6f}\n",
                (long long)st.seed_lookups, (long long)st.seed_hits, (long long)st.hits_extended, (long long)st.ungapped_cols,
                (long long)st.hsps, (long long)st.anchors, (long long)st.anchors_skipped, (long long)st.dp_sides,
                (long long)st.dp_cells, (long long)st.dp_rows, (long long)st.alignments, (long long)st.dp_cells_run,
                (long long)st.gapped_rounds, st.t_index, st.t_seed, st.t_gapped, st.t_total);
    miblast_multi_destroy(ctx);
    return 0;
}

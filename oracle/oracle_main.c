/*
 * oracle_main.c -- command-line front end of the CPU ORACLE (test infrastructure).
 * Accepts the argv that run_lastz builds (/root/reference/src/cactus/paf/local_alignment.py:60-68):
 *   oracle_lastz A.fa[multiple][nameparse=darkspace] B.fa[nameparse=darkspace] --format=paf:wfmash <params>
 * and writes PAF to stdout; --counters prints the oracle counters to stderr as JSON.
 */
#define _POSIX_C_SOURCE 200809L
#include "lastz_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char *strip_actions(const char *arg) {
    char *s = strdup(arg);
    char *b = strchr(s, '[');
    if (b) *b = 0;
    return s;
}

int main(int argc, char **argv) {
    olz_params p;
    olz_params_default(&p);
    const char *files[2] = {0, 0};
    int nf = 0, counters = 0;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-') { if (nf < 2) files[nf++] = a; else { fprintf(stderr, "too many files\n"); return 2; } continue; }
        if (!strncmp(a, "--step=", 7)) p.step = atoi(a + 7);
        else if (!strncmp(a, "--ydrop=", 8)) p.ydrop = atoi(a + 8);
        else if (!strncmp(a, "--xdrop=", 8)) p.xdrop = atoi(a + 8);
        else if (!strncmp(a, "--hspthresh=", 12)) p.hspthresh = atoi(a + 12);
        else if (!strncmp(a, "--gappedthresh=", 15)) p.gappedthresh = atoi(a + 15);
        else if (!strncmp(a, "--queryhspbest=", 15)) p.queryhspbest = atoi(a + 15);
        else if (!strcmp(a, "--notransition")) p.transitions = 0;
        else if (!strcmp(a, "--transition")) p.transitions = 1;
        else if (!strcmp(a, "--noentropy")) p.entropy = 0;
        else if (!strcmp(a, "--entropy")) p.entropy = 1;
        else if (!strcmp(a, "--ungapped") || !strcmp(a, "--nogapped")) p.gapped = 0;
        else if (!strncmp(a, "--ambiguous=", 12)) p.ambiguous_n = 1;
        else if (!strncmp(a, "--format=", 9)) {
            if (!strcmp(a + 9, "paf:wfmash")) p.format = 0;
            else if (!strcmp(a + 9, "general:name1,zstart1,end1,name2,zstart2+,end2+")) p.format = 1;
            else { fprintf(stderr, "unsupported format %s\n", a + 9); return 2; }
        }
        else if (!strcmp(a, "--markend")) p.markend = 1;
        else if (!strcmp(a, "--miblast-diag=hash16")) p.diag_hash16 = 1;
        else if (!strcmp(a, "--miblast-diag=exact")) p.diag_hash16 = 0;
        else if (!strcmp(a, "--miblast-walls")) p.walls = 1;
        else if (!strcmp(a, "--strand=both")) p.strands = 0;
        else if (!strcmp(a, "--strand=plus")) p.strands = 1;
        else if (!strcmp(a, "--strand=minus")) p.strands = 2;
        else if (!strncmp(a, "--queryhsplimit=keep,nowarn:", 28)) p.queryhsplimit = atoi(a + 28);
        else if (!strncmp(a, "--querydepth=keep,nowarn:", 25)) { /* no effect with --ungapped (cactus_lastzRepeatMask.py:100) */ }
        /* the named switches of SURVEY A.9 (lastz_oracle.h); --allocate:traceback is lastz's own spelling of the last one */
        else if (!strcmp(a, "--oracle-query-softmask=ignore")) p.query_softmask = 1;
        else if (!strcmp(a, "--oracle-step-origin=sequence")) p.step_origin = 1;
        else if (!strcmp(a, "--oracle-xdrop=le") || !strcmp(a, "--miblast-xdrop=le")) p.xdrop_le = 1;          /* (--miblast-...: the spelling the MI355X front ends take: one argv serves both) */
        else if (!strcmp(a, "--miblast-xdrop=lt")) p.xdrop_le = 0;
        else if (!strcmp(a, "--oracle-hspbest-ties=later") || !strcmp(a, "--miblast-hspbest-ties=later")) p.hspbest_ties = 1;
        else if (!strcmp(a, "--miblast-hspbest-ties=earlier")) p.hspbest_ties = 0;
        else if (!strncmp(a, "--oracle-traceback-cells=", 25)) p.traceback_cells = atoll(a + 25);
        else if (!strncmp(a, "--allocate:traceback=", 21)) {
            char *end = NULL;
            double v = strtod(a + 21, &end);
            if (end && (*end == 'K' || *end == 'k')) v *= 1024.0; else if (end && (*end == 'M' || *end == 'm')) v *= 1048576.0; else if (end && (*end == 'G' || *end == 'g')) v *= 1073741824.0;
            p.traceback_cells = (int64_t)v;                  /* (one byte of traceback per DP cell) */
        }
        else if (!strcmp(a, "--counters")) counters = 1;
        else { fprintf(stderr, "unknown option %s\n", a); return 2; }
    }
    if (nf != 2) { fprintf(stderr, "usage: oracle_lastz target.fa query.fa [options]\n"); return 2; }
    char *tf = strip_actions(files[0]), *qf = strip_actions(files[1]);
    olz_seqset *T = olz_seqset_from_fasta_file(tf), *Q = olz_seqset_from_fasta_file(qf);
    if (!T || !Q) { fprintf(stderr, "cannot read input\n"); return 1; }
    olz_result *r = NULL;
    olz_align(T, Q, &p, &r);
    fwrite(r->paf, 1, r->paf_len, stdout);
    if (counters) {
        fprintf(stderr, "{\"seed_lookups\":%lld,\"seed_hits\":%lld,\"hits_extended\":%lld,\"ungapped_cols\":%lld,"
                "\"hsps\":%lld,\"anchors\":%lld,\"anchors_skipped\":%lld,\"dp_sides\":%lld,\"dp_cells\":%lld,\"dp_rows\":%lld,"
                "\"alignments\":%lld,\"t_index\":%.6f,\"t_seed\":%.6f,\"t_gapped\":%.6f,\"t_total\":%.6f}\n",
                (long long)r->c.seed_lookups, (long long)r->c.seed_hits, (long long)r->c.hits_extended, (long long)r->c.ungapped_cols,
                (long long)r->c.hsps, (long long)r->c.anchors, (long long)r->c.anchors_skipped, (long long)r->c.dp_sides,
                (long long)r->c.dp_cells, (long long)r->c.dp_rows, (long long)r->c.alignments,
                r->c.t_index, r->c.t_seed, r->c.t_gapped, r->c.t_total);
    }
    olz_result_free(r); olz_seqset_free(T); olz_seqset_free(Q); free(tf); free(qf);
    return 0;
}

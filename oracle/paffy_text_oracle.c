/* paffy_text_oracle.c -- TEST INFRASTRUCTURE (a checker; only tests/ may run it; nothing of the product links or calls it).
 *
 * Independent restatement of the text steps either side of the blast call, written from the
 * reference's call sites rather than from the product's code, and with other algorithms -- per-base counters where the product
 * sorts and merges intervals, one pass over characters where the product splits fields:
 *
 *   to_bed_extract PAF FASTA MINSIZE FLANK     `paffy to_bed --excludeAligned --binary --minSize N -i PAF --queryFastaFile FASTA`
 *                                              piped into `faffy extract -i BED FASTA --flank F`
 *                                              (/root/reference/src/cactus/paf/local_alignment.py:460-488; "paffy to_bed creates
 *                                              SequenceCountArray (2 bytes per base) for query (ingroup) sequences", :451)
 *   dechunk PAF [--query]                      `paffy dechunk -i PAF [--query]`  (:352, :515)
 *   chunk FASTA CHUNKSIZE OVERLAP              `faffy chunk -c C -o O --dir D FASTA` (:378-387) as one listing: a line "== file k"
 *                                              before the records of every chunk file
 *   trim_aligned PAF FLANK FASTA...            trim_unaligned_sequences (:861-904): `paffy to_bed --binary --excludeUnaligned
 *                                              --includeInverted` | `faffy extract --skipMissing --minSize 1 --flank F` per file |
 *                                              `paffy upconvert`, as one listing
 *
 * PARITY UNPINNED: the paffy submodule is empty in the reference tree; the record naming
 * NAME|SEQLEN|START is SURVEY Appendix B's [MEMORY] of the tool.  What this file pins is that the product's two implementations
 * (host text code, device coverage) and this third, differently built one agree.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void die(const char *msg) { fprintf(stderr, "paffy_text_oracle: %s\n", msg); exit(1); }

static char *slurp(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) die("cannot open input");
    size_t cap = 1 << 16, n = 0;
    char *buf = (char *)malloc(cap + 1);
    for (;;) {
        if (n == cap) { cap *= 2; buf = (char *)realloc(buf, cap + 1); }
        size_t got = fread(buf + n, 1, cap - n, f);
        if (!got) break;
        n += got;
    }
    fclose(f);
    buf[n] = 0;
    *len = n;
    return buf;
}

/* FASTA records: name = header up to the first blank, sequence = every non-blank character of the following lines */
typedef struct { char *name; char *seq; int64_t len; } rec_t;
static rec_t *read_fasta(const char *path, int *n_out) {
    size_t len;
    char *txt = slurp(path, &len);
    rec_t *recs = NULL;
    int n = 0, cap = 0;
    size_t i = 0;
    while (i < len) {
        if (txt[i] != '>') { while (i < len && txt[i] != '\n') i++; i++; continue; }
        size_t h0 = ++i;
        while (h0 < len && (txt[h0] == ' ' || txt[h0] == '\t')) h0++;
        size_t h1 = h0;
        while (h1 < len && txt[h1] != '\n' && txt[h1] != ' ' && txt[h1] != '\t' && txt[h1] != '\r') h1++;
        while (i < len && txt[i] != '\n') i++;
        i++;
        if (n == cap) { cap = cap ? 2 * cap : 16; recs = (rec_t *)realloc(recs, (size_t)cap * sizeof *recs); }
        rec_t *r = &recs[n++];
        r->name = (char *)malloc(h1 - h0 + 1);
        memcpy(r->name, txt + h0, h1 - h0); r->name[h1 - h0] = 0;
        size_t scap = 1024; r->seq = (char *)malloc(scap); r->len = 0;
        int at_line_start = 1;
        while (i < len && !(at_line_start && txt[i] == '>')) {
            char c = txt[i++];
            if (c == '\n') { at_line_start = 1; continue; }
            at_line_start = 0;
            if (c == '\r' || c == ' ' || c == '\t') continue;
            if ((size_t)r->len + 1 >= scap) { scap *= 2; r->seq = (char *)realloc(r->seq, scap); }
            r->seq[r->len++] = c;
        }
    }
    free(txt);
    *n_out = n;
    return recs;
}

static void put_record(const char *name, int64_t seq_len, int64_t start, const char *seq, int64_t n, int width) {
    printf(">%s|%lld|%lld\n", name, (long long)seq_len, (long long)start);
    for (int64_t x = 0; x < n; x += width) { fwrite(seq + x, 1, (size_t)((n - x) < width ? (n - x) : width), stdout); fputc('\n', stdout); }
}

static int cmd_to_bed_extract(const char *paf_path, const char *fa_path, int64_t min_size, int64_t flank) {
    int n;
    rec_t *recs = read_fasta(fa_path, &n);
    for (int a = 0; a < n; a++) for (int b = a + 1; b < n; b++) if (!strcmp(recs[a].name, recs[b].name)) die("a sequence name occurs twice");
    /* one saturating 16-bit counter per query base, like the tool's SequenceCountArray */
    uint16_t **cnt = (uint16_t **)malloc(((size_t)n + 1) * sizeof *cnt);
    for (int k = 0; k < n; k++) cnt[k] = (uint16_t *)calloc((size_t)recs[k].len + 1, 2);
    size_t len;
    char *paf = slurp(paf_path, &len);
    char *line = paf;
    while (line < paf + len) {
        char *eol = memchr(line, '\n', (size_t)(paf + len - line));
        if (!eol) eol = paf + len;
        *eol = 0;
        int blank = 1;
        for (char *c = line; *c; c++) if (*c != ' ' && *c != '\t' && *c != '\r') blank = 0;
        if (!blank) {
            char name[4096];
            long long qlen, qs, qe;
            if (sscanf(line, "%4095[^\t]\t%lld\t%lld\t%lld", name, &qlen, &qs, &qe) != 4) die("PAF line with fewer than 4 columns");
            int k = 0;
            while (k < n && strcmp(recs[k].name, name)) k++;
            if (k == n) die("PAF query name is not in the FASTA file");
            for (long long p = qs < 0 ? 0 : qs; p < qe && p < recs[k].len; p++) if (cnt[k][p] != 0xFFFF) cnt[k][p]++;
        }
        line = eol + 1;
    }
    /* uncovered runs of at least min_size bases (the BED), each widened by flank; a base is written iff some widened run holds it */
    for (int k = 0; k < n; k++) {
        const int64_t L = recs[k].len;
        uint8_t *keep = (uint8_t *)calloc((size_t)L + 1, 1);
        for (int64_t p = 0; p < L;) {
            if (cnt[k][p]) { p++; continue; }
            int64_t e = p;
            while (e < L && !cnt[k][e]) e++;
            if (e - p >= min_size)
                for (int64_t x = (p - flank < 0 ? 0 : p - flank); x < (e + flank > L ? L : e + flank); x++) keep[x] = 1;
            p = e;
        }
        /* a widened run that ends where the next one starts is the same record */
        for (int64_t p = 0; p < L;) {
            if (!keep[p]) { p++; continue; }
            int64_t e = p;
            while (e < L && keep[e]) e++;
            put_record(recs[k].name, L, p, recs[k].seq + p, e - p, 60);
            p = e;
        }
        free(keep);
    }
    return 0;
}

/* NAME|SEQLEN|START -> NAME, SEQLEN, START (the last two '|' fields) */
static int split_chunk_name(char *name, long long *seq_len, long long *start) {
    char *b = strrchr(name, '|');
    if (!b || b == name) return 0;
    *b = 0;
    char *a = strrchr(name, '|');
    if (!a) { *b = '|'; return 0; }
    char *end;
    *start = strtoll(b + 1, &end, 10);
    if (*end || end == b + 1) return 0;
    *seq_len = strtoll(a + 1, &end, 10);
    if (*end || end == a + 1) return 0;
    *a = 0;
    return 1;
}

static int cmd_dechunk(const char *paf_path, int query_only) {
    size_t len;
    char *paf = slurp(paf_path, &len);
    char *line = paf;
    while (line < paf + len) {
        char *eol = memchr(line, '\n', (size_t)(paf + len - line));
        if (!eol) eol = paf + len;
        *eol = 0;
        if (eol > line && eol[-1] == '\r') eol[-1] = 0;
        if (*line) {
            /* columns 1-4 query name / length / start / end, 5 strand, 6-9 the same for the target, then the rest untouched */
            char *col[10];
            int nc = 0;
            char *c = line;
            while (nc < 9) {
                col[nc++] = c;
                char *t = strchr(c, '\t');
                if (!t) { c = NULL; break; }
                *t = 0; c = t + 1;
            }
            if (nc < 9) die("PAF line with fewer than 9 columns");
            long long ql, qo, tl = 0, to = 0;
            if (!split_chunk_name(col[0], &ql, &qo)) die("query name is not NAME|LENGTH|START");
            if (!query_only && !split_chunk_name(col[5], &tl, &to)) die("target name is not NAME|LENGTH|START");
            printf("%s\t%lld\t%lld\t%lld\t%s\t", col[0], ql, atoll(col[2]) + qo, atoll(col[3]) + qo, col[4]);
            if (query_only) printf("%s\t%s\t%s\t%s", col[5], col[6], col[7], col[8]);
            else printf("%s\t%lld\t%lld\t%lld", col[5], tl, atoll(col[7]) + to, atoll(col[8]) + to);
            if (c) printf("\t%s", c);
            fputc('\n', stdout);
        }
        line = eol + 1;
    }
    return 0;
}

static int cmd_chunk(const char *fa_path, int64_t chunk, int64_t overlap) {
    int n;
    rec_t *recs = read_fasta(fa_path, &n);
    int64_t room = 0;               /* bases the open file may still take */
    int file_no = -1;
    for (int k = 0; k < n; k++)
        for (int64_t s = 0; s < recs[k].len; s += chunk) {
            int64_t e = s + chunk + overlap;
            if (e > recs[k].len) e = recs[k].len;
            if (file_no < 0 || room <= 0) { printf("== file %d\n", ++file_no); room = chunk; }
            put_record(recs[k].name, recs[k].len, s, recs[k].seq + s, e - s, 100);
            room -= e - s;
        }
    return 0;
}

/* trim_unaligned_sequences (/root/reference/src/cactus/paf/local_alignment.py:861-904) in one go:
 *   paffy to_bed --binary --excludeUnaligned --includeInverted -i PAF          -> coverage of every sequence by query AND target intervals
 *   faffy extract -i BED FASTA_k --skipMissing --minSize 1 --flank F           -> per file, the covered stretches widened by F
 *   paffy upconvert -i PAF trimmed_1 trimmed_2 ...                             -> the PAF in the coordinates of the extracted records
 * Output: "== file k" before the records of every trimmed file, "== paf" before the converted alignments.  Per-base counters and a
 * per-base keep mask; an alignment finds its record by walking the mask from its own first base. */
static int cmd_trim_aligned(const char *paf_path, int64_t flank, int n_files, char **paths) {
    rec_t **files = (rec_t **)malloc((size_t)n_files * sizeof *files);
    int *n_recs = (int *)malloc((size_t)n_files * sizeof *n_recs);
    uint8_t ***keep = (uint8_t ***)malloc((size_t)n_files * sizeof *keep);
    uint16_t ***cnt = (uint16_t ***)malloc((size_t)n_files * sizeof *cnt);
    for (int f = 0; f < n_files; f++) {
        files[f] = read_fasta(paths[f], &n_recs[f]);
        keep[f] = (uint8_t **)malloc(((size_t)n_recs[f] + 1) * sizeof **keep);
        cnt[f] = (uint16_t **)malloc(((size_t)n_recs[f] + 1) * sizeof **cnt);
        for (int k = 0; k < n_recs[f]; k++) { keep[f][k] = (uint8_t *)calloc((size_t)files[f][k].len + 1, 1); cnt[f][k] = (uint16_t *)calloc((size_t)files[f][k].len + 1, 2); }
    }
    size_t len;
    char *paf = slurp(paf_path, &len);
    char *copy = (char *)malloc(len + 1);
    memcpy(copy, paf, len + 1);
    /* pass 1: counters */
    for (char *line = paf; line < paf + len;) {
        char *eol = memchr(line, '\n', (size_t)(paf + len - line));
        if (!eol) eol = paf + len;
        *eol = 0;
        char qn[4096], tn[4096], strand[8];
        long long ql, qs, qe, tl, ts, te;
        if (sscanf(line, "%4095[^\t]\t%lld\t%lld\t%lld\t%7[^\t]\t%4095[^\t]\t%lld\t%lld\t%lld", qn, &ql, &qs, &qe, strand, tn, &tl, &ts, &te) == 9) {
            for (int side = 0; side < 2; side++) {
                const char *nm = side ? tn : qn;
                long long s0 = side ? ts : qs, e0 = side ? te : qe;
                for (int f = 0; f < n_files; f++) for (int k = 0; k < n_recs[f]; k++) if (!strcmp(files[f][k].name, nm))
                    for (long long p = s0 < 0 ? 0 : s0; p < e0 && p < files[f][k].len; p++) if (cnt[f][k][p] != 0xFFFF) cnt[f][k][p]++;
            }
        } else {
            int blank = 1;
            for (char *c = line; *c; c++) if (*c != ' ' && *c != '\t' && *c != '\r') blank = 0;
            if (!blank) die("PAF line with fewer than 9 columns");
        }
        line = eol + 1;
    }
    /* the records */
    for (int f = 0; f < n_files; f++) {
        printf("== file %d\n", f);
        for (int k = 0; k < n_recs[f]; k++) {
            const int64_t L = files[f][k].len;
            for (int64_t p = 0; p < L;) {
                if (!cnt[f][k][p]) { p++; continue; }
                int64_t e = p;
                while (e < L && cnt[f][k][e]) e++;
                for (int64_t x = (p - flank < 0 ? 0 : p - flank); x < (e + flank > L ? L : e + flank); x++) keep[f][k][x] = 1;
                p = e;
            }
            for (int64_t p = 0; p < L;) {
                if (!keep[f][k][p]) { p++; continue; }
                int64_t e = p;
                while (e < L && keep[f][k][e]) e++;
                put_record(files[f][k].name, L, p, files[f][k].seq + p, e - p, 60);
                p = e;
            }
        }
    }
    /* pass 2: the alignments in the records' coordinates */
    printf("== paf\n");
    for (char *line = copy; line < copy + len;) {
        char *eol = memchr(line, '\n', (size_t)(copy + len - line));
        if (!eol) eol = copy + len;
        *eol = 0;
        if (eol > line && eol[-1] == '\r') eol[-1] = 0;
        char *col[10];
        int nc = 0;
        char *c = line;
        while (nc < 9 && c) {
            col[nc++] = c;
            char *t = strchr(c, '\t');
            if (!t) { c = NULL; break; }
            *t = 0; c = t + 1;
        }
        if (nc == 9) {
            long long v[2][3] = {{atoll(col[1]), atoll(col[2]), atoll(col[3])}, {atoll(col[6]), atoll(col[7]), atoll(col[8])}};
            char name[2][4200];
            for (int side = 0; side < 2; side++) {
                const char *nm = col[side ? 5 : 0];
                snprintf(name[side], sizeof name[side], "%s", nm);
                for (int f = 0; f < n_files; f++) for (int k = 0; k < n_recs[f]; k++) if (!strcmp(files[f][k].name, nm)) {
                    const int64_t L = files[f][k].len;
                    int64_t a = v[side][1], b = v[side][1];
                    if (a >= L || !keep[f][k][a]) die("an alignment starts outside every extracted record");
                    while (a > 0 && keep[f][k][a - 1]) a--;
                    while (b < L && keep[f][k][b]) b++;
                    if (v[side][2] > b) die("an alignment ends outside its extracted record");
                    snprintf(name[side], sizeof name[side], "%s|%lld|%lld", nm, (long long)L, (long long)a);
                    v[side][0] = b - a; v[side][1] -= a; v[side][2] -= a;
                }
            }
            printf("%s\t%lld\t%lld\t%lld\t%s\t%s\t%lld\t%lld\t%lld", name[0], v[0][0], v[0][1], v[0][2], col[4], name[1], v[1][0], v[1][1], v[1][2]);
            if (c) printf("\t%s", c);
            fputc('\n', stdout);
        }
        line = eol + 1;
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 6 && !strcmp(argv[1], "to_bed_extract")) return cmd_to_bed_extract(argv[2], argv[3], atoll(argv[4]), atoll(argv[5]));
    if (argc >= 3 && !strcmp(argv[1], "dechunk")) return cmd_dechunk(argv[2], argc > 3 && !strcmp(argv[3], "--query"));
    if (argc >= 5 && !strcmp(argv[1], "chunk")) return cmd_chunk(argv[2], atoll(argv[3]), atoll(argv[4]));
    if (argc >= 5 && !strcmp(argv[1], "trim_aligned")) return cmd_trim_aligned(argv[2], atoll(argv[3]), argc - 4, argv + 4);
    die("usage: to_bed_extract PAF FASTA MINSIZE FLANK | dechunk PAF [--query] | chunk FASTA CHUNKSIZE OVERLAP | trim_aligned PAF FLANK FASTA...");
    return 2;
}

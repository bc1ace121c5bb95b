/*
 * paffy_oracle.c -- CPU ORACLE of the chaining stage (test infrastructure, NOT product code).
 *
 * Restates the `paffy` sub-commands that chain_alignments / chain_tile_trim_filter_one_contig run right after the
 * blast phase (/root/reference/src/cactus/paf/local_alignment.py:607-727):
 *
 *     paffy invert | chain | tile | trim | filter | split_file
 *
 * PARITY UNPINNED.  paffy (https://github.com/ComparativeGenomicsToolkit/paffy, Makefile:310-316 of the reference) is
 * an un-vendored submodule: /root/reference/submodules/paffy is empty, no paffy binary exists in this image and
 * the reference holds no golden PAF for these steps.  What the reference does fix is the pipeline (the call sites
 * above), the options and their values (cactus_progressive_config.xml:104-113: chainMaxGapLength 1000000,
 * chainGapOpen 5000, chainGapExtend 1, chainTrimFraction 1.0, pafTrimIdentity 0.2, minPrimaryChainScore 10000), and
 * the tags later stages read (tp:A:P / tp:A:S, tl:i:N : local_alignment.py:705-706, pipeline/cactus_workflow.py:135).
 * Everything else is a RULE fixed here (R-C*, R-T*, R-R*, R-F*; listed in DESIGN.md section 11) from the published
 * description of the tool, so that "GPU result == oracle result" is well defined; each rule is to be confirmed
 * against a real paffy binary when one is available.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may execute this program.
 * Straightforward sequential code on purpose.
 */
#define _POSIX_C_SOURCE 200809L
#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    char *qn, *tn;
    int64_t ql, qs, qe, tl, ts, te, nm, nb, mq;
    int same;                    /* 1: '+', 0: '-' (query reverse-complemented relative to the target) */
    char tp;                     /* tp:A: or 0 */
    int has_as;
    int64_t as, tile, cn, s1;    /* AS:i: ; tl:i: cn:i: s1:i: (-1 = absent) */
    uint32_t *ops;               /* len << 3 | code ; codes: 0 '=', 1 'X', 2 'M', 3 'I' (query only), 4 'D' (target only) */
    int64_t nops;
    int64_t idx;                 /* input order */
    /* scratch of the sub-commands */
    int64_t cs, pred, tqs, tqe, tts, tte, sortpos;
} Paf;

static const char OPCH[5] = {'=', 'X', 'M', 'I', 'D'};

static void die(const char *msg) {
    fprintf(stderr, "oracle_paffy: %s\n", msg);
    exit(2);
}

/* ---- PAF text <-> record.  Columns as in SURVEY.md Appendix B; of the optional tags only the ones paffy itself
 * writes are kept (tp, AS, tl, cn, s1, cg), in that order on output. ---------------------------------------------- */
static int parse_line(char *line, Paf *p) {
    char *col[64];
    int n = 0;
    for (char *s = line; n < 64;) {
        col[n++] = s;
        char *t = strchr(s, '\t');
        if (!t) break;
        *t = 0;
        s = t + 1;
    }
    if (n < 12) return -1;
    memset(p, 0, sizeof *p);
    p->qn = strdup(col[0]); p->ql = atoll(col[1]); p->qs = atoll(col[2]); p->qe = atoll(col[3]);
    p->same = col[4][0] == '+';
    p->tn = strdup(col[5]); p->tl = atoll(col[6]); p->ts = atoll(col[7]); p->te = atoll(col[8]);
    p->nm = atoll(col[9]); p->nb = atoll(col[10]); p->mq = atoll(col[11]);
    p->tile = p->cn = p->s1 = -1;
    for (int i = 12; i < n; i++) {
        char *t = col[i];
        if (!strncmp(t, "tp:A:", 5)) p->tp = t[5];
        else if (!strncmp(t, "AS:i:", 5)) { p->has_as = 1; p->as = atoll(t + 5); }
        else if (!strncmp(t, "tl:i:", 5)) p->tile = atoll(t + 5);
        else if (!strncmp(t, "cn:i:", 5)) p->cn = atoll(t + 5);
        else if (!strncmp(t, "s1:i:", 5)) p->s1 = atoll(t + 5);
        else if (!strncmp(t, "cg:Z:", 5)) {
            const char *c = t + 5;
            int64_t cap = 16;
            p->ops = malloc(cap * sizeof(uint32_t));
            while (*c && *c != '\n') {
                int64_t len = 0;
                if (*c < '0' || *c > '9') return -1;
                while (*c >= '0' && *c <= '9') len = len * 10 + (*c++ - '0');
                int code = *c == '=' ? 0 : *c == 'X' ? 1 : *c == 'M' ? 2 : *c == 'I' ? 3 : *c == 'D' ? 4 : -1;
                if (code < 0 || len <= 0 || len >= (1 << 28)) return -1;
                c++;
                if (p->nops == cap) { cap *= 2; p->ops = realloc(p->ops, cap * sizeof(uint32_t)); }
                p->ops[p->nops++] = (uint32_t)(len << 3) | (uint32_t)code;
            }
        }
    }
    return 0;
}

static void print_paf(FILE *f, const Paf *p) {
    fprintf(f, "%s\t%" PRId64 "\t%" PRId64 "\t%" PRId64 "\t%c\t%s\t%" PRId64 "\t%" PRId64 "\t%" PRId64 "\t%" PRId64 "\t%" PRId64 "\t%" PRId64,
            p->qn, p->ql, p->qs, p->qe, p->same ? '+' : '-', p->tn, p->tl, p->ts, p->te, p->nm, p->nb, p->mq);
    if (p->tp || p->tile != -1) fprintf(f, "\ttp:A:%c", p->tp ? p->tp : (p->tile > 1 ? 'S' : 'P'));
    if (p->has_as) fprintf(f, "\tAS:i:%" PRId64, p->as);
    if (p->tile != -1) fprintf(f, "\ttl:i:%" PRId64, p->tile);
    if (p->cn != -1) fprintf(f, "\tcn:i:%" PRId64, p->cn);
    if (p->s1 != -1) fprintf(f, "\ts1:i:%" PRId64, p->s1);
    if (p->ops) {
        fputs("\tcg:Z:", f);
        for (int64_t i = 0; i < p->nops; i++) fprintf(f, "%u%c", p->ops[i] >> 3, OPCH[p->ops[i] & 7]);
    }
    fputc('\n', f);
}

static Paf *read_all(FILE *f, int64_t *n_out) {
    int64_t n = 0, cap = 1024;
    Paf *v = malloc(cap * sizeof(Paf));
    char *line = NULL;
    size_t lcap = 0;
    ssize_t got;
    while ((got = getline(&line, &lcap, f)) > 0) {
        while (got > 0 && (line[got - 1] == '\n' || line[got - 1] == '\r')) line[--got] = 0;
        if (got == 0) continue;
        if (n == cap) { cap *= 2; v = realloc(v, cap * sizeof(Paf)); }
        if (parse_line(line, &v[n]) != 0) die("malformed PAF line");
        v[n].idx = n;
        n++;
    }
    free(line);
    *n_out = n;
    return v;
}

/* ---- invert (local_alignment.py:624, :411-418): swap query and target; on the '-' strand the op list is reversed
 * so that it still runs along the (new) target; I <-> D. ---------------------------------------------------------- */
static void invert(Paf *p) {
    char *s = p->qn; p->qn = p->tn; p->tn = s;
    int64_t t;
    t = p->ql; p->ql = p->tl; p->tl = t;
    t = p->qs; p->qs = p->ts; p->ts = t;
    t = p->qe; p->qe = p->te; p->te = t;
    if (!p->same)
        for (int64_t i = 0, j = p->nops - 1; i < j; i++, j--) { uint32_t o = p->ops[i]; p->ops[i] = p->ops[j]; p->ops[j] = o; }
    for (int64_t i = 0; i < p->nops; i++) {
        uint32_t c = p->ops[i] & 7;
        if (c == 3 || c == 4) p->ops[i] = (p->ops[i] & ~7u) | (c == 3 ? 4u : 3u);
    }
}

/* ---- chain (local_alignment.py:672-677).  Rules:
 * R-C1  order: (query name, target name by strcmp; '-' before '+'; query start; target start; input order).
 * R-C2  an alignment takes part with its ends pulled in by trim = (int64)(length * trimFraction / 2) on each axis
 *       (trimFraction 1.0: every alignment shrinks to its midpoint, so overlapping neighbours can still be chained).
 * R-C3  j may precede i iff same query, target and strand, j is before i in R-C1 order, and both gaps are in
 *       [0, maxGapLength]: gq = qs'_i - qe'_j ; gt = ts'_i - te'_j on '+', ts'_j - te'_i on '-'.
 * R-C4  joining costs chainGapOpen + chainGapExtend * (gq + gt).
 * R-C5  cs_i = score_i + max(0, max_j (cs_j - cost(j,i))); the predecessor is the j of the maximum when that is > 0,
 *       the first such j in R-C1 order on ties.  score = AS:i: (column 10 when the tag is absent).
 * R-C6  chains are peeled off in order of (cs descending, R-C1 order): an alignment not yet in a chain opens chain
 *       number k = 0, 1, 2 ... with chain score cs, and takes its predecessors along until one already belongs to a chain.
 * R-C7  output: chains by number, members in R-C1 order; every record gets cn:i:<k> and s1:i:<chain score>.         */
static int cmp_loc(const void *a, const void *b) {
    const Paf *x = *(Paf *const *)a, *y = *(Paf *const *)b;
    int c = strcmp(x->qn, y->qn);
    if (c) return c;
    c = strcmp(x->tn, y->tn);
    if (c) return c;
    if (x->same != y->same) return x->same - y->same;
    if (x->qs != y->qs) return x->qs < y->qs ? -1 : 1;
    if (x->ts != y->ts) return x->ts < y->ts ? -1 : 1;
    return x->idx < y->idx ? -1 : x->idx > y->idx ? 1 : 0;
}
static int cmp_cs(const void *a, const void *b) {
    const Paf *x = *(Paf *const *)a, *y = *(Paf *const *)b;
    if (x->cs != y->cs) return x->cs > y->cs ? -1 : 1;
    return x->sortpos < y->sortpos ? -1 : x->sortpos > y->sortpos ? 1 : 0;
}
static int cmp_chain_out(const void *a, const void *b) {
    const Paf *x = *(Paf *const *)a, *y = *(Paf *const *)b;
    if (x->cn != y->cn) return x->cn < y->cn ? -1 : 1;
    return x->sortpos < y->sortpos ? -1 : x->sortpos > y->sortpos ? 1 : 0;
}

static void chain(Paf *v, int64_t n, int64_t max_gap, int64_t gap_open, int64_t gap_extend, double trim_fraction, FILE *out) {
    Paf **s = malloc((n + 1) * sizeof(Paf *));
    for (int64_t i = 0; i < n; i++) s[i] = &v[i];
    qsort(s, n, sizeof(Paf *), cmp_loc);
    for (int64_t i = 0; i < n; i++) {
        Paf *p = s[i];
        p->sortpos = i;
        int64_t tq = (int64_t)((double)(p->qe - p->qs) * trim_fraction / 2.0);
        int64_t tt = (int64_t)((double)(p->te - p->ts) * trim_fraction / 2.0);
        p->tqs = p->qs + tq; p->tqe = p->qe - tq; p->tts = p->ts + tt; p->tte = p->te - tt;
    }
    for (int64_t i = 0; i < n; i++) {
        Paf *p = s[i];
        int64_t sc = p->has_as ? p->as : p->nm;
        int64_t best = 0, pred = -1;
        for (int64_t j = i - 1; j >= 0; j--) {
            Paf *q = s[j];
            if (strcmp(q->qn, p->qn) || strcmp(q->tn, p->tn) || q->same != p->same) break;
            int64_t gq = p->tqs - q->tqe;
            int64_t gt = p->same ? p->tts - q->tte : q->tts - p->tte;
            if (gq < 0 || gt < 0 || gq > max_gap || gt > max_gap) continue;
            int64_t val = q->cs - (gap_open + gap_extend * (gq + gt));
            if (val > best || (val == best && val > 0)) { best = val; pred = j; }      /* descending j: ties end at the first */
        }
        p->cs = sc + best;
        p->pred = pred;
    }
    Paf **r = malloc((n + 1) * sizeof(Paf *));
    memcpy(r, s, n * sizeof(Paf *));
    qsort(r, n, sizeof(Paf *), cmp_cs);
    for (int64_t i = 0; i < n; i++) v[i].cn = -1;
    int64_t next = 0;
    for (int64_t k = 0; k < n; k++) {
        Paf *p = r[k];
        if (p->cn != -1) continue;
        int64_t id = next++, score = p->cs;
        for (;;) {
            p->cn = id; p->s1 = score;
            if (p->pred < 0 || s[p->pred]->cn != -1) break;
            p = s[p->pred];
        }
    }
    qsort(s, n, sizeof(Paf *), cmp_chain_out);
    for (int64_t i = 0; i < n; i++) print_paf(out, s[i]);
    free(s); free(r);
}

/* ---- tile (local_alignment.py:678).  Rules:
 * R-T1  order: (s1:i: if present else AS:i: else column 10) descending, input order on ties.
 * R-T2  one counter per query base and query sequence (saturating at 32767), all 0 at the start.
 * R-T3  level = 1 + the median of the counters under the alignment's aligned query bases (ops = X M): the smallest L with
 *       2 * #(bases with counter <= L) >= #bases; 1 when there is no aligned base.
 * R-T4  after the level is taken, the counters under the aligned bases go up by one.
 * R-T5  output in R-T1 order with tl:i:<level> and tp:A:P (level 1) or tp:A:S.                                       */
typedef struct { char *name; uint16_t *cnt; int64_t len; } QSeq;
static int64_t tile_key(const Paf *p) { return p->s1 != -1 ? p->s1 : p->has_as ? p->as : p->nm; }
static int cmp_tile(const void *a, const void *b) {
    const Paf *x = *(Paf *const *)a, *y = *(Paf *const *)b;
    int64_t kx = tile_key(x), ky = tile_key(y);
    if (kx != ky) return kx > ky ? -1 : 1;
    return x->idx < y->idx ? -1 : x->idx > y->idx ? 1 : 0;
}
static void tile(Paf *v, int64_t n, FILE *out) {
    Paf **s = malloc((n + 1) * sizeof(Paf *));
    for (int64_t i = 0; i < n; i++) s[i] = &v[i];
    qsort(s, n, sizeof(Paf *), cmp_tile);
    QSeq *seqs = NULL;
    int64_t nseq = 0;
    int64_t *hist = calloc(32768, sizeof(int64_t));
    for (int64_t k = 0; k < n; k++) {
        Paf *p = s[k];
        QSeq *qs = NULL;
        for (int64_t i = 0; i < nseq; i++) if (!strcmp(seqs[i].name, p->qn)) { qs = &seqs[i]; break; }
        if (!qs) {
            seqs = realloc(seqs, (nseq + 1) * sizeof(QSeq));
            qs = &seqs[nseq++];
            qs->name = p->qn; qs->len = p->ql; qs->cnt = calloc(p->ql > 0 ? p->ql : 1, sizeof(uint16_t));
        }
        int64_t aligned = 0, maxl = 0;
        for (int pass = 0; pass < 2; pass++) {
            int64_t q = p->same ? p->qs : p->qe;                 /* on '-' the op list runs down the query */
            for (int64_t i = 0; i < p->nops; i++) {
                int64_t len = p->ops[i] >> 3;
                int code = p->ops[i] & 7;
                if (code == 4) continue;
                if (code <= 2)
                    for (int64_t j = 0; j < len; j++) {
                        int64_t b = p->same ? q + j : q - 1 - j;
                        if (b < 0 || b >= qs->len) die("tile: cigar leaves the query sequence");
                        if (pass == 0) { hist[qs->cnt[b]]++; aligned++; if (qs->cnt[b] > maxl) maxl = qs->cnt[b]; }
                        else if (qs->cnt[b] < 32767) qs->cnt[b]++;
                    }
                q += p->same ? len : -len;
            }
            if (pass == 0) {
                int64_t level = 0, cum = 0;
                if (aligned > 0)
                    for (level = 0; level <= maxl; level++) { cum += hist[level]; if (2 * cum >= aligned) break; }
                for (int64_t l = 0; l <= maxl; l++) hist[l] = 0;
                p->tile = level + 1;
                p->tp = p->tile == 1 ? 'P' : 'S';
            }
        }
    }
    for (int64_t i = 0; i < n; i++) print_paf(out, s[i]);
    for (int64_t i = 0; i < nseq; i++) free(seqs[i].cnt);
    free(seqs); free(hist); free(s);
}

/* ---- trim --trimIdentity x (local_alignment.py:679; x = pafTrimIdentity 0.2).  Rules:
 * R-R1  alignment columns: '=' / 'M' bases are matches; 'X', 'I', 'D' bases are non-matching columns.
 * R-R2  the prefix cut is the LONGEST prefix (in columns) whose identity matches/columns is < x, x taken as the exact
 *       decimal fraction num/den of its text; 0 columns when no prefix qualifies.  The suffix cut is the same on the
 *       reversed op list.  Both are taken on the untrimmed alignment.
 * R-R3  if the cuts meet or overlap the record is dropped; otherwise the cut columns are removed, the coordinates
 *       advance by the bases they held (the prefix sits at the query END on the '-' strand), columns 10 / 11 become the
 *       remaining matches / columns; AS and the other tags stay.                                                      */
static int64_t cut_columns(const uint32_t *ops, int64_t nops, int rev, int64_t num, int64_t den) {
    int64_t m = 0, c = 0, cut = 0;
    for (int64_t k = 0; k < nops; k++) {
        uint32_t o = ops[rev ? nops - 1 - k : k];
        int64_t len = o >> 3;
        int code = o & 7;
        if (code == 0 || code == 2) {
            /* (m + t) / (c + t) < num / den  <=>  t * (den - num) < num * c - den * m */
            int64_t rhs = num * c - den * m, t = 0;
            if (den == num) t = rhs > 0 ? len : 0;
            else if (rhs > 0) { t = (rhs + (den - num) - 1) / (den - num) - 1; if (t > len) t = len; }
            if (t >= 1) cut = c + t;
            m += len; c += len;
        } else {
            c += len;
            if (m * den < num * c) cut = c;
        }
    }
    return cut;
}
static void remove_columns(Paf *p, int64_t cut, int rev, int64_t *qdel, int64_t *tdel) {
    *qdel = *tdel = 0;
    int64_t k = 0;
    while (cut > 0 && k < p->nops) {
        int64_t at = rev ? p->nops - 1 - k : k;
        int64_t len = p->ops[at] >> 3;
        int code = p->ops[at] & 7;
        int64_t take = len < cut ? len : cut;
        if (code != 4) *qdel += take;
        if (code != 3) *tdel += take;
        cut -= take;
        if (take < len) { p->ops[at] = (uint32_t)((len - take) << 3) | (uint32_t)code; break; }
        k++;
    }
    if (!rev) memmove(p->ops, p->ops + k, (p->nops - k) * sizeof(uint32_t));
    p->nops -= k;
}
static void trim(Paf *v, int64_t n, int64_t num, int64_t den, FILE *out) {
    for (int64_t i = 0; i < n; i++) {
        Paf *p = &v[i];
        if (!p->ops) { print_paf(out, p); continue; }
        int64_t cols = 0;
        for (int64_t k = 0; k < p->nops; k++) cols += p->ops[k] >> 3;
        int64_t pre = cut_columns(p->ops, p->nops, 0, num, den), suf = cut_columns(p->ops, p->nops, 1, num, den);
        if (pre + suf >= cols) continue;
        int64_t qa, ta, qb, tb;
        remove_columns(p, pre, 0, &qa, &ta);
        remove_columns(p, suf, 1, &qb, &tb);
        p->ts += ta; p->te -= tb;
        if (p->same) { p->qs += qa; p->qe -= qb; } else { p->qe -= qa; p->qs += qb; }
        p->nm = p->nb = 0;
        for (int64_t k = 0; k < p->nops; k++) {
            int64_t len = p->ops[k] >> 3;
            int code = p->ops[k] & 7;
            if (code == 0 || code == 2) p->nm += len;
            p->nb += len;
        }
        print_paf(out, p);
    }
}

/* ---- filter (local_alignment.py:680-681, :696, :710-715).  Rules:
 * R-F1  a record passes iff (no --maxTileLevel or tl <= it) and (no --minChainScore or s1 >= it); absent tags count as
 *       tl = -1, s1 = -1.  --invert writes the records that do not pass.  Input order is kept.                          */
static void filter(Paf *v, int64_t n, int64_t max_tile, int64_t min_chain, int inv, FILE *out) {
    for (int64_t i = 0; i < n; i++) {
        int ok = (max_tile < 0 || v[i].tile <= max_tile) && (min_chain < 0 || v[i].s1 >= min_chain);
        if (ok != inv) print_paf(out, &v[i]);
    }
}

/* ---- split_file --query --prefix P --minLength N (local_alignment.py:638-642).  Rules:
 * R-S1  query sequences in order of first appearance; a part is closed as soon as the summed length (column 2) of its
 *       sequences reaches N; part k is written to P<k>.paf with its records in input order.                           */
static void split_file(Paf *v, int64_t n, const char *prefix, int64_t min_len) {
    char **names = NULL;
    int64_t *part_of = NULL, nn = 0, part = 0, acc = 0;
    FILE **fs = NULL;
    int64_t nfs = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t k;
        for (k = 0; k < nn; k++) if (!strcmp(names[k], v[i].qn)) break;
        if (k == nn) {
            names = realloc(names, (nn + 1) * sizeof(char *));
            part_of = realloc(part_of, (nn + 1) * sizeof(int64_t));
            names[nn] = v[i].qn; part_of[nn] = part; nn++;
            acc += v[i].ql;
            if (part == nfs) {
                char path[4096];
                snprintf(path, sizeof path, "%s%" PRId64 ".paf", prefix, part);
                fs = realloc(fs, (nfs + 1) * sizeof(FILE *));
                if (!(fs[nfs++] = fopen(path, "w"))) die("split_file: cannot open output");
            }
            if (acc >= min_len) { part++; acc = 0; }
        }
        print_paf(fs[part_of[k]], &v[i]);
    }
    for (int64_t k = 0; k < nfs; k++) fclose(fs[k]);
}

static void parse_fraction(const char *s, int64_t *num, int64_t *den) {
    *num = 0; *den = 1;
    int seen_dot = 0, digits = 0;
    for (; *s; s++) {
        if (*s == '.' && !seen_dot) { seen_dot = 1; continue; }
        if (*s < '0' || *s > '9' || ++digits > 15) die("bad fraction");
        *num = *num * 10 + (*s - '0');
        if (seen_dot) *den *= 10;
    }
    if (!digits || *num > *den) die("bad fraction (must be in [0, 1])");
}

int main(int argc, char **argv) {
    if (argc < 2) die("usage: oracle_paffy <invert|chain|tile|trim|filter|split_file> [options]");
    const char *cmd = argv[1], *input = NULL, *prefix = "split_";
    int64_t max_gap = 50000, gap_open = 5000, gap_extend = 1, max_tile = -1, min_chain = -1, min_len = 0, num = 0, den = 1;
    double trim_fraction = 1.0;
    int inv = 0;
    for (int i = 2; i < argc; i++) {
        const char *a = argv[i];
        const char *val = i + 1 < argc ? argv[i + 1] : NULL;
        if (!strcmp(a, "--invert")) inv = 1;
        else if (!strcmp(a, "--query")) ;
        else if (!val) die("option needs a value");
        else if (!strcmp(a, "--inputFile") || !strcmp(a, "-i")) { input = val; i++; }
        else if (!strcmp(a, "--maxGapLength")) { max_gap = atoll(val); i++; }
        else if (!strcmp(a, "--chainGapOpen")) { gap_open = atoll(val); i++; }
        else if (!strcmp(a, "--chainGapExtend")) { gap_extend = atoll(val); i++; }
        else if (!strcmp(a, "--trimFraction")) { trim_fraction = atof(val); i++; }
        else if (!strcmp(a, "--trimIdentity")) { parse_fraction(val, &num, &den); i++; }
        else if (!strcmp(a, "--maxTileLevel")) { max_tile = atoll(val); i++; }
        else if (!strcmp(a, "--minChainScore")) { min_chain = atoll(val); i++; }
        else if (!strcmp(a, "--prefix")) { prefix = val; i++; }
        else if (!strcmp(a, "--minLength")) { min_len = atoll(val); i++; }
        else if (!strcmp(a, "--logLevel")) i++;
        else die("unknown option");
    }
    FILE *in = input ? fopen(input, "r") : stdin;
    if (!in) die("cannot open input");
    int64_t n;
    Paf *v = read_all(in, &n);
    if (!strcmp(cmd, "invert")) { for (int64_t i = 0; i < n; i++) { invert(&v[i]); print_paf(stdout, &v[i]); } }
    else if (!strcmp(cmd, "chain")) chain(v, n, max_gap, gap_open, gap_extend, trim_fraction, stdout);
    else if (!strcmp(cmd, "tile")) tile(v, n, stdout);
    else if (!strcmp(cmd, "trim")) trim(v, n, num, den, stdout);
    else if (!strcmp(cmd, "filter")) filter(v, n, max_tile, min_chain, inv, stdout);
    else if (!strcmp(cmd, "split_file")) split_file(v, n, prefix, min_len);
    else die("unknown sub-command");
    return 0;
}

/*
 * lastz_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See lastz_oracle.h for provenance and the "PARITY UNPINNED" statement.
 *
 * Every function names the rule of SURVEY.md Appendix A it restates and the
 * reference call site that fixes its parameters.  Straightforward sequential
 * code on purpose: this is the checker, never the thing measured or shipped.
 */
#define _POSIX_C_SOURCE 200809L
#include "lastz_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define SEED_SPAN   19
#define SEED_WEIGHT 12
#define WORD_BITS   24
#define N_BUCKETS   (1u << WORD_BITS)
#define NEG         (-(1 << 29))

/* 12of19 = 1110100110010101111; offsets of the care positions */
static const int CARE[SEED_WEIGHT] = {0, 1, 2, 4, 7, 8, 11, 13, 15, 16, 17, 18};

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------ */
/* parameters: lastz defaults ; Cactus overrides them per
 * divergence via cactus_progressive_config.xml:130-137                      */
void olz_params_default(olz_params *p) {
    p->step = 1;
    p->transitions = 1;
    p->xdrop = 910;          /* 10 * sub[A][A] */
    p->ydrop = 9400;         /* O + 300 E      */
    p->hspthresh = 3000;
    p->gappedthresh = -1;    /* -1 => = hspthresh */
    p->gap_open = 400;
    p->gap_extend = 30;
    p->entropy = 1;
    p->queryhspbest = 0;
    p->ambiguous_n = 1;
    p->gapped = 1;
    p->format = 0;
    p->markend = 0;
    p->queryhsplimit = 0;
    p->diag_hash16 = 0; p->walls = 0; p->strands = 0;
    p->query_softmask = 0; p->step_origin = 0; p->xdrop_le = 0; p->hspbest_ties = 0; p->traceback_cells = 0;
}

/* ------------------------------------------------------------------------ */
/* scoring: HOXD70, N row/col = -100 under --ambiguous=iupac,100,100 */
static const int32_t HOXD70[4][4] = {
    {  91, -114,  -31, -123},
    {-114,  100, -125,  -31},
    { -31, -125,  100, -114},
    {-123,  -31, -114,   91},
};

int32_t olz_score(uint8_t a, uint8_t b, int ambiguous_n) {
    (void)ambiguous_n;
    unsigned x = a & 7u, y = b & 7u;
    if (x > 3 || y > 3) return -100;
    return HOXD70[x][y];
}

/* ------------------------------------------------------------------------ */
/* FASTA -> concatenated code array ([multiple] + nameparse=darkspace,
 * local_alignment.py:60-62 ; alphabet ACGTNacgtn after
 * preprocessor/cactus_sanitizeFastaHeaders.c:35-49 )            */
static uint8_t encode_char(unsigned char ch) {
    switch (ch) {
        case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
        case 'a': return 8; case 'c': return 9; case 'g': return 10; case 't': return 11;
        default: break;
    }
    if (ch >= 'a' && ch <= 'z') return 12;   /* n / lowercase IUPAC */
    return 4;                                /* N / anything else   */
}

olz_seqset *olz_seqset_from_fasta_mem(const char *buf, size_t len) {
    olz_seqset *s = (olz_seqset *)calloc(1, sizeof *s);
    size_t cap_c = 16;
    s->names = (char **)malloc(cap_c * sizeof(char *));
    s->starts = (int64_t *)malloc(cap_c * sizeof(int64_t));
    s->lens = (int64_t *)malloc(cap_c * sizeof(int64_t));
    s->codes = (uint8_t *)malloc(len + 2);
    int64_t pos = 0;
    int cur = -1;
    size_t i = 0;
    while (i < len) {
        if (buf[i] == '>') {
            /* close previous contig; drop it if empty */
            if (cur >= 0 && s->lens[cur] == 0) { free(s->names[cur]); s->n_contigs--; cur--; if (cur >= 0) pos = s->starts[cur] + s->lens[cur]; else pos = 0; }
            size_t j = i + 1, e;
            while (j < len && buf[j] != '\n' && buf[j] != ' ' && buf[j] != '\t' && buf[j] != '\r') j++;
            e = j;
            while (j < len && buf[j] != '\n') j++;
            if ((size_t)s->n_contigs == cap_c) {
                cap_c *= 2;
                s->names = (char **)realloc(s->names, cap_c * sizeof(char *));
                s->starts = (int64_t *)realloc(s->starts, cap_c * sizeof(int64_t));
                s->lens = (int64_t *)realloc(s->lens, cap_c * sizeof(int64_t));
            }
            cur = s->n_contigs++;
            s->names[cur] = (char *)malloc(e - (i + 1) + 1);
            memcpy(s->names[cur], buf + i + 1, e - (i + 1));
            s->names[cur][e - (i + 1)] = 0;
            if (cur > 0) s->codes[pos++] = OLZ_SEP;
            s->starts[cur] = pos;
            s->lens[cur] = 0;
            i = j + 1;
            continue;
        }
        unsigned char ch = (unsigned char)buf[i++];
        if (ch == '\n' || ch == '\r' || ch == ' ' || ch == '\t') continue;
        if (cur < 0) continue;               /* junk before first header */
        s->codes[pos++] = encode_char(ch);
        s->lens[cur]++;
    }
    if (cur >= 0 && s->lens[cur] == 0) { free(s->names[cur]); s->n_contigs--; cur--; if (cur >= 0) pos = s->starts[cur] + s->lens[cur]; else pos = 0; }
    s->total = pos;
    s->codes[pos] = OLZ_SEP;
    return s;
}

olz_seqset *olz_seqset_from_fasta_file(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)n + 1);
    size_t got = fread(buf, 1, (size_t)n, f);
    fclose(f);
    olz_seqset *s = olz_seqset_from_fasta_mem(buf, got);
    free(buf);
    return s;
}

void olz_seqset_free(olz_seqset *s) {
    if (!s) return;
    for (int i = 0; i < s->n_contigs; i++) free(s->names[i]);
    free(s->names); free(s->starts); free(s->lens); free(s->codes); free(s);
}

void olz_free(void *p) { free(p); }

static int contig_of(const olz_seqset *s, int64_t pos) {
    int lo = 0, hi = s->n_contigs - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) / 2;
        if (s->starts[mid] <= pos) lo = mid; else hi = mid - 1;
    }
    return lo;
}

/* reverse complement of each contig in place of its own slot (strand '-') */
static uint8_t *revcomp_codes(const olz_seqset *s) {
    uint8_t *r = (uint8_t *)malloc((size_t)s->total + 1);
    memcpy(r, s->codes, (size_t)s->total + 1);
    for (int c = 0; c < s->n_contigs; c++) {
        int64_t st = s->starts[c], n = s->lens[c];
        for (int64_t k = 0; k < n; k++) {
            uint8_t v = s->codes[st + n - 1 - k];
            if ((v & 7u) < 4) v = (uint8_t)((v & 8u) | (3u - (v & 7u)));
            r[st + k] = v;
        }
    }
    return r;
}

/* ------------------------------------------------------------------------ */
/* seed words: valid iff all 19 window chars are uppercase ACGT */
static inline int seedable(uint8_t c) { return c < 4; }

/* valid_from[p] helper: fills ok[p]=1 iff window [p,p+19) seedable */
/* (ignore_mask: the query_softmask switch -- lowercase ACGT seeds too; separators and N never do) */
static uint8_t *window_valid_map_m(const uint8_t *codes, int64_t n, int ignore_mask) {
    uint8_t *ok = (uint8_t *)calloc((size_t)n + 1, 1);
    int64_t run = 0;                       /* seedable run length ending at i */
    for (int64_t i = 0; i < n; i++) {
        run = (ignore_mask ? (codes[i] != OLZ_SEP && (codes[i] & 7u) < 4) : seedable(codes[i])) ? run + 1 : 0;
        if (run >= SEED_SPAN) ok[i - SEED_SPAN + 1] = 1;
    }
    return ok;
}

static inline uint32_t seed_word(const uint8_t *codes, int64_t p) {
    uint32_t w = 0;
    for (int k = 0; k < SEED_WEIGHT; k++) w = (w << 2) | (uint32_t)(codes[p + CARE[k]] & 3u);
    return w;
}

static uint8_t *window_valid_map(const uint8_t *codes, int64_t n) { return window_valid_map_m(codes, n, 0); }

/* INDEX(T, step) of SURVEY A.10: CSR, ascending p inside a bucket */
static int build_index_origin(const olz_seqset *T, int32_t step, int per_sequence, uint32_t **offsets_out, uint32_t **positions_out);
int olz_build_index(const olz_seqset *T, int32_t step, uint32_t **offsets_out, uint32_t **positions_out) {
    return build_index_origin(T, step, 0, offsets_out, positions_out);
}
/* (per_sequence: the step_origin switch -- positions are taken every `step` from the start of each sequence, not of the concatenation) */
static int build_index_origin(const olz_seqset *T, int32_t step, int per_sequence, uint32_t **offsets_out, uint32_t **positions_out) {
    int64_t n = T->total;
    uint8_t *ok = window_valid_map(T->codes, n);
    if (per_sequence && step > 1) {
        for (int c = 0; c < T->n_contigs; c++)
            for (int64_t p = T->starts[c]; p < T->starts[c] + T->lens[c]; p++)
                if ((p - T->starts[c]) % step) ok[p] = 0;
        step = 1;                                           /* (what survived above IS the per-sequence lattice) */
        for (int c = 0; c + 1 < T->n_contigs; c++) ok[T->starts[c] + T->lens[c]] = 0;
    }
    uint32_t *off = (uint32_t *)calloc((size_t)N_BUCKETS + 1, sizeof(uint32_t));
    for (int64_t p = 0; p + SEED_SPAN <= n; p += step)
        if (ok[p]) off[seed_word(T->codes, p) + 1]++;
    for (uint32_t b = 0; b < N_BUCKETS; b++) off[b + 1] += off[b];
    uint32_t total = off[N_BUCKETS];
    uint32_t *posv = (uint32_t *)malloc(((size_t)total + 1) * sizeof(uint32_t));
    uint32_t *cur = (uint32_t *)malloc((size_t)N_BUCKETS * sizeof(uint32_t));
    memcpy(cur, off, (size_t)N_BUCKETS * sizeof(uint32_t));
    for (int64_t p = 0; p + SEED_SPAN <= n; p += step)
        if (ok[p]) posv[cur[seed_word(T->codes, p)]++] = (uint32_t)p;
    free(cur); free(ok);
    *offsets_out = off; *positions_out = posv;
    return 0;
}

/* ------------------------------------------------------------------------ */
typedef struct {
    const olz_seqset *T, *Q;
    const uint8_t *tc, *qc;          /* target codes, query codes of current strand */
    const olz_params *p;
    int32_t K, L;
    olz_counters *c;
} ctx_t;

/* UNGAPPED(t_end,q_end) of SURVEY A.10 (x-drop both ways from the seed end) */
static olz_hsp ungapped(ctx_t *x, int64_t t_end, int64_t q_end) {
    const olz_seqset *T = x->T, *Q = x->Q;
    int tcg = contig_of(T, t_end - 1), qcg = contig_of(Q, q_end - 1);
    int64_t tlo = T->starts[tcg], thi = tlo + T->lens[tcg];
    int64_t qlo = Q->starts[qcg], qhi = qlo + Q->lens[qcg];
    int32_t xdrop = x->p->xdrop;
    const int32_t le = x->p->xdrop_le ? 1 : 0;          /* (run <= best - xdrop  <=>  run < best - xdrop + 1) */
    int32_t run = 0, bestL = 0, bestR = 0;
    int64_t bl = 0, br = 0;
    for (int64_t k = 1;; k++) {
        int64_t i = t_end - k, j = q_end - k;
        if (i < tlo || j < qlo) break;
        run += olz_score(x->tc[i], x->qc[j], 1);
        x->c->ungapped_cols++;
        if (run > bestL) { bestL = run; bl = k; }
        else if (run < bestL - xdrop + le) break;
    }
    run = 0;
    for (int64_t k = 0;; k++) {
        int64_t i = t_end + k, j = q_end + k;
        if (i >= thi || j >= qhi) break;
        run += olz_score(x->tc[i], x->qc[j], 1);
        x->c->ungapped_cols++;
        if (run > bestR) { bestR = run; br = k + 1; }
        else if (run < bestR - xdrop + le) break;
    }
    olz_hsp h;
    memset(&h, 0, sizeof h);
    h.t_start = (int32_t)(t_end - bl);
    h.q_start = (int32_t)(q_end - bl);
    h.len = (int32_t)(bl + br);
    h.score = bestL + bestR;
    h.seed_t_end = (int32_t)t_end;
    h.seed_q_end = (int32_t)q_end;
    h.q_contig = qcg;
    return h;
}

/* ENTROPY_OK(h) of SURVEY A.10 -- IEEE double, host libm */
static int entropy_ok(ctx_t *x, olz_hsp *h) {
    for (int k = 0; k < 4; k++) h->cnt[k] = 0;
    for (int32_t k = 0; k < h->len; k++) {
        uint8_t a = x->tc[h->t_start + k] & 7u, b = x->qc[h->q_start + k] & 7u;
        if (a < 4 && a == b) h->cnt[a]++;
    }
    if (!x->p->entropy) return 1;
    int64_t n = (int64_t)h->cnt[0] + h->cnt[1] + h->cnt[2] + h->cnt[3];
    if (n == 0) return 0;
    double H = 0.0;
    for (int k = 0; k < 4; k++) {
        if (h->cnt[k] > 0) {
            double pr = (double)h->cnt[k] / (double)n;
            H -= pr * log(pr);
        }
    }
    H /= log(4.0);
    return (double)h->score * H >= (double)x->K;
}

/* ANCHOR(h) of SURVEY A.10 / A.6 */
static void anchor_of(ctx_t *x, const olz_hsp *h, int32_t *at, int32_t *aq) {
    int32_t off;
    if (h->len <= 31) off = h->len / 2;
    else {
        int64_t sum = 0, bestsum;
        int32_t bestc = 0;
        for (int32_t k = 0; k < 31; k++) sum += olz_score(x->tc[h->t_start + k], x->qc[h->q_start + k], 1);
        bestsum = sum;
        for (int32_t c = 1; c + 31 <= h->len; c++) {
            sum += olz_score(x->tc[h->t_start + c + 30], x->qc[h->q_start + c + 30], 1);
            sum -= olz_score(x->tc[h->t_start + c - 1], x->qc[h->q_start + c - 1], 1);
            if (sum > bestsum) { bestsum = sum; bestc = c; }
        }
        off = bestc + 15;
    }
    *at = h->t_start + off;
    *aq = h->q_start + off;
}

/* ------------------------------------------------------------------------ */
/* ONE_SIDED of SURVEY A.10 / A.7.  a = target along columns, b = query along
 * rows; dir=+1 forward from (t0,q0) inclusive, dir=-1 backward from (t0-1,q0-1).
 * trace byte per cell: bits0-1 src (0 diag,1 D vertical,2 I horizontal,3 origin),
 * bit2 Dext, bit3 Iext.  ops are emitted walking back from the best cell.    */
typedef struct {
    int32_t best, bi, bj;
    int64_t cells, rows;
    uint8_t *ops;        /* one op per alignment column, in walk-back order: 0 M,2 I(query only),3 D(target only) */
    int64_t n_ops;
} side_t;

/* walls (olz_params.walls): per earlier alignment, the target interval its path occupies in every query row it spans */
typedef struct { int64_t q_lo, q_hi; int32_t *tmin, *tmax; } wall_t;
typedef struct { const wall_t *w; int64_t n; } walls_t;

static int walled(const walls_t *ws, int64_t t, int64_t q) {
    for (int64_t m = 0; m < ws->n; m++) {
        const wall_t *w = &ws->w[m];
        if (q >= w->q_lo && q < w->q_hi && w->tmin[q - w->q_lo] >= 0 && t >= w->tmin[q - w->q_lo] && t <= w->tmax[q - w->q_lo]) return 1;
    }
    return 0;
}

static side_t one_sided(ctx_t *x, int64_t t0, int64_t q0, int dir, int64_t na, int64_t nb, const walls_t *ws) {
    const int32_t O = x->p->gap_open, E = x->p->gap_extend, Y = x->p->ydrop;
    const uint8_t *tc = x->tc, *qc = x->qc;
    side_t r;
    memset(&r, 0, sizeof r);
    /* row storage sized to the widest possible row (na+1) is wasteful; grow on demand */
    int64_t cap = 4096;
    int32_t *Cp = (int32_t *)malloc((size_t)cap * 4), *Dp = (int32_t *)malloc((size_t)cap * 4);
    int32_t *Cc = (int32_t *)malloc((size_t)cap * 4), *Dc = (int32_t *)malloc((size_t)cap * 4);
    /* trace: per row offset + LY */
    int64_t rcap = 1024, tcap = 1 << 16, tlen = 0;
    int64_t *row_off = (int64_t *)malloc((size_t)rcap * 8);
    int64_t *row_ly = (int64_t *)malloc((size_t)rcap * 8);
    uint8_t *tr = (uint8_t *)malloc((size_t)tcap);
    int32_t best = 0; int64_t bi = 0, bj = 0;
    /* row 0 */
    int64_t R0 = 0;
    if (Y >= O) { R0 = (Y - O) / E; if (R0 > na) R0 = na; }
    int64_t LY = 0, RY = R0 + 1;           /* prev-row alive window [LY,RY) ; buffers hold cols LY.. at index j-LY */
    while (RY - LY + 2 > cap) { cap *= 2; Cp = realloc(Cp, (size_t)cap * 4); Dp = realloc(Dp, (size_t)cap * 4); Cc = realloc(Cc, (size_t)cap * 4); Dc = realloc(Dc, (size_t)cap * 4); }
    row_off[0] = 0; row_ly[0] = 0;
    for (int64_t j = 0; j <= R0; j++) {
        Cp[j] = (j == 0) ? 0 : -(O + (int32_t)j * E);
        Dp[j] = NEG;
        if (tlen + 1 > tcap) { tcap *= 2; tr = realloc(tr, (size_t)tcap); }
        tr[tlen++] = (j == 0) ? 3 : (uint8_t)(2 | (j >= 2 ? 8 : 0));
    }
    r.cells += R0 + 1;
    int64_t nrows = 1;
    for (int64_t i = 1; i <= nb; i++) {
        if (x->p->traceback_cells > 0 && tlen >= x->p->traceback_cells) break;      /* the traceback_cells switch: the memory of one traceback is spent */
        uint8_t bq = qc[dir > 0 ? q0 + i - 1 : q0 - i];
        if (nrows + 1 > rcap) { rcap *= 2; row_off = realloc(row_off, (size_t)rcap * 8); row_ly = realloc(row_ly, (size_t)rcap * 8); }
        row_off[nrows] = tlen; row_ly[nrows] = LY;
        int32_t Iv = NEG, Cleft = NEG;
        int64_t first_alive = -1, last_alive = -1;
        int64_t j;
        for (j = LY; j <= na; j++) {
            int64_t idx = j - LY;
            if (idx + 2 > cap) { cap *= 2; Cp = realloc(Cp, (size_t)cap * 4); Dp = realloc(Dp, (size_t)cap * 4); Cc = realloc(Cc, (size_t)cap * 4); Dc = realloc(Dc, (size_t)cap * 4); }
            int32_t diag = NEG, Dv = NEG, Dext = 0, Iext = 0;
            if (j - 1 >= LY && j - 1 < RY) {
                uint8_t at = tc[dir > 0 ? t0 + j - 1 : t0 - j];
                diag = Cp[idx - 1] + olz_score(at, bq, 1);
            }
            if (j < RY) {
                int32_t ext = Dp[idx] - E, opn = Cp[idx] - O - E;
                if (ext >= opn) { Dv = ext; Dext = 1; } else { Dv = opn; }
            }
            {
                int32_t ext = Iv - E, opn = Cleft - O - E;
                if (ext >= opn) { Iv = ext; Iext = 1; } else { Iv = opn; }
            }
            int32_t Cv; uint8_t src;
            if (diag >= Dv && diag >= Iv) { Cv = diag; src = 0; }
            else if (Dv >= Iv) { Cv = Dv; src = 1; }
            else { Cv = Iv; src = 2; }
            r.cells++;
            /* walls switch: the cell pairs target base j with query base i; if that pair lies on an earlier path the cell is
             * dead and neither gap state survives it */
            int blocked = ws && ws->n && j >= 1 && walled(ws, dir > 0 ? t0 + j - 1 : t0 - j, dir > 0 ? q0 + i - 1 : q0 - i);
            if (blocked) { Cv = NEG; Dv = NEG; Iv = NEG; }
            if (Cv > best) { best = Cv; bi = i; bj = j; }
            int alive = !blocked && (Cv >= best - Y);
            if (!alive) Cv = NEG;
            Cc[idx] = Cv; Dc[idx] = Dv; Cleft = Cv;
            if (tlen + 1 > tcap) { tcap *= 2; tr = realloc(tr, (size_t)tcap); }
            tr[tlen++] = (uint8_t)(src | (Dext ? 4 : 0) | (Iext ? 8 : 0));
            if (alive) { if (first_alive < 0) first_alive = j; last_alive = j; }
            else if (j >= RY) break;
        }
        nrows++;
        if (first_alive < 0) break;
        /* re-base buffers to the new window */
        int64_t nLY = first_alive, nRY = last_alive + 1;
        memmove(Cp, Cc + (nLY - LY), (size_t)(nRY - nLY) * 4);
        memmove(Dp, Dc + (nLY - LY), (size_t)(nRY - nLY) * 4);
        LY = nLY; RY = nRY;
    }
    r.rows = nrows;
    r.best = best; r.bi = (int32_t)bi; r.bj = (int32_t)bj;
    /* traceback */
    r.ops = (uint8_t *)malloc((size_t)(bi + bj + 1));
    {
        int64_t i = bi, j = bj; int state = 0;       /* 0 C, 1 D, 2 I */
        while (i > 0 || j > 0) {
            uint8_t tb = tr[row_off[i] + (j - row_ly[i])];
            if (state == 0) {
                int src = tb & 3;
                if (src == 0) { r.ops[r.n_ops++] = 0; i--; j--; }
                else if (src == 1) state = 1;
                else if (src == 2) state = 2;
                else break;
            } else if (state == 1) {
                r.ops[r.n_ops++] = 2; if (!(tb & 4)) state = 0; i--;
            } else {
                r.ops[r.n_ops++] = 3; if (!(tb & 8)) state = 0; j--;
            }
        }
    }
    free(Cp); free(Dp); free(Cc); free(Dc); free(row_off); free(row_ly); free(tr);
    return r;
}

/* ------------------------------------------------------------------------ */
typedef struct { int32_t t, q, score, hsp; } anchor_t;

static int cmp_anchor(const void *a, const void *b) {
    const anchor_t *x = (const anchor_t *)a, *y = (const anchor_t *)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    if (x->q != y->q) return x->q < y->q ? -1 : 1;
    return 0;
}

typedef struct { int32_t score; int64_t ord; } rank_t;
static int cmp_rank(const void *a, const void *b) {
    const rank_t *x = (const rank_t *)a, *y = (const rank_t *)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    return x->ord < y->ord ? -1 : (x->ord > y->ord ? 1 : 0);
}
static int cmp_rank_later(const void *a, const void *b) {      /* the hspbest_ties switch: of equal scores the later found first */
    const rank_t *x = (const rank_t *)a, *y = (const rank_t *)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    return x->ord > y->ord ? -1 : (x->ord < y->ord ? 1 : 0);
}
static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* growable arrays */
#define PUSH(arr, n, cap, val) do { if ((n) == (cap)) { (cap) = (cap) ? (cap) * 2 : 256; (arr) = realloc((arr), (size_t)(cap) * sizeof *(arr)); } (arr)[(n)++] = (val); } while (0)

typedef struct { char *s; size_t n, cap; } sbuf;
static void sb_put(sbuf *b, const char *s, size_t n) {
    if (b->n + n + 1 > b->cap) { while (b->n + n + 1 > b->cap) b->cap = b->cap ? b->cap * 2 : 4096; b->s = realloc(b->s, b->cap); }
    memcpy(b->s + b->n, s, n); b->n += n; b->s[b->n] = 0;
}

int olz_align(const olz_seqset *T, const olz_seqset *Q, const olz_params *pp, olz_result **out) {
    olz_result *res = (olz_result *)calloc(1, sizeof *res);
    olz_params p = *pp;
    if (p.gappedthresh < 0) p.gappedthresh = p.hspthresh;
    if (p.step < 1) p.step = 1;
    ctx_t x;
    x.T = T; x.Q = Q; x.p = &p; x.K = p.hspthresh; x.L = p.gappedthresh; x.c = &res->c;
    x.tc = T->codes;
    double t_begin = now_s();

    uint32_t *off = NULL, *posv = NULL;
    build_index_origin(T, p.step, p.step_origin, &off, &posv);
    res->c.t_index = now_s() - t_begin;

    uint8_t *qrc = revcomp_codes(Q);
    int64_t ndiag = T->total + Q->total + 2;
    if (ndiag < 65536) ndiag = 65536;                      /* diag_hash16 indexes 2^16 entries */
    int32_t *extent = (int32_t *)malloc((size_t)ndiag * 4);

    olz_hsp *hsps = NULL; int64_t nh = 0, caph = 0;          /* all strands, filtered */
    olz_aln *alns = NULL; int64_t na_ = 0, capa = 0;
    uint32_t *ops = NULL; int64_t nops = 0, capo = 0;
    sbuf paf = {0, 0, 0};

    /* per-strand HSP lists, kept to run gapped after both searches (order of
     * computation is irrelevant; output order is fixed below: per query contig,
     * '+' then '-', anchor-processing order -- SURVEY A.8) */
    typedef struct { int64_t a0, a1; } range_t;
    range_t *aln_ranges = (range_t *)calloc((size_t)Q->n_contigs * 2 + 1, sizeof(range_t));

    for (int strand = 0; strand < 2; strand++) {
        if ((p.strands == 1 && strand == 1) || (p.strands == 2 && strand == 0)) continue;      /* --strand=plus / minus */
        double t0s = now_s();
        x.qc = strand ? qrc : Q->codes;
        memset(extent, 0, (size_t)ndiag * 4);
        uint8_t *qok = window_valid_map_m(x.qc, Q->total, p.query_softmask);
        olz_hsp *sh = NULL; int64_t nsh = 0, capsh = 0;       /* this strand, found order */
        /* SEARCH(Qs) of SURVEY A.10 */
        for (int64_t q = 0; q + SEED_SPAN <= Q->total; q++) {
            if (!qok[q]) continue;
            uint32_t w = seed_word(x.qc, q);
            int nvar = p.transitions ? 1 + SEED_WEIGHT : 1;
            for (int v = 0; v < nvar; v++) {
                uint32_t wv = (v == 0) ? w : (w ^ (2u << (2 * (SEED_WEIGHT - v))));
                res->c.seed_lookups++;
                uint32_t b0 = off[wv], b1 = off[wv + 1];
                for (uint32_t k = b1; k > b0; k--) {          /* descending target position */
                    int64_t pt = posv[k - 1];
                    int64_t t_end = pt + SEED_SPAN, q_end = q + SEED_SPAN;
                    res->c.seed_hits++;
                    int64_t d = p.diag_hash16 ? ((t_end - q_end) & 0xFFFF) : t_end - q_end + Q->total;      /* A.4: exact / lastz's 16-bit hash */
                    if (q_end <= extent[d]) continue;          /* suppression rule (A.9 #4) */
                    olz_hsp h = ungapped(&x, t_end, q_end);
                    res->c.hits_extended++;
                    extent[d] = h.q_start + h.len;
                    if (h.score >= x.K) {
                        res->c.hsps_pre_entropy++;
                        h.strand = strand;
                        if (entropy_ok(&x, &h)) PUSH(sh, nsh, capsh, h);
                    }
                }
            }
        }
        free(qok);
        res->c.t_seed += now_s() - t0s;

        /* --queryhsplimit=keep,nowarn:N (cactus_lastzRepeatMask.py:36 lastzOpts): the search of a query stops once it
         * has N HSPs; what was found until then is kept -- i.e. the first N in found order, per query contig & strand */
        if (p.queryhsplimit > 0) {
            int64_t *seen = (int64_t *)calloc((size_t)Q->n_contigs + 1, sizeof(int64_t));
            int64_t w = 0;
            for (int64_t k = 0; k < nsh; k++)
                if (seen[sh[k].q_contig]++ < p.queryhsplimit) sh[w++] = sh[k];
            nsh = w;
            free(seen);
        }
        /* --queryhspbest=N : per query contig & strand keep the N best, ties by found order */
        if (p.queryhspbest > 0) {
            olz_hsp *kept = NULL; int64_t nk = 0, capk = 0;
            for (int qc_i = 0; qc_i < Q->n_contigs; qc_i++) {
                rank_t *rk = NULL; int64_t nr = 0, capr = 0;
                for (int64_t k = 0; k < nsh; k++) if (sh[k].q_contig == qc_i) { rank_t e = {sh[k].score, k}; PUSH(rk, nr, capr, e); }
                if (nr > p.queryhspbest) {
                    qsort(rk, (size_t)nr, sizeof *rk, p.hspbest_ties ? cmp_rank_later : cmp_rank);
                    int64_t *keep = (int64_t *)malloc((size_t)p.queryhspbest * 8);
                    for (int64_t k = 0; k < p.queryhspbest; k++) keep[k] = rk[k].ord;
                    qsort(keep, (size_t)p.queryhspbest, 8, cmp_i64);
                    for (int64_t k = 0; k < p.queryhspbest; k++) PUSH(kept, nk, capk, sh[keep[k]]);
                    free(keep);
                } else {
                    for (int64_t k = 0; k < nr; k++) PUSH(kept, nk, capk, sh[rk[k].ord]);
                }
                free(rk);
            }
            free(sh); sh = kept; nsh = nk; capsh = capk;
        }
        for (int64_t k = 0; k < nsh; k++) PUSH(hsps, nh, caph, sh[k]);
        res->c.hsps += nsh;

        /* gapped stage, per query contig */
        double t0g = now_s();
        if (p.gapped) {
            for (int qc_i = 0; qc_i < Q->n_contigs; qc_i++) {
                anchor_t *an = NULL; int64_t nan_ = 0, capan = 0;
                for (int64_t k = 0; k < nsh; k++) if (sh[k].q_contig == qc_i) {
                    anchor_t a; a.score = sh[k].score; a.hsp = (int32_t)k;
                    anchor_of(&x, &sh[k], &a.t, &a.q);
                    PUSH(an, nan_, capan, a);
                }
                qsort(an, (size_t)nan_, sizeof *an, cmp_anchor);
                res->c.anchors += nan_;
                int64_t first_aln = na_;
                int64_t qlo = Q->starts[qc_i], qhi = qlo + Q->lens[qc_i];
                wall_t *wl = NULL; int64_t nwl = 0, capwl = 0;         /* paths of this unit's alignments (walls switch) */
                for (int64_t k = 0; k < nan_; k++) {
                    int32_t at = an[k].t, aq = an[k].q;
                    int covered = 0;
                    for (int64_t m = first_aln; m < na_ && !covered; m++) {
                        const olz_aln *A = &alns[m];
                        int32_t d = at - aq;
                        if (at >= A->t_lo && at < A->t_hi && aq >= A->q_lo && aq < A->q_hi && d >= A->dmin && d <= A->dmax) covered = 1;
                    }
                    if (covered) { res->c.anchors_skipped++; continue; }
                    int tcg = contig_of(T, at);
                    int64_t tlo = T->starts[tcg], thi = tlo + T->lens[tcg];
                    walls_t ws; ws.w = wl; ws.n = p.walls ? nwl : 0;
                    side_t R = one_sided(&x, at, aq, +1, thi - at, qhi - aq, &ws);
                    side_t Ls = one_sided(&x, at, aq, -1, at - tlo, aq - qlo, &ws);
                    res->c.dp_sides += 2;
                    res->c.dp_cells += R.cells + Ls.cells;
                    res->c.dp_rows += R.rows + Ls.rows;
                    int32_t score = R.best + Ls.best;
                    if (score >= x.L) {
                        olz_aln A;
                        memset(&A, 0, sizeof A);
                        A.strand = strand; A.q_contig = qc_i; A.t_contig = tcg;
                        A.t_lo = at - Ls.bj; A.t_hi = at + R.bj;
                        A.q_lo = aq - Ls.bi; A.q_hi = aq + R.bi;
                        A.score = score; A.anchor_t = at; A.anchor_q = aq;
                        A.ops_off = nops;
                        /* columns in forward order: left side walk-back order is already
                         * forward; right side walk-back order must be reversed */
                        int64_t tt = A.t_lo, qq = A.q_lo;
                        int32_t dmin = 0x7fffffff, dmax = -0x7fffffff - 1;
                        int64_t ncol = Ls.n_ops + R.n_ops;
                        uint32_t cur_op = 0, cur_len = 0;
                        wall_t W; W.q_lo = A.q_lo; W.q_hi = A.q_hi; W.tmin = W.tmax = NULL;
                        if (p.walls) {
                            W.tmin = (int32_t *)malloc((size_t)(A.q_hi - A.q_lo + 1) * 4); W.tmax = (int32_t *)malloc((size_t)(A.q_hi - A.q_lo + 1) * 4);
                            for (int64_t r = 0; r <= A.q_hi - A.q_lo; r++) { W.tmin[r] = -1; W.tmax[r] = -1; }
                        }
                        for (int64_t c = 0; c < ncol; c++) {
                            uint8_t o = (c < Ls.n_ops) ? Ls.ops[c] : R.ops[R.n_ops - 1 - (c - Ls.n_ops)];
                            uint32_t op;
                            if (o == 0) {
                                uint8_t a = x.tc[tt] & 7u, b = x.qc[qq] & 7u;
                                op = (a < 4 && a == b) ? 0u : 1u;
                                int32_t d = (int32_t)(tt - qq);
                                if (d < dmin) dmin = d;
                                if (d > dmax) dmax = d;
                                if (p.walls) { int64_t r = qq - A.q_lo; if (W.tmin[r] < 0) W.tmin[r] = (int32_t)tt; W.tmax[r] = (int32_t)tt; }
                                tt++; qq++;
                            } else if (o == 2) { op = 2; qq++; }
                            else { op = 3; tt++; }
                            if (cur_len && op == cur_op) cur_len++;
                            else { if (cur_len) PUSH(ops, nops, capo, (cur_len << 2) | cur_op); cur_op = op; cur_len = 1; }
                        }
                        if (cur_len) PUSH(ops, nops, capo, (cur_len << 2) | cur_op);
                        A.n_ops = nops - A.ops_off;
                        A.dmin = dmin; A.dmax = dmax;
                        PUSH(alns, na_, capa, A);
                        if (p.walls) PUSH(wl, nwl, capwl, W);
                    }
                    free(R.ops); free(Ls.ops);
                }
                aln_ranges[qc_i * 2 + strand].a0 = first_aln;
                aln_ranges[qc_i * 2 + strand].a1 = na_;
                for (int64_t m = 0; m < nwl; m++) { free(wl[m].tmin); free(wl[m].tmax); }
                free(wl);
                free(an);
            }
        }
        res->c.t_gapped += now_s() - t0g;
        free(sh);
    }

    /* output order + PAF text (`--format=paf:wfmash`,
     * local_alignment.py:68): query = 2nd file, target = 1st file */
    olz_aln *ordered = (olz_aln *)malloc((size_t)(na_ + 1) * sizeof *ordered);
    int64_t no = 0;
    for (int qc_i = 0; qc_i < Q->n_contigs; qc_i++)
        for (int strand = 0; strand < 2; strand++)
            for (int64_t m = aln_ranges[qc_i * 2 + strand].a0; m < aln_ranges[qc_i * 2 + strand].a1; m++)
                ordered[no++] = alns[m];
    char line[512];
    for (int64_t m = 0; m < no; m++) {
        const olz_aln *A = &ordered[m];
        int64_t qst = Q->starts[A->q_contig], qlen = Q->lens[A->q_contig];
        int64_t tst = T->starts[A->t_contig], tlen = T->lens[A->t_contig];
        int64_t qs = A->q_lo - qst, qe = A->q_hi - qst;
        if (A->strand) { int64_t s2 = qlen - qe, e2 = qlen - qs; qs = s2; qe = e2; }
        int64_t nmatch = 0, alen = 0;
        for (int64_t k = 0; k < A->n_ops; k++) {
            uint32_t o = ops[A->ops_off + k];
            alen += o >> 2;
            if ((o & 3u) == 0) nmatch += o >> 2;
        }
        int n = snprintf(line, sizeof line, "%s\t%lld\t%lld\t%lld\t%c\t%s\t%lld\t%lld\t%lld\t%lld\t%lld\t255\tAS:i:%d\tcg:Z:",
                         Q->names[A->q_contig], (long long)qlen, (long long)qs, (long long)qe, A->strand ? '-' : '+',
                         T->names[A->t_contig], (long long)tlen, (long long)(A->t_lo - tst), (long long)(A->t_hi - tst),
                         (long long)nmatch, (long long)alen, A->score);
        sb_put(&paf, line, (size_t)n);
        for (int64_t k = 0; k < A->n_ops; k++) {
            uint32_t o = ops[A->ops_off + k];
            n = snprintf(line, sizeof line, "%u%c", o >> 2, "=XID"[o & 3u]);
            sb_put(&paf, line, (size_t)n);
        }
        sb_put(&paf, "\n", 1);
    }
    if (p.format == 1) {
        /* --format=general:name1,zstart1,end1,name2,zstart2+,end2+ (cactus_lastzRepeatMask.py:104): one line per HSP,
         * target interval then query interval on the '+' strand; per query contig, '+' strand HSPs then '-' strand */
        const char *hdr = "#name1\tzstart1\tend1\tname2\tzstart2+\tend2+\n";
        sb_put(&paf, hdr, strlen(hdr));
        /* hsps[] holds strand 0 then strand 1, each in found order (q ascending => contig ascending): merge the two */
        int64_t split = 0;
        while (split < nh && hsps[split].strand == 0) split++;
        int64_t a = 0, b = split;
        while (a < split || b < nh) {
            int take_a;
            if (a >= split) take_a = 0;
            else if (b >= nh) take_a = 1;
            else take_a = hsps[a].q_contig <= hsps[b].q_contig;
            const olz_hsp *h = take_a ? &hsps[a++] : &hsps[b++];
            int qc_i = h->q_contig, strand = h->strand;
            int tcg = contig_of(T, h->t_start);
            int64_t qst = Q->starts[qc_i], qlen = Q->lens[qc_i];
            int64_t qs = h->q_start - qst, qe = qs + h->len;
            if (strand) { int64_t s2 = qlen - qe, e2 = qlen - qs; qs = s2; qe = e2; }
            int n = snprintf(line, sizeof line, "%s\t%lld\t%lld\t%s\t%lld\t%lld\n", T->names[tcg],
                             (long long)(h->t_start - T->starts[tcg]), (long long)(h->t_start - T->starts[tcg] + h->len),
                             Q->names[qc_i], (long long)qs, (long long)qe);
            sb_put(&paf, line, (size_t)n);
        }
    }
    if (p.markend) sb_put(&paf, "# lastz end-of-file\n", 20);
    if (!paf.s) { paf.s = (char *)calloc(1, 1); }
    free(alns);
    res->alns = ordered; res->n_alns = no;
    res->c.alignments = no;
    res->hsps = hsps; res->n_hsps = nh;
    res->ops = ops; res->n_ops = nops;
    res->paf = paf.s; res->paf_len = paf.n;
    res->c.t_total = now_s() - t_begin;
    free(aln_ranges); free(extent); free(qrc); free(off); free(posv);
    *out = res;
    return 0;
}

void olz_result_free(olz_result *r) {
    if (!r) return;
    free(r->paf); free(r->hsps); free(r->alns); free(r->ops); free(r);
}

/*
 * lastz_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-threaded restatement of the lastz seeded gapped aligner
 * that Cactus's blast phase shells out to
 *   (/root/reference/src/cactus/paf/local_alignment.py:29-97 `run_lastz`,
 *    command line built at :60-68, parameter sets at
 *    /root/reference/src/cactus/cactus_progressive_config.xml:130-137).
 *
 * The arithmetic itself lives in the third-party `lastz` submodule
 * (/root/reference/.gitmodules:25-27), whose directory is EMPTY in the
 * reference tree, and in no other file of the reference.  This oracle therefore
 * restates lastz's published algorithm as fixed in SURVEY.md Appendix A.10
 * ("normative pseudo-code").
 *
 * *** PARITY UNPINNED ***: the reference holds no golden vector, known-answer
 * test or fixture for this path at PAF/cigar level, and no
 * lastz binary can be built or run here.  The only pins that exist are the
 * parameter strings (api/tests/cactusParamsTest.c:16-17) and the structural
 * PAF contract enforced by caf (caf/impl/pinchIterator.c:59-121); both are
 * checked in tests/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use this code.  The product (cactus_amd/csrc) never links or calls it.
 */
#ifndef LASTZ_ORACLE_H
#define LASTZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- parameters (lastz command-line options used by Cactus) ------------- */
typedef struct olz_params {
    int32_t step;          /* --step=N          (default 1)                     */
    int32_t transitions;   /* 1 = 12of19 with one transition, 0 = --notransition */
    int32_t xdrop;         /* ungapped x-drop   (default 10*sub[A][A] = 910)    */
    int32_t ydrop;         /* --ydrop=N         (default O+300E = 9400)         */
    int32_t hspthresh;     /* --hspthresh=N  K  (default 3000)                  */
    int32_t gappedthresh;  /* --gappedthresh=N L (default = K)                  */
    int32_t gap_open;      /* O = 400                                           */
    int32_t gap_extend;    /* E = 30                                            */
    int32_t entropy;       /* 1 = entropy-adjusted HSP filter (default)         */
    int32_t queryhspbest;  /* --queryhspbest=N (0 = unlimited)                  */
    int32_t ambiguous_n;   /* 1 = --ambiguous=iupac,100,100 (N scores -100)     */
    int32_t gapped;        /* 1 = gapped stage on; 0 = --ungapped / --nogapped  */
    /* repeat-masker call site (cactus_lastzRepeatMask.py:97-105) */
    int32_t format;        /* 0 paf:wfmash ; 1 general:name1,zstart1,end1,name2,zstart2+,end2+ (ungapped HSPs) */
    int32_t markend;       /* --markend */
    int32_t queryhsplimit; /* --queryhsplimit=keep,nowarn:N (0 = off): first N HSPs found per query contig and strand */
    /* Named switches for the two points of SURVEY A.9 (#4, #8) where a real lastz may differ from the rules fixed in A.10.
     * Both default to 0 (= A.10, what the MI355X path implements and every parity test uses); they exist so that the day a
     * lastz binary is at hand (tests/test_p1_lastz_binary.py) the oracle can be flipped to the other reading at once.
     * (The bounded traceback -- --allocate:traceback, default 80 MiB, an alignment whose DP outgrows it is truncated -- is the
     * traceback_cells switch further down; it can only matter for sides of ~5e5 rows and more, i.e. at chunk scale and low
     * divergence: DESIGN.md section 3.)                                                                                       */
    int32_t diag_hash16;   /* 1: diagonal suppression state indexed by (t_end - q_end) & 0xFFFF as lastz's diagEnd[] (A.4): hits on
                              colliding diagonals are silently dropped; sequential in hit generation order                   */
    int32_t walls;         /* 1: base pairs on the path of an earlier alignment of the same query sequence and strand are hard
                              walls for later DPs (A.7): a cell pairing such bases is dead and no gap passes through it      */
    int32_t strands;       /* --strand=both|plus|minus: 0 both, 1 only '+', 2 only '-' (the strands are searched and extended
                              independently of each other: A.8)                                                                */
    /* Further named switches for SURVEY A.9 (round 5; oracle side only -- the product implements the A.10 reading, and the two above).
     * Each is a reading of upstream lastz that A.10 decided AGAINST on a "medium / low confidence" item; 0 = A.10.  With a lastz
     * binary at hand tests/test_p1_lastz_binary.py runs every combination that explains a difference (bisection by switch).    */
    int32_t query_softmask;  /* A.9 #2.  0: lowercase query bases do not seed (A.1 rule "apply to both"); 1: only the TARGET's soft-masking
                                keeps a position out of the seed search -- a lowercase query window of ACGT seeds like an uppercase one   */
    int32_t step_origin;     /* A.9 #5.  0: --step counts from position 0 of the concatenated target; 1: from the start of every sequence of
                                a [multiple] target (indexed positions p with (p - start of p's sequence) % step == 0)                   */
    int32_t xdrop_le;        /* A.9 #9.  0: an ungapped walk stops when run < best - xdrop; 1: when run <= best - xdrop                     */
    int32_t hspbest_ties;    /* A.9 #11. 0: --queryhspbest keeps the EARLIER found of equal-scoring HSPs at the cut; 1: the later found    */
    int64_t traceback_cells; /* lastz's bounded traceback (--allocate:traceback=<bytes>, default 80 MiB).  > 0: a
                                one-sided DP that has stored this many cells evaluates no further row -- the alignment ends at the best
                                cell found until then and the rest of the homology is left to a later anchor.  0: unbounded (A.10).       */
} olz_params;

void olz_params_default(olz_params *p);

/* ---- sequences ---------------------------------------------------------- */
/* code byte: bits 0..2 = base (0..3 = A,C,G,T ; 4 = N/other), bit 3 = lowercase
 * (soft-masked); 0xFF = separator between contigs of a concatenated set.     */
#define OLZ_SEP 0xFFu

typedef struct olz_seqset {
    int32_t   n_contigs;
    char    **names;      /* nameparse=darkspace: header up to first blank     */
    int64_t  *starts;     /* start of contig in the concatenation              */
    int64_t  *lens;
    int64_t   total;      /* concatenated length incl. one separator between   */
    uint8_t  *codes;      /* total+1 bytes (trailing separator)                */
} olz_seqset;

olz_seqset *olz_seqset_from_fasta_mem(const char *buf, size_t len);
olz_seqset *olz_seqset_from_fasta_file(const char *path);
void        olz_seqset_free(olz_seqset *s);

/* ---- stage records (shared layout with the product's debug exports) ----- */
typedef struct olz_hsp {
    int32_t t_start, q_start, len, score;   /* concatenated coords; q on the strand searched */
    int32_t seed_t_end, seed_q_end;         /* the seed hit that produced it   */
    int32_t cnt[4];                         /* identical A,C,G,T columns       */
    int32_t strand;                         /* 0 '+', 1 '-'                    */
    int32_t q_contig;                       /* query contig index              */
} olz_hsp;

typedef struct olz_aln {
    int32_t strand, q_contig, t_contig;
    int32_t t_lo, t_hi, q_lo, q_hi;         /* concatenated, strand coords     */
    int32_t score;
    int32_t dmin, dmax;
    int32_t anchor_t, anchor_q;
    int64_t ops_off, n_ops;                 /* into olz_result.ops             */
} olz_aln;

typedef struct olz_counters {
    int64_t seed_lookups, seed_hits, hits_extended, ungapped_cols;
    int64_t hsps_pre_entropy, hsps, anchors, anchors_skipped;
    int64_t dp_sides, dp_cells, dp_rows, alignments;
    double  t_index, t_seed, t_gapped, t_total;    /* seconds (CPU wall)       */
} olz_counters;

typedef struct olz_result {
    char     *paf;      size_t paf_len;
    olz_hsp  *hsps;     int64_t n_hsps;     /* after entropy + queryhspbest, found order */
    olz_aln  *alns;     int64_t n_alns;     /* output order                     */
    uint32_t *ops;      int64_t n_ops;      /* (len<<2)|op ; op 0 '=',1 'X',2 'I',3 'D' */
    olz_counters c;
} olz_result;

int  olz_align(const olz_seqset *T, const olz_seqset *Q, const olz_params *p, olz_result **out);
void olz_result_free(olz_result *r);

/* seed index export (stage parity): offsets[2^24+1], positions[offsets[2^24]] */
int  olz_build_index(const olz_seqset *T, int32_t step, uint32_t **offsets, uint32_t **positions);
void olz_free(void *p);

/* substitution score between two code bytes (case-insensitive) */
int32_t olz_score(uint8_t a, uint8_t b, int ambiguous_n);

#ifdef __cplusplus
}
#endif
#endif

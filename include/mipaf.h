/*
 * mipaf.h -- C ABI of the chaining stage of libmiblast.so (MI355X / gfx950).
 *
 * The step right after the blast phase: chain_alignments and
 * chain_tile_trim_filter_one_contig (/root/reference/src/cactus/paf/local_alignment.py:607-727) pipe the PAF of
 * all chunk pairs through
 *
 *     paffy invert | paffy chain | paffy tile | paffy trim | paffy filter | paffy chain | paffy filter
 *
 * The reference binds these as SUBPROCESSES (cactus_call, local_alignment.py:624,:684-691); bin/paffy honours that argv
 * for the sub-commands named here, and the functions below are the same code for in-process use.  Each function
 * cites the call site it replaces.  paffy itself is an absent submodule of the reference: the rules these functions
 * implement are the ones written down in DESIGN.md section 11 and restated by oracle/paffy_oracle.c.
 *
 * Conventions are those of miblast.h: int return (MIBLAST_OK or < 0), miblast_last_error(), no exceptions and no
 * abort() across the ABI, caller-owned inputs, library-allocated outputs freed with the matching *_free /
 * miblast_free.  chain, tile and trim run on the GPU of the context and return MIBLAST_ENODEV without one (there is no
 * CPU path); parse, invert, filter, split and write are text handling on the host.
 */
#ifndef MIPAF_H
#define MIPAF_H

#include "miblast.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A list of PAF records in order (paffy's in-memory stList of Paf).  Of the optional tags the ones paffy itself
 * writes are kept: tp:A, AS:i, tl:i, cn:i, s1:i, cg:Z -- written back in that order.                              */
typedef struct mipaf_set mipaf_set;

int mipaf_set_from_mem(const char *text, size_t len, mipaf_set **out);
int mipaf_set_from_file(const char *path, mipaf_set **out);             /* paffy's --inputFile                       */
void mipaf_set_free(mipaf_set *s);
int64_t mipaf_set_size(const mipaf_set *s);
/* PAF text of the set; *text is allocated by the library (miblast_free).                                         */
int mipaf_set_text(const mipaf_set *s, char **text, size_t *len);
int mipaf_set_write(const mipaf_set *s, int fd);                        /* paffy's stdout / --outputFile             */

/* `paffy invert` (local_alignment.py:624 and :411-418): query <-> target.                                        */
int mipaf_invert(mipaf_set *s);

/* `paffy dechunk [--query]` (local_alignment.py:352 and :515) on PAF text: NAME|SEQLEN|CHUNKSTART -> NAME, start / end
 * shifted by CHUNKSTART, length restored, on the query and (unless query_only) the target; every other column and
 * tag passes through untouched, so this works on the text and not on a mipaf_set.  *out is freed with miblast_free. */
int mipaf_dechunk_text(const char *paf, size_t len, int32_t query_only, char **out, size_t *out_len);

/* `paffy to_bed --excludeAligned --binary --minSize N -i paf --queryFastaFile fa` followed by `faffy extract -i bed fa --flank F`
 * (make_ingroup_to_outgroup_alignments_2, local_alignment.py:469-489) on text: the stretches of the QUERY sequences of `fasta`
 * that no alignment of `paf` covers, at least min_size long, widened by flank (widened stretches that touch are merged), as
 * FASTA records NAME|SEQLEN|START with 60 columns per line, in file order.  Host-side text code like mipaf_dechunk_text: this
 * is the step between two outgroup calls of a chain, and its latency is part of the blast phase.  *out is freed with
 * miblast_free; a PAF query name that is not in `fasta` is an error.                                                        */
int mipaf_unaligned_fasta(const char *paf, size_t paf_len, const char *fasta, size_t fasta_len, int64_t min_size, int64_t flank,
                          char **out, size_t *out_len);

/* ---- the other text steps of the blast phase (mp_text.cpp; front ends bin/paffy to_bed|upconvert, bin/faffy chunk|extract) ----------
 * `paffy to_bed --binary {--excludeAligned|--excludeUnaligned} [--includeInverted] [--minSize N] -i paf [--queryFastaFile fa]`
 * (local_alignment.py:191-204, :476-480, :878-880): BED lines NAME<tab>START<tab>END of the stretches no alignment covers
 * (exclude_aligned) or of the stretches some alignment covers (exclude_unaligned), at least min_size long.  Coverage is by the query
 * intervals and, with include_inverted, by the target intervals as well.  With a FASTA text its sequences are reported in its order
 * (a query that is not in it is an error); without one (fasta NULL) the sequences of the PAF in order of first appearance, lengths
 * from the PAF.  *out is freed with miblast_free.                                                                                  */
int mipaf_to_bed_text(const char *paf, size_t paf_len, const char *fasta, size_t fasta_len, int32_t exclude_aligned, int32_t exclude_unaligned,
                      int32_t include_inverted, int64_t min_size, char **out, size_t *out_len);
/* `faffy extract -i bed fa [--flank F] [--minSize N] [--skipMissing]` (:208-216, :485-488, :890-893): the BED intervals of at least
 * min_size bases, widened by flank (widened intervals that touch are one), as FASTA records NAME|SEQLEN|START with 60 columns per line,
 * in file order.  skip_missing: BED lines naming a sequence that is not in `fasta` are passed over instead of being an error.       */
int mipaf_fasta_extract_text(const char *bed, size_t bed_len, const char *fasta, size_t fasta_len, int64_t flank, int64_t min_size, int32_t skip_missing,
                             char **out, size_t *out_len);
/* `paffy upconvert -i paf trimmed_1.fa trimmed_2.fa ...` (:899-900): the alignments rewritten to refer to the extracted sub-sequences
 * (records NAME|SEQLEN|START of the given FASTA texts) -- the inverse of dechunk; other columns and tags pass through.              */
int mipaf_upconvert_text(const char *paf, size_t paf_len, const char *const *fastas, const size_t *fasta_lens, size_t n_fastas, char **out, size_t *out_len);
/* `faffy chunk -c chunkSize -o overlapSize --dir D fa` (:378-387): records NAME|SEQLEN|START of chunkSize + overlapSize bases, 100
 * columns per line, packed into files D/chunk_<k>.fa of about chunkSize bases; *n_files (may be NULL) = files written.             */
int mipaf_fasta_chunk_files(const char *fasta, size_t fasta_len, const char *out_dir, int64_t chunk_size, int64_t overlap, int32_t *n_files);

typedef struct mipaf_chain_params {     /* `paffy chain` options (local_alignment.py:672-677, values xml:108-111)  */
    int64_t max_gap_length;             /* --maxGapLength  (chainMaxGapLength 1000000)                            */
    int64_t gap_open;                   /* --chainGapOpen  (chainGapOpen 5000)                                    */
    int64_t gap_extend;                 /* --chainGapExtend (chainGapExtend 1)                                    */
    double trim_fraction;               /* --trimFraction  (chainTrimFraction 1.0)                                */
} mipaf_chain_params;
void mipaf_chain_params_default(mipaf_chain_params *p);                 /* the values of cactus_progressive_config.xml */

typedef struct mipaf_stats {            /* of the last chain / tile / trim call; milliseconds are HIP-event times  */
    int64_t records, groups, query_sequences, ops;
    int64_t chain_pairs;                /* predecessor candidates inspected by the chain DP                       */
    double t_sort_ms, t_chain_dp_ms, t_tile_ms, t_trim_ms;
    double t_total_s;                   /* host wall time of the call                                             */
} mipaf_stats;

/* `paffy chain` (local_alignment.py:672-677, first and second use :684-690): orders the records, links them into
 * chains and writes cn:i / s1:i (rules R-C1..R-C7).  stats may be NULL.                                          */
int mipaf_chain(miblast_ctx *ctx, mipaf_set *s, const mipaf_chain_params *p, mipaf_stats *stats);
/* `paffy tile` (:678): tl:i / tp:A from the median cover of the query bases (R-T1..R-T5).  hist_bins = 0: the
 * sort-based levelling (all alignments at once; falls back to the counter walk when a pile-up would need more than
 * $MIPAF_TILE_MAX_PIECES, default 2^28, pieces).  hist_bins > 0 forces the counter walk with an LDS histogram of
 * that many bins.  The result does not depend on the choice.                                                     */
int mipaf_tile(miblast_ctx *ctx, mipaf_set *s, int32_t hist_bins, mipaf_stats *stats);
/* `paffy trim --trimIdentity x` (:679; x = pafTrimIdentity "0.2"): x is the decimal text, at most 6 digits.      */
int mipaf_trim(miblast_ctx *ctx, mipaf_set *s, const char *trim_identity, mipaf_stats *stats);
/* `paffy filter [--maxTileLevel L] [--minChainScore S] [--invert]` (:680-681,:696,:710-715); -1 = option absent. */
int mipaf_filter(mipaf_set *s, int64_t max_tile_level, int64_t min_chain_score, int32_t invert);
/* `paffy split_file --query --prefix P --minLength N` (:638-642): writes P<k>.paf, returns the number of parts.  */
int mipaf_split_by_query(const mipaf_set *s, const char *prefix, int64_t min_length, int32_t *n_parts);

/* One chain_tile_trim_filter_one_contig job (local_alignment.py:660-727) without the pipes: chain | tile | trim |
 * filter --maxTileLevel 1 | chain | filter --minChainScore S, and with output_secondary != 0 the second branch of
 * that function (secondaries first, then the primaries that keep their chain score, then the demoted ones as
 * tp:A:S tl:i:2).  The set is replaced by the job's output.                                                      */
int mipaf_chain_tile_trim_filter(miblast_ctx *ctx, mipaf_set *s, const mipaf_chain_params *p, const char *trim_identity,
                                 int64_t min_primary_chain_score, int32_t output_secondary, mipaf_stats *stats);

#ifdef __cplusplus
}
#endif
#endif

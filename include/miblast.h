/*
 * miblast.h -- C ABI of libmiblast.so, the MI355X-native replacement for the external
 * `lastz` / `run_kegalign` executables that Cactus's blast phase shells out to.
 *
 * The reference has NO FFI for this path: the boundary is a subprocess command line built by
 *   run_lastz                /root/reference/src/cactus/paf/local_alignment.py:29-97
 *     argv                   local_alignment.py:60-68   (lastz A.fa[multiple][nameparse=darkspace] B.fa[...] --format=paf:wfmash <opts>)
 *     GPU argv               local_alignment.py:54-58   (run_kegalign A.fa B.fa --format=paf:wfmash <opts> --num_gpu G --num_threads C)
 *     process launch         /root/reference/src/cactus/shared/common.py:732-994 (cactus_call: stdout -> file, exit code, stderr)
 * Each entry point below names the piece of that interface it replaces.  The same code is also
 * linked into bin/lastz and bin/run_kegalign (cactus_amd/csrc/mb_lastz_main.cpp) so that the
 * unmodified run_lastz job works with CACTUS_BINARIES_MODE=local; INTEGRATION.md shows the
 * ctypes binding a maintainer would add for in-process use.
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative
 * MIBLAST_E* code and never throws, aborts or writes to stderr; miblast_last_error() returns a
 * thread-local message.  Callers own the buffers they pass in; buffers returned by the library
 * stay valid until the owning handle is freed.
 */
#ifndef MIBLAST_H
#define MIBLAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIBLAST_OK         0
#define MIBLAST_EINVAL    (-1)   /* bad argument / unknown option (lastz: exit != 0, common.py:962-988) */
#define MIBLAST_EIO       (-2)   /* cannot read input / write output                                    */
#define MIBLAST_ENODEV    (-3)   /* no usable gfx950 device: the library has NO CPU fallback             */
#define MIBLAST_EHIP      (-4)   /* a HIP runtime call returned non-success                               */
#define MIBLAST_ELIMIT    (-5)   /* input exceeds an implementation limit               */

/* ---- alignment parameters = the lastz options Cactus passes ------------------------------
 * (cactus_progressive_config.xml:130-146 <lastzArguments>/<kegalignArguments>; lastz defaults
 * per SURVEY.md A.2).  Same field order as the oracle's olz_params on purpose: tests memcpy. */
typedef struct miblast_params {
    int32_t step;          /* --step=N                                   default 1    */
    int32_t transitions;   /* 0 = --notransition                         default 1    */
    int32_t xdrop;         /* ungapped x-drop                            default 910  */
    int32_t ydrop;         /* --ydrop=N                                  default 9400 */
    int32_t hspthresh;     /* --hspthresh=N                              default 3000 */
    int32_t gappedthresh;  /* --gappedthresh=N (-1: same as hspthresh)   default -1   */
    int32_t gap_open;      /* 400                                                     */
    int32_t gap_extend;    /* 30                                                      */
    int32_t entropy;       /* entropy-adjusted HSP filter                default 1    */
    int32_t queryhspbest;  /* --queryhspbest=N, 0 = unlimited            default 0    */
    int32_t ambiguous_n;   /* --ambiguous=iupac,100,100                  default 1    */
    int32_t gapped;        /* 0 = --ungapped / --nogapped                default 1    */
    /* second call site of the same binary, the lastz repeat masker
     * (/root/reference/src/cactus/preprocessor/lastzRepeatMasking/cactus_lastzRepeatMask.py:97-105): */
    int32_t format;        /* 0 = paf:wfmash ; 1 = general:name1,zstart1,end1,name2,zstart2+,end2+ (HSP list, needs gapped=0) */
    int32_t markend;       /* --markend: terminate the output with "# lastz end-of-file"                  */
    int32_t queryhsplimit; /* --queryhsplimit=keep,nowarn:N : per query sequence and strand keep only the first N HSPs found; 0 = off */
    /* The oracle's named comparison switches (oracle/lastz_oracle.h), kept in the same place so that the
     * two structs stay copy-compatible.  The MI355X path implements both (mb_hash16.h; the WALLS variant of the 4-wave DP kernel) and
     * gives the oracle's bytes with each; a call they cannot serve (diag=hash16 beyond one seed batch, more than 1 024 earlier
     * alignments per unit with walls, blocked inputs) is refused with MIBLAST_ELIMIT, never answered in the default reading.     */
    int32_t diag_hash16;   /* --miblast-diag=hash16 */
    int32_t walls;         /* --miblast-walls       */
    /* --strand=both|plus|minus (lastz's own option; Cactus never passes it): 0 both, 1 only the query's '+' strand, 2 only its '-'
     * strand.  A query strand's HSPs and alignments do not depend on the other strand's, and a pair's PAF is, query sequence by query
     * sequence, the '+' lines followed by the '-' lines: (chunk pair, strand) is the exact work unit below the chunk pair
     * (cactus_amd.multigpu.merge_strand_pafs puts the halves together).                                                 */
    int32_t strands;
    /* Round 5: the oracle's further comparison switches for SURVEY A.9 (oracle/lastz_oracle.h), same names, same order -- the structs stay
     * copy-compatible up to here.  The MI355X path implements the two that cost nothing: xdrop_le (A.9 #9: an ungapped walk stops at
     * run <= best - xdrop; --miblast-xdrop=le) and hspbest_ties (A.9 #11: --queryhspbest keeps the LATER found of equal scores at the cut;
     * --miblast-hspbest-ties=later).  query_softmask (#2) and step_origin (#5) are oracle-side only: a call that sets them is refused
     * with MIBLAST_EINVAL, never answered in the default reading.                                                                      */
    int32_t query_softmask;
    int32_t step_origin;
    int32_t xdrop_le;
    int32_t hspbest_ties;
} miblast_params;

void miblast_params_default(miblast_params *p);
/* sizeof(miblast_params) as THIS build of the library has it.  The struct has grown between rounds (18 -> 22 fields) and carries no version
 * field: a binding checks its own struct against this once at load (cactus_amd/miblast.py does) instead of letting
 * miblast_params_from_argv write past a shorter one.                                                                                       */
size_t miblast_params_size(void);
/* What the shipped front ends (bin/lastz, bin/run_kegalign, bin/paffy) call first thing in main(), BEFORE the first device call: with
 * `threads` host threads (<= 0: the cores of the affinity mask) of 24 or more the ROCm runtime is told to poll its completion signals
 * (HSA_ENABLE_INTERRUPT=0: ~25 us less per wait, a spinning core per waiting thread) -- bench.py's own rule -- unless the caller's
 * environment has set the variable or MIBLAST_POLL=0.  The LIBRARY does not do this on its own any more (round 5 did, in its static
 * constructor: a host application that merely loaded libmiblast.so got busy-spinning waits); MIBLAST_POLL=1 asks a library user's
 * process for it explicitly.  Returns 1 when polling was switched on, else 0.                                                          */
int miblast_frontend_runtime_defaults(int threads);

/* Replaces lastz's own option parser for the argv run_lastz builds (local_alignment.py:60-68).
 * argv[0] is ignored.  On return files[0]/files[1] point INTO argv (target, query) with any
 * trailing [actions] left in place; num_gpu / num_threads receive --num_gpu / --num_threads
 * (local_alignment.py:58) or 1 / 1.  Unknown option -> MIBLAST_EINVAL.                        */
int miblast_params_from_argv(int argc, char **argv, miblast_params *p, const char *files[2],
                             int *num_gpu, int *num_threads);

/* ---- device context ---------------------------------------------------------------------- */
typedef struct miblast_ctx miblast_ctx;

/* Replaces `count_nvidia_gpus` (/root/reference/src/cactus/shared/configWrapper.py:307).      */
int miblast_device_count(void);                                        /* < 0: $MIBLAST_DEVICE_MAP names a device that is not there */
/* Host threads of this process that parse, sort anchors, merge traces and format PAF beside the GPU (the calling thread
 * included).  Replaces KegAlign's `--num_threads C`, which run_lastz sets to job.cores (local_alignment.py:58): a job
 * must not occupy cores the workflow engine gave to other jobs.  n = 0: automatic = min(16, cores of the affinity mask
 * and of the cgroup CPU quota), or $MIBLAST_THREADS.  Returns the number now in use, or MIBLAST_EINVAL while an
 * alignment call is running in another thread.                                                                     */
int miblast_set_host_threads(int n);
/* One context per process per GPU (ordinal after HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES). */
int miblast_ctx_create(int device, miblast_ctx **ctx);
/* Several contexts on one GPU serve independent lastz jobs at the same time (a Toil node runs as many as its cores allow,
 * local_alignment.py:395-405).  level < 0: this context's launches yield to those of the other contexts on the device -- for jobs
 * nothing waits for, beside a chain of jobs that depend on one another (an ingroup against its outgroups, :460-526); level > 0: they
 * go first; 0: the default.  Not while a call is running on the context.                                              */
int miblast_ctx_set_priority(miblast_ctx *ctx, int level);
void miblast_ctx_destroy(miblast_ctx *ctx);

/* ---- sequence sets ------------------------------------------------------------------------
 * A FASTA file as lastz loads it with [multiple][nameparse=darkspace] (local_alignment.py:60-62):
 * all records concatenated with one separator, names cut at the first blank.  The set is encoded
 * on the host and made resident in HBM at creation, so miblast_align() starts from device memory. */
typedef struct miblast_seqset miblast_seqset;

int miblast_seqset_from_fasta_file(miblast_ctx *ctx, const char *path, miblast_seqset **out);
int miblast_seqset_from_fasta_mem(miblast_ctx *ctx, const char *buf, size_t len, miblast_seqset **out);
void miblast_seqset_free(miblast_seqset *s);
/* What the library derives from a resident set stays resident with it: the seed position table of a target per --step (SURVEY 8e:
 * "index built once per owned target chunk, stays resident" while the query chunks of make_chunked_alignments' job list stream
 * through it, /root/reference/src/cactus/paf/local_alignment.py:395-405), the '-' strand and the packed form of a query.  They go
 * with the set (miblast_seqset_free); this call frees all of them at once -- they are made again by the next job that needs them
 * (a benchmark step that must pay for its tables calls it first).  Not while an alignment call is running.                       */
void miblast_drop_derived(void);
/* Outgroup trimming between two blast calls, on the device.  Replaces, for the n ingroup -> outgroup chains of a dependency level at
 * once, the pipe `paffy to_bed --excludeAligned --binary --minSize N` | `faffy extract --flank F` and the re-reading of its output
 * (/root/reference/src/cactus/paf/local_alignment.py:460-499, trimMinSize / trimFlanking of cactus_progressive_config.xml:116-117):
 * the per-base coverage of the RESIDENT query set queries[k] by the query intervals of the alignments in pafs[k] is taken on the
 * device; what no alignment covers -- stretches of at least min_size bases, widened by flank on both sides, touching ones merged --
 * is gathered on the device into a new resident set whose records are named NAME|SEQLEN|START (faffy's sub-sequence names, undone
 * by `paffy dechunk --query`).  out[k] = that set, or NULL when nothing is left.  Same names, lengths and bases as
 * mipaf_unaligned_fasta + miblast_seqset_from_fasta_mem on the same inputs.  A PAF query name that is not in the set: MIBLAST_EINVAL. */
int miblast_seqsets_unaligned(miblast_ctx *ctx, size_t n, const miblast_seqset *const *queries, const char *const *pafs, const size_t *paf_lens,
                              int64_t min_size, int64_t flank, miblast_seqset **out);
/* FASTA text of a set (60 columns per line; ACGTN, lower case where soft-masked): free with miblast_free.                          */
int miblast_seqset_fasta(const miblast_seqset *s, char **text, size_t *len);
int32_t miblast_seqset_n_contigs(const miblast_seqset *s);
int64_t miblast_seqset_total(const miblast_seqset *s);           /* concatenated length */
const char *miblast_seqset_name(const miblast_seqset *s, int32_t i);
int64_t miblast_seqset_start(const miblast_seqset *s, int32_t i);
int64_t miblast_seqset_len(const miblast_seqset *s, int32_t i);

/* ---- results ------------------------------------------------------------------------------- */
typedef struct miblast_hsp {            /* an ungapped HSP that passed --hspthresh (+entropy, +queryhspbest) */
    int32_t t_start, q_start, len, score;   /* concatenated coordinates; q on the strand searched */
    int32_t seed_t_end, seed_q_end;
    int32_t cnt[4];                         /* identical A,C,G,T columns (entropy input)          */
    int32_t strand;                         /* 0 '+', 1 '-'                                        */
    int32_t q_contig;
} miblast_hsp;

typedef struct miblast_aln {            /* one output alignment = one PAF line                    */
    int32_t strand, q_contig, t_contig;
    int32_t t_lo, t_hi, q_lo, q_hi;         /* concatenated, strand coordinates, half-open         */
    int32_t score;
    int32_t dmin, dmax;                     /* diagonal band of its aligned columns                */
    int32_t anchor_t, anchor_q;
    int64_t ops_off, n_ops;                 /* into miblast_result_ops()                           */
} miblast_aln;

typedef struct miblast_stats {          /* counters defined by SURVEY.md section 8(d); seconds are GPU-side wall time */
    int64_t seed_lookups, seed_hits, hits_extended, ungapped_cols;
    int64_t hsps_pre_entropy, hsps, anchors, anchors_skipped;
    int64_t dp_sides, dp_cells, dp_rows, alignments;
    double  t_index, t_seed, t_gapped, t_total;
    /* implementation-side extras (not part of the oracle's counter set) */
    int64_t dp_sides_run, dp_cells_run;     /* incl. speculative / re-run work                    */
    int64_t gapped_rounds, seed_batches;
    double  t_dp_kernel_ms;  int64_t dp_kernel_launches;     /* HIP-event time of k_ydrop launches  */
    double  t_ungapped_kernel_ms; int64_t ungapped_kernel_launches;
    double  t_sort_ms, t_seedfill_ms;
    int64_t dp_rows_run;                    /* DP rows evaluated incl. speculative work                */
    int64_t relay_accepted, relay_rejected; /* hand-overs between concurrently evaluated pieces of long DPs */
    double  t_traceback_ms, t_merge_ms;     /* host wall time of the traceback (kernels + copies) and of the trace merge  */
    int64_t dp_reruns;                      /* pieces rerun with a wider DP kernel (window outgrew the one-wave kernel)    */
    double  t_dp_busy_ms;                   /* time during which at least one DP launch of the call was running: the union of the launches'
                                             * HIP-event intervals (the groups of a call's pairs launch on streams of their own and overlap;
                                             * t_dp_kernel_ms is the SUM of the launch durations)                            */
    int64_t relay_inline_checks;            /* hand-overs a piece checked itself inside its DP launch ...              */
    int64_t relay_inline_continued;         /* ... and pieces that went on past their first stop row there instead of in a launch of their own */
    int64_t seed_binned;                    /* strands whose seed hits were grouped by diagonal through bins + LDS (mb_seed_bin.h) instead of the device-wide radix sort */
} miblast_stats;

typedef struct miblast_result miblast_result;

/* The blast job itself: replaces one `lastz target query --format=paf:wfmash <opts>` process
 * (local_alignment.py:65-73).  Output order and bytes follow SURVEY.md Appendix B / A.8.       */
int miblast_align(miblast_ctx *ctx, const miblast_seqset *target, const miblast_seqset *query,
                  const miblast_params *p, miblast_result **out);
/* Several chunk pairs of one GPU in one call (the reference runs one lastz process per pair and lets Toil put many of
 * them on a node, local_alignment.py:395-405).  Seed stages run back to back; the gapped stages are merged, so the
 * speculative DPs of all pairs share kernel launches and fill the GPU together.  results[i] is byte-identical to what
 * miblast_align(targets[i], queries[i]) returns.                                                  */
int miblast_align_pairs(miblast_ctx *ctx, const miblast_seqset *const *targets, const miblast_seqset *const *queries,
                        size_t n_pairs, const miblast_params *p, miblast_result **results);
void miblast_result_free(miblast_result *r);
/* PAF text that the reference reads from the process's stdout (common.py:875-878).            */
const char *miblast_result_paf(const miblast_result *r, size_t *len);
const miblast_stats *miblast_result_stats(const miblast_result *r);
const miblast_hsp *miblast_result_hsps(const miblast_result *r, int64_t *n);
const miblast_aln *miblast_result_alns(const miblast_result *r, int64_t *n);
const uint32_t *miblast_result_ops(const miblast_result *r, int64_t *n);   /* (len<<2)|op, op 0 '=',1 'X',2 'I',3 'D' */

/* File-level convenience: load both FASTA files, align on the context's device, write PAF to fd.
 * Replaces the whole cactus_call(lastz ...) invocation (local_alignment.py:72).  Inputs of any size (blocks
 * of whole sequences, see miblast_multi below).                                                  */
int miblast_align_files(miblast_ctx *ctx, const char *target_fa, const char *query_fa,
                        const miblast_params *p, int out_fd, miblast_stats *stats);

/* ---- several GPUs from one process, inputs of any size ----------------------------------------
 * The reference's GPU branch starts ONE process per genome pair and gives it every GPU of the job:
 *   run_kegalign A.fa B.fa --format=paf:wfmash <opts> --num_gpu G --num_threads C   (local_alignment.py:54-58),
 *   accelerators 'cuda:G' per job (local_alignment.py:393,405), inputs up to bigChunkSize = 6 000 000 000 bases
 *   (cactus_progressive_config.xml:47,91; local_alignment.py:376).
 * A miblast_multi owns one context per device 0..num_gpu-1.  Its calls cut both inputs into blocks of whole
 * sequences (at most 2^30 bases each), deal the block pairs to the devices longest-first (no data-path collective;
 * chunk pairs are independent) and assemble the PAF in the order ONE lastz process over the whole files
 * writes it: the bytes do not depend on num_gpu or on the block size.  $MIBLAST_DEVICE_MAP
 * ("0,0,1": logical -> physical ordinals) lets several logical devices share a GPU (tests on a one-GPU box).     */
typedef struct miblast_multi miblast_multi;
int miblast_multi_create(int num_gpu, miblast_multi **out);          /* num_gpu <= miblast_device_count() */
void miblast_multi_destroy(miblast_multi *m);
int miblast_multi_num_gpu(const miblast_multi *m);
/* = one `run_kegalign target query <opts> --num_gpu G` process (PAF to out_fd); also what bin/lastz runs with G = 1 */
int miblast_multi_align_files(miblast_multi *m, const char *target_fa, const char *query_fa, const miblast_params *p,
                              int out_fd, miblast_stats *stats);
/* A list of chunk pairs (FASTA text in memory) sharded over the devices: SURVEY 8b's multi-GPU entry.  *paf receives
 * the pairs' PAF in pair order (free with miblast_free); stats (may be NULL) = totals over the call.            */
typedef struct miblast_fasta_pair { const char *target; size_t target_len; const char *query; size_t query_len; } miblast_fasta_pair;
int miblast_multi_align_fasta_pairs(miblast_multi *m, const miblast_fasta_pair *pairs, size_t n_pairs, const miblast_params *p,
                                    char **paf, size_t *paf_len, miblast_stats *stats);

/* Stage export for parity tests: the target seed position table (lastz pos_table)
 * as CSR.  offsets has 2^24+1 entries; positions are ascending inside a bucket.  Caller frees
 * both with miblast_free().                                                                    */
int miblast_build_index(miblast_ctx *ctx, const miblast_seqset *target, int32_t step,
                        uint32_t **offsets, uint32_t **positions);
void miblast_free(void *p);

const char *miblast_last_error(void);
/* Diagnostics: hipMalloc + hipFree calls the library has made in this process so far.  A device allocation in the middle of a job stalls every
 * concurrent job on the device (hipFree waits for an idle device); bench.py reads this around its timed steps and reports the difference as
 * device_allocs_in_timed_steps, which is 0 once the workspaces have met the workload.                                                  */
long long miblast_debug_device_allocs(void);
const char *miblast_version(void);

#ifdef __cplusplus
}
#endif
#endif

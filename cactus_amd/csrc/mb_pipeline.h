// mb_pipeline.h -- host-side objects behind the opaque handles of include/miblast.h
#pragma once

#include <atomic>
#include "mb_common.h"

#include <functional>

namespace mb {

struct Workspace;
Workspace *workspace_create();
void workspace_destroy(Workspace *w);

// HIP-event intervals of the DP launches of one align_pairs call (the groups of its pairs launch on streams of their own): start and
// end of every launch in ms after `base`, collected from all streams; their union is miblast_stats.t_dp_busy_ms
struct DpSpans;

struct Ctx {
    int device = 0;
    int priority = 0;                   // HIP priority of the context's streams (ctx_set_priority)
    Workspace *ws = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr, ev4 = nullptr;
    void *chain_cache = nullptr;        // device buffers the chaining stage keeps between calls (mp_chain.cpp)
    DpSpans *spans = nullptr;           // where this context's DP launches report their intervals during a call (mb_pipeline.cpp)
    std::atomic<unsigned> *arena_scale = nullptr;   // what the trace arenas' estimate is multiplied by (x / 256): the context's own, shared with its lanes (mb_pipeline.cpp)
    std::atomic<size_t> *arena_class = nullptr;     // ... and the largest arena a stage of the context has asked for
    std::atomic<unsigned long long> *hits_hint = nullptr;   // a lane: the largest strand (seed hits) any lane of its context has met -- its buffers are sized for that before a call's kernels are queued
};
void chain_cache_destroy(void *cache);  // mp_chain.cpp

// --queryhspbest (and --queryhsplimit in front of a gapped stage) over a target that needs several blocks (mb_multi.cpp): the last HSP the whole
// target keeps for a query sequence and strand -- its score and its place in the order a sequential search finds HSPs (query position, word
// variant, whole-target position descending)
struct HspBestCut {
    int32_t active = 0, score = 0, q_end = 0, rank = 0; int64_t neg_t = 0;
    // --queryhsplimit in front of it (or of a gapped stage): the last HSP of the whole target's first N in found order
    int32_t lim_active = 0, lim_q_end = 0, lim_rank = 0; int64_t lim_neg_t = 0;
};

struct Result {
    // set by mb_multi.cpp before the call, read by the seed stage's host half instead of ranking the block's own HSPs: cut[2 * contig + strand]
    // for the contigs of the query block; t_origin: the target block's first base in the whole target
    const std::vector<HspBestCut> *best_cut = nullptr;
    int64_t best_cut_t_origin = 0;
    std::string paf;
    std::vector<miblast_hsp> hsps;
    std::vector<miblast_aln> alns;
    std::vector<uint32_t> ops;
    miblast_stats stats{};
    std::vector<int32_t> aln_anchor_score;   // per alignment: score of the HSP its anchor came from
    std::vector<size_t> line_off;            // PAF line k of alns[k] = paf[line_off[k], line_off[k+1])
};

int ctx_set_priority(Ctx &ctx, int level);
void ctx_pair_streams(Ctx &ctx);     // the context's stream and its lanes' on hardware queues of their own
void upload_seqset(SeqSet &s, int device);
void release_seqset(SeqSet &s);
void drop_derived();                 // frees what the library keeps with resident sets: seed tables, '-' strands, packed strands (made again on demand)
// outgroup trimming on the device (mb_pipeline.cpp): what no alignment of `paf` covers of the resident query set, as a new resident set
// (n items in one call; outs[k] is left empty and nothing_left[k] set when every base of Qs[k] is covered)
int seqset_unaligned(Ctx &ctx, size_t n, const SeqSet *const *Qs, const char *const *pafs, const size_t *paf_lens, int64_t min_size, int64_t flank,
                     SeqSet *const *outs, bool *nothing_left);
int align(Ctx &ctx, const SeqSet &T, const SeqSet &Q, const miblast_params &p, Result &res);
int align_pairs(Ctx &ctx, const SeqSet *const *Ts, const SeqSet *const *Qs, size_t n, const miblast_params &p, Result **results);
// mb_multi.cpp: n_pairs (target, query) sets parsed on the host (not uploaded), cut into blocks of whole contigs, the block pairs
// dealt to the contexts' devices; `paf` = the pairs' outputs in pair order, each as one lastz process over the whole files writes it
int align_blocked(const std::vector<Ctx *> &ctxs, const SeqSet *const *Ts, const SeqSet *const *Qs, size_t n_pairs,
                  const miblast_params &p, std::string &paf, miblast_stats *stats);
// independent work items on the library's persistent worker threads (the caller takes part; nested calls run inline)
void host_parallel_for(size_t n, const std::function<void(size_t)> &f);
struct HostHot {                        // keeps the workers spinning for the duration of a job (they sleep otherwise)
    HostHot();
    ~HostHot();
    HostHot(const HostHot &) = delete;
    HostHot &operator=(const HostHot &) = delete;
private:
    alignas(8) unsigned char impl[8];
};
int set_host_threads(int n);      // 0 = automatic; returns the threads in use or -1 (a job is running)
int host_threads();
int export_index(Ctx &ctx, const SeqSet &T, int step, uint32_t **offsets, uint32_t **positions);
// rank of the word variant behind a seed hit (0: the words are equal; 1 + k: a transition at care position k; 99: neither) from the 19 code
// bytes before the seed end in the target and in the searched strand of the query: the middle key of the order a sequential search finds hits in
int seed_variant_rank(const uint8_t *t19, const uint8_t *q19);

}  // namespace mb

// the opaque handles of include/miblast.h (shared by mb_capi.cpp and mp_chain.cpp)
struct miblast_ctx { mb::Ctx c; };
struct miblast_seqset { mb::SeqSet s; };
struct miblast_result { mb::Result r; };
struct miblast_multi { std::vector<miblast_ctx *> ctxs; };

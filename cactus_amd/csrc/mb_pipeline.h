// mb_pipeline.h -- host-side objects behind the opaque handles of include/miblast.h
#pragma once

#include "mb_common.h"

namespace mb {

struct Workspace;
Workspace *workspace_create();
void workspace_destroy(Workspace *w);

struct Ctx {
    int device = 0;
    Workspace *ws = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr, ev4 = nullptr;
};

struct Result {
    std::string paf;
    std::vector<miblast_hsp> hsps;
    std::vector<miblast_aln> alns;
    std::vector<uint32_t> ops;
    miblast_stats stats{};
};

void upload_seqset(SeqSet &s, int device);
void release_seqset(SeqSet &s);
int align(Ctx &ctx, const SeqSet &T, const SeqSet &Q, const miblast_params &p, Result &res);
int align_pairs(Ctx &ctx, const SeqSet *const *Ts, const SeqSet *const *Qs, size_t n, const miblast_params &p, Result **results);
int set_host_threads(int n);      // 0 = automatic; returns the threads in use or -1 (a job is running)
int host_threads();
int export_index(Ctx &ctx, const SeqSet &T, int step, uint32_t **offsets, uint32_t **positions);

}  // namespace mb

// the opaque handles of include/miblast.h (shared by mb_capi.cpp and mp_chain.cpp)
struct miblast_ctx { mb::Ctx c; };
struct miblast_seqset { mb::SeqSet s; };
struct miblast_result { mb::Result r; };

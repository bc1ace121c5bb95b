// mb_multi.cpp -- one blast job over several MI355X and over inputs of any size, from ONE process.
//
// The reference's GPU branch hands one `run_kegalign A.fa B.fa ... --num_gpu G` process two whole genomes
// (bigChunkSize = 6 000 000 000, /root/reference/src/cactus/cactus_progressive_config.xml:47,91;
// /root/reference/src/cactus/paf/local_alignment.py:54-58,376) and all G GPUs of the job
// (local_alignment.py:393,405).  This file is that process's inside:
//
//   * both files are cut into BLOCKS of whole contigs (at most kBlockCap bases, so that block coordinates stay
//     int32), the job becomes the grid target blocks x query blocks, the block pairs are dealt to the devices
//     longest-first, and every device (one host thread, one context, its own stream and workspace) runs its share
//     through align_pairs() in batches;
//   * the result is assembled in the order ONE lastz process over the whole files would have written it, so the
//     bytes do not depend on the number of devices, on the block size or on the dealing:
//       - query blocks are exact by construction: lastz handles one query sequence at a time;
//       - target blocks: no seed hit, x-drop or y-drop extension crosses a contig separator and the
//         diagonal-suppression state of a diagonal never crosses one either (a later hit on the same diagonal in
//         another contig lies beyond the separator that stopped the earlier extension), so HSPs and alignments of a
//         target contig do not depend on which other contigs share its block; the --step phase is kept by indexing
//         positions with (block origin + p) % step == 0; what does depend on the whole target is the ORDER in which
//         a query sequence's alignments are written (anchor order: HSP score descending, then t, then q -- A.6/A.8),
//         so the blocks' lists are merged on that key with t in whole-file coordinates;
//       - --queryhspbest=N ranks a query sequence's HSPs over the whole target: when the target needs more than one block the seed
//         stages of all its blocks run once WITHOUT the limit, the HSPs of a query sequence and strand are ranked over the blocks as one
//         search over the whole target ranks them (score; of equal scores the earlier found), and the jobs proper are given the last
//         HSP kept (HspBestCut) instead of ranking their own block's (round 6; refused until then).  --queryhsplimit=N in front of a
//         gapped stage (or of --queryhspbest) the same way: the whole target's first N in found order, the last of them the cut.
//
// No data-path collective: block pairs are independent; the only exchange is this in-process gather of
// the PAF lines.  The one-process-per-GPU form of the same sharding (torch.distributed, RCCL gather) is bench.py /
// cactus_amd/multigpu.py.
#include "mb_pipeline.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

namespace mb {

namespace {

constexpr int64_t kBlockCap = (1ll << 30) - 64;          // |T block| + |Q block| + 4 < 2^31 (seed_phase limit)

long env_ll(const char *name, long long dflt) {
    const char *v = getenv(name);
    return v && *v ? atoll(v) : dflt;
}

struct Block { int c0 = 0, c1 = 0; int64_t origin = 0, total = 0; };

// whole contigs packed into blocks of at most `limit` bases (separators included); a contig longer than `limit` but
// within `cap` gets a block of its own
int partition(const SeqSet &S, int64_t cap, int64_t limit, std::vector<Block> &out) {
    out.clear();
    const int n = (int)S.names.size();
    if (n == 0) { out.push_back(Block{0, 0, 0, 0}); return MIBLAST_OK; }
    limit = std::max<int64_t>(1, std::min(limit, cap));
    int c0 = 0;
    while (c0 < n) {
        if (S.lens[(size_t)c0] > cap) {
            set_error("sequence " + S.names[(size_t)c0] + " is longer than a block (" + std::to_string(cap) + " bases): split it first (faffy chunk does, local_alignment.py:380-387)");
            return MIBLAST_ELIMIT;
        }
        int c1 = c0 + 1;
        while (c1 < n && S.starts[(size_t)c1] + S.lens[(size_t)c1] - S.starts[(size_t)c0] <= limit) c1++;
        Block b;
        b.c0 = c0; b.c1 = c1; b.origin = S.starts[(size_t)c0];
        b.total = S.starts[(size_t)c1 - 1] + S.lens[(size_t)c1 - 1] - b.origin;
        out.push_back(b);
        c0 = c1;
    }
    return MIBLAST_OK;
}

void make_view(const SeqSet &parent, const Block &b, SeqSet &v) {
    v.names.assign(parent.names.begin() + b.c0, parent.names.begin() + b.c1);
    v.starts.clear(); v.lens.clear();
    for (int c = b.c0; c < b.c1; c++) { v.starts.push_back(parent.starts[(size_t)c] - b.origin); v.lens.push_back(parent.lens[(size_t)c]); }
    v.total = b.total;
    v.view = parent.host() + b.origin;            // separators between the block's contigs are in place; the pads are added on upload
    v.origin = parent.origin + b.origin;
}

struct Job {
    int pair = 0, tb = 0, qb = 0, device = 0;
    double cost = 0;
    Result res;
};

// launch-level figures are shared by the pairs of one align_pairs() batch (gapped_phase copies them to every pair)
void add_stats(miblast_stats &a, const miblast_stats &s, bool first_of_batch) {
    a.seed_lookups += s.seed_lookups; a.seed_hits += s.seed_hits; a.hits_extended += s.hits_extended; a.ungapped_cols += s.ungapped_cols;
    a.hsps_pre_entropy += s.hsps_pre_entropy; a.hsps += s.hsps; a.anchors += s.anchors; a.anchors_skipped += s.anchors_skipped;
    a.dp_sides += s.dp_sides; a.dp_cells += s.dp_cells; a.dp_rows += s.dp_rows; a.alignments += s.alignments;
    a.t_index += s.t_index; a.t_seed += s.t_seed;
    a.seed_batches += s.seed_batches; a.seed_binned += s.seed_binned;
    a.t_ungapped_kernel_ms += s.t_ungapped_kernel_ms; a.ungapped_kernel_launches += s.ungapped_kernel_launches;
    a.t_sort_ms += s.t_sort_ms; a.t_seedfill_ms += s.t_seedfill_ms;
    if (first_of_batch) {
        a.t_gapped += s.t_gapped; a.gapped_rounds += s.gapped_rounds; a.dp_sides_run += s.dp_sides_run; a.dp_cells_run += s.dp_cells_run;
        a.dp_rows_run += s.dp_rows_run; a.t_dp_kernel_ms += s.t_dp_kernel_ms; a.t_dp_busy_ms += s.t_dp_busy_ms; a.dp_kernel_launches += s.dp_kernel_launches;
        a.relay_accepted += s.relay_accepted; a.relay_rejected += s.relay_rejected; a.relay_inline_checks += s.relay_inline_checks; a.relay_inline_continued += s.relay_inline_continued; a.dp_reruns += s.dp_reruns;
        a.t_traceback_ms += s.t_traceback_ms; a.t_merge_ms += s.t_merge_ms;
    }
}

}  // namespace

int align_blocked(const std::vector<Ctx *> &ctxs, const SeqSet *const *Ts, const SeqSet *const *Qs, size_t n_pairs,
                  const miblast_params &p, std::string &paf, miblast_stats *stats) {
    const double t_begin = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const int n_dev = (int)ctxs.size();
    if (n_dev <= 0 || n_pairs == 0) { set_error("align_blocked: nothing to do"); return MIBLAST_EINVAL; }
    const int64_t cap = std::min<int64_t>(kBlockCap, std::max<int64_t>(64, env_ll("MIBLAST_BLOCK_BASES", kBlockCap)));
    // one header / end marker per job, suppression state shared by diagonals 65536 apart (diag=hash16), earlier alignments as walls
    // for later ones: such jobs are not assembled from blocks
    // (round 6: --format=general and --markend ARE assembled from blocks now -- the HSP lists of a query block's target blocks merged in the
    //  order one search over the whole target finds them, below)
    const bool one_block_only = p.diag_hash16 || p.walls;
    const bool general = p.format != 0;
    const int step = std::max(1, p.step);

    // ---- blocks and the job grid ---------------------------------------------------------------------------------
    std::vector<std::vector<Block>> tblk(n_pairs), qblk(n_pairs);
    std::vector<std::unique_ptr<Job>> jobs;
    bool any_cut = false;                                  // --queryhspbest over a target in several blocks: two passes (below)
    for (size_t k = 0; k < n_pairs; k++) {
        const SeqSet &T = *Ts[k], &Q = *Qs[k];
        // Enough block pairs to occupy every device when the contigs allow it: the query side is split first (exact by
        // construction and free of any merge), the target side only for size or when the query is a single sequence.
        int64_t q_want = 1, t_want = 1;
        if (n_dev > 1 && n_pairs < (size_t)(2 * n_dev) && !one_block_only) {
            const int64_t want = (2 * n_dev + (int64_t)n_pairs - 1) / (int64_t)n_pairs;
            q_want = std::min<int64_t>(want, (int64_t)Q.names.size());
            if (q_want < want && p.queryhspbest <= 0 && p.queryhsplimit <= 0) t_want = std::min<int64_t>((want + std::max<int64_t>(1, q_want) - 1) / std::max<int64_t>(1, q_want), (int64_t)T.names.size());
        }
        const int64_t q_limit = std::max<int64_t>(1, (Q.total + std::max<int64_t>(1, q_want) - 1) / std::max<int64_t>(1, q_want));
        const int64_t t_limit = std::max<int64_t>(1, (T.total + std::max<int64_t>(1, t_want) - 1) / std::max<int64_t>(1, t_want));
        int rc = partition(T, cap, one_block_only ? cap : t_limit, tblk[k]);
        if (rc == MIBLAST_OK) rc = partition(Q, cap, one_block_only ? cap : q_limit, qblk[k]);
        if (rc != MIBLAST_OK) return rc;
        if (one_block_only && (tblk[k].size() > 1 || qblk[k].size() > 1)) {
            set_error("--miblast-diag=hash16 / --miblast-walls jobs are not assembled from blocks: input longer than 2^30 bases");
            return MIBLAST_ELIMIT;
        }
        if (tblk[k].size() > 1 && (p.queryhspbest > 0 || (p.queryhsplimit > 0 && !general))) any_cut = true;
        for (size_t qb = 0; qb < qblk[k].size(); qb++)
            for (size_t tb = 0; tb < tblk[k].size(); tb++) {
                std::unique_ptr<Job> j(new Job());
                j->pair = (int)k; j->tb = (int)tb; j->qb = (int)qb;
                const double tn = (double)tblk[k][tb].total, qn = (double)qblk[k][qb].total;
                j->cost = tn / step * qn + 4096.0 * (tn + qn) + 1.0;         // chance seed hits grow with the product, everything else with the sum
                jobs.push_back(std::move(j));
            }
    }
    // ---- longest first onto the least loaded device ----------------------------------------------------------------
    {
        std::vector<size_t> order(jobs.size());
        for (size_t x = 0; x < order.size(); x++) order[x] = x;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return jobs[a]->cost > jobs[b]->cost; });
        std::vector<double> load((size_t)n_dev, 0.0);
        for (size_t x : order) {
            int best = 0;
            for (int d = 1; d < n_dev; d++) if (load[(size_t)d] < load[(size_t)best]) best = d;
            jobs[x]->device = best; load[(size_t)best] += jobs[x]->cost;
        }
    }
    // ---- one host thread per device ----------------------------------------------------------------------------------
    const int64_t batch_bases = std::max<int64_t>(1, env_ll("MIBLAST_BATCH_BASES", 512ll << 20));
    const size_t batch_pairs = (size_t)std::max<long long>(1, env_ll("MIBLAST_BATCH_PAIRS", 64));
    std::vector<int> dev_rc((size_t)n_dev, MIBLAST_OK);
    std::vector<std::string> dev_err((size_t)n_dev);
    std::vector<std::vector<std::pair<size_t, bool>>> dev_done((size_t)n_dev);      // (job, first of its batch)
    std::vector<char> active(jobs.size(), 1);              // the jobs of the current pass
    miblast_params p_run = p;                              // ... and its options
    auto device_main = [&](int d) {
        try {
            Ctx &ctx = *ctxs[(size_t)d];
            MB_HIP(hipSetDevice(ctx.device));
            std::vector<size_t> mine;
            for (size_t x = 0; x < jobs.size(); x++) if (jobs[x]->device == d && active[x]) mine.push_back(x);     // (pair, query block, target block) order
            size_t at = 0;
            while (at < mine.size()) {
                // a batch: block pairs that share one align_pairs() call (merged gapped launches), every block uploaded once
                std::map<std::pair<int, int>, std::unique_ptr<SeqSet>> tsets, qsets;
                std::vector<const SeqSet *> bt, bq;
                std::vector<Result *> br;
                int64_t bases = 0;
                const size_t first = at;
                struct Release { std::map<std::pair<int, int>, std::unique_ptr<SeqSet>> &a, &b; ~Release() { for (auto &e : a) release_seqset(*e.second); for (auto &e : b) release_seqset(*e.second); } } rel{tsets, qsets};
                while (at < mine.size() && br.size() < batch_pairs) {
                    Job &j = *jobs[mine[at]];
                    const Block &tb = tblk[(size_t)j.pair][(size_t)j.tb], &qb = qblk[(size_t)j.pair][(size_t)j.qb];
                    const bool have_t = tsets.count({j.pair, j.tb}) != 0, have_q = qsets.count({j.pair, j.qb}) != 0;
                    const int64_t extra = (have_t ? 0 : tb.total) + (have_q ? 0 : qb.total);
                    if (!br.empty() && bases + extra > batch_bases) break;
                    if (!have_t) {
                        std::unique_ptr<SeqSet> v(new SeqSet());
                        make_view(*Ts[(size_t)j.pair], tb, *v);
                        upload_seqset(*v, ctx.device);
                        tsets[{j.pair, j.tb}] = std::move(v);
                    }
                    if (!have_q) {
                        std::unique_ptr<SeqSet> v(new SeqSet());
                        make_view(*Qs[(size_t)j.pair], qb, *v);
                        upload_seqset(*v, ctx.device);
                        qsets[{j.pair, j.qb}] = std::move(v);
                    }
                    bases += extra;
                    bt.push_back(tsets[{j.pair, j.tb}].get()); bq.push_back(qsets[{j.pair, j.qb}].get()); br.push_back(&j.res);
                    at++;
                }
                const int rc = align_pairs(ctx, bt.data(), bq.data(), br.size(), p_run, br.data());
                if (rc != MIBLAST_OK) { dev_rc[(size_t)d] = rc; dev_err[(size_t)d] = last_error_text(); return; }
                for (size_t x = first; x < at; x++) dev_done[(size_t)d].push_back({mine[x], x == first});
            }
        } catch (const HipFailure &e) {
            dev_rc[(size_t)d] = (e.code == hipErrorNoDevice || e.code == hipErrorInvalidDevice) ? MIBLAST_ENODEV : MIBLAST_EHIP;
            dev_err[(size_t)d] = std::string("HIP call did not succeed: ") + e.what + " -> " + hipGetErrorString(e.code) + " (" + e.file + ":" + std::to_string(e.line) + ")";
        } catch (const std::bad_alloc &) {
            dev_rc[(size_t)d] = MIBLAST_ELIMIT; dev_err[(size_t)d] = "out of host memory";
        } catch (const std::exception &e) {
            dev_rc[(size_t)d] = MIBLAST_EHIP; dev_err[(size_t)d] = std::string("internal: ") + e.what();
        }
    };
    auto run_pass = [&]() -> int {
        for (auto &v : dev_done) v.clear();
        if (n_dev == 1) device_main(0);
        else {
            std::vector<std::thread> th;
            for (int d = 0; d < n_dev; d++) th.emplace_back(device_main, d);
            for (std::thread &t : th) t.join();
        }
        for (int d = 0; d < n_dev; d++)
            if (dev_rc[(size_t)d] != MIBLAST_OK) { set_error(dev_err[(size_t)d]); return dev_rc[(size_t)d]; }
        return MIBLAST_OK;
    };
    // the word variant that made an HSP's seed hit (0 exact, 1 + k a transition at care position k): the 19 bases before the seed end in the
    // target and in the searched strand of the query -- the '-' strand read off the '+' strand's codes, contig-wise mirrored
    auto variant_of = [&](size_t k, const Block &TB, const Block &QB, const miblast_hsp &h) -> int {
        const SeqSet &T = *Ts[k], &Q = *Qs[k];
        const uint8_t *th = T.host(), *qh = Q.host();
        uint8_t tw[kSeedSpan], qw[kSeedSpan];
        const int64_t t_end = TB.origin + h.seed_t_end;
        for (int c = 0; c < kSeedSpan; c++) tw[c] = th[t_end - kSeedSpan + c];
        const int qc_g = QB.c0 + h.q_contig;
        const int64_t cst = Q.starts[(size_t)qc_g], cln = Q.lens[(size_t)qc_g];
        const int64_t q_in = (int64_t)h.seed_q_end - (cst - QB.origin);          // seed end inside the contig, on the searched strand
        for (int c = 0; c < kSeedSpan; c++) {
            const int64_t ps = q_in - kSeedSpan + c;
            if (!h.strand) qw[c] = qh[cst + ps];
            else { const uint8_t b = qh[cst + (cln - 1 - ps)]; qw[c] = (uint8_t)((b & 4u) ? b : (b & ~3u) | (3u - (b & 3u))); }
        }
        return seed_variant_rank(tw, qw);
    };
    // ---- --queryhspbest over a target in several blocks: what the whole target keeps ------------------------------------
    // The seed stages of those jobs once without the limit (and without a gapped stage); per query sequence and strand the HSPs of all target
    // blocks ranked as seed_host ranks one block's (mb_pipeline.cpp: score descending, of equal scores the earlier found -- query position,
    // word variant, target position descending --, with hspbest_ties the later found); the N-th is the cut the jobs proper are given.
    std::vector<std::vector<HspBestCut>> cuts;             // per (pair, query block) in job order: 2 x contigs of the block
    std::vector<size_t> cut_of(jobs.size(), (size_t)-1);
    if (any_cut) {
        p_run.queryhspbest = 0; p_run.gapped = 0;          // (--queryhsplimit stays: a block's own first N hold the whole target's that lie in it)
        {
            size_t x = 0;
            for (size_t k = 0; k < n_pairs; k++) {
                const size_t cnt = tblk[k].size() * qblk[k].size();
                for (size_t e = 0; e < cnt; e++) active[x + e] = tblk[k].size() > 1;
                x += cnt;
            }
        }
        int rc = run_pass();
        if (rc != MIBLAST_OK) return rc;
        size_t x = 0;
        for (size_t k = 0; k < n_pairs; k++) {
            const size_t n_tb = tblk[k].size(), n_qb = qblk[k].size();
            if (n_tb > 1)
                for (size_t qb = 0; qb < n_qb; qb++) {
                    const Block &QB = qblk[k][qb];
                    struct Ent { int32_t qc, strand, score, q_end, rank; int64_t neg_t; };
                    std::vector<Ent> ents;
                    for (size_t tb = 0; tb < n_tb; tb++) {
                        const Block &TB = tblk[k][tb];
                        for (const miblast_hsp &h : jobs[x + qb * n_tb + tb]->res.hsps)
                            ents.push_back(Ent{h.q_contig, h.strand, h.score, h.seed_q_end, variant_of(k, TB, QB, h), -(TB.origin + (int64_t)h.seed_t_end)});
                    }
                    const bool later = p.hspbest_ties != 0;
                    auto found_before = [](const Ent &a, const Ent &b) {
                        if (a.q_end != b.q_end) return a.q_end < b.q_end;
                        if (a.rank != b.rank) return a.rank < b.rank;
                        return a.neg_t < b.neg_t;
                    };
                    auto ranked_before = [later, found_before](const Ent &a, const Ent &b) {
                        if (a.score != b.score) return a.score > b.score;
                        return later ? found_before(b, a) : found_before(a, b);
                    };
                    std::sort(ents.begin(), ents.end(), [ranked_before](const Ent &a, const Ent &b) {
                        if (a.qc != b.qc) return a.qc < b.qc;
                        if (a.strand != b.strand) return a.strand < b.strand;
                        return ranked_before(a, b);
                    });
                    std::vector<HspBestCut> cut(2 * (size_t)(QB.c1 - QB.c0));
                    for (size_t i = 0; i < ents.size();) {
                        size_t j = i;
                        while (j < ents.size() && ents[j].qc == ents[i].qc && ents[j].strand == ents[i].strand) j++;
                        HspBestCut &c = cut[2 * (size_t)ents[i].qc + (size_t)ents[i].strand];
                        size_t n_in = j - i;
                        if (p.queryhsplimit > 0 && (int64_t)n_in > (int64_t)p.queryhsplimit) {
                            // the whole target's first N in found order (every block kept its own first N: they are among those), the rest leaves the ranking
                            std::sort(ents.begin() + (long)i, ents.begin() + (long)j, found_before);
                            n_in = (size_t)p.queryhsplimit;
                            const Ent &e = ents[i + n_in - 1];
                            c.lim_active = 1; c.lim_q_end = e.q_end; c.lim_rank = e.rank; c.lim_neg_t = e.neg_t;
                            std::sort(ents.begin() + (long)i, ents.begin() + (long)(i + n_in), ranked_before);
                        }
                        if (p.queryhspbest > 0 && (int64_t)n_in > (int64_t)p.queryhspbest) {
                            const Ent &e = ents[i + (size_t)p.queryhspbest - 1];
                            c.active = 1; c.score = e.score; c.q_end = e.q_end; c.rank = e.rank; c.neg_t = e.neg_t;
                        }
                        i = j;
                    }
                    for (size_t tb = 0; tb < n_tb; tb++) cut_of[x + qb * n_tb + tb] = cuts.size();
                    cuts.push_back(std::move(cut));
                }
            x += n_tb * n_qb;
        }
        x = 0;
        for (size_t k = 0; k < n_pairs; k++)
            for (size_t qb = 0; qb < qblk[k].size(); qb++)
                for (size_t tb = 0; tb < tblk[k].size(); tb++, x++) {
                    jobs[x]->res = Result();
                    if (cut_of[x] != (size_t)-1) { jobs[x]->res.best_cut = &cuts[cut_of[x]]; jobs[x]->res.best_cut_t_origin = tblk[k][tb].origin; }
                }
        p_run = p;
        std::fill(active.begin(), active.end(), 1);
    }
    {
        const int rc = run_pass();
        if (rc != MIBLAST_OK) return rc;
    }

    // ---- assembly in the order of one job over the whole files ------------------------------------------------------
    paf.clear();
    size_t x0 = 0;
    if (general) {
        // --format=general:name1,zstart1,end1,name2,zstart2+,end2+ [--markend] (the repeat masker's call, cactus_lastzRepeatMask.py:97-105): ONE
        // header line, then per query sequence (file order) the '+' HSPs and the '-' HSPs in the order one search over the WHOLE target finds
        // them -- query position ascending, word variant, target position descending (seed_host, mb_pipeline.cpp) --, at most --queryhsplimit
        // per query sequence and strand (every block kept its own first N: the first N of the whole target are among them), one end marker.
        auto put_num = [&](int64_t v) { char buf[24]; int n = snprintf(buf, sizeof buf, "%lld", (long long)v); paf.append(buf, (size_t)n); };
        for (size_t k = 0; k < n_pairs; k++) {
            const SeqSet &T = *Ts[k], &Q = *Qs[k];
            const size_t n_tb = tblk[k].size(), n_qb = qblk[k].size();
            if (n_tb == 1 && n_qb == 1) { paf += jobs[x0]->res.paf; x0 += 1; continue; }      // (one block pair: the job's own text)
            paf += "#name1\tzstart1\tend1\tname2\tzstart2+\tend2+\n";
            for (size_t qb = 0; qb < n_qb; qb++) {
                const Block &QB = qblk[k][qb];
                struct Ent { int32_t qc, strand, q_end, rank; int64_t neg_t; const miblast_hsp *h; int64_t t_origin; };
                std::vector<Ent> ents;
                for (size_t tb = 0; tb < n_tb; tb++) {
                    const Result &r = jobs[x0 + qb * n_tb + tb]->res;
                    const Block &TB = tblk[k][tb];
                    for (const miblast_hsp &h : r.hsps) {
                        const int64_t t_end = TB.origin + h.seed_t_end;
                        const int qc_g = QB.c0 + h.q_contig;
                        ents.push_back(Ent{qc_g, h.strand, h.seed_q_end, variant_of(k, TB, QB, h), -t_end, &h, TB.origin});
                    }
                }
                std::stable_sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) {
                    if (a.qc != b.qc) return a.qc < b.qc;
                    if (a.strand != b.strand) return a.strand < b.strand;
                    if (a.q_end != b.q_end) return a.q_end < b.q_end;
                    if (a.rank != b.rank) return a.rank < b.rank;
                    return a.neg_t < b.neg_t;
                });
                int cur_qc = -1, cur_strand = -1; int64_t taken = 0;
                for (const Ent &e : ents) {
                    if (e.qc != cur_qc || e.strand != cur_strand) { cur_qc = e.qc; cur_strand = e.strand; taken = 0; }
                    if (p.queryhsplimit > 0 && taken >= p.queryhsplimit) continue;
                    taken++;
                    const miblast_hsp &h = *e.h;
                    const int64_t t_start = e.t_origin + h.t_start;
                    const int tcg = T.contig_of(t_start);
                    const int64_t cst = Q.starts[(size_t)e.qc], cln = Q.lens[(size_t)e.qc];
                    int64_t qs = (int64_t)h.q_start - (cst - QB.origin), qe = qs + h.len;
                    if (h.strand) { const int64_t s2 = cln - qe, e2 = cln - qs; qs = s2; qe = e2; }
                    paf += T.names[(size_t)tcg]; paf.push_back('\t');
                    put_num(t_start - T.starts[(size_t)tcg]); paf.push_back('\t');
                    put_num(t_start - T.starts[(size_t)tcg] + h.len); paf.push_back('\t');
                    paf += Q.names[(size_t)e.qc]; paf.push_back('\t');
                    put_num(qs); paf.push_back('\t'); put_num(qe); paf.push_back('\n');
                }
            }
            if (p.markend) paf += "# lastz end-of-file\n";
            x0 += n_tb * n_qb;
        }
    } else
    for (size_t k = 0; k < n_pairs; k++) {
        const size_t n_tb = tblk[k].size(), n_qb = qblk[k].size();
        for (size_t qb = 0; qb < n_qb; qb++) {
            Job *const *row = nullptr;
            std::vector<Job *> rowv(n_tb);
            for (size_t tb = 0; tb < n_tb; tb++) rowv[tb] = jobs[x0 + qb * n_tb + tb].get();
            row = rowv.data();
            if (n_tb == 1) { paf += row[0]->res.paf; continue; }
            // per query sequence and strand the alignments of every target block, in anchor order over the whole target
            struct Ent { int32_t qc, strand, nscore; int64_t tg; int32_t aq; const Result *r; size_t k; };
            std::vector<Ent> ents;
            for (size_t tb = 0; tb < n_tb; tb++) {
                const Result &r = row[tb]->res;
                for (size_t a = 0; a < r.alns.size(); a++) {
                    const miblast_aln &A = r.alns[a];
                    ents.push_back(Ent{A.q_contig, A.strand, -r.aln_anchor_score[a], (int64_t)A.anchor_t + tblk[k][tb].origin, A.anchor_q, &r, a});
                }
            }
            std::stable_sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) {
                if (a.qc != b.qc) return a.qc < b.qc;
                if (a.strand != b.strand) return a.strand < b.strand;
                if (a.nscore != b.nscore) return a.nscore < b.nscore;
                if (a.tg != b.tg) return a.tg < b.tg;
                return a.aq < b.aq;
            });
            for (const Ent &e : ents) paf.append(e.r->paf, e.r->line_off[e.k], e.r->line_off[e.k + 1] - e.r->line_off[e.k]);
        }
        x0 += n_tb * n_qb;
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (int d = 0; d < n_dev; d++)
            for (const auto &e : dev_done[(size_t)d]) add_stats(*stats, jobs[e.first]->res.stats, e.second);
        stats->t_total = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_begin;
    }
    return MIBLAST_OK;
}

}  // namespace mb

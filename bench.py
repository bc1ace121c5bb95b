#!/usr/bin/env python3
"""bench.py -- blast-phase throughput of the MI355X-native lastz replacement.

Workload of the headline line (BASELINE.json configs[2], SURVEY.md 8d "Config 3"): the blast phase of a whole progressive run
over the evolverMammals guide tree (/root/reference/examples/evolverMammals.txt:1) on a synthetic stand-in for its five genomes
(the FASTA files are URLs; 600 kb ancestor, seed 2001, cactus_amd/gen.py): per internal node (mr, Anc1, Anc2, Anc0) one lastz call
per ingroup pair and, per ingroup, a chain of calls to the <= 3 nearest outgroups with the still-unaligned sequence only
(/root/reference/src/cactus/paf/local_alignment.py:806-835, 421-526) -- 20 lastz calls of ~0.6 Mb x <= 0.6 Mb,
each with the option set its phylogenetic distance selects (cactus_progressive_config.xml:10-13,130-137).  A step = that whole
phase once, with the genomes already resident in HBM: the calls of one dependency level that share an option set go through ONE
miblast_align_pairs call (they are independent Toil jobs in the reference); trimmed sub-sequences are products of the step and are
uploaded inside it.  `--workload pair` is the bring-up configuration configs[1] (one 1 Mb x 1 Mb pair per GPU).

N > 1: one process per GPU (`--gpus N` spawns them when not started under torchrun), each running the same phase on its own
copy of the data (weak scaling, per-GPU work held constant, no data-path collective); the only exchange is the gather of the
final PAF bytes to rank 0 over RCCL, inside the timed region.

Prints ONE JSON line on rank 0.  value = dp_cells (the oracle-defined counter: cells of committed anchors only, speculative
work NOT counted) per second of whole-job wall time.  cpu_baseline runs the CPU oracle on the same calls and reports whether
every call's PAF bytes are equal (`same_bytes`).
"""
from __future__ import annotations

import argparse
import json
import os
import queue
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# A process of this bench holds a dozen HIP streams (contexts of concurrent calls, the lanes of their batched calls); with the
# runtime's default of 4 hardware queues two lanes of one call can land on the same queue, where their launches run one after the
# other instead of side by side (batched_pairs leg: 34 instead of 31 ms, whichever way the streams happen to be dealt).  Set before
# the HIP runtime starts; an explicit setting of the caller's wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# A step of the phase waits for the device some thirty times on its critical path (seed stage hand-overs, every DP launch's
# results, traceback), each time for a millisecond or less.  The ROCm runtime wakes a waiting thread through an interrupt by
# default; polling the completion signal instead takes ~25 us off every wait (19.1 -> 18.3 ms per step) at the price of a
# spinning core per waiting thread -- the library's host threads spin between their tasks anyway.  Caller's setting wins.
def _cores_per_rank():
    ranks = int(os.environ.get("LOCAL_WORLD_SIZE", "0") or 0)
    if not ranks and "--gpus" in sys.argv[1:-1]:
        try:
            ranks = int(sys.argv[sys.argv.index("--gpus") + 1])
        except ValueError:
            ranks = 0
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return cores / max(1, ranks)


if _cores_per_rank() >= 24:                       # (a rank keeps about twenty threads busy then: not on a node that has fewer per GPU)
    os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")

DEFAULT_ARGS = "--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=100000"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
TRACE_BYTES_PER_CELL = 0.5     # SURVEY 8d "Gapped: write 0.5*C (4-bit traceback)": the algorithmic figure roofline.achieved is computed from
TIMELINE = bool(os.environ.get("MIBLAST_BENCH_TIMELINE"))       # per-call wall times of a step on stderr
TRACE_BYTES_WRITTEN = 0.5      # what the DP kernels store per evaluated cell: 4-bit trace codes, two columns per byte
PER_PAIR = ("seed_lookups", "seed_hits", "hits_extended", "ungapped_cols", "hsps", "anchors", "dp_sides", "dp_cells", "dp_rows", "alignments",
            "t_index", "t_seed", "t_ungapped_kernel_ms", "t_sort_ms", "t_seedfill_ms", "seed_binned")
PER_BATCH = ("t_gapped", "gapped_rounds", "dp_sides_run", "dp_cells_run", "dp_rows_run", "t_dp_kernel_ms", "t_dp_busy_ms", "dp_kernel_launches",
             "relay_accepted", "relay_rejected", "t_traceback_ms", "t_merge_ms", "dp_reruns", "relay_inline_checks", "relay_inline_continued")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("evolver", "pair", "chr20", "hm", "hm30"), default="evolver",
                    help="evolver: BASELINE configs[2] stand-in (default; weak scaling: every GPU its own phase); pair: configs[1], one synthetic "
                         "chunk pair per GPU; chr20: configs[3], ONE genome pair whose chunk pairs are dealt to the GPUs (strong scaling); hm: the "
                         "scaled human-mouse stand-in of configs[4], same sharding; hm30: the same genome pair cut at the reference's own chunk size (30 Mb: 2 chunk pairs, one of them 32.6 Mb x 32.0 Mb)")
    ap.add_argument("--split-strands", default="auto", choices=("auto", "0", "1"),
                    help="chunk-scale workloads: deal (chunk pair, query strand) units instead of whole chunk pairs (exact: miblast_params.strands; the halves of a "
                         "pair are put together on rank 0) -- auto: when there are fewer than four chunk pairs per GPU")
    ap.add_argument("--chunk-legs", type=int, default=2, help="evolver workload: also time the chunk-scale configurations chr20 (configs[3]) and hm (configs[4] stand-in) "
                                                              "on this GPU, every chunk pair checked against its oracle digest (0 = skip; 1 = chr20 + hm; 2, the default, = + hm30: "
                                                              "the stand-in genome pair at the reference's own 30 Mb chunk size); reported under chr20 / hm / hm30")
    ap.add_argument("--chr20-bases", type=int, default=64_444_167, help="chr20 workload: bases of the synthetic chromosome (SURVEY 8d config 4)")
    ap.add_argument("--chr20-chunk", type=int, default=30_000_000,
                    help="chr20 workload: chunkSize (cactus_progressive_config.xml:90; overlap 10 000, :92).  A finer chunking gives more pairs to deal "
                         "but is not bit-comparable to the CPU path's 30 Mb chunks")
    ap.add_argument("--primates-leg", type=int, default=1, help="evolver workload: also time the evolverPrimates stand-in, BASELINE configs[0] (0 = skip); reported under primates")
    ap.add_argument("--ancestor", type=int, default=600_000, help="ancestor length of the evolverMammals stand-in (SURVEY 8d config 3: 600 kb)")
    ap.add_argument("--size", type=int, default=1_000_000, help="bases per chunk of the pair workload / pair leg (config 2: 1 Mb)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--random-pair", action="store_true", help="pair workload: pure-random pair (seed/ungapped isolation)")
    ap.add_argument("--cpu-sample", type=int, default=1, help="0 = skip the CPU-oracle baseline leg")
    ap.add_argument("--chain-leg", type=int, default=400,
                    help="syntenic series in the synthetic PAF of the chaining-stage leg (0 = skip); reported under chain_stage, never in value")
    ap.add_argument("--lastz-args", default=DEFAULT_ARGS, help="option set of the pair workload")
    ap.add_argument("--pairs-per-gpu", type=int, default=1, help="pair workload: chunk pairs per GPU per step, aligned in ONE batched call")
    ap.add_argument("--pair-leg", type=int, default=1, help="evolver workload: also time the 1 Mb x 1 Mb pair of configs[1] (0 = skip); reported under pair_1mb")
    ap.add_argument("--batch-leg", type=int, default=16,
                    help="1 Mb x 1 Mb chunk pairs aligned in ONE miblast_align_pairs call (0 = skip); reported under batched_pairs, never in value")
    ap.add_argument("--seed-leg", type=int, default=8_000_000,
                    help="chunk size of the extra seed-stage leg on a pure-random pair (0 = skip); reported under seed_stage, never in value")
    ap.add_argument("--full-out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_full.json"),
                    help="file the FULL result object goes to (every leg with its counters, notes and per-pair figures); stdout's one line is the compact form of it")
    return ap.parse_args()


LINE_LIMIT = 6000              # bytes of the printed line (the driver reads an 8 KB tail of stdout: round 5's 20 KB line did not parse)


def _num(x, digits=5):
    """floats at `digits` significant figures (the full file keeps every digit)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _short(text, n=200):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def compact_line(out, full_path):
    """The ONE line of stdout: the contract's keys, the roofline of the dominant kernel, the CPU baseline, and one small object per leg.  Everything
    else (per-stage counters, notes, per-pair CPU seconds, the legs' own baselines) is in the file `full` names."""
    def pick(d, keys):
        return {k: _num(d[k]) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}

    line = {k: _num(out[k], 7) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in out}
    cfg = out.get("config", {})
    line["config"] = {"workload": _short(cfg.get("workload", ""), 420), "collective_backend": cfg.get("collective_backend"),
                      "timed_region": "sequence sets resident in HBM: FASTA parse + host-to-device copy are OUTSIDE the step (every run_lastz job pays them; DESIGN.md section 6 has the inclusive rate)"}
    for k in ("paf_md5", "units_per_rank"):
        if k in cfg:
            line["config"][k] = cfg[k]
    sp = out.get("step_ms_spread") or {}
    line["step_ms"] = pick(sp, ("min", "median", "max"))
    r = out.get("roofline") or {}
    line["roofline"] = dict(pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "algorithmic_bytes_per_launch", "cells_per_launch", "bytes_per_cell")),
                            kernel="k_ydrop2")
    if isinstance(r.get("valu"), dict):
        line["roofline"]["valu"] = pick(r["valu"], ("frac", "frac_of_measured_peak", "cells_evaluated_per_s_per_gpu"))
    if isinstance(r.get("saturated"), dict):
        line["roofline"]["saturated"] = pick(r["saturated"], ("frac", "launch_ms", "gapped_gcells_per_s_kernel"))
    if isinstance(out.get("hbm_read"), dict):
        line["hbm_read"] = pick(out["hbm_read"], ("frac", "achieved_GBps", "peak_GBps", "bytes_total"))
    for k in ("speculation_factor", "gapped_gcells_per_s_kernel", "seeds_per_s", "device_allocs_in_timed_steps"):
        if k in out:
            line[k] = _num(out[k])
    if isinstance(out.get("parity"), dict):
        line["parity"] = pick(out["parity"], ("same_bytes", "pairs_checked", "pairs_differing"))
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = pick(cb, ("value", "unit", "cores", "kind", "seconds", "calls", "same_bytes", "same_dp_cells", "calls_differing"))
        line["cpu_baseline"]["sample"] = _short(cb.get("sample", ""), 160)
        if isinstance(cb.get("node"), dict):
            line["cpu_baseline"]["node"] = pick(cb["node"], ("value", "cores", "cores_available", "seconds_wall", "same_bytes"))
    legs = {}
    for name in ("primates", "pair_1mb", "batched_pairs", "seed_stage", "chr20", "hm", "hm30", "chain_stage"):
        leg = out.get(name)
        if not isinstance(leg, dict):
            continue
        o = pick(leg, ("ms_per_step", "ms_per_call", "steps", "value", "unit", "seeds_per_s", "records_per_s", "seconds", "frac", "chunk_pairs", "gapped_gcells_per_s_kernel",
                       "device_allocs_in_timed_steps"))
        if isinstance(leg.get("step_ms_spread"), dict):
            o["step_ms"] = pick(leg["step_ms_spread"], ("min", "median", "max"))
        if isinstance(leg.get("hbm_read"), dict):
            o["hbm_read_frac"] = _num(leg["hbm_read"].get("frac"))
        if isinstance(leg.get("roofline_dp") or leg.get("roofline"), dict):
            o["dp_frac"] = _num((leg.get("roofline_dp") or leg.get("roofline")).get("frac"))
        if isinstance(leg.get("parity"), dict):
            o["parity"] = pick(leg["parity"], ("same_bytes", "pairs_checked", "pairs_differing"))
        if isinstance(leg.get("cpu_baseline"), dict):
            o["cpu_baseline"] = pick(leg["cpu_baseline"], ("value", "unit", "records_per_s", "cores", "seconds", "same_bytes"))
        legs[name] = o
    if legs:
        line["legs"] = legs
    if "legs_note" in out:
        line["legs_note"] = _short(out["legs_note"], 200)
    line["full"] = full_path
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:                               # never again a line the driver cannot read: the legs go first, then the notes
        for k in ("legs_note", "legs", "step_ms", "hbm_read"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= LINE_LIMIT:
                break
    return text


def emit(out, full_path):
    """full object -> file, compact line -> stdout (exactly one line)"""
    try:
        tmp = full_path + ".tmp%d" % os.getpid()
        with open(tmp, "w") as f:
            json.dump(out, f)
            f.write("\n")
        os.replace(tmp, full_path)
    except OSError as e:                                     # (a read-only tree: the line still goes out)
        print(f"bench.py: could not write {full_path}: {e}", file=sys.stderr)
        full_path = None
    print(compact_line(out, full_path), flush=True)


def spawn_ranks(a) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves (one process per GPU, the launch contract's env
    variables), stream rank 0's line through, fail if any rank does."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def add_stats(agg, stats_list):
    """counters of one batched call: per-pair counters add up, launch-level figures (shared by the pairs in flight) count once"""
    for k in PER_PAIR:
        agg[k] = agg.get(k, 0) + sum(s[k] for s in stats_list)
    for k in PER_BATCH:
        agg[k] = agg.get(k, 0) + stats_list[0][k]


def newick_to_gen_tree(node):
    """cactus_amd.blast_phase.Node -> the nested tuples cactus_amd.gen.make_tree_genomes walks"""
    return (node.iD, node.distance, [newick_to_gen_tree(c) for c in node.children])


class EvolverPhase:
    """The blast phase of a whole progressive run on one GPU: BASELINE configs[2] (evolverMammals, the headline) or configs[0]
    (evolverPrimates) -- see the module docstring."""

    def __init__(self, a, ctx, rank, which="mammals"):
        from cactus_amd import blast_phase as bp, gen, miblast
        from cactus_amd.paf.local_alignment import select_lastz_params
        from cactus_amd.shared.configWrapper import load_config
        self.bp, self.miblast, self.ctx = bp, miblast, ctx
        cfg = load_config()
        self.options = lambda d: select_lastz_params(d, cfg, 0)
        blast = cfg.find("blast")
        self.trim = (int(blast.attrib["trimMinSize"]), int(blast.attrib["trimFlanking"]))
        max_div = float(cfg.find("constants").find("divergences").attrib["five"])
        if which == "mammals":
            tree = bp.parse_newick(bp.EVOLVER_MAMMALS_TREE)
            genomes = gen.make_tree_genomes(a.ancestor, 2001, ancestors=True)
        else:
            # "primates-like" set, 600 kb ancestor, the leaves at the tree distances of examples/evolverPrimates.txt:1, seed 1001
            tree = bp.parse_newick(bp.EVOLVER_PRIMATES_TREE)
            t = newick_to_gen_tree(tree)
            genomes = gen.make_tree_genomes(a.ancestor, 1001, tree=("root", t[2]), ancestors=True)
        self.calls = bp.blast_phase_calls(tree, max_div=max_div, max_outgroups=3)
        # every rank runs the same phase (weak scaling with per-GPU work held exactly constant); only the names differ
        self.fasta = {k: gen.fasta_bytes([("id=%s|%s_r%d" % (k, k, rank), v)]) for k, v in genomes.items()}
        self.resident = {fa: ctx.seqset_from_fasta_bytes(fa) for fa in self.fasta.values()}      # genomes resident in HBM before the timed region
        for fa in self.fasta.values():
            bp.parsed_records(fa, keep=True)                                                     # ... and parsed once on the host, like their upload
        self.params = {}
        # the option sets of a dependency level are independent jobs too (1 x "four" beside 9 x "default" at level 0), and so are the
        # ingroup pairs nothing waits for beside the chains of outgroup calls: each job in flight gets its own context (stream +
        # workspace) on this GPU and they run concurrently, as Toil runs independent jobs of a node.
        self.contexts = [ctx] + [miblast.Context(ctx.device) for _ in range(max(0, int(os.environ.get("MIBLAST_BENCH_CONTEXTS", "4")) - 1))]
        self.free_contexts = queue.Queue()                     # align_batch blocks on it: never more calls in flight than contexts
        self.background_contexts = queue.Queue()               # ... those of the jobs nothing waits for: their launches yield to the chains'
        n_bg = min(len(self.contexts) - 1, int(os.environ.get("MIBLAST_BENCH_BACKGROUND", "2"))) if len(self.contexts) > 2 else 0
        for k, cx in enumerate(self.contexts):
            if k >= len(self.contexts) - n_bg:
                self.background_contexts.put(cx.set_priority(-1))
            else:
                self.free_contexts.put(cx)
        sets = sorted({self.options(c.distance).split()[0] + " ..." for c in self.calls})
        if which == "mammals":
            self.describe = (f"evolverMammals blast phase stand-in (BASELINE configs[2], SURVEY 8d config 3): {len(self.calls)} lastz calls over the guide tree of "
                             f"examples/evolverMammals.txt:1, synthetic genomes from a {a.ancestor} bp ancestor (seed 2001), ingroup trimming between outgroups "
                             "on the device, option set per call by distance (1 x \"four\", rest \"default\")")
        else:
            self.describe = (f"evolverPrimates blast phase stand-in (BASELINE configs[0], SURVEY 8d config 1): {len(self.calls)} lastz calls over the guide tree of "
                             f"examples/evolverPrimates.txt:1, synthetic genomes from a {a.ancestor} bp ancestor (seed 1001), every call option set \"one\" ({', '.join(sets)})")

    def step(self, keep=None):
        agg = {}
        made = []
        t_step = time.perf_counter()

        lock = threading.Lock()

        def align_batch(pairs, opts, pool=None):
            pool = pool or self.free_contexts
            cx = pool.get()
            try:
                with lock:
                    pm = self.params.get(opts)
                    if pm is None:
                        pm = self.params[opts] = self.miblast.params_from_args(opts.split())
                sets = []
                for tf, qf in pairs:
                    s = []
                    for fa in (tf, qf):
                        h = fa if isinstance(fa, self.miblast.SeqSet) else self.resident.get(fa)      # (a trimmed query is a resident set already)
                        if h is None:
                            h = cx.seqset_from_fasta_bytes(fa)
                            with lock:
                                made.append(h)
                        s.append(h)
                    sets.append(tuple(s))
                t0 = time.perf_counter()
                rs = cx.align_pairs(sets, pm)
            finally:
                pool.put(cx)
            with lock:
                add_stats(agg, [r.stats for r in rs])
                # (bases of the call's targets and queries: SURVEY 8d's read terms -- every call builds its table and packs its strands)
                agg["t_bases"] = agg.get("t_bases", 0) + sum(t.total for t, _ in sets)
                agg["q_bases"] = agg.get("q_bases", 0) + sum(q.total for _, q in sets)
            if TIMELINE:
                print(f"[bench] align_pairs of {len(pairs)} pairs: {(time.perf_counter() - t0) * 1e3:.2f} ms (since step start {(time.perf_counter() - t_step) * 1e3:.2f}); "
                      f"t_gapped {rs[0].stats['t_gapped'] * 1e3:.2f}, dp {rs[0].stats['t_dp_kernel_ms']:.2f}, max t_seed+index {max(r.stats['t_seed'] + r.stats['t_index'] for r in rs) * 1e3:.2f}", file=sys.stderr)
            return [r.paf for r in rs]

        def trim_resident(items, min_size, flank):
            """what is left of every chain's ingroup after its previous call, cut out on the device (miblast_seqsets_unaligned)"""
            t0 = time.perf_counter()
            qs = [q if isinstance(q, self.miblast.SeqSet) else self.resident[q] for q, _ in items]
            cx = self.free_contexts.get()                      # (a context serves one call at a time, and jobs nothing waits for may be out)
            try:
                outs = cx.seqsets_unaligned(qs, [paf for _, paf in items], min_size, flank)
            finally:
                self.free_contexts.put(cx)
            made.extend(o for o in outs if o is not None)
            if TIMELINE:
                print(f"[bench] trim_resident of {len(items)} chains: {(time.perf_counter() - t0) * 1e3:.2f} ms", file=sys.stderr)
            return outs

        if os.environ.get("MIBLAST_BENCH_TEXT_TRIM", "0") == "0":
            align_batch.trim_resident = trim_resident
        if not self.background_contexts.empty():
            align_batch.background = lambda pairs, opts: align_batch(pairs, opts, self.background_contexts)
        align_batch.concurrent = len(self.contexts)
        align_batch.split_above = int(os.environ.get("MIBLAST_BENCH_SPLIT", "0")) if len(self.contexts) > 1 else 0
        res = self.bp.run_blast_phase(self.fasta, self.calls, self.options, align_batch, *self.trim,
                                      on_call=(lambda c, tf, qf, paf: keep.append((tf, qf, self.options(c.distance), paf))) if keep is not None else None)
        for h in made:
            h.close()
        if TIMELINE:
            print(f"[bench] step total {(time.perf_counter() - t_step) * 1e3:.2f} ms", file=sys.stderr)
        paf = b"".join(part for v in res.values() for part in (v["ingroup"], v["outgroup"]))
        return agg, paf


class PairWorkload:
    """BASELINE configs[1]: P synthetic chunk pairs per GPU (SURVEY 8d config 2 recipe), one batched call per step."""

    def __init__(self, a, ctx, rank):
        from cactus_amd import gen, miblast
        self.ctx = ctx
        self.pm = miblast.params_from_args(a.lastz_args.split())
        self.sets, self.fasta = [], []
        for k in range(max(1, a.pairs_per_gpu)):
            t, q = gen.make_pair(a.size, a.seed + k, homologous=not a.random_pair)
            tf, qf = gen.fasta_bytes([(f"id=simT{rank}_{k}|chr1", t)]), gen.fasta_bytes([(f"id=simQ{rank}_{k}|chr1", q)])
            self.fasta.append((tf, qf))
            self.sets.append((ctx.seqset_from_fasta_bytes(tf), ctx.seqset_from_fasta_bytes(qf)))
        self.args = a.lastz_args
        self.describe = (f"{len(self.sets)} x ({a.size} x {a.size}) synthetic chunk pair(s) per GPU (BASELINE configs[1], SURVEY 8d config 2"
                         f"{', pure-random variant' if a.random_pair else ''}), seed {a.seed}+pair index, identical on every rank")

    def step(self, keep=None):
        agg = {}
        rs = self.ctx.align_pairs(self.sets, self.pm) if len(self.sets) > 1 else [self.ctx.align(*self.sets[0], self.pm, details=False)]
        add_stats(agg, [r.stats for r in rs])
        if keep is not None:
            keep.extend((tf, qf, self.args, r.paf) for (tf, qf), r in zip(self.fasta, rs))
        return agg, b"".join(r.paf for r in rs)


class ChunkWorkload:
    """The chunk-scale configurations (cactus_amd/workloads.py): `chr20` = BASELINE configs[3], `hm` = the scaled
    stand-in for configs[4] (config 5).  ONE genome pair, chunked exactly as the CPU path chunks it (faffy chunk -c chunkSize -o 10000:
    cactus_progressive_config.xml:90-92, /root/reference/src/cactus/paf/local_alignment.py:378-387), every (target chunk, query chunk)
    pair an independent job (:395-405) with the option set of its divergence.  The work units -- chunk pairs, or with --split-strands
    (chunk pair, query strand) -- are owned TARGET-MAJOR (cactus_amd.multigpu.assign_target_major): rank g owns the target
    chunks i mod N with all their units (fewer target chunks than ranks: a chunk's column is shared by a group of ranks, longest unit
    first); a rank aligns its share in ONE batched call: a target chunk's seed table is built once per step -- on one rank when there
    are at least as many target chunks as ranks, never more than ceil(Na / N) per rank -- and stays resident while the query chunks
    stream through it (the library keeps it with the set; miblast_drop_derived at the start of every step makes the step pay for it).  The only exchange
    is the gather of the framed PAFs to rank 0, which strings them together in chunk-pair order -- the bytes do not depend on the number
    of GPUs.  Total work is fixed: STRONG scaling."""

    def __init__(self, a, ctx, rank, world, which):
        from cactus_amd import miblast, workloads
        from cactus_amd.multigpu import assign_target_major
        self.ctx, self.rank, self.world, self.miblast = ctx, rank, world, miblast
        kw = {}
        if which == "chr20" and (a.chr20_bases, a.chr20_chunk) != (64_444_167, 30_000_000):
            kw = dict(bases=a.chr20_bases, chunk=a.chr20_chunk)
        self.w = w = workloads.by_name(which, **kw)
        self.OPTIONS = w.options
        self.pm = miblast.params_from_args(w.options.split())
        self.tfa, self.qfa, self.pairs = w.tfa, w.qfa, w.pairs
        # work units: whole chunk pairs (strand code 0), or -- few pairs per GPU -- (chunk pair, query strand): 1 '+', 2 '-'
        mode = getattr(a, "split_strands", "auto")
        self.split = mode == "1" or (mode == "auto" and world > 1 and len(w.pairs) < 4 * world)
        self.units = [(k, sc) for k in range(len(w.pairs)) for sc in ((1, 2) if self.split else (0,))]
        self.weights = [w.weights()[k] * (0.5 if sc else 1.0) for k, sc in self.units]
        # target-major ownership: rank g owns the target chunks i mod N with all their units; fewer target chunks than ranks:
        # a chunk's column is shared by a group of ranks (cactus_amd.multigpu.assign_target_major)
        self.shares = assign_target_major([w.pairs[k][0] for k, _ in self.units], self.weights, world)
        self.mine = self.shares[rank]
        self.tables_per_rank = [len({w.pairs[self.units[u][0]][0] for u in sh}) for sh in self.shares]
        self.pm_strand = {sc: miblast.params_from_args(w.options.split() + ([] if sc == 0 else ["--strand=" + ("plus" if sc == 1 else "minus")])) for sc in (0, 1, 2)}
        need_t, need_q = sorted({self.pairs[self.units[u][0]][0] for u in self.mine}), sorted({self.pairs[self.units[u][0]][1] for u in self.mine})
        self.T = {i: ctx.seqset_from_fasta_bytes(self.tfa[i]) for i in need_t}                 # only this rank's chunks go to its HBM
        self.Q = {j: ctx.seqset_from_fasta_bytes(self.qfa[j]) for j in need_q}
        self.describe = w.describe + f", dealt longest first to {world} GPU(s)"

    def close(self):
        for h in list(self.T.values()) + list(self.Q.values()):
            h.close()

    def step(self, keep=None):
        from cactus_amd.multigpu import _frame
        agg = {}
        blob = b""
        self.miblast.drop_derived()                        # a step builds every seed table, '-' strand and packed strand it uses (once)
        if self.mine:
            frames = []
            for sc in (0, 1, 2):                            # one batched call per strand selection of this rank's units
                us = [u for u in self.mine if self.units[u][1] == sc]
                if not us:
                    continue
                sets = [(self.T[self.pairs[self.units[u][0]][0]], self.Q[self.pairs[self.units[u][0]][1]]) for u in us]
                rs = self.ctx.align_pairs(sets, self.pm_strand[sc]) if len(sets) > 1 else [self.ctx.align(*sets[0], self.pm_strand[sc], details=False)]
                call = {}
                add_stats(call, [r.stats for r in rs])
                for key, v in call.items():
                    agg[key] = agg.get(key, 0) + v
                frames += [_frame(u, r.paf) for u, r in zip(us, rs)]
                if keep is not None and sc == 0:
                    keep.extend((self.tfa[self.pairs[self.units[u][0]][0]], self.qfa[self.pairs[self.units[u][0]][1]], self.OPTIONS, r.paf) for u, r in zip(us, rs))
            blob = b"".join(frames)
        else:
            for k in PER_PAIR + PER_BATCH:
                agg[k] = 0
        return agg, blob

    def assemble(self, gathered):
        """rank 0: the ranks' framed PAFs -> {pair index: PAF} and one PAF in chunk-pair order"""
        from cactus_amd.multigpu import _unframe, fasta_names, merge_strand_pafs
        by_unit = {}
        for part in gathered:
            for idx, paf in _unframe(part):
                by_unit[idx] = paf
        assert sorted(by_unit) == list(range(len(self.units))), "a work unit is missing from the gather"
        by_index = {}
        names = self.__dict__.setdefault("_qnames", {})    # record names of a query chunk, read off its FASTA once (not once per step: 30 Mb of text)
        for u, (k, sc) in enumerate(self.units):
            if sc == 0:
                by_index[k] = by_unit[u]
            elif sc == 1:                                   # (its '-' half is the next unit)
                qi = self.pairs[k][1]
                if qi not in names:
                    names[qi] = fasta_names(self.qfa[qi])
                by_index[k] = merge_strand_pafs(by_unit[u], by_unit[u + 1], names[qi])
        self.by_index = by_index
        return b"".join(by_index[k] for k in range(len(self.pairs)))

    def digest_check(self, by_index):
        """every chunk pair's PAF against the CPU oracle's digest of that pair (tests/golden/<key>_pairs.json, written by
        scripts/oracle_chunk_digests.py: the oracle needs 1 to 60 s per pair, so its runs are committed, not repeated in the bench)"""
        import hashlib
        path = os.path.join(ROOT, "tests", "golden", f"{self.w.key}_pairs.json")
        if not os.path.exists(path):
            return {"same_bytes": None, "note": f"no committed oracle digests for this variant of the workload ({os.path.basename(path)})"}
        gold = json.load(open(path))
        if gold.get("fasta_md5") != hashlib.md5(b"".join(self.tfa + self.qfa)).hexdigest():
            return {"same_bytes": None, "note": "the committed digests were made from other FASTA bytes (numpy version?)"}
        bad = [k for k in range(len(self.pairs)) if gold["pairs"][k] is None or hashlib.md5(by_index[k]).hexdigest() != gold["pairs"][k]["paf_md5"]]
        return {"same_bytes": not bad, "pairs_checked": len(self.pairs), "pairs_differing": len(bad), "kind": "md5 of every chunk pair's PAF against the CPU oracle's (tests/golden/%s)" % os.path.basename(path),
                "oracle_cpu_seconds_all_pairs": sum(p["oracle_seconds"] for p in gold["pairs"] if p),
                "oracle_dp_cells": sum(p["dp_cells"] for p in gold["pairs"] if p), "oracle_seed_hits": sum(p["seed_hits"] for p in gold["pairs"] if p)}

    def cpu_sample(self, by_index, budget_s=75.0):
        """SURVEY 8d's CPU baseline of a chunk-scale workload, live on this box in this run: the chunk pairs as min(pairs, cores)
        single-threaded oracle processes side by side, each pinned to a core (the reference's CPU path: one lastz process per chunk
        pair, one core each -- cactus_progressive_config.xml:4, local_alignment.py:402), every PAF diffed byte for byte against the
        GPU's, cells and hits from the processes' own counters.  With fewer cores than pairs the longest pairs go first; if the
        whole list would take more than ~budget_s of wall time the sample is the heaviest pairs that fit (never fewer than the cores
        hold, at least eight, the pairs WITH homology first -- they carry the DP cells)."""
        path = os.path.join(ROOT, "tests", "golden", f"{self.w.key}_pairs.json")
        n = len(self.pairs)
        est = [float(self.weights[k]) for k in range(n)] if not self.split else [float(self.w.weights()[k]) for k in range(n)]
        if os.path.exists(path):
            gold = json.load(open(path))["pairs"]
            est = [gold[k]["oracle_seconds"] if gold[k] else 60.0 for k in range(n)]
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        order = sorted(range(n), key=lambda k: (-est[k], k))
        pick = list(order)
        if os.path.exists(path) and max(est) >= budget_s and any(e < budget_s for e in est):
            # (a 30 Mb x 30 Mb pair under the default option set keeps the oracle busy for ten minutes: the live sample is the pairs that fit the
            #  budget; every pair's digest of the committed oracle run is checked in `parity` all the same)
            order = [k for k in order if est[k] < budget_s]
            pick = list(order)
        if os.path.exists(path) and sum(est) / max(1, min(cores, n)) > budget_s and max(est) < budget_s:
            pick, t = [], 0.0
            for k in order:
                if len(pick) >= max(8, min(cores, n)) and (t + est[k]) / max(1, min(cores, n)) > budget_s:
                    break
                pick.append(k); t += est[k]
        pick.sort()
        calls = [(self.tfa[self.pairs[k][0]], self.qfa[self.pairs[k][1]], self.OPTIONS, by_index[k]) for k in pick]
        node = cpu_baseline_concurrent(calls, None, None, order=[est[k] for k in pick])
        out = {"value": node["value"], "unit": "Gcell/s", "seeds_per_s": node["seeds_per_s"], "cores": node["cores"], "kind": "port",
               "sample": ("all %d chunk pairs" % n if len(pick) == n else "%d of the %d chunk pairs (the heaviest for the oracle: pairs %s)" % (len(pick), n, pick))
                         + " of: " + self.describe + " -- one single-threaded oracle process per pair, min(pairs, cores) at a time, pinned (SURVEY 8d)",
               "seconds": node["seconds_wall"], "cpu_seconds": node["cpu_seconds"], "same_bytes": node["same_bytes"], "sample_pairs": pick, "node": node}
        return out

    def b_read(self, tot, per, elapsed_step_s):
        """SURVEY 8d's algorithmic READ bytes of one step, per stage and whole-leg, against the HBM roof.  T, Q in bases; a target's table is
        built once per step (target-major), a query's strands are packed once."""
        step, nvar = self.pm.step, (13 if self.pm.transitions else 1)
        t_bases = sum(len(x) for x in self.tfa) * 60 / 61.0        # (FASTA bytes -> bases: 60 columns + newline)
        q_bases = sum(len(x) for x in self.qfa) * 60 / 61.0
        look, hits, cols, rows = tot["seed_lookups"] / per, tot["seed_hits"] / per, tot["ungapped_cols"] / per, tot["dp_rows"] / per
        stages = {
            "index": 1.375 * t_bases,                                         # 1 B/base codes in, 0.375 B/base packed read back for the words
            "seed_search": 0.375 * q_bases * 2 * len(self.tfa) + 8.0 * look + 4.0 * hits,      # packed strands streamed per target chunk, two bucket bounds per look-up, a position per hit
            "ungapped": 8.0 * hits + 0.5 * cols,                              # the hit keys + 2 x 0.25 B per column
            "gapped": 0.75 * rows,                                           # 2 x 0.375 B per DP row (both sequences); the traceback's read-back is of the same order
        }
        total = sum(stages.values())
        return {"bytes_per_step": {k: float(v) for k, v in stages.items()}, "bytes_total": float(total),
                "achieved_GBps": total / elapsed_step_s / 1e9, "peak_GBps": HBM_PEAK_GBS * self.world, "frac": total / elapsed_step_s / 1e9 / (HBM_PEAK_GBS * self.world),
                "note": "SURVEY 8d read terms only, nothing padded; the gapped DP is VALU / issue bound (its bytes are negligible), so a step in which it is "
                        "a third of the time cannot come near the roof: see per-stage kernel times"}


def phase_b_read(tot, per, elapsed_step_s, world, work):
    """SURVEY 8d's algorithmic READ bytes of one step of the blast phase, per stage and whole-phase, against the HBM roof (the same
    terms as the chunk legs' b_read; a call of the phase builds its own table and packs its own strands, so T and Q count per call)."""
    nvar_note = "13 or 1 look-ups per query position and strand are in seed_lookups"
    t_bases, q_bases = tot["t_bases"] / per, tot["q_bases"] / per
    look, hits, cols, rows = tot["seed_lookups"] / per, tot["seed_hits"] / per, tot["ungapped_cols"] / per, tot["dp_rows"] / per
    stages = {"index": 1.375 * t_bases, "seed_search": 0.375 * q_bases * 2 + 8.0 * look + 4.0 * hits, "ungapped": 8.0 * hits + 0.5 * cols, "gapped": 0.75 * rows}
    total = sum(stages.values())
    return {"bytes_per_step": {k: float(v) for k, v in stages.items()}, "bytes_total": float(total), "achieved_GBps": total / elapsed_step_s / 1e9,
            "peak_GBps": HBM_PEAK_GBS * world, "frac": total / elapsed_step_s / 1e9 / (HBM_PEAK_GBS * world),
            "note": "SURVEY 8d read terms only, nothing padded (" + nvar_note + "); a phase of twenty 0.6 Mb calls is latency bound (SURVEY 8d's worked example: "
                    "a 1 Mb pair's reads are 30 us at the roof), and its DP is VALU / issue bound: the whole-phase fraction says how far a small "
                    "configuration is from the roof, not how good the kernels are -- see roofline (DP kernel) and the chunk legs"}


def reduce_totals(tot, elapsed, dist, coll_dev):
    """counters summed over the ranks, the barrier-to-barrier time as the maximum over the ranks"""
    import torch
    keys = sorted(tot)
    vec = torch.tensor([float(tot[k]) for k in keys] + [elapsed], dtype=torch.float64, device=coll_dev)
    if dist is not None:
        tmax = vec[-1:].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        elapsed = float(tmax.item())
    return elapsed, {k: float(v) for k, v in zip(keys, vec[:-1].tolist())}


def cpu_throttle_state():
    """(nr_throttled, throttled_usec) of this container's CPU cgroup -- the bench box runs under a CPU quota; a process whose threads
    exceed it is frozen for the rest of the 100 ms period, which shows up as outlier steps"""
    try:
        kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
    except Exception:                       # noqa: BLE001
        return 0, 0


def timed_steps(work, steps, warmup, sync, gather):
    for _ in range(warmup):
        gather(work.step()[1])
    sync()
    import gc
    nogc = os.environ.get("MIBLAST_BENCH_NOGC", "1") != "0"           # the harness's own garbage collector stays out of the timed steps (as timeit keeps it)
    if nogc:
        gc.collect(); gc.disable()
    thr0, cpu0 = cpu_throttle_state(), time.process_time()
    from cactus_amd import miblast as _mbl
    allocs0 = _mbl.device_allocs()                     # hipMalloc + hipFree calls of the library so far: none are wanted inside the timed steps
    t0 = time.perf_counter()
    tot, keep = {}, None
    marks = [t0]
    for k in range(steps):
        keep = [] if k == steps - 1 else None          # the calls of the last timed step are kept for the byte diff with the oracle
        agg, paf = work.step(keep)
        gather(paf)
        for key, v in agg.items():
            tot[key] = tot.get(key, 0) + v
        marks.append(time.perf_counter())              # (a step ends with its PAF on the host: the spread of the steps, for the record)
    sync()
    elapsed = time.perf_counter() - t0
    if nogc:
        gc.enable()
    each = sorted((b - a) * 1e3 for a, b in zip(marks, marks[1:]))
    if os.environ.get("MIBLAST_BENCH_STEP_TIMES"):
        print("[bench] step times (ms): " + " ".join("%.1f" % ((b - a) * 1e3) for a, b in zip(marks, marks[1:])), file=sys.stderr)
    timed_steps.last_spread = {"min": each[0], "median": each[len(each) // 2], "max": each[-1]} if each else {}      # (of this rank)
    thr1 = cpu_throttle_state()
    tot["host_cpu_seconds"] = time.process_time() - cpu0
    tot["device_allocs_in_timed_steps"] = _mbl.device_allocs() - allocs0
    tot["host_throttled_periods"] = thr1[0] - thr0[0]
    tot["host_throttled_ms"] = (thr1[1] - thr0[1]) / 1e3
    return elapsed, tot, keep


def dp_roofline(tot, profile_name):
    """Dominant kernel (k_ydrop2) against both roofs.  HBM: algorithmic bytes per launch = SURVEY 8d's 0.5 B per evaluated cell
    (4-bit trace codes) + 16 B row record + ~2 sequence bytes per row, / the average launch duration measured with HIP events
    on the library's own stream.  bytes_per_cell_written is what the kernel stores per cell; traffic is the PMC figure."""
    launches = max(1.0, tot["dp_kernel_launches"])
    cells, rows = tot["dp_cells_run"] / launches, tot["dp_rows_run"] / launches
    dp_ms = tot["t_dp_kernel_ms"] / launches
    algo = cells * TRACE_BYTES_PER_CELL                        # SURVEY 8d, strictly: 0.5 B of trace per evaluated cell, nothing else
    algo_rows = algo + rows * 18.0                             # ... + what the kernel also moves per row: the 16-byte row record and ~2 sequence bytes
    achieved = algo / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0
    achieved_rows = algo_rows / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0
    traffic, note = None, "no committed PMC summary for this workload"
    path = os.path.join(ROOT, "profiles", profile_name)
    if os.path.exists(path):
        ks = json.load(open(path))["kernels"]
        dp = [v for name, v in ks.items() if "k_ydrop" in name]
        calls = sum(v["calls"] for v in dp)
        if calls:
            pmc = sum(v["calls"] * (v["fetch_bytes_corrected_per_call"] + v["write_size_bytes_per_call"]) for v in dp) / calls
            traffic = pmc / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else None
            note = ("PMC bytes per DP launch (corrected FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, profiles/%s: %.1f MB averaged over the %d "
                    "launches profiled) / this run's average launch duration" % (profile_name, pmc / 1e6, calls))
    busy_ms = tot.get("t_dp_busy_ms", 0.0)
    busy = {"achieved": tot["dp_cells_run"] * TRACE_BYTES_PER_CELL / (busy_ms * 1e-3) / 1e9, "busy_ms": busy_ms,
            "note": "all launches' algorithmic bytes / the time at least one DP launch was running (union of the HIP-event intervals): what the kernel "
                    "moves per second of its own time when launches of two streams overlap"} if busy_ms > 0 else None
    if busy:
        busy["frac"] = busy["achieved"] / HBM_PEAK_GBS
    return {"bound": "hbm", "kernel": "k_ydrop2 (one-sided Y-drop DP, one wave per piece; k_ydrop1 / k_ydrop for wider windows)", "over_busy_time": busy,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            # the same with the cells of COMMITTED anchors only (the oracle's dp_cells; speculative and repeated evaluations left out)
            "committed_only": {"achieved": (tot["dp_cells"] / launches) * TRACE_BYTES_PER_CELL / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0,
                               "frac": ((tot["dp_cells"] / launches) * TRACE_BYTES_PER_CELL / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0) / HBM_PEAK_GBS,
                               "note": "frac above counts every evaluated cell (what the kernel wrote in that time); this one only the oracle-defined dp_cells"},
            "algorithmic_bytes_per_launch": algo, "bytes_per_cell": TRACE_BYTES_PER_CELL, "bytes_per_cell_written": TRACE_BYTES_WRITTEN, "cells_per_launch": cells, "rows_per_launch": rows,
            "with_row_records": {"achieved": achieved_rows, "frac": achieved_rows / HBM_PEAK_GBS, "bytes_per_launch": algo_rows,
                                 "note": "the same plus 18 B per DP row (16-byte row record + sequence bytes); frac above is SURVEY 8d's 0.5 B per evaluated cell alone"},
            "launch_ms": dp_ms, "traffic_note": note,
            "note": "the DP is bound by instruction issue and latency per row, not by HBM (SURVEY 8d caveat): see roofline.valu for its meaningful ceiling"}


def run_rank(a):
    import torch
    from cactus_amd import miblast

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MIBLAST_BENCH_SINGLE_DEVICE"):      # test hook: several ranks on one GPU (use with MIBLAST_BENCH_BACKEND=gloo)
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus}, or without a launcher")
    dist = None
    coll_backend = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libmiblast has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        coll_backend = os.environ.get("MIBLAST_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        dist.init_process_group(backend=coll_backend, device_id=torch.device("cuda", local_rank) if coll_backend == "nccl" else None)

    from cactus_amd.multigpu import gather_bytes, share_host_cores
    host_threads = share_host_cores(int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))) if world > 1 else miblast.set_host_threads(0)
    ctx = miblast.Context(local_rank)
    coll_dev = torch.device("cuda", local_rank) if coll_backend != "gloo" else torch.device("cpu")
    sharded = a.workload in ("chr20", "hm", "hm30")
    work = EvolverPhase(a, ctx, rank) if a.workload == "evolver" else ChunkWorkload(a, ctx, rank, world, a.workload) if sharded else PairWorkload(a, ctx, rank)
    gathered = {}

    def gather(paf: bytes):
        """final hit list -> rank 0 (RCCL over xGMI); the sharded workload strings the ranks' shares together there"""
        gathered["last"] = gather_bytes(paf, dist, rank, world, coll_dev)
        if sharded and rank == 0:
            gathered["paf"] = work.assemble(gathered["last"])

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    elapsed, tot, keep = timed_steps(work, a.steps, a.warmup, sync, gather)
    step_spread = dict(timed_steps.last_spread, note="wall time of the single steps of the timed region on rank 0 (ms_per_step is their mean over all ranks' barrier-to-barrier time); "
                                                     "Python's cyclic garbage collector is switched off for the timed steps, as timeit does (MIBLAST_BENCH_NOGC=0 leaves it on)")
    elapsed, tot = reduce_totals(tot, elapsed, dist, coll_dev)
    # the chunk-scale configurations, sharded over ALL ranks (strong scaling; at N = 1 the same legs on one GPU): every rank takes part
    sharded_legs = {}
    if a.workload == "evolver" and a.chunk_legs > 0:
        for which in ("chr20", "hm", "hm30")[:2 + (1 if a.chunk_legs > 1 else 0)]:
            sharded_legs[which] = chunk_leg(a, ctx, which, rank, world, dist, coll_dev, sync)

    if rank == 0:
        per = a.steps * (1 if sharded else world)      # per-step figures: of one rank's phase (weak) / of the whole sharded job (strong)
        out = {
            "metric": "gapped X-drop Gcell/s (blast phase, whole job)",
            "value": tot["dp_cells"] / elapsed / 1e9,
            "unit": "Gcell/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "step_ms_spread": step_spread,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": work.describe,
                       "sharding": ("the chunk pairs of ONE genome pair dealt longest first to the GPUs (independent jobs, no data-path collective); framed PAFs "
                                    "gathered to rank 0 and strung together in chunk-pair order") if sharded else
                                   "every GPU runs its own copy of the phase (chunk pairs are independent jobs); gather of the PAF to rank 0",
                       "collective_backend": coll_backend, "host_threads_per_rank": host_threads,
                       "paf_bytes_gathered_per_step": sum(len(x) for x in gathered["last"]) if gathered.get("last") else 0},
            "seeds_per_s": tot["seed_hits"] / elapsed,
            "seed_lookups_per_s": tot["seed_lookups"] / elapsed,
            "dp_cells_per_step": tot["dp_cells"] / per, "seed_hits_per_step": tot["seed_hits"] / per, "alignments_per_step": tot["alignments"] / per,
            "strands_grouped_in_lds_per_step": tot["seed_binned"] / per,      # (mb_seed_bin.h: bins + LDS instead of the device-wide radix sort; the other strands went through rocprim)
            "stage_seconds_per_step": {k: tot[k] / per for k in ("t_index", "t_seed", "t_gapped")},
            "stage_kernel_ms_per_step": {"ydrop": tot["t_dp_kernel_ms"] / per, "ydrop_busy": tot["t_dp_busy_ms"] / per, "ungapped": tot["t_ungapped_kernel_ms"] / per,
                                         "sort": tot["t_sort_ms"] / per, "seed_fill": tot["t_seedfill_ms"] / per},
            # evaluated cells (speculative ones included) per second of DP kernel time: over the time at least one DP launch was running
            # (union of the launches' HIP-event intervals), and over the sum of the launch durations (launches of two streams overlap)
            "gapped_gcells_per_s_kernel": tot["dp_cells_run"] / max(1e-9, tot["t_dp_busy_ms"] * 1e-3 / world) / 1e9,
            "gapped_gcells_per_s_sum_of_launches": tot["dp_cells_run"] / max(1e-9, tot["t_dp_kernel_ms"] * 1e-3 / world) / 1e9,
            "speculation_factor": tot["dp_cells_run"] / max(1.0, tot["dp_cells"]),
            # hipMalloc + hipFree calls of the library inside the timed steps, all ranks (miblast_debug_device_allocs): a device allocation in the middle
            # of a step stalls every lane; 0 once the workspaces have met the workload in the warm-up steps
            "device_allocs_in_timed_steps": int(tot["device_allocs_in_timed_steps"]),
            "relay": {"pieces_per_step": tot["dp_sides_run"] / per, "dp_launches_per_step": tot["dp_kernel_launches"] / per,
                      "handovers_accepted_per_step": tot["relay_accepted"] / per, "handovers_rejected_per_step": tot["relay_rejected"] / per,
                      "checked_inside_the_launch_per_step": tot["relay_inline_checks"] / per, "pieces_that_went_on_inside_per_step": tot["relay_inline_continued"] / per,
                      "reruns_per_step": tot["dp_reruns"] / per,
                      "traceback_ms_per_step": tot["t_traceback_ms"] / per, "merge_ms_per_step": tot["t_merge_ms"] / per},
            "roofline": dp_roofline(tot, "r06_hbm_traffic_pmc.json" if a.workload == "evolver" else "r03_pair_hbm_traffic_pmc.json"),
            "hbm_read": phase_b_read(tot, per, elapsed / a.steps, world, work) if "t_bases" in tot else None,
            "host": {"cpu_seconds_per_step": tot["host_cpu_seconds"] / per, "busy_threads_avg": tot["host_cpu_seconds"] / world / max(1e-9, elapsed),
                     "cgroup_throttled_periods": tot["host_throttled_periods"], "cgroup_throttled_ms": tot["host_throttled_ms"],
                     "note": "all ranks; a CPU-quota container freezes the process when its threads exceed the quota (outlier steps)"},
        }
        try:
            # the DP is VALU / issue bound, so its cells/s are also put against the int32 VALU peak: evaluated cells
            # (speculative ones included) x the ~10 integer operations the recurrence needs per cell (3 max, 3 add, score lookup,
            # 2 compares for the y-drop test, trace code) / (CUs x 4 SIMD-32 units x 32 lanes x clock = the int32 VALU peak,
            # half the 157 TFLOP/s fp32 FMA figure of MI355X_MICROARCH.md)
            prop = torch.cuda.get_device_properties(local_rank)
            clock_hz = float(getattr(prop, "clock_rate", 2_400_000)) * 1e3
            peak_ops = prop.multi_processor_count * 4 * 32 * clock_hz
            cells_per_s = tot["dp_cells_run"] / max(1e-12, tot["t_dp_busy_ms"] * 1e-3 / world)
            out["roofline"]["valu"] = {"cells_evaluated_per_s_per_gpu": cells_per_s, "min_int_ops_per_cell": 10, "peak_lane_ops_per_s": peak_ops,
                                       "frac": cells_per_s * 10 / peak_ops, "cus": prop.multi_processor_count, "clock_ghz": clock_hz / 1e9,
                                       "peak_measured_lane_ops_per_s": peak_ops / 2, "frac_of_measured_peak": cells_per_s * 10 / (peak_ops / 2),
                                       "peak_note": "peak = CUs x 4 SIMD-32 x 32 lanes x clock (one int32 VALU instruction per SIMD and 2 cycles); measured on this chip an int32 "
                                                    "VALU instruction of a wave64 takes ~4 cycles (DESIGN.md section 5): half that peak, given as peak_measured"}
        except Exception as e:                               # noqa: BLE001  (never lose the line over a device-property quirk)
            out["roofline"]["valu"] = {"error": str(e)}
        if sharded:
            import hashlib
            out["config"]["paf_md5"] = hashlib.md5(gathered["paf"]).hexdigest()
            out["config"]["paf_bytes"] = len(gathered["paf"])
            out["config"]["work_unit"] = "(chunk pair, query strand)" if work.split else "chunk pair"
            out["config"]["units_per_rank"] = [len(x) for x in work.shares]
            out["config"]["tables_built_per_step"] = {"per_rank": work.tables_per_rank, "total": sum(work.tables_per_rank), "target_chunks": len(work.tfa),
                                                      "bound_per_rank": -(-len(work.tfa) // world)}
            out["parity"] = work.digest_check(work.by_index)
            out["hbm_read"] = work.b_read(tot, per, elapsed / a.steps)
            out["config"]["residency"] = ("chunks resident in HBM before the timed region (parse + upload excluded); seed tables, '-' strands and packed strands "
                                          "are dropped at the start of every step and built once per step and chunk (target-major, SURVEY 8e)")
        out["roofline"]["overlap_note"] = ("a call of several pairs runs the gapped stages of two groups of its pairs on two streams (MIBLAST_GAPPED_LANES=2): their DP launches "
                                           "share the GPU, so a launch's HIP-event duration -- the denominator here -- is longer than it would be alone, and the "
                                           "durations add up to more than the wall time they cover")
        if world > 1:
            # the extra legs and the CPU baseline are single-GPU figures: measured at N = 1 only (the ranks of a scaling run do not wait
            # for rank 0 to time them)
            a.primates_leg = a.pair_leg = a.batch_leg = a.seed_leg = a.chain_leg = 0
            if not sharded:
                a.cpu_sample = 0
            out["legs_note"] = ("primates / pair_1mb / batched_pairs / seed_stage / chain_stage / cpu_baseline are measured at --gpus 1 only; chr20 / hm are the "
                                "SHARDED chunk-scale configurations at every N (strong scaling: the chunk pairs of one genome pair dealt over the ranks, framed PAFs gathered)")
        if a.workload == "evolver" and a.primates_leg > 0:
            out["primates"] = primates_leg(a, ctx)
        if a.workload == "evolver" and a.pair_leg > 0:
            out["pair_1mb"] = pair_leg(a, ctx)
        if a.batch_leg > 1:
            out["batched_pairs"] = batch_leg(a, ctx)
            # the dominant kernel on launches that fill the GPU (16 pairs in one call): the like-for-like figure across rounds
            sat = out["batched_pairs"].get("roofline") or {}
            out["roofline"]["saturated"] = dict(sat, source="batched_pairs leg: 16 x (1 Mb x 1 Mb) in one miblast_align_pairs call",
                                                gapped_gcells_per_s_kernel=out["batched_pairs"].get("gapped_gcells_per_s_kernel"))
        if a.seed_leg > 0 and not a.random_pair:
            out["seed_stage"] = seed_stage_leg(a, ctx)
        for which, leg in sharded_legs.items():
            out[which] = leg
        if a.chain_leg > 0:
            out["chain_stage"] = chain_stage_leg(a, ctx)
        if a.cpu_sample > 0:
            if sharded:
                out["cpu_baseline"] = work.cpu_sample(work.by_index)
            else:
                out["cpu_baseline"] = cpu_baseline(keep, tot["dp_cells"] / per, work.describe)
        emit(out, a.full_out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pair_leg(a, ctx):
    """BASELINE configs[1], the bring-up configuration and round 1's headline: one 1 Mb x 1 Mb pair, one call per step; diffed
    against the oracle (1.6 s)."""
    import types
    b = types.SimpleNamespace(size=a.size, seed=a.seed, random_pair=False, lastz_args=DEFAULT_ARGS, pairs_per_gpu=1)
    w = PairWorkload(b, ctx, 0)
    elapsed, tot, keep = timed_steps(w, a.steps, a.warmup, lambda: None, lambda paf: None)
    r = dp_roofline(tot, "r03_pair_hbm_traffic_pmc.json")
    out = {"workload": w.describe, "ms_per_step": 1e3 * elapsed / a.steps, "value": tot["dp_cells"] / elapsed / 1e9, "unit": "Gcell/s",
           "seeds_per_s": tot["seed_hits"] / elapsed, "speculation_factor": tot["dp_cells_run"] / max(1.0, tot["dp_cells"]),
           "gapped_gcells_per_s_kernel": tot["dp_cells_run"] / max(1e-9, tot["t_dp_busy_ms"] * 1e-3) / 1e9,
           "stage_kernel_ms_per_step": {"ydrop": tot["t_dp_kernel_ms"] / a.steps, "ungapped": tot["t_ungapped_kernel_ms"] / a.steps,
                                        "sort": tot["t_sort_ms"] / a.steps, "seed_fill": tot["t_seedfill_ms"] / a.steps},
           "roofline": {k: r[k] for k in ("achieved", "frac", "traffic", "launch_ms", "cells_per_launch")}}
    if a.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(keep, tot["dp_cells"] / a.steps, w.describe)
    for t, q in w.sets:
        t.close(); q.close()
    return out


def primates_leg(a, ctx):
    """BASELINE configs[0] (the reference's own CPU-runnable case): the blast phase of the evolverPrimates run -- every pair within
    divergence "one" -- on the same terms as the headline: whole phase per step, genomes resident, oracle diff of every call."""
    w = EvolverPhase(a, ctx, 0, which="primates")
    elapsed, tot, keep = timed_steps(w, max(3, a.steps // 2), 2, lambda: None, lambda paf: None)
    steps = max(3, a.steps // 2)
    out = {"workload": w.describe, "ms_per_step": 1e3 * elapsed / steps, "value": tot["dp_cells"] / elapsed / 1e9, "unit": "Gcell/s", "calls": len(w.calls),
           "seeds_per_s": tot["seed_hits"] / elapsed, "speculation_factor": tot["dp_cells_run"] / max(1.0, tot["dp_cells"]),
           "dp_cells_per_step": tot["dp_cells"] / steps, "alignments_per_step": tot["alignments"] / steps,
           "stage_kernel_ms_per_step": {"ydrop": tot["t_dp_kernel_ms"] / steps, "ungapped": tot["t_ungapped_kernel_ms"] / steps,
                                        "sort": tot["t_sort_ms"] / steps, "seed_fill": tot["t_seedfill_ms"] / steps}}
    if a.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(keep, tot["dp_cells"] / steps, w.describe)
    for h in w.resident.values():
        h.close()
    return out


def batch_leg(a, ctx):
    """Many chunk pairs of one GPU in one call (the regime of BASELINE configs[3..4]: a genome pair is tens to thousands of
    30 Mb chunk pairs): P x (1 Mb x 1 Mb) pairs of the config-2 recipe through ONE miblast_align_pairs call -- seed stages on
    concurrent lanes, gapped stages merged into shared DP launches."""
    import types
    b = types.SimpleNamespace(size=a.size, seed=a.seed, random_pair=False, lastz_args=DEFAULT_ARGS, pairs_per_gpu=a.batch_leg)
    w = PairWorkload(b, ctx, 0)
    elapsed, tot, _ = timed_steps(w, 3, 1, lambda: None, lambda paf: None)
    r = dp_roofline(tot, "none")
    out = {"workload": w.describe, "ms_per_call": 1e3 * elapsed / 3, "value": tot["dp_cells"] / elapsed / 1e9, "unit": "Gcell/s",
           "seeds_per_s": tot["seed_hits"] / elapsed, "speculation_factor": tot["dp_cells_run"] / max(1.0, tot["dp_cells"]),
           "gapped_gcells_per_s_kernel": tot["dp_cells_run"] / max(1e-9, tot["t_dp_busy_ms"] * 1e-3) / 1e9,
           "dp_launches_per_call": tot["dp_kernel_launches"] / 3, "pieces_per_call": tot["dp_sides_run"] / 3,
           "roofline": {k: r[k] for k in ("achieved", "frac", "launch_ms", "cells_per_launch")}}
    for t, q in w.sets:
        t.close(); q.close()
    return out


def chunk_leg(a, ctx, which, rank=0, world=1, dist=None, coll_dev=None, sync=None):
    """A chunk-scale configuration as a leg of the default line: configs[3] (chr20) or the configs[4] stand-in (hm), on the same terms as
    `--workload chr20|hm` -- the chunk pairs of ONE genome pair dealt over the `world` ranks (every rank calls this; strong scaling: the
    work is fixed, ms_per_step is the maximum over the ranks, barrier to barrier, gather included), each rank its share in one batched
    call per step, target-major, every pair's PAF checked against the oracle's digest on rank 0, a bounded live CPU sample at N = 1,
    SURVEY 8d's read bytes against the HBM roof of the GPUs in use.  Returns the leg on rank 0, None elsewhere."""
    import hashlib
    from cactus_amd.multigpu import gather_bytes
    # (a context of its own: its streams and its lanes' are made together, now -- the runtime deals streams to hardware queues in the order
    #  they are made, and lanes added to a context that has been in use since the start of the process can end up sharing queues)
    from cactus_amd import miblast as _mb
    own = _mb.Context(ctx.device)
    w = ChunkWorkload(a, own, rank, world, which)
    steps, warm = 10, 3                                   # (three untimed steps: the lanes' buffers and the pool's trace arenas have met the heaviest pairs -- the lanes size their
                                                          #  buffers at the start of the second for what the first one met, the gapped stages' tables follow the groups the lanes happen to take)
    box = {}

    def gather(paf):
        box["last"] = gather_bytes(paf, dist, rank, world, coll_dev) if world > 1 else [paf]
    elapsed, tot, _ = timed_steps(w, steps, warm, sync or (lambda: None), gather)
    if world > 1:
        elapsed, tot = reduce_totals(tot, elapsed, dist, coll_dev)
    if rank != 0:
        w.close(); own.close()
        return None
    paf = w.assemble(box["last"])
    by_index = w.by_index
    r = dp_roofline(tot, "none")
    shares = w.shares
    load = [sum(w.weights[k] for k in sh) for sh in shares]
    unit_kind = "(chunk pair, query strand)" if w.split else "chunk pair"
    out = {"workload": w.describe, "chunk_pairs": len(w.pairs), "n_gpus": world, "scaling": "strong", "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
           # (rank 0's single steps: a step in which a lane's tables or trace arena grow -- a device allocation while the other lanes keep the device
           #  busy -- takes 0.3 - 0.6 s longer than the others; the mean above includes it, the median does not)
           "step_ms_spread": dict(timed_steps.last_spread),
           "value": tot["dp_cells"] / elapsed / 1e9, "unit": "Gcell/s",
           "work_unit": unit_kind, "units_per_rank": [len(sh) for sh in shares], "balance_by_weight": (sum(load) / len(load)) / max(load) if max(load) > 0 else 1.0,
           "ownership": "target-major (SURVEY 8e): rank g owns target chunks i mod N; fewer target chunks than ranks: a chunk's column shared by a group of ranks",
           "tables_built_per_step": {"per_rank": w.tables_per_rank, "total": sum(w.tables_per_rank), "target_chunks": len(w.tfa), "bound_per_rank": -(-len(w.tfa) // world)},
           "seeds_per_s": tot["seed_hits"] / elapsed, "seed_lookups_per_s": tot["seed_lookups"] / elapsed,
           "dp_cells_per_step": tot["dp_cells"] / steps, "seed_hits_per_step": tot["seed_hits"] / steps, "alignments_per_step": tot["alignments"] / steps,
           "strands_grouped_in_lds_per_step": tot["seed_binned"] / steps,      # (mb_seed_bin.h: bins + LDS instead of the device-wide radix sort; the rest went through rocprim)
           "speculation_factor": tot["dp_cells_run"] / max(1.0, tot["dp_cells"]),
           "device_allocs_in_timed_steps": int(tot["device_allocs_in_timed_steps"]),
           "gapped_gcells_per_s_kernel": tot["dp_cells_run"] / max(1e-9, tot["t_dp_busy_ms"] * 1e-3 / world) / 1e9,
           "stage_kernel_ms_per_step": {"ydrop": tot["t_dp_kernel_ms"] / steps, "ydrop_busy": tot["t_dp_busy_ms"] / steps, "ungapped": tot["t_ungapped_kernel_ms"] / steps,
                                        "sort": tot["t_sort_ms"] / steps, "seed_search": tot["t_seedfill_ms"] / steps,
                                        "note": "HIP-event durations summed over the pairs of the call (all ranks); the seed stages of up to twelve pairs share a GPU, so the sums exceed the wall time they cover"},
           "paf_md5": hashlib.md5(paf).hexdigest(), "paf_bytes": len(paf),
           "parity": w.digest_check(by_index), "hbm_read": w.b_read(tot, steps, elapsed / steps),
           "roofline_dp": {k: r[k] for k in ("achieved", "frac", "launch_ms", "cells_per_launch")}}
    if a.cpu_sample > 0 and world == 1 and which != "hm30":          # (hm30: the oracle's ten minutes are in tests/golden/hm30_pairs.json, `parity` checks against them)
        out["cpu_baseline"] = w.cpu_sample(by_index)
    w.close(); own.close()
    return out


def seed_stage_leg(a, ctx):
    """seeds/s where it means something: the headline workloads are dominated by long gapped extensions, so the seed half of
    BASELINE.json's metric is measured on the pure-random variant of the config-2 recipe (SURVEY 8d "pure-random pair ... to
    isolate seed/ungapped throughput"), large enough not to be launch bound.  One untimed + one timed job.  The algorithmic
    bytes are exactly SURVEY 8d's list (no sort term, packed sequences)."""
    from cactus_amd import gen, miblast
    import numpy as np
    n = a.seed_leg
    pm = miblast.params_from_args(DEFAULT_ARGS.split())
    rng = np.random.default_rng(43)
    t, q = gen.random_sequence(n, rng), gen.random_sequence(n, rng)
    T = ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("id=randT|chr1", t)]))
    Q = ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("id=randQ|chr1", q)]))
    ctx.align(T, Q, pm, details=False)
    miblast.drop_derived()                                  # (the timed job builds its seed table, '-' strand and packed strands like the first one did)
    t0 = time.perf_counter()
    r = ctx.align(T, Q, pm, details=False)
    dt = time.perf_counter() - t0
    s = r.stats
    T.close(); Q.close()
    hits, look, cols = s["seed_hits"], s["seed_lookups"], s["ungapped_cols"]
    # SURVEY 8d, read + write terms: index build 1*T + 0.375*T + 0.375*T + 4*(T/step) + 2*64 MiB; seed search 0.375*Q*S + 8 B per
    # lookup + 4 B per hit read, 8 B per hit written; ungapped 8 B per hit + 2 * 0.25 B per column, 24 B per HSP written
    algo = (1.75 * n + 4.0 * n + 2 * 64 * 2**20) + (0.375 * n * 2 + 8.0 * look + 12.0 * hits) + (8.0 * hits + 0.5 * cols + 24.0 * s["hsps"])
    t_stage = max(1e-9, s["t_seed"] + s["t_index"])
    return {"workload": f"{n} x {n} pure-random pair, seed 43, lastz default option set", "seeds_per_s": hits / dt, "seed_lookups_per_s": look / dt,
            "seconds": dt, "seed_hits": hits, "t_seed_s": s["t_seed"], "t_index_s": s["t_index"],
            "kernel_ms": {"ungapped": s["t_ungapped_kernel_ms"], "sort": s["t_sort_ms"], "seed_fill": s["t_seedfill_ms"]},
            "algorithmic_bytes": algo, "algorithmic_GBps": algo / t_stage / 1e9, "hbm_peak_GBps": HBM_PEAK_GBS, "frac": algo / t_stage / 1e9 / HBM_PEAK_GBS,
            "bytes_note": "SURVEY 8d terms only: no sort passes, 0.375 B/base packed sequences, 0.5 B per ungapped column",
            "chance_alignments": s["alignments"]}


def chain_stage_leg(a, ctx):
    """The step after the blast phase (local_alignment.py:660-727): one chain | tile | trim | filter | chain |
    filter job on a synthetic PAF (both orientations, as chain_alignments feeds it), text in and text out, with the HIP-event
    times of its kernels; beside it the oracle's six piped processes on the same text (1 core each, as paffy runs)."""
    from cactus_amd import gen, mipaf
    text = gen.random_paf(1234, n_series=a.chain_leg, per_series=(20, 60), n_q=2, n_t=2, contig_len=20_000_000, noise=10 * a.chain_leg, ragged=False)
    text += mipaf.PafSet.from_text(text).invert().text()
    n = len(text.splitlines())

    def job():
        t0 = time.perf_counter()
        s = mipaf.PafSet.from_text(text)
        t1 = time.perf_counter()
        s.chain_tile_trim_filter(ctx, None, "0.2", 10000)
        t2 = time.perf_counter()
        out = s.text()
        t3 = time.perf_counter()
        st = s.stats
        s.close()
        return out, st, (t1 - t0, t2 - t1, t3 - t2)

    job()
    best = None
    for _ in range(3):
        out, st, (tp, tj, tw) = job()
        if best is None or tp + tj + tw < sum(best[2]):
            best = (out, st, (tp, tj, tw))
    out, st, (tp, tj, tw) = best
    oracle = os.path.join(ROOT, "oracle", "oracle_paffy")
    chain = f"{oracle} chain --maxGapLength 1000000 --chainGapOpen 5000 --chainGapExtend 1 --trimFraction 1.0"
    cmd = f"{chain} | {oracle} tile | {oracle} trim --trimIdentity 0.2 | {oracle} filter --maxTileLevel 1 | {chain} | {oracle} filter --minChainScore 10000"
    t0 = time.perf_counter()
    p = subprocess.run(["bash", "-o", "pipefail", "-c", cmd], input=text.encode(), capture_output=True)
    t_cpu = time.perf_counter() - t0
    return {"workload": f"{n} PAF records ({len(text) / 1e6:.1f} MB; {a.chain_leg} syntenic series + noise, both orientations), Cactus's chaining parameters",
            "records_per_s": n / (tp + tj + tw), "seconds": tp + tj + tw,
            "seconds_parse_job_write": [tp, tj, tw], "records_out": len(out.splitlines()),
            "kernel_ms": {"sorts": st["t_sort_ms"], "chain_dp": st["t_chain_dp_ms"], "tile": st["t_tile_ms"], "trim": st["t_trim_ms"]},
            "cpu_baseline": {"seconds": t_cpu, "records_per_s": n / t_cpu, "cores": 6, "kind": "port",
                             "sample": "the same text through the oracle's six piped processes", "same_bytes": p.returncode == 0 and p.stdout.decode() == out}}


def cpu_baseline(kept_calls, dp_cells_gpu, describe, node=True):
    """CPU oracle (kind "port": the in-repo C restatement) on the lastz calls of the last timed step -- the same FASTA bytes and
    option strings, call by call -- timed on this box's host cores, every PAF compared with the GPU's.  Two figures: one core,
    the calls one after the other (1 thread like a lastz job; this run makes the byte diff), and SURVEY 8d's CPU throughput model
    -- min(calls, cores) single-threaded oracle processes at a time, each pinned to a core with taskset, as Toil schedules lastz
    jobs on a node (defaultCpu="1", cactus_progressive_config.xml:4)."""
    from cactus_amd import miblast
    from oracle import olz
    cells = hits = 0
    same = True
    n_diff = 0
    t0 = time.perf_counter()
    for tf, qf, opts, paf in kept_calls:
        pm = miblast.params_from_args(opts.split())
        o = olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}), details=False)
        cells += o["counters"]["dp_cells"]; hits += o["counters"]["seed_hits"]
        if o["paf"] != paf:
            same = False
            n_diff += 1
    dt = time.perf_counter() - t0
    out = {"value": cells / dt / 1e9, "unit": "Gcell/s", "cores": 1, "kind": "port",
           "sample": f"all {len(kept_calls)} lastz calls of one step of: {describe} ({dt:.1f} s of CPU work, one after the other on one core)",
           "seeds_per_s": hits / dt, "seconds": dt, "calls": len(kept_calls),
           "same_bytes": same, "calls_differing": n_diff,
           "same_dp_cells": None if dp_cells_gpu is None else int(cells) == int(round(dp_cells_gpu))}
    if not node:
        return out
    try:
        out["node"] = cpu_baseline_concurrent(kept_calls, cells, hits)
    except Exception as e:                                   # noqa: BLE001  (the figure above stands on its own)
        out["node"] = {"error": str(e)}
    return out


def cpu_baseline_concurrent(kept_calls, cells, hits, order=None):
    """min(calls, cores) oracle processes at a time (oracle/oracle_lastz, the lastz-argv front end of the same restatement), one core
    each: wall time of the whole list, longest calls first (`order`: estimated seconds per call, else by size).  cells / hits None:
    taken from the processes' own counters (--counters), with the CPU seconds of every call."""
    import shutil
    import tempfile
    exe = os.path.join(ROOT, "oracle", "oracle_lastz")
    if not os.path.exists(exe):
        raise RuntimeError("oracle/oracle_lastz is not built")
    cores = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    width = max(1, min(len(kept_calls), len(cores)))
    work = tempfile.mkdtemp(prefix="miblast_cpu_")
    try:
        jobs = []
        for k, (tf, qf, opts, paf) in enumerate(kept_calls):
            tp, qp = os.path.join(work, f"{k}.t.fa"), os.path.join(work, f"{k}.q.fa")
            open(tp, "wb").write(tf); open(qp, "wb").write(qf)
            jobs.append((order[k] if order else len(tf) * len(qf), k, [exe, tp + "[multiple][nameparse=darkspace]", qp + "[nameparse=darkspace]", "--format=paf:wfmash", "--counters"] + opts.split(), paf))
        jobs.sort(key=lambda x: (-x[0], x[1]))
        taskset = shutil.which("taskset")
        free, running, same = list(cores[:width]), {}, True
        per_call = {}
        t0 = time.perf_counter()
        pending = list(jobs)
        while pending or running:
            while pending and free:
                _, k, cmd, paf = pending.pop(0)
                core = free.pop(0)
                outp, errp = os.path.join(work, f"{k}.paf"), os.path.join(work, f"{k}.err")            # (to files: a finished job never waits on a full pipe)
                p = subprocess.Popen(([taskset, "-c", str(core)] if taskset else []) + cmd, stdout=open(outp, "wb"), stderr=open(errp, "wb"))
                running[p.pid] = (p, core, paf, outp, errp, k)
            pid, status = os.wait()                              # whichever job ends first gives its core to the next one
            if pid not in running:
                continue
            p, core, paf, outp, errp, k = running.pop(pid)
            p.returncode = os.waitstatus_to_exitcode(status)
            same = same and p.returncode == 0 and open(outp, "rb").read() == paf
            try:
                per_call[k] = json.loads(open(errp).read().strip().splitlines()[-1])
            except Exception:                                    # noqa: BLE001
                per_call[k] = {}
            free.append(core)
        wall = time.perf_counter() - t0
    finally:
        shutil.rmtree(work, ignore_errors=True)
    if cells is None:
        cells, hits = sum(c.get("dp_cells", 0) for c in per_call.values()), sum(c.get("seed_hits", 0) for c in per_call.values())
    cpu_s = [round(per_call.get(k, {}).get("t_total", 0.0), 3) for k in range(len(kept_calls))]
    return {"value": cells / wall / 1e9, "unit": "Gcell/s", "seeds_per_s": hits / wall, "cores": width, "cores_available": len(cores), "seconds_wall": wall,
            "cpu_seconds_per_call": cpu_s, "cpu_seconds": sum(cpu_s), "dp_cells": int(cells), "seed_hits": int(hits),
            "pinned_with_taskset": bool(taskset), "same_bytes": same,
            "model": "SURVEY 8d: min(calls, cores) single-threaded processes at a time, one lastz job per core"}


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    run_rank(a)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- blast-phase throughput of the MI355X-native lastz replacement.

Workload (BASELINE.json configs[1], SURVEY.md 8d "Config 2"): one 1 Mb x 1 Mb synthetic chunk pair
per GPU (ancestor + mutated copy: 15 % substitutions, 1 % indels, one inversion, one replaced
segment, 20 % soft-masked, two N runs), lastz "default" parameter set of
cactus_progressive_config.xml:136.  A step = one full blast job (index build, seed search both
strands, ungapped extension, gapped Y-drop extension, PAF) with both sequence sets already
resident in HBM.  N>1: one process per GPU, each with its own copy of the chunk pair (weak scaling with the
per-GPU work held exactly constant, no data-path collective); the only exchange is the gather of the final PAF bytes to rank 0 over RCCL,
inside the timed region.

Prints ONE JSON line on rank 0.  metric value = dp_cells (the oracle-defined counter: cells of
committed anchors only, speculative work NOT counted) per second, whole job.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_ARGS = "--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=100000"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=1_000_000, help="bases per chunk (config 2: 1 Mb)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--random-pair", action="store_true", help="pure-random pair (seed/ungapped isolation)")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="chunk size for the CPU-oracle baseline leg (0 = skip); repeated until ~10 s of CPU work")
    ap.add_argument("--chain-leg", type=int, default=400,
                    help="syntenic series in the synthetic PAF of the chaining-stage leg (0 = skip); reported under chain_stage, never in value")
    ap.add_argument("--lastz-args", default=DEFAULT_ARGS)
    ap.add_argument("--pairs-per-gpu", type=int, default=1,
                    help="chunk pairs per GPU per step, aligned in ONE batched call (merged gapped launches).  The default 1 is "
                         "BASELINE.json configs[1]; larger values are the many-pairs regime of configs[2..4]")
    ap.add_argument("--seed-leg", type=int, default=8_000_000,
                    help="chunk size of the extra seed-stage leg on a pure-random pair (0 = skip); reported under seed_stage, never in value")
    a = ap.parse_args()

    import torch
    from cactus_amd import gen, miblast

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MIBLAST_BENCH_SINGLE_DEVICE"):      # test hook: several ranks on one GPU (use with MIBLAST_BENCH_BACKEND=gloo)
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    coll_backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        coll_backend = os.environ.get("MIBLAST_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        dist.init_process_group(backend=coll_backend, device_id=torch.device("cuda", local_rank) if coll_backend == "nccl" else None)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libmiblast has no CPU path")
    torch.cuda.set_device(local_rank)

    pm = miblast.params_from_args(a.lastz_args.split())
    from cactus_amd.multigpu import share_host_cores
    host_threads = share_host_cores(int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))) if world > 1 else miblast.set_host_threads(0)
    ctx = miblast.Context(local_rank)
    # each rank aligns its own chunk pair (chunk-pair sharding)
    P = max(1, a.pairs_per_gpu)
    sets = []
    for k in range(P):
        # weak scaling with per-GPU work held exactly constant: every rank gets the same P chunk pairs (same seeds), only
        # the sequence names differ; different seeds would turn the max-over-ranks time into a lottery over the longest DP
        t, q = gen.make_pair(a.size, a.seed + k, homologous=not a.random_pair)
        sets.append((ctx.seqset_from_fasta_bytes(gen.fasta_bytes([(f"id=simT{rank}_{k}|chr1", t)])),
                     ctx.seqset_from_fasta_bytes(gen.fasta_bytes([(f"id=simQ{rank}_{k}|chr1", q)]))))
    T, Q = sets[0]

    from cactus_amd.multigpu import gather_bytes
    coll_dev = torch.device("cuda", local_rank) if coll_backend != "gloo" else torch.device("cpu")

    def gather_paf(paf: bytes):
        """final hit list -> rank 0 (RCCL over xGMI)"""
        return gather_bytes(paf, dist, rank, world, coll_dev)

    def step():
        if P == 1:
            r = ctx.align(T, Q, pm, details=False)
            pafs = gather_paf(r.paf)
            return r, pafs
        rs = ctx.align_pairs(sets, pm)
        pafs = gather_paf(b"".join(x.paf for x in rs))
        # per-pair counters add up; launch-level figures (shared by the pairs in flight) are taken once
        shared = ("t_gapped", "gapped_rounds", "dp_sides_run", "dp_cells_run", "dp_rows_run", "t_dp_kernel_ms", "dp_kernel_launches",
                  "relay_accepted", "relay_rejected", "t_traceback_ms", "t_merge_ms")
        merged = {k: (rs[0].stats[k] if k in shared else sum(x.stats[k] for x in rs)) for k in rs[0].stats}
        rs[0].stats.update(merged)
        return rs[0], pafs

    for _ in range(a.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agg = None
    for _ in range(a.steps):
        r, pafs = step()
        if agg is None:
            agg = {k: 0 for k in r.stats}
        for k, v in r.stats.items():
            agg[k] += v
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    keys = ["dp_cells", "seed_hits", "seed_lookups", "ungapped_cols", "alignments", "dp_cells_run", "dp_rows_run", "t_dp_kernel_ms",
            "dp_kernel_launches", "t_index", "t_seed", "t_gapped", "t_total", "t_ungapped_kernel_ms", "t_sort_ms", "t_seedfill_ms",
            "relay_accepted", "relay_rejected", "t_traceback_ms", "t_merge_ms", "dp_sides_run"]
    vec = torch.tensor([float(agg[k]) for k in keys] + [elapsed], dtype=torch.float64, device=coll_dev)
    if dist is not None:
        tmax = vec[-1:].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        elapsed = float(tmax.item())
    tot = {k: float(v) for k, v in zip(keys, vec[:-1].tolist())}

    if rank == 0:
        launches = max(1.0, tot["dp_kernel_launches"])
        cells_per_launch = tot["dp_cells_run"] / launches
        rows_per_launch = tot["dp_rows_run"] / launches
        # Algorithmic HBM bytes of one k_ydrop launch (SURVEY 8d "Gapped"): one trace byte
        # written per evaluated cell + one 16-byte {offset, LY} record per row + ~2 sequence bytes read per row
        # (one query base, ~one new target column).  The launch duration is measured inside libmiblast with HIP
        # events on its own stream.
        algo_bytes = cells_per_launch * 1.0 + rows_per_launch * (16.0 + 2.0)
        dp_ms = tot["t_dp_kernel_ms"] / launches
        achieved = algo_bytes / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0
        traffic, traffic_note = None, "no committed PMC summary"
        pmc_path = os.path.join(ROOT, "profiles", "r01_hbm_traffic_pmc.json")
        if os.path.exists(pmc_path) and not a.random_pair and a.size == 1_000_000 and a.lastz_args == DEFAULT_ARGS:
            ks = json.load(open(pmc_path))["kernels"]
            dp = [v for name, v in ks.items() if "k_ydrop" in name]         # every DP kernel variant (one wave / four waves per piece)
            calls = sum(v["calls"] for v in dp)
            if calls:
                pmc_bytes = sum(v["calls"] * (v["fetch_bytes_corrected_per_call"] + v["write_size_bytes_per_call"]) for v in dp) / calls
                traffic = pmc_bytes / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else None
                traffic_note = ("PMC bytes per DP launch (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes, "
                                "profiles/r01_hbm_traffic_pmc.json: %.1f MB averaged over the %d launches profiled) / this run's average launch duration"
                                % (pmc_bytes / 1e6, calls))
        out = {
            "metric": "gapped X-drop Gcell/s (blast phase, whole job)",
            "value": tot["dp_cells"] / elapsed / 1e9,
            "unit": "Gcell/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{a.size} x {a.size} synthetic chunk pair per GPU (SURVEY 8d config 2"
                                   f"{', pure-random variant' if a.random_pair else ''}), seed {a.seed}+pair index, identical on every rank",
                       "lastz_args": a.lastz_args, "chunk_pairs": world * P, "pairs_per_gpu": P,
                       "sharding": "chunk pairs sharded over GPUs (batched per GPU when pairs_per_gpu > 1), gather of PAF to rank 0",
                       "collective_backend": coll_backend, "host_threads_per_rank": host_threads},
            "seeds_per_s": tot["seed_hits"] / elapsed,
            "seed_lookups_per_s": tot["seed_lookups"] / elapsed,
            "stage_seconds_per_step": {k: tot[k] / a.steps / world for k in ("t_index", "t_seed", "t_gapped", "t_total")},
            "stage_kernel_ms_per_step": {"ydrop": tot["t_dp_kernel_ms"] / a.steps / world, "ungapped": tot["t_ungapped_kernel_ms"] / a.steps / world,
                                         "sort": tot["t_sort_ms"] / a.steps / world, "seed_fill": tot["t_seedfill_ms"] / a.steps / world},
            "gapped_gcells_per_s_kernel": tot["dp_cells_run"] / max(1e-9, tot["t_dp_kernel_ms"] * 1e-3 / world) / 1e9,
            "speculation_factor": tot["dp_cells_run"] / max(1.0, tot["dp_cells"]),
            "alignments_per_step": tot["alignments"] / a.steps,
            "relay": {"pieces_per_step": tot["dp_sides_run"] / a.steps / world, "handovers_accepted_per_step": tot["relay_accepted"] / a.steps / world,
                      "handovers_rejected_per_step": tot["relay_rejected"] / a.steps / world,
                      "traceback_ms_per_step": tot["t_traceback_ms"] / a.steps / world, "merge_ms_per_step": tot["t_merge_ms"] / a.steps / world,
                      "note": "long one-sided DPs run as concurrently evaluated pieces with verified hand-overs (DESIGN.md section 5)"},
            "roofline": {"bound": "hbm", "kernel": "k_ydrop2 (one-sided Y-drop DP, one wave per piece; k_ydrop1 / k_ydrop for wider windows)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "launch_ms": dp_ms, "launches_per_step": launches / a.steps / world,
                         "traffic_note": traffic_note,
                         "note": "a launch is bound by latency per row (~2100 clocks for a lone wave, ~3500 with thousands resident) and instruction issue, not HBM; the roofline fraction rises with the number of pieces in flight (SURVEY 8d caveat, DESIGN.md section 5)"},
        }
        try:
            # the DP is VALU / issue bound, so its cells/s are also put against the int32 VALU peak: evaluated cells
            # (speculative ones included) x the ~10 integer operations the recurrence needs per cell (3 max, 3 add, score lookup,
            # 2 compares for the y-drop test, trace code) / (CUs x 4 SIMDs x 16 lanes x clock)
            prop = torch.cuda.get_device_properties(local_rank)
            clock_hz = float(getattr(prop, "clock_rate", 2_400_000)) * 1e3
            peak_ops = prop.multi_processor_count * 4 * 16 * clock_hz
            cells_per_s = tot["dp_cells_run"] / max(1e-12, tot["t_dp_kernel_ms"] * 1e-3)
            out["roofline"]["valu"] = {"cells_evaluated_per_s": cells_per_s, "min_int_ops_per_cell": 10, "peak_lane_ops_per_s": peak_ops,
                                       "frac": cells_per_s * 10 / peak_ops, "cus": prop.multi_processor_count, "clock_ghz": clock_hz / 1e9,
                                       "note": "k_ydrop2 issues ~330 wave instructions per DP row of up to 256 column slots (scans, pruning, trace codes, window bookkeeping included); "
                                               "with 2-3 resident waves per SIMD a row takes ~3500 clocks, i.e. the SIMDs that hold pieces issue close to one instruction per 4 clocks"}
        except Exception as e:                               # noqa: BLE001  (never lose the line over a device-property quirk)
            out["roofline"]["valu"] = {"error": str(e)}
        if a.seed_leg > 0 and not a.random_pair:
            out["seed_stage"] = seed_stage_leg(a, pm, ctx)
        if a.chain_leg > 0:
            out["chain_stage"] = chain_stage_leg(a, ctx)
        if a.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(a, pm)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def seed_stage_leg(a, pm, ctx):
    """seeds/s where it means something: the benchmark pair is dominated by a few long gapped extensions, so the seed
    half of BASELINE.json's metric is measured on the pure-random variant of the recipe (SURVEY 8d "pure-random pair ...
    to isolate seed/ungapped throughput"), large enough not to be launch bound.  One untimed + one timed job."""
    from cactus_amd import gen
    import numpy as np
    n = a.seed_leg
    rng = np.random.default_rng(43)
    t, q = gen.random_sequence(n, rng), gen.random_sequence(n, rng)
    T = ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("id=randT|chr1", t)]))
    Q = ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("id=randQ|chr1", q)]))
    ctx.align(T, Q, pm, details=False)
    t0 = time.perf_counter()
    r = ctx.align(T, Q, pm, details=False)
    dt = time.perf_counter() - t0
    s = r.stats
    T.close(); Q.close()
    hits, look = s["seed_hits"], s["seed_lookups"]
    # algorithmic HBM bytes of the seed stage: index build 1*T + 4*T + 2*64 MiB; search 1*Q*S + 8 B per lookup
    # + 4 B per hit read, 8 B per hit written; sort ~ 6 passes x 16 B per hit; ungapped 8 B per hit + 2 B per column
    algo = (5.0 * n + 2 * 64 * 2**20) + (2.0 * n + 8.0 * look + 12.0 * hits) + 96.0 * hits + (8.0 * hits + 2.0 * s["ungapped_cols"])
    return {"workload": f"{n} x {n} pure-random pair, seed 43, same lastz options", "seeds_per_s": hits / dt, "seed_lookups_per_s": look / dt,
            "seconds": dt, "seed_hits": hits, "t_seed_s": s["t_seed"], "t_index_s": s["t_index"],
            "kernel_ms": {"ungapped": s["t_ungapped_kernel_ms"], "sort": s["t_sort_ms"], "seed_fill": s["t_seedfill_ms"]},
            "algorithmic_GBps": algo / max(1e-9, s["t_seed"] + s["t_index"]) / 1e9, "hbm_peak_GBps": HBM_PEAK_GBS,
            "chance_alignments": s["alignments"], "gapped_gcells_per_s_kernel": s["dp_cells_run"] / max(1e-9, s["t_dp_kernel_ms"] * 1e-3) / 1e9}


def chain_stage_leg(a, ctx):
    """The step after the blast phase (local_alignment.py:660-727): one chain | tile | trim | filter | chain |
    filter job on a synthetic PAF (both orientations, as chain_alignments feeds it), text in and text out, with the HIP-event
    times of its kernels; beside it the oracle's six piped processes on the same text (1 core each, as paffy runs)."""
    import subprocess
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


def cpu_baseline(a, pm):
    """CPU oracle (kind "port": the in-repo C restatement, 1 thread like a lastz job) on a bounded
    sample of the same recipe, timed on this box's host cores."""
    from cactus_amd import gen
    from oracle import olz
    n = min(a.size, a.cpu_sample)
    t, q = gen.make_pair(n, a.seed, homologous=not a.random_pair)
    tf, qf = gen.fasta_bytes([("id=simT|chr1", t)]), gen.fasta_bytes([("id=simQ|chr1", q)])
    po = olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_})
    cells = hits = 0
    reps = 0
    t0 = time.perf_counter()
    while True:
        o = olz.align(tf, qf, po, details=False)
        c = o["counters"]
        cells += c["dp_cells"]; hits += c["seed_hits"]; reps += 1
        dt = time.perf_counter() - t0
        if dt >= 10.0 or reps >= 8:
            break
    return {"value": cells / dt / 1e9, "unit": "Gcell/s", "cores": 1, "kind": "port",
            "sample": f"{n} x {n} pair of the same recipe, seed {a.seed}, whole job {reps} times in {dt:.1f} s",
            "seeds_per_s": hits / dt, "seconds": dt, "seconds_per_job": dt / reps,
            "stage_seconds": {k: c[k] for k in ("t_index", "t_seed", "t_gapped", "t_total")}}


if __name__ == "__main__":
    main()

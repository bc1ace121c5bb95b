"""-m gpu : the measurement harness and the one-process-per-GPU form of the sharding.  bench.py must really start N ranks for
--gpus N (the driver's scaling runs depend on it), attach the oracle's bytes to the benched workload, and the gloo/RCCL gather
must carry the product's output, not a stand-in's.  A one-GPU box hosts both ranks ($MIBLAST_BENCH_SINGLE_DEVICE, gloo)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--ancestor", "60000", "--steps", "1", "--warmup", "1", "--seed-leg", "0", "--chain-leg", "0", "--pair-leg", "0", "--batch-leg", "0", "--primates-leg", "0", "--chunk-legs", "0"]


def _bench(args, **env):
    """Runs bench.py; checks the printed line (ONE line, < 8 000 bytes, the contract's keys, the figures of the full object) and returns the
    FULL object of the file the line names."""
    import tempfile
    fd, full = tempfile.mkstemp(suffix=".json", prefix="bench_full_")
    os.close(fd)
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--full-out", full], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, **env), cwd=ROOT)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout
        assert len(lines[0].encode()) < 8000, len(lines[0])
        line = json.loads(lines[0])
        out = json.load(open(full))
    finally:
        os.unlink(full)
    assert line["full"] == full
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == out[k], k
    for k in ("value", "ms_per_step"):
        assert abs(line[k] - out[k]) <= 1e-5 * abs(out[k]), k
    assert line["config"]["workload"][:100] == out["config"]["workload"][:100] and "OUTSIDE the step" in line["config"]["timed_region"]
    r = line["roofline"]
    assert r["kernel"] == "k_ydrop2" and r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 * r["frac"] + 1e-12
    if "cpu_baseline" in out:
        assert line["cpu_baseline"]["same_bytes"] == out["cpu_baseline"]["same_bytes"] and line["cpu_baseline"]["cores"] == out["cpu_baseline"]["cores"]
    return out


def test_bench_line_carries_the_oracle_diff_of_the_benched_workload():
    out = _bench(SMALL)
    assert out["n_gpus"] == 1 and out["unit"] == "Gcell/s" and out["value"] > 0
    assert "evolverMammals" in out["config"]["workload"] and "configs[2]" in out["config"]["workload"]
    cb = out["cpu_baseline"]
    assert cb["same_bytes"] is True and cb["calls_differing"] == 0 and cb["same_dp_cells"] is True and cb["calls"] >= 14
    r = out["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0 < r["valu"]["frac"] < 1


def test_bench_gpus_2_really_runs_two_ranks():
    one = _bench(SMALL + ["--cpu-sample", "0"])
    two = _bench(SMALL + ["--cpu-sample", "0", "--gpus", "2"], MIBLAST_BENCH_SINGLE_DEVICE="1", MIBLAST_BENCH_BACKEND="gloo")
    assert two["n_gpus"] == 2 and two["config"]["collective_backend"] == "gloo"
    # both ranks' PAF reached rank 0 inside the timed region: twice the bytes, twice the work
    assert two["config"]["paf_bytes_gathered_per_step"] == 2 * one["config"]["paf_bytes_gathered_per_step"] > 0
    assert two["dp_cells_per_step"] == one["dp_cells_per_step"]              # per rank and step
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL], capture_output=True, text=True,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr                   # a launcher that started the wrong number of ranks is an error


def test_primates_leg_and_the_node_wide_cpu_baseline():
    out = _bench([x if x != "0" or prev != "--primates-leg" else "1" for prev, x in zip([""] + SMALL, SMALL)])
    pr = out["primates"]
    assert "evolverPrimates" in pr["workload"] and "configs[0]" in pr["workload"] and pr["calls"] == 9
    assert pr["cpu_baseline"]["same_bytes"] is True and pr["cpu_baseline"]["same_dp_cells"] is True
    node = out["cpu_baseline"]["node"]
    assert node["same_bytes"] is True and node["cores"] >= 1 and node["value"] > 0 and "min(calls, cores)" in node["model"]


def test_chr20_workload_shards_one_genome_pair_and_the_bytes_do_not_depend_on_the_ranks(olz):
    """BASELINE configs[3] at a two-hundredth of the size: the chunk pairs of ONE genome pair dealt over 1, 2 and 3 ranks (all on
    this box's GPU, gloo) -- strong scaling, the assembled PAF identical for every world size and equal to the oracle's per-pair
    outputs strung together in chunk-pair order."""
    import hashlib
    sys.path.insert(0, ROOT)
    from cactus_amd import gen, miblast
    args = ["--workload", "chr20", "--chr20-bases", "300000", "--chr20-chunk", "120000", "--steps", "1", "--warmup", "1", "--seed-leg", "0", "--chain-leg", "0",
            "--batch-leg", "0", "--cpu-sample", "0"]
    one = _bench(args)
    assert one["scaling"] == "strong" and "configs[3]" in one["config"]["workload"] and one["config"]["units_per_rank"] == [9] and one["config"]["work_unit"] == "chunk pair"
    t, q = gen.make_pair(300000, 3001, sub_rate=0.013, indel_rate=0.002, mask_frac=0.5)
    chunks = lambda name, seq: [gen.fasta_bytes([(f"{name}|{len(seq)}|{s0}", seq[s0:s0 + 130000])]) for s0 in range(0, len(seq), 120000)]      # noqa: E731
    pm = miblast.params_from_args("--step=2 --ambiguous=iupac,100,100 --ydrop=3000 --notransition --queryhspbest=100000".split())
    po = olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_})
    want = b"".join(olz.align(tf, qf, po, details=False)["paf"] for tf in chunks("id=simT|chr20", t) for qf in chunks("id=simQ|chr20", q))
    assert one["config"]["paf_md5"] == hashlib.md5(want).hexdigest() and one["config"]["paf_bytes"] == len(want) > 1000
    for n in (2, 3):
        many = _bench(args + ["--gpus", str(n)], MIBLAST_BENCH_SINGLE_DEVICE="1", MIBLAST_BENCH_BACKEND="gloo")
        # (fewer than four chunk pairs per GPU -- nine over three ranks -- and the units are the 18 (chunk pair, query strand) halves)
        split = 9 < 4 * n
        assert many["n_gpus"] == n and many["scaling"] == "strong" and many["config"]["work_unit"] == ("(chunk pair, query strand)" if split else "chunk pair")
        assert sum(many["config"]["units_per_rank"]) == (18 if split else 9) and len(many["config"]["units_per_rank"]) == n
        assert many["config"]["paf_md5"] == one["config"]["paf_md5"]
        assert many["dp_cells_per_step"] == one["dp_cells_per_step"]                     # the whole job's work, whoever did it


def _rank_main(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cases import CASES, DEFAULT
    from cactus_amd import miblast
    from cactus_amd.multigpu import blast_pairs_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = miblast.Context(0)
    pm = miblast.params_from_args(DEFAULT)
    pairs = [(c[1], c[2]) for c in CASES if c[0] in ("homolog_20k_default", "random_50k", "multi_contig_ragged", "tandem_repeats", "empty_query", "revcomp_query")]

    def align(pair):
        T, Q = ctx.seqset_from_fasta_bytes(pair[0]), ctx.seqset_from_fasta_bytes(pair[1])
        try:
            return ctx.align(T, Q, pm, details=False).paf
        finally:
            T.close(); Q.close()

    out = blast_pairs_sharded(pairs, [float(len(a) * len(b)) for a, b in pairs], align, dist if world > 1 else None, rank, world, torch.device("cpu"))
    if rank == 0:
        open(out_path, "wb").write(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_of_the_real_library_gather_the_oracle_bytes(olz, tmp_path):
    """SURVEY 8e determinism with the PRODUCT in the loop: chunk pairs sharded over two ranks (both on this box's GPU), PAF
    gathered to rank 0 in pair order == the single-rank run == the oracle."""
    import torch.multiprocessing as mp
    from cases import CASES, DEFAULT
    from cactus_amd import miblast
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    outs = {}
    for world in (1, 2):
        path = str(tmp_path / f"w{world}.paf")
        procs = [ctx.Process(target=_rank_main, args=(r, world, port + world, path)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=600)
            assert p.exitcode == 0
        outs[world] = open(path, "rb").read()
    pm = miblast.params_from_args(DEFAULT)
    po = olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_})
    want = b"".join(olz.align(c[1], c[2], po, details=False)["paf"]
                    for c in CASES if c[0] in ("homolog_20k_default", "random_50k", "multi_contig_ragged", "tandem_repeats", "empty_query", "revcomp_query"))
    assert outs[1] == outs[2] == want and want.count(b"\n") > 5


def test_chunk_scale_legs_are_sharded_over_the_ranks_and_every_pair_equals_its_oracle_digest():
    """The default line's chr20 (BASELINE configs[3]) and hm (configs[4] stand-in) legs at N = 2 (two ranks on this box's GPU, gloo):
    the chunk pairs of each genome pair dealt over the ranks, framed PAFs gathered to rank 0, and EVERY chunk pair's PAF equal to
    the CPU oracle's committed digest (tests/golden/chr20_pairs.json, hm_pairs.json) -- the bytes do not depend on the number of ranks."""
    args = [x if x != "0" or prev != "--chunk-legs" else "1" for prev, x in zip([""] + SMALL, SMALL)] + ["--cpu-sample", "0", "--gpus", "2"]
    out = _bench(args, MIBLAST_BENCH_SINGLE_DEVICE="1", MIBLAST_BENCH_BACKEND="gloo")
    for which, n_pairs in (("chr20", 9), ("hm", 42)):
        leg = out[which]
        assert leg["n_gpus"] == 2 and leg["scaling"] == "strong" and leg["chunk_pairs"] == n_pairs
        assert leg["work_unit"] == "chunk pair" and sum(leg["units_per_rank"]) == n_pairs      # (nine pairs over two ranks: whole pairs; the halves are dealt below four pairs per GPU)
        assert leg["parity"]["same_bytes"] is True and leg["parity"]["pairs_checked"] == n_pairs and leg["parity"]["pairs_differing"] == 0
        assert leg["dp_cells_per_step"] == leg["parity"]["oracle_dp_cells"] and leg["seed_hits_per_step"] == leg["parity"]["oracle_seed_hits"]
        assert 0 < leg["hbm_read"]["frac"] < 1


def test_full_size_chr20_dealt_as_strand_halves_equals_the_digests():
    """BASELINE configs[3] at full size with the (chunk pair, query strand) units forced on one GPU: 18 half-pair jobs in two batched
    calls, the halves put together on the host -- every chunk pair's PAF still equals the CPU oracle's digest of the WHOLE pair."""
    out = _bench(["--workload", "chr20", "--split-strands", "1", "--steps", "1", "--warmup", "0", "--seed-leg", "0", "--chain-leg", "0", "--batch-leg", "0", "--cpu-sample", "0"])
    assert out["config"]["work_unit"] == "(chunk pair, query strand)" and out["config"]["units_per_rank"] == [18]
    assert out["parity"]["same_bytes"] is True and out["parity"]["pairs_checked"] == 9
    assert out["dp_cells_per_step"] == out["parity"]["oracle_dp_cells"] and out["seed_hits_per_step"] == out["parity"]["oracle_seed_hits"]


def test_human_mouse_stand_in_at_the_reference_chunk_size_equals_the_digests():
    """BASELINE configs[4] at the reference's own chunk size (cactus_progressive_config.xml:90 chunkSize 30 Mb): the stand-in genome pair cut
    into 2 chunk pairs, one of them 32.6 Mb x 32.0 Mb under the DEFAULT option set (step 1, 13 word variants: 854 million seed hits, several
    q batches per strand) -- both pairs' PAFs equal the CPU oracle's digests (tests/golden/hm30_pairs.json: one oracle run of 610 s), hits and
    cells counted as the oracle counts them."""
    out = _bench(["--workload", "hm30", "--steps", "1", "--warmup", "0", "--seed-leg", "0", "--chain-leg", "0", "--batch-leg", "0", "--cpu-sample", "0"])
    assert "chunkSize 30000000" in out["config"]["workload"] and out["config"]["units_per_rank"] == [2]
    assert out["parity"]["same_bytes"] is True and out["parity"]["pairs_checked"] == 2 and out["parity"]["pairs_differing"] == 0
    assert out["dp_cells_per_step"] == out["parity"]["oracle_dp_cells"] and out["seed_hits_per_step"] == out["parity"]["oracle_seed_hits"] > 9e8

"""-m gpu : the parity tests proper.  Every case goes through the C ABI of libmiblast.so on a real
MI355X and is compared with the CPU oracle on the same bytes -- bit-exact: PAF text, HSP records,
alignment records, run-length ops and every oracle-defined counter (integer scoring)."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from cases import CASES, CASE_IDS

pytestmark = pytest.mark.gpu

COUNTERS = ["seed_lookups", "seed_hits", "hits_extended", "ungapped_cols", "hsps_pre_entropy", "hsps", "anchors",
            "anchors_skipped", "dp_sides", "dp_cells", "dp_rows", "alignments"]


def _params(args):
    from cactus_amd import miblast
    return miblast.params_from_args(args)


def _oracle_params(olz, pm):
    return olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_})


@pytest.mark.parametrize("name,tf,qf,args", CASES, ids=CASE_IDS)
def test_case_matches_oracle(gpu_ctx, olz, monkeypatch, name, tf, qf, args):
    """Every case pair by pair (the path of a pair too large for the shared seed stage: dense seed table, seed search and sort per
    strand); small single pairs take the shared seed stage by default -- the test after the next one runs every case through it."""
    monkeypatch.setenv("MIBLAST_SEED_BATCHED", "0")
    pm = _params(args)
    T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
    got = gpu_ctx.align(T, Q, pm)
    want = olz.align(tf, qf, _oracle_params(olz, pm))
    assert got.paf == want["paf"]
    assert sorted(got.hsps) == sorted(want["hsps"])
    assert got.hsps == want["hsps"], "HSP list must also come out in the oracle's discovery order"
    assert got.alns == want["alns"]
    assert got.ops == want["ops"]
    for k in COUNTERS:
        assert got.stats[k] == want["counters"][k], k


@pytest.mark.parametrize("env", [
    {"MIBLAST_SEED_PACKED": "2"},                                       # seed words of target and both strands from the packed form (2 bits + mask bit), whatever the size
    {"MIBLAST_SEED_PACKED": "0"},                                       # ... from the code bytes
    {"MIBLAST_SEED_PACKED": "2", "MIBLAST_SEED_FUSED": "0"},            # strands one after the other
    {"MIBLAST_SEED_ORDERED": "0"},                                      # round 3's search: keys in arrival order, sorted as whole keys
    {"MIBLAST_DIAG_SCRAMBLE": "0"},                                     # keys sorted by the plain diagonal
    {"MIBLAST_RESIDENT_TABLES": "0"},                                   # a table per call in the context's own memory
    {"MIBLAST_SEED_PACKED": "2", "MIBLAST_HIT_CAP": "3000"},            # q batches (two-pass path, extent[] carried from batch to batch)
    {"MIBLAST_BIN_MEAN": "40"},                                         # keys grouped by diagonal through MANY bins + LDS (mb_seed_bin.h; the default mean gives these cases one bin or two)
    {"MIBLAST_BIN_MEAN": "40", "MIBLAST_SEED_FUSED": "0"},              # ... strands one after the other
    {"MIBLAST_SORT_BIN": "0"},                                          # ... and not at all: rocprim's radix sort + k_keys_unhash, as before round 5
    {"MIBLAST_SEED_PACKED": "2", "MIBLAST_UX_PACKED": "2"},             # round 6: the windows of the ungapped extension from the packed strands (k_ux_extend_pk) whatever the size ...
    {"MIBLAST_SEED_PACKED": "2", "MIBLAST_UX_PACKED": "2", "MIBLAST_UNGAPPED": "ux"},      # ... with the level-synchronous pipeline forced (small cases take the run-per-lane kernel otherwise)
    {"MIBLAST_SEED_PACKED": "2", "MIBLAST_UX_PACKED": "0", "MIBLAST_UNGAPPED": "ux"},      # ... and from the code bytes
    {"MIBLAST_SEED_PACKED": "2", "MIBLAST_HIT_CAP": "3000", "MIBLAST_UX_PACKED": "2", "MIBLAST_UNGAPPED": "ux"},      # q batches: extent[] in the keys' scrambled order, batches through the bins, packed windows
    {"MIBLAST_SEED_PACKED": "2", "MIBLAST_HIT_CAP": "3000", "MIBLAST_EXTENT_SCRAMBLE": "0", "MIBLAST_BATCH_BINS": "0"},      # ... a slot per plain diagonal, batches through the radix sort
], ids=lambda e: ",".join(f"{k[8:].lower()}={v}" for k, v in e.items()))
def test_dense_seed_path_switches_match_oracle(gpu_ctx, olz, monkeypatch, env):
    """The seed stage of a large pair (mb_seed_dense.h: packed strands, q-ordered one-pass search -- k_seed_hits + k_seed_keys --, scrambled
    diagonals, tables and '-' strands resident with their sets) on every case, with each of its switches: same bytes, HSP list in
    discovery order and counters as the oracle's.  Every case runs twice on the same resident sets, so the second call takes the
    both-strands-in-one-go path with the tables of the first."""
    monkeypatch.setenv("MIBLAST_SEED_BATCHED", "0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    binned = 0
    for name, tf, qf, args in CASES:
        pm = _params(args)
        T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
        want = olz.align(tf, qf, _oracle_params(olz, pm))
        for rep in range(2):
            got = gpu_ctx.align(T, Q, pm)
            assert got.paf == want["paf"], (name, rep)
            assert got.hsps == want["hsps"], (name, rep)
            assert got.alns == want["alns"] and got.ops == want["ops"], (name, rep)
            for k in COUNTERS:
                assert got.stats[k] == want["counters"][k], (name, rep, k)
            binned += got.stats["seed_binned"]
        T.close(); Q.close()
    if env.get("MIBLAST_SORT_BIN") == "0" or env.get("MIBLAST_SEED_ORDERED") == "0":
        assert binned == 0
    elif "MIBLAST_HIT_CAP" not in env and "MIBLAST_DIAG_SCRAMBLE" not in env:
        assert binned > 0, "no strand took the bins + LDS path"


def test_grouping_by_diagonal_in_lds_equals_the_radix_sort_on_a_large_pair(gpu_ctx, olz, monkeypatch):
    """mb_seed_bin.h at the size it is for: a 1.5 Mb pair at 3 % divergence (1.5 x 10^6 hits per strand: hundreds of bins, diagonals of real
    homology with hundreds of hits each -- long rank loops -- next to chance hits) gives the bytes, HSPs and counters of the rocprim path
    and of the oracle, with the default mean (bins of ~ 8 000 keys: the staged scatter, the large sorter), with bins for the small sorter and with 8 192 bins (the direct scatter); a pair aligned to ITSELF (one diagonal
    holds every hit of the + strand: no LDS holds that) falls back to rocprim for that strand and still agrees."""
    from cactus_amd import gen
    from cases import DEFAULT
    monkeypatch.setenv("MIBLAST_SEED_BATCHED", "0")
    pm = _params(DEFAULT)
    t, q = gen.make_pair(1_500_000, 77, sub_rate=0.03, indel_rate=0.001)
    tf, qf = gen.fasta_bytes([("id=T|c", t)]), gen.fasta_bytes([("id=Q|c", q)])
    want = olz.align(tf, qf, _oracle_params(olz, pm))
    T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
    runs = {}
    for label, env in (("radix", {"MIBLAST_SORT_BIN": "0"}), ("bins", {}), ("small bins", {"MIBLAST_BIN_MEAN": "2800"}), ("many bins", {"MIBLAST_BIN_MEAN": "300"})):
        for k in ("MIBLAST_SORT_BIN", "MIBLAST_BIN_MEAN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for rep in range(2):                                            # (the second call takes the both-strands-in-one-go path)
            runs[label] = gpu_ctx.align(T, Q, pm)
        got = runs[label]
        assert got.paf == want["paf"] and got.hsps == want["hsps"] and got.alns == want["alns"], label
        for k in COUNTERS:
            assert got.stats[k] == want["counters"][k], (label, k)
        assert (got.stats["seed_binned"] == 0) if label == "radix" else (got.stats["seed_binned"] >= 1), (label, got.stats["seed_binned"])
    T.close(); Q.close()
    monkeypatch.delenv("MIBLAST_BIN_MEAN", raising=False)
    t = gen.random_sequence(60_000, __import__("numpy").random.default_rng(5))
    tf, qf = gen.fasta_bytes([("id=S|c", t)]), gen.fasta_bytes([("id=S2|c", t)])
    want = olz.align(tf, qf, _oracle_params(olz, pm))
    T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
    for rep in range(2):
        got = gpu_ctx.align(T, Q, pm)
        assert got.paf == want["paf"] and got.hsps == want["hsps"], rep
        assert got.stats["seed_binned"] <= 1, "the + strand of a self alignment cannot have gone through LDS"
    T.close(); Q.close()


def test_strand_halves_of_a_pair_put_together_equal_the_whole(gpu_ctx, olz, monkeypatch):
    """--strand=plus / minus (miblast_params.strands, lastz's own option): each half equals the oracle's half -- PAF, HSPs, counters --
    and the two halves interleaved by query sequence in file order (cactus_amd.multigpu.merge_strand_pafs) are the whole pair's PAF:
    (chunk pair, strand) is an exact work unit below the chunk pair.  Multi-contig, busy-diagonal and plain cases, through
    the one-pair call and through a batched call."""
    from cases import DEFAULT, multi_contig, pair
    from cactus_amd.multigpu import fasta_names, merge_strand_pafs
    for tf, qf in (multi_contig(9), pair(120000, 3), pair(50000, 12, sub_rate=0.03)):
        T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
        whole = gpu_ctx.align(T, Q, _params(DEFAULT))
        halves = []
        for opt in ("--strand=plus", "--strand=minus"):
            pm = _params(DEFAULT + [opt])
            got = gpu_ctx.align(T, Q, pm)
            want = olz.align(tf, qf, _oracle_params(olz, pm))
            assert got.paf == want["paf"] and got.hsps == want["hsps"] and got.alns == want["alns"], opt
            for k in COUNTERS:
                assert got.stats[k] == want["counters"][k], (opt, k)
            assert gpu_ctx.align_pairs([(T, Q), (T, Q)], pm)[1].paf == got.paf
            halves.append(got)
        assert merge_strand_pafs(halves[0].paf, halves[1].paf, fasta_names(qf)) == whole.paf
        for k in COUNTERS:
            assert halves[0].stats[k] + halves[1].stats[k] == whole.stats[k], k
        T.close(); Q.close()


def _ref_argv():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "ref_argv.json")))["run_lastz"]


@pytest.mark.parametrize("rec", _ref_argv(), ids=lambda r: f"d{r['distance']}-gpu{r['gpu']}")
def test_command_lines_built_by_the_reference_run_on_the_front_ends(olz, tmp_path, monkeypatch, rec):
    """tests/golden/ref_argv.json holds the argv that the REFERENCE's unmodified run_lastz builds for every divergence class, CPU and
    GPU branch (written by tests/golden/make_ref_argv.py in the build container, where /root/reference exists; the same functions are
    run end to end there by tests/test_reference_jobs_cpu.py).  Here the recorded command lines run against the real bin/lastz and
    bin/run_kegalign on the MI355X, from a work directory like the job's: PAF on stdout equal to the oracle's, empty stderr."""
    import subprocess
    from cases import pair
    tf, qf = pair(40000, 77)
    (tmp_path / "A.fa").write_bytes(tf); (tmp_path / "B.fa").write_bytes(qf)
    argv = list(rec["argv"])
    env = dict(os.environ, PATH=os.path.join(ROOT, "bin") + os.pathsep + os.environ.get("PATH", ""))
    if rec["gpu"] > 1:
        env["MIBLAST_DEVICE_MAP"] = ",".join(["0"] * rec["gpu"])       # (a one-GPU box: the job's logical devices all map to it)
    p = subprocess.run(argv, cwd=str(tmp_path), env=env, capture_output=True, timeout=600)
    assert p.returncode == 0 and p.stderr == b"", p.stderr.decode()
    pm = _params([a for a in argv[4:] if a.startswith("--") and not a.startswith("--num_")])
    want = olz.align(tf, qf, _oracle_params(olz, pm), details=False)["paf"]
    assert p.stdout == want and want.count(b"\n") >= 1


@pytest.mark.parametrize("kernel", ["lane", "ux", "grp"])
def test_every_ungapped_kernel_matches_oracle(gpu_ctx, olz, monkeypatch, kernel):
    """The short diagonal runs have three interchangeable kernels (launch_ungapped: a run per lane; the level-synchronous
    pipeline of mb_ungapped_ux.h that dense hit sets get by default; eight lanes per run).  Each is forced in turn on pairs
    with busy diagonals, many contigs, chance hits only, soft-masked and N stretches, and on the q-batched path (MIBLAST_HIT_CAP)
    where later batches start from the extents of earlier ones: HSP list in discovery order and every counter as the oracle's."""
    from cases import DEFAULT, multi_contig, pair
    from cactus_amd import gen
    monkeypatch.setenv("MIBLAST_UNGAPPED", kernel)
    monkeypatch.setenv("MIBLAST_SEED_BATCHED", "0")             # (pair by pair: the q-batched path lives there; the shared seed stage has its own tests)
    monkeypatch.setenv("MIBLAST_CHECK_ANCHORS", "1")             # k_hsp_anchor against the host's column-by-column scan (the call fails on a difference)
    t, q = gen.make_pair(150000, 17, homologous=False)
    chance = (gen.fasta_bytes([("T|c0", t)]), gen.fasta_bytes([("Q|c0", q)]))
    for (tf, qf), args, cap in ((pair(120000, 3), DEFAULT, None), (multi_contig(9), DEFAULT, None), (chance, ["--hspthresh=1500"], None),
                                (pair(80000, 11, sub_rate=0.02), ["--step=1", "--hspthresh=2200"], "20000"), (chance, ["--hspthresh=1500", "--xdrop=300"], "3000")):
        if cap is None:
            monkeypatch.delenv("MIBLAST_HIT_CAP", raising=False)
        else:
            monkeypatch.setenv("MIBLAST_HIT_CAP", cap)
        pm = _params(args)
        T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
        got = gpu_ctx.align(T, Q, pm)
        want = olz.align(tf, qf, _oracle_params(olz, pm))
        assert got.hsps == want["hsps"]
        assert got.paf == want["paf"]
        for k in COUNTERS:
            assert got.stats[k] == want["counters"][k], (k, args, cap)


@pytest.mark.parametrize("name,tf,qf,args", CASES, ids=CASE_IDS)
def test_case_matches_oracle_through_the_batched_seed_stage(gpu_ctx, olz, monkeypatch, name, tf, qf, args):
    """The seed stage that the pairs of a batched call share (seed_phase_batched: sparse seed tables, one seed search, one sort, one
    launch of the ungapped kernels over all (pair, strand) units) forced on every single case: same bytes, HSP list in discovery
    order and counters as the oracle's."""
    monkeypatch.setenv("MIBLAST_SEED_BATCHED", "2")
    monkeypatch.setenv("MIBLAST_CHECK_ANCHORS", "1")
    pm = _params(args)
    T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
    got = gpu_ctx.align(T, Q, pm)
    want = olz.align(tf, qf, _oracle_params(olz, pm))
    assert got.hsps == want["hsps"]
    assert got.paf == want["paf"]
    assert got.alns == want["alns"] and got.ops == want["ops"]
    for k in COUNTERS:
        assert got.stats[k] == want["counters"][k], k


@pytest.mark.parametrize("kernel", ["lane", "ux", "grp"])
@pytest.mark.parametrize("batched", ["1", "0"])
def test_batched_call_with_every_ungapped_kernel_matches_oracle(gpu_ctx, olz, monkeypatch, kernel, batched):
    """Several pairs in one call -- shared targets, a pair without a hit, an empty query, many contigs, two pairs of one genome
    against itself -- with each short-run kernel forced in turn, through the shared seed stage and (MIBLAST_SEED_BATCHED=0) pair by
    pair on the lanes: every pair's bytes, HSPs and counters are the oracle's for that pair alone."""
    from cases import DEFAULT, multi_contig, pair
    from cactus_amd import gen
    monkeypatch.setenv("MIBLAST_UNGAPPED", kernel)
    monkeypatch.setenv("MIBLAST_SEED_BATCHED", batched)
    monkeypatch.setenv("MIBLAST_CHECK_ANCHORS", "1")
    a, b, c = pair(60000, 3), pair(40000, 5, sub_rate=0.05), multi_contig(9)
    t, q = gen.make_pair(50000, 17, homologous=False)
    chance = (gen.fasta_bytes([("T|c0", t)]), gen.fasta_bytes([("Q|c0", q)]))
    fastas = [a, (a[0], b[1]), c, chance, (b[0], b""), (a[0], a[0]), b, (c[0], a[1])]
    for args in (DEFAULT, ["--step=2", "--notransition", "--ydrop=3000"]):
        pm = _params(args)
        cache = {}
        sets = []
        for tf, qf in fastas:
            for fa in (tf, qf):
                if fa not in cache:
                    cache[fa] = gpu_ctx.seqset_from_fasta_bytes(fa)
            sets.append((cache[tf], cache[qf]))
        got = gpu_ctx.align_pairs(sets, pm, details=True)
        for (tf, qf), r in zip(fastas, got):
            want = olz.align(tf, qf, _oracle_params(olz, pm))
            assert r.hsps == want["hsps"]
            assert r.paf == want["paf"]
            for k in COUNTERS:
                assert r.stats[k] == want["counters"][k], (k, args)
        for h in cache.values():
            h.close()


@pytest.mark.parametrize("step", [1, 2, 5])
def test_seed_index_matches_oracle(gpu_ctx, olz, step):
    from cases import multi_contig, pair
    import numpy as np
    for tf in (pair(40000, 31)[0], multi_contig(32)[0]):
        T = gpu_ctx.seqset_from_fasta_bytes(tf)
        off, pos = gpu_ctx.build_index(T, step)
        ooff, opos = olz.build_index(tf, step)
        assert np.array_equal(off, ooff)
        assert np.array_equal(pos, opos)


def test_deterministic_across_runs_and_batch_sizes(gpu_ctx, monkeypatch):
    """Speculation policy (batch size, spatial thinning), seed-hit batch capacity and trace-arena size (forces the
    grow-and-retry path) must not change a single byte."""
    from cases import pair, DEFAULT
    tf, qf = pair(60000, 33)
    pm = _params(DEFAULT)
    T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
    base = gpu_ctx.align(T, Q, pm)
    assert base.paf.count(b"\n") >= 3
    for env in ({"MIBLAST_GAPPED_BATCH_MAX": "1"}, {"MIBLAST_SHADOW_Q": "0", "MIBLAST_SHADOW_D": "0"},
                {"MIBLAST_SHADOW_Q": "100000000"}, {"MIBLAST_HIT_CAP": "3000"}, {"MIBLAST_SEED_BATCHED": "0", "MIBLAST_HIT_CAP": "3000"}, {"MIBLAST_ARENA_MB": "1"},
                # DP kernel: one wave per piece with 2 x 4, 4 or 8 columns per lane (windows that outgrow the lanes are rerun with
                # the 4-wave LDS-ring kernel), or the 4-wave kernel from the start
                {"MIBLAST_DP_KERNEL": "2"}, {"MIBLAST_DP_KERNEL": "4"}, {"MIBLAST_DP_KERNEL": "8"}, {"MIBLAST_DP_KERNEL": "100"},
                {"MIBLAST_DP_WAVES": "4"}, {"MIBLAST_DP_WAVES": "5"},    # the two builds of the one-wave DP kernel: 4 waves per SIMD (the default) / held to 96 VGPRs = 5 waves
                {"MIBLAST_SEED_BATCHED": "0"},                  # pair by pair instead of the shared seed stage (the default for a pair this small)
                {"MIBLAST_SEED_BATCHED": "0", "MIBLAST_SEED_ONE_PASS": "0"},      # ... with the two-pass seed search (count, scan, fill) instead of the fused one
                {"MIBLAST_SEED_BATCHED": "0", "MIBLAST_SEED_FUSED": "0"},         # ... strands one after the other instead of both in one go
                {"MIBLAST_LONG_RUN": "4"}, {"MIBLAST_LONG_RUN": "32"},      # which diagonal runs go to the wave-per-run ungapped kernel
                {"MIBLAST_RELAY_CKPT": "0"},                    # rejected hand-overs continue to the next relay instead of retrying at a later snapshot
                # traceback: no join walks from predicted entries (the sides walk themselves) / every other prediction made wrong on purpose
                {"MIBLAST_TRACE_PREJOIN": "0"}, {"MIBLAST_TRACE_PREJOIN": "2"},
                {"MIBLAST_TRACE_PREJOIN": "2", "MIBLAST_RELAY_S0": "64", "MIBLAST_RELAY_S": "256", "MIBLAST_RELAY_W": "64"},
                {"MIBLAST_CHAIN_HEADS": "0"},                   # first-round nomination by spatial thinning instead of one head per colinear anchor group
                {"MIBLAST_GROUP_GAP": "200", "MIBLAST_GROUP_TOL": "8"}, {"MIBLAST_GROUP_GAP": "1000000", "MIBLAST_GROUP_TOL": "100000"},
                {"MIBLAST_DP_KERNEL": "4", "MIBLAST_RELAY_S0": "64", "MIBLAST_RELAY_S": "256", "MIBLAST_RELAY_W": "64", "MIBLAST_RELAY_FORCE_REJECT": "3"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        again = gpu_ctx.align(T, Q, pm)
        for k in env:
            monkeypatch.delenv(k)
        assert again.paf == base.paf, env
        assert again.hsps == base.hsps and again.alns == base.alns
        for k in COUNTERS:
            assert again.stats[k] == base.stats[k], (env, k)


RELAY_CONFIGS = [
    {"MIBLAST_RELAY_S0": "0"},                                                                  # relays off: one piece per side
    {"MIBLAST_RELAY_S0": "64", "MIBLAST_RELAY_S": "256", "MIBLAST_RELAY_W": "64"},
    {"MIBLAST_RELAY_S0": "128", "MIBLAST_RELAY_S": "512", "MIBLAST_RELAY_W": "128", "MIBLAST_RELAY_MAX": "3"},      # capped chains are re-planted
    {"MIBLAST_RELAY_S0": "64", "MIBLAST_RELAY_S": "256", "MIBLAST_RELAY_W": "64", "MIBLAST_RELAY_FORCE_REJECT": "2"},
    {"MIBLAST_RELAY_S0": "32", "MIBLAST_RELAY_S": "300", "MIBLAST_RELAY_W": "70", "MIBLAST_RELAY_FORCE_REJECT": "3", "MIBLAST_RELAY_TOL": "40"},
    {"MIBLAST_RELAY_S0": "512", "MIBLAST_RELAY_S": "2048", "MIBLAST_RELAY_W": "256", "MIBLAST_RELAY_FORCE_REJECT": "5"},
    # the policy of batched calls on a single pair: a side first has to survive S0 rows, relays are planted at its first stop
    {"MIBLAST_RELAY_S0": "48", "MIBLAST_RELAY_S": "256", "MIBLAST_RELAY_W": "64", "MIBLAST_RELAY_PLANT_AT_ONCE": "0"},
    {"MIBLAST_RELAY_S0": "256", "MIBLAST_RELAY_S": "640", "MIBLAST_RELAY_W": "128", "MIBLAST_RELAY_PLANT_AT_ONCE": "0", "MIBLAST_RELAY_FORCE_REJECT": "3"},
    # retries at the relays' later entry snapshots switched off / forced through all three checkpoints
    {"MIBLAST_RELAY_S0": "64", "MIBLAST_RELAY_S": "512", "MIBLAST_RELAY_W": "64", "MIBLAST_RELAY_FORCE_REJECT": "2", "MIBLAST_RELAY_CKPT": "0"},
    {"MIBLAST_RELAY_S0": "64", "MIBLAST_RELAY_S": "768", "MIBLAST_RELAY_W": "48", "MIBLAST_RELAY_FORCE_REJECT": "2"},
    # traceback: the join walks k_trace_prejoin makes from predicted entries switched off / half of the predictions made wrong
    {"MIBLAST_RELAY_S0": "64", "MIBLAST_RELAY_S": "256", "MIBLAST_RELAY_W": "64", "MIBLAST_TRACE_PREJOIN": "0"},
    {"MIBLAST_RELAY_S0": "64", "MIBLAST_RELAY_S": "256", "MIBLAST_RELAY_W": "64", "MIBLAST_TRACE_PREJOIN": "2"},
    {"MIBLAST_RELAY_S0": "32", "MIBLAST_RELAY_S": "300", "MIBLAST_RELAY_W": "70", "MIBLAST_RELAY_FORCE_REJECT": "3", "MIBLAST_TRACE_PREJOIN": "2"},
    # round 5, the hand-over inside the DP launch (mb_ydrop2.h): off (every rejected hand-over a launch of its own, as before); every second
    # piece's first check rejected on purpose (the piece goes on to the relay's next snapshot in the same wave); the first three checks of
    # EVERY piece rejected (past the relay's last snapshot, on to the relay after); next to no room to go on (the host's continuation
    # takes over); both kinds of forced rejection at once
    {"MIBLAST_RELAY_S0": "64", "MIBLAST_RELAY_S": "256", "MIBLAST_RELAY_W": "64", "MIBLAST_RELAY_INLINE": "0", "MIBLAST_RELAY_FORCE_REJECT": "2"},
    {"MIBLAST_RELAY_S0": "64", "MIBLAST_RELAY_S": "256", "MIBLAST_RELAY_W": "64", "MIBLAST_RELAY_INLINE_FORCE_REJECT": "2"},
    {"MIBLAST_RELAY_S0": "64", "MIBLAST_RELAY_S": "512", "MIBLAST_RELAY_W": "64", "MIBLAST_RELAY_INLINE_FORCE_REJECT": "-3"},
    {"MIBLAST_RELAY_S0": "64", "MIBLAST_RELAY_S": "256", "MIBLAST_RELAY_W": "64", "MIBLAST_RELAY_INLINE_FORCE_REJECT": "-1", "MIBLAST_RELAY_INLINE_ROWS": "100"},
    {"MIBLAST_RELAY_S0": "48", "MIBLAST_RELAY_S": "300", "MIBLAST_RELAY_W": "70", "MIBLAST_RELAY_INLINE_FORCE_REJECT": "3", "MIBLAST_RELAY_FORCE_REJECT": "4", "MIBLAST_RELAY_CKPT": "0"},
]


@pytest.mark.parametrize("env", RELAY_CONFIGS, ids=lambda e: ",".join(v for v in e.values()))
def test_relay_handover_matches_oracle(gpu_ctx, olz, monkeypatch, env):
    """Long one-sided DPs are cut into concurrently evaluated pieces (relays started at downstream anchors, accepted
    only when the hand-over states match).  Wherever the pieces start, whether hand-overs are
    accepted, rejected (forced here) or chains re-planted, every byte and counter must equal the sequential oracle."""
    from cases import DEFAULT
    from cactus_amd import gen
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for n, seed, sub, indel, args in ((60000, 33, 0.15, 0.01, DEFAULT), (90000, 77, 0.05, 0.02, ["--ydrop=4000", "--hspthresh=2200", "--gappedthresh=2400"]),
                                      (30000, 5, 0.25, 0.03, DEFAULT),
                                      # lastz's own y-drop (9400): windows of ~400 columns, 8 columns per lane, the widest rows rerun with the LDS ring
                                      (50000, 91, 0.10, 0.02, ["--ambiguous=iupac,100,100", "--hspthresh=2200"])):
        t, q = gen.make_pair(n, seed, sub_rate=sub, indel_rate=indel)
        tf, qf = gen.fasta_bytes([("T|c0", t)]), gen.fasta_bytes([("Q|c0", q)])
        pm = _params(args)
        T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
        got = gpu_ctx.align(T, Q, pm)
        want = olz.align(tf, qf, _oracle_params(olz, pm))
        assert got.paf == want["paf"]
        assert got.alns == want["alns"] and got.ops == want["ops"]
        for k in COUNTERS:
            assert got.stats[k] == want["counters"][k], k
        if env["MIBLAST_RELAY_S0"] == "0":
            assert got.stats["relay_accepted"] == 0 and got.stats["relay_rejected"] == 0
        elif n >= 60000:
            assert got.stats["relay_accepted"] > 0, "the case must exercise hand-overs"
            if "MIBLAST_RELAY_FORCE_REJECT" in env:
                assert got.stats["relay_rejected"] > 0
            if env.get("MIBLAST_RELAY_INLINE") == "0":
                assert got.stats["relay_inline_checks"] == 0 and got.stats["relay_inline_continued"] == 0
            if "MIBLAST_RELAY_INLINE_FORCE_REJECT" in env and args is DEFAULT:                # (--ydrop=4000: the one-wave kernel that has the check)
                assert got.stats["relay_inline_checks"] > 0
                if "MIBLAST_RELAY_INLINE_ROWS" not in env:
                    assert got.stats["relay_inline_continued"] > 0, "no piece went on inside its launch"


def test_full_size_properties_1mb(gpu_ctx, olz, monkeypatch):
    """BASELINE config 2 at full size (1 Mb x 1 Mb): the oracle's bytes and counters (1.6 s of CPU), and size-independent
    properties: every record passes the caf walk (pinchIterator.c:59-121) and its AS score re-derives from the
    sequences; the run is reproducible; aligning the reverse-complemented query yields the same alignments with the
    strand flipped."""
    from cactus_amd import gen, pafcheck
    from cases import DEFAULT
    t, q = gen.make_pair(1_000_000, 42)
    tf = gen.fasta_bytes([("id=simT|chr1", t)])
    qf = gen.fasta_bytes([("id=simQ|chr1", q)])
    qrf = gen.fasta_bytes([("id=simQ|chr1", gen.revcomp(q))])
    pm = _params(DEFAULT)
    T, Q, QR = (gpu_ctx.seqset_from_fasta_bytes(x) for x in (tf, qf, qrf))
    r1 = gpu_ctx.align(T, Q, pm, details=False)
    r2 = gpu_ctx.align(T, Q, pm, details=False)
    assert r1.paf == r2.paf
    want = olz.align(tf, qf, _oracle_params(olz, pm), details=False)
    assert r1.paf == want["paf"]
    for k in COUNTERS:
        assert r1.stats[k] == want["counters"][k], k
    n = pafcheck.check_paf(r1.paf.decode(), pafcheck.read_fasta(tf), pafcheck.read_fasta(qf))
    assert n == r1.stats["alignments"] and n > 5
    rr = gpu_ctx.align(T, QR, pm, details=False)

    def canon(paf, flip):
        out = []
        for line in paf.decode().splitlines():
            f = line.split("\t")
            if flip:
                qlen, qs, qe = int(f[1]), int(f[2]), int(f[3])
                f[2], f[3] = str(qlen - qe), str(qlen - qs)
                f[4] = "+" if f[4] == "-" else "-"
            out.append("\t".join(f))
        return sorted(out)

    assert canon(r1.paf, False) == canon(rr.paf, True)
    assert r1.stats["dp_cells"] == rr.stats["dp_cells"] and r1.stats["seed_hits"] == rr.stats["seed_hits"]
    # the relay path (hundreds of concurrently evaluated pieces per long alignment) against the plain sequential DPs
    assert r1.stats["relay_accepted"] > 100
    monkeypatch.setenv("MIBLAST_RELAY_S0", "0")
    seq = gpu_ctx.align(T, Q, pm, details=False)
    monkeypatch.delenv("MIBLAST_RELAY_S0")
    assert seq.stats["relay_accepted"] == 0 and seq.paf == r1.paf
    for k in COUNTERS:
        assert seq.stats[k] == r1.stats[k], k


def test_cli_front_ends_match_library(gpu_ctx, tmp_path):
    """bin/lastz and bin/run_kegalign with the exact argv run_lastz builds (local_alignment.py:60-68, :54-58):
    same bytes as the in-process call, stderr empty on success, non-zero exit on a bad option."""
    import subprocess, os
    from cactus_amd.shared.common import BIN_DIR
    from cases import pair, DEFAULT, KEG_DEFAULT
    tf, qf = pair(25000, 34)
    (tmp_path / "A_0.fa").write_bytes(tf)
    (tmp_path / "B_0.fa").write_bytes(qf)
    T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
    want = gpu_ctx.align(T, Q, _params(DEFAULT), details=False).paf
    p = subprocess.run([os.path.join(BIN_DIR, "lastz"), "A_0.fa[multiple][nameparse=darkspace]", "B_0.fa[nameparse=darkspace]",
                        "--format=paf:wfmash"] + DEFAULT, cwd=tmp_path, capture_output=True)
    assert p.returncode == 0 and p.stderr == b"" and p.stdout == want
    want_k = gpu_ctx.align(T, Q, _params(KEG_DEFAULT), details=False).paf
    p = subprocess.run([os.path.join(BIN_DIR, "run_kegalign"), "A_0.fa", "B_0.fa", "--format=paf:wfmash"] + KEG_DEFAULT +
                       ["--num_gpu", "1", "--num_threads", "2"], cwd=tmp_path, capture_output=True)
    assert p.returncode == 0 and p.stderr == b"" and p.stdout == want_k
    p = subprocess.run([os.path.join(BIN_DIR, "lastz"), "A_0.fa", "B_0.fa", "--format=paf:wfmash", "--bogus=1"], cwd=tmp_path, capture_output=True)
    assert p.returncode != 0 and p.stdout == b""
    p = subprocess.run([os.path.join(BIN_DIR, "lastz"), "missing.fa", "B_0.fa", "--format=paf:wfmash"], cwd=tmp_path, capture_output=True)
    assert p.returncode != 0


def test_run_lastz_job_interface(gpu_ctx, olz, tmp_path, monkeypatch):
    """The Toil job function with the reference's signature, CPU-style and GPU-style config, subprocess and
    in-process boundary: identical PAF, equal to the oracle run with the parameter set the distance selects."""
    from cactus_amd.paf.local_alignment import run_lastz, select_lastz_params
    from cactus_amd.shared.configWrapper import load_config
    from localjob import LocalJob, LocalFileStore, FileID
    from cactus_amd import miblast
    from cases import pair
    tf, qf = pair(30000, 35, sub_rate=0.1, indel_rate=0.005)
    a, b = tmp_path / "a.fa", tmp_path / "b.fa"
    a.write_bytes(tf); b.write_bytes(qf)
    job = LocalJob(LocalFileStore(str(tmp_path / "js")) if (tmp_path / "js").mkdir() is None else None)
    for gpu in (0, 1):
        cfg = load_config()
        cfg.find("blast").attrib["gpu"] = str(gpu)
        for distance in (0.176, 0.4):
            args = select_lastz_params(distance, cfg, gpu).split(" ")
            pm = miblast.params_from_args(args)
            want = olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}), details=False)["paf"]
            for inproc in ("0", "1"):
                monkeypatch.setenv("MIBLAST_INPROCESS", inproc)
                fid = run_lastz(job, "A_0", FileID.of(str(a)), "B_0", FileID.of(str(b)), distance, cfg)
                assert open(str(fid), "rb").read() == want, (gpu, distance, inproc)


def test_evolver_like_blast_phase_cigar_diff(gpu_ctx, olz):
    """BASELINE config 3 stand-in (evolverMammals is a set of URLs): five leaves evolved on the evolverMammals guide
    tree from a 120 kb ancestor; every ingroup pair is run through the job interface's parameter selection (mouse-rat
    at distance 0.176 -> set "four", the others -> "default") and the PAF is diffed byte-for-byte against the oracle."""
    from cactus_amd import gen, miblast
    from cactus_amd.paf.local_alignment import select_lastz_params
    from cactus_amd.shared.configWrapper import load_config
    cfg = load_config()
    leaves = gen.make_tree_genomes(120_000, 2001)
    fa = {k: gen.fasta_bytes([("id=%s|%s" % (k, k), v)]) for k, v in leaves.items()}
    sets = {k: gpu_ctx.seqset_from_fasta_bytes(v) for k, v in fa.items()}
    pairs = [("simMouse_chr6", "simRat_chr6", 0.176098), ("simHuman_chr6", "simMouse_chr6", 0.500501),
             ("simCow_chr6", "simDog_chr6", 0.35211), ("simHuman_chr6", "simDog_chr6", 0.360539),
             ("simRat_chr6", "simCow_chr6", 0.606134)]
    total = 0
    for a, b, dist in pairs:
        args = select_lastz_params(dist, cfg, 0).split(" ")
        pm = miblast.params_from_args(args)
        got = gpu_ctx.align(sets[a], sets[b], pm, details=False)
        want = olz.align(fa[a], fa[b], olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}), details=False)
        assert got.paf == want["paf"], (a, b)
        assert got.stats["dp_cells"] == want["counters"]["dp_cells"] and got.stats["seed_hits"] == want["counters"]["seed_hits"]
        total += got.paf.count(b"\n")
    assert total >= 2 * len(pairs)


def test_repeat_mask_call_site_general_format(gpu_ctx, olz, tmp_path):
    """the repeat masker's lastz invocation (cactus_lastzRepeatMask.py:97-105) -- query = 200 bp fragments
    (hundreds of records), --ungapped --queryhsplimit=keep,nowarn:N, general six-column output, --markend -- byte-identical
    to the oracle, through the library and through bin/lastz; then the whole job masks a planted high-copy repeat."""
    import subprocess, os
    import numpy as np
    from cactus_amd import gen, miblast
    from cactus_amd.preprocessor.lastz_repeat_mask import LastzRepeatMaskJob, RepeatMaskOptions, fasta_fragments
    from cactus_amd.shared.common import BIN_DIR
    from localjob import LocalFileStore, FileID
    rng = np.random.default_rng(9)
    unit = gen.random_sequence(600, rng)
    parts, truth = [], []
    pos = 0
    for k in range(40):
        flank = gen.random_sequence(int(rng.integers(300, 900)), rng)
        parts.append(flank); pos += len(flank)
        if k % 2 == 0:
            copy = gen.mutate(unit, rng, 0.03, 0.0)
            truth.append((pos, pos + len(copy))); parts.append(copy); pos += len(copy)
    genome = np.concatenate(parts)
    qfa = gen.fasta_bytes([("id=E|chrR", genome)])
    frags = fasta_fragments(qfa.decode(), 200, 100, "zero").encode()
    args = "--step=3 --ambiguous=iupac,100,100 --ungapped --queryhsplimit=keep,nowarn:7 --querydepth=keep,nowarn:53 --format=general:name1,zstart1,end1,name2,zstart2+,end2+ --markend".split()
    pm = miblast.params_from_args(args)
    assert (pm.gapped, pm.format, pm.markend, pm.queryhsplimit) == (0, 1, 1, 7)
    T, Q = gpu_ctx.seqset_from_fasta_bytes(qfa), gpu_ctx.seqset_from_fasta_bytes(frags)
    got = gpu_ctx.align(T, Q, pm)
    want = olz.align(qfa, frags, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}))
    assert got.paf == want["paf"] and got.hsps == want["hsps"]
    assert got.paf.count(b"\n") > 200 and got.paf.endswith(b"# lastz end-of-file\n")
    (tmp_path / "t.fa").write_bytes(qfa); (tmp_path / "f.fa").write_bytes(frags)
    p = subprocess.run([os.path.join(BIN_DIR, "lastz"), "t.fa[multiple][nameparse=darkspace]", "f.fa[nameparse=darkspace]"] + args, cwd=tmp_path, capture_output=True)
    assert p.returncode == 0 and p.stderr == b"" and p.stdout == got.paf
    # end to end: fragments -> lastz -> covered intervals -> softmask
    fs = LocalFileStore(str(tmp_path / "js")) if (tmp_path / "js").mkdir() is None else None
    src = tmp_path / "g.fa"; src.write_bytes(qfa)
    job = LastzRepeatMaskJob(RepeatMaskOptions(fragment=200, minPeriod=5, eventName="E"), FileID.of(str(src)), [FileID.of(str(src))])
    masked = open(str(job.run(fs))).read()
    seq = "".join(masked.splitlines()[1:])
    assert seq.upper() == genome.tobytes().decode()
    low = np.array([c.islower() for c in seq])
    inside = np.mean([low[a + 50:b - 50].mean() for a, b in truth])
    outside_mask = np.ones(len(seq), bool)
    for a, b in truth:
        outside_mask[max(0, a - 30):b + 30] = False
    assert inside > 0.9 and low[outside_mask].mean() < 0.02


def test_ingroup_to_outgroup_trimming_chain(gpu_ctx, tmp_path):
    """SURVEY section 8 f4 (local_alignment.py:421-526): align the ingroup to the nearest outgroup, extract what stayed
    unaligned (>= trimMinSize, + trimFlanking), align only that to the next outgroup, fix coordinates with dechunk --query,
    invert.  Every record of the final PAF must validate against the FULL sequences, and the second outgroup must
    only pick up what the first one did not cover."""
    import numpy as np
    from cactus_amd import gen, pafcheck
    from cactus_amd.paf.local_alignment import make_ingroup_to_outgroup_alignments_0
    from cactus_amd.shared.configWrapper import load_config
    from localjob import LocalJob, LocalFileStore, FileID
    rng = np.random.default_rng(21)
    anc = gen.random_sequence(60000, rng)
    ingroup = gen.mutate(anc, rng, 0.03, 0.002)
    og1 = np.concatenate([gen.mutate(anc[:28000], rng, 0.08, 0.004), gen.random_sequence(5000, rng)])      # shares the left part
    og2 = np.concatenate([gen.random_sequence(4000, rng), gen.mutate(anc[20000:], rng, 0.08, 0.004)])      # shares the right part (+ overlap)
    paths = {}
    for name, seq in (("I", ingroup), ("O1", og1), ("O2", og2)):
        p = tmp_path / (name + ".fa"); gen.write_fasta(str(p), [("id=%s|chr1" % name, seq)]); paths[name] = p
    (tmp_path / "js").mkdir()
    job = LocalJob(LocalFileStore(str(tmp_path / "js")))
    seqs = {k: FileID.of(str(v)) for k, v in paths.items()}
    dist = {("I", "O1"): 0.3, ("I", "O2"): 0.3}
    out = make_ingroup_to_outgroup_alignments_0(job, "I", ["O1", "O2"], dict(seqs), dist, load_config())
    text = open(str(out)).read()
    full = {}
    for v in paths.values():
        full.update(pafcheck.read_fasta(str(v)))
    recs = [pafcheck.parse_line(l) for l in text.splitlines()]
    assert recs and all(r["tname"] == "id=I|chr1" and r["tlen"] == len(ingroup) for r in recs)      # inverted: ingroup is the target
    n = pafcheck.check_paf(text, full, full)
    cov = {"id=O1|chr1": np.zeros(len(ingroup), bool), "id=O2|chr1": np.zeros(len(ingroup), bool)}
    for r in recs:
        cov[r["qname"]][r["tstart"]:r["tend"]] = True
    assert cov["id=O1|chr1"][:25000].mean() > 0.9 and cov["id=O1|chr1"][30000:].mean() < 0.01
    assert cov["id=O2|chr1"][32000:].mean() > 0.9
    # O2 was only offered what O1 left unaligned (plus 100 bp flanks): no O2 alignment deep inside O1's territory
    assert cov["id=O2|chr1"][:20000].mean() < 0.01 and (cov["id=O1|chr1"] & cov["id=O2|chr1"]).sum() <= 2 * 100 * n


@pytest.mark.parametrize("lanes", ["4", "1", "3"])
def test_batched_pairs_equal_single_calls(gpu_ctx, monkeypatch, lanes):
    """miblast_align_pairs: several chunk pairs in one call (seed stages dealt to concurrent lanes, merged gapped launches) must
    return, pair by pair, exactly the bytes and counters of separate miblast_align calls -- including pairs with no alignment
    and different contig sets; the number of lanes must not matter."""
    from cases import CASES, DEFAULT
    from cactus_amd import miblast
    monkeypatch.setenv("MIBLAST_SEED_LANES", lanes)
    pm = miblast.params_from_args(DEFAULT)
    chosen = [c for c in CASES if c[0] in ("homolog_20k_default", "random_50k", "multi_contig_ragged", "tandem_repeats", "revcomp_query", "empty_query")]
    sets = [(gpu_ctx.seqset_from_fasta_bytes(c[1]), gpu_ctx.seqset_from_fasta_bytes(c[2])) for c in chosen]
    single = [gpu_ctx.align(t, q, pm) for t, q in sets]
    batched = gpu_ctx.align_pairs(sets, pm, details=True)
    assert len(batched) == len(single)
    for name, a, b in zip([c[0] for c in chosen], single, batched):
        assert a.paf == b.paf, name
        assert a.hsps == b.hsps and a.alns == b.alns and a.ops == b.ops, name
        for k in ("seed_hits", "hits_extended", "ungapped_cols", "hsps", "anchors", "anchors_skipped", "dp_sides", "dp_cells", "dp_rows", "alignments"):
            assert a.stats[k] == b.stats[k], (name, k)


@pytest.mark.parametrize("seed", range(8))
def test_outgroup_trimming_on_the_device_equals_the_text_path_and_the_oracle(gpu_ctx, tmp_path, seed):
    """miblast_seqsets_unaligned (per-base coverage, uncovered stretches, gather -- all on the resident set) against
    oracle/paffy_text_oracle.c (`paffy to_bed --excludeAligned --minSize N | faffy extract --flank F` restated with per-base
    counters) and against the product's own text path: same records, names, lengths and bases; several items per call; a second
    round on an already trimmed set (nested NAME|LEN|START names); nothing left -> no set."""
    from test_text_oracle_cpu import random_case, oracle, built      # noqa: F401  (the oracle is built on demand)
    import subprocess, os
    subprocess.run(["make", "-C", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"), "oracle_paffy_text"], check=True, capture_output=True)
    from cactus_amd import mipaf
    fa, paf, recs = random_case(seed, n_contigs=1 + seed % 5, nested=seed % 3 == 2)
    (tmp_path / "q.fa").write_bytes(fa)
    (tmp_path / "a.paf").write_bytes(paf)
    Q = gpu_ctx.seqset_from_fasta_bytes(fa)
    settings = ((100, 100), (1, 0), (37, 5), (5000, 10), (10, 3000))
    for min_size, flank in settings:
        want = oracle("to_bed_extract", tmp_path / "a.paf", tmp_path / "q.fa", min_size, flank)
        got = gpu_ctx.seqsets_unaligned([Q, Q], [paf, b""], min_size, flank)                   # (an item without alignments: everything >= min_size is left)
        if not want:
            assert got[0] is None
        else:
            assert got[0].fasta_bytes() == want == mipaf.unaligned_fasta(paf, fa, min_size, flank)
            ref = gpu_ctx.seqset_from_fasta_bytes(want)
            assert got[0].contigs == ref.contigs and got[0].total == ref.total
            # second round on the trimmed set: what a later outgroup leaves of it
            inner = []
            for name, start, n in got[0].contigs:
                if n > 30:
                    inner.append(f"{name}\t{n}\t{n // 4}\t{n // 2}\t+\tid=T|x\t9999\t0\t{n // 2 - n // 4}\t1\t1\t255\n")
            paf2 = "".join(inner).encode()
            (tmp_path / "q2.fa").write_bytes(want)
            (tmp_path / "a2.paf").write_bytes(paf2)
            want2 = oracle("to_bed_extract", tmp_path / "a2.paf", tmp_path / "q2.fa", 8, 3)
            got2 = gpu_ctx.seqsets_unaligned([got[0]], [paf2], 8, 3)[0]
            assert (got2.fasta_bytes() if got2 is not None else b"") == want2
            ref.close()
        empty = tmp_path / "none.paf"
        empty.write_bytes(b"")
        want_all = oracle("to_bed_extract", empty, tmp_path / "q.fa", min_size, flank)
        assert (got[1].fasta_bytes() if got[1] is not None else b"") == want_all
    from cactus_amd import miblast
    with pytest.raises(miblast.MiblastError):
        gpu_ctx.seqsets_unaligned([Q], [b"nobody\t10\t0\t5\t+\tt\t10\t0\t5\t5\t5\t255\n"], 1, 0)
    Q.close()


def test_evolver_phase_with_trimming_on_the_device_equals_the_oracle_call_by_call(gpu_ctx, olz):
    """The phase as bench.py runs it -- genomes resident, every chain's leftover cut out on the device between the outgroup calls
    (align_batch.trim_resident -> miblast_seqsets_unaligned) -- at a tenth of the size: every call's bytes and counters are the
    oracle's on the FASTA text of what the device left over, and the assembled files equal those of the text path."""
    from cactus_amd import blast_phase as bp, gen, miblast
    from cactus_amd.paf.local_alignment import select_lastz_params
    from cactus_amd.shared.configWrapper import load_config
    cfg = load_config()
    calls = bp.blast_phase_calls(bp.parse_newick(bp.EVOLVER_MAMMALS_TREE))
    genomes = gen.make_tree_genomes(60_000, 2001, ancestors=True)
    fasta = {k: gen.fasta_bytes([("id=%s|%s" % (k, k), v)]) for k, v in genomes.items()}
    resident = {fa: gpu_ctx.seqset_from_fasta_bytes(fa) for fa in fasta.values()}
    made, seen = [], []

    def align_batch(pairs, opts):
        pm = miblast.params_from_args(opts.split())
        sets = [(resident[t], q if isinstance(q, miblast.SeqSet) else resident[q]) for t, q in pairs]
        return [r.paf for r in gpu_ctx.align_pairs(sets, pm)]

    def trim_resident(items, min_size, flank):
        outs = gpu_ctx.seqsets_unaligned([q if isinstance(q, miblast.SeqSet) else resident[q] for q, _ in items], [p for _, p in items], min_size, flank)
        made.extend(o for o in outs if o is not None)
        return outs

    options = lambda d: select_lastz_params(d, cfg, 0)      # noqa: E731
    align_batch.trim_resident = trim_resident
    res = bp.run_blast_phase(fasta, calls, options, align_batch, on_call=lambda c, tf, qf, paf: seen.append((c, tf, qf, paf)))
    assert len(seen) == 20 and made
    for c, tf, qf, paf in seen:
        pm = miblast.params_from_args(options(c.distance).split())
        want = olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}), details=False)
        assert paf == want["paf"], (c.node, c.kind, c.level)

    def text_batch(pairs, opts):
        pm = miblast.params_from_args(opts.split())
        out = []
        for t, q in pairs:
            T, Q = gpu_ctx.seqset_from_fasta_bytes(t), gpu_ctx.seqset_from_fasta_bytes(q)
            out.append(gpu_ctx.align(T, Q, pm, details=False).paf)
            T.close(); Q.close()
        return out

    assert bp.run_blast_phase(fasta, calls, options, text_batch) == res
    for h in list(resident.values()) + made:
        h.close()


def test_evolver_primates_phase_at_full_size_equals_the_oracle_call_by_call(gpu_ctx, olz):
    """BASELINE configs[0] as a parity test: the evolverPrimates stand-in of SURVEY 8d config 1 (600 kb ancestor, seed 1001, the guide
    tree of /root/reference/examples/evolverPrimates.txt:1) -- nine lastz calls, every one with option set "one" -- through the
    batched calls and the device trimming the bench uses, every call diffed byte for byte and counter for counter."""
    from cactus_amd import blast_phase as bp, gen, miblast
    from cactus_amd.paf.local_alignment import select_lastz_params
    from cactus_amd.shared.configWrapper import load_config
    cfg = load_config()
    tree = bp.parse_newick(bp.EVOLVER_PRIMATES_TREE)
    calls = bp.blast_phase_calls(tree)

    def nested(n):
        return (n.iD, n.distance, [nested(c) for c in n.children])

    genomes = gen.make_tree_genomes(600_000, 1001, tree=("root", nested(tree)[2]), ancestors=True)
    assert set(genomes) == {"simOrang", "simChimp", "simHuman", "simGorilla", "cb", "hcb", "Anc0"}
    fasta = {k: gen.fasta_bytes([("id=%s|%s" % (k, k), v)]) for k, v in genomes.items()}
    resident = {fa: gpu_ctx.seqset_from_fasta_bytes(fa) for fa in fasta.values()}
    made, seen = [], []

    def align_batch(pairs, opts):
        assert opts.startswith("--step=2 ") and "--notransition" in opts                      # set "one" everywhere
        pm = miblast.params_from_args(opts.split())
        sets = [(resident[t], q if isinstance(q, miblast.SeqSet) else resident[q]) for t, q in pairs]
        rs = gpu_ctx.align_pairs(sets, pm)
        seen.extend((opts, r.stats) for r in rs)
        return [r.paf for r in rs]

    def trim_resident(items, min_size, flank):
        outs = gpu_ctx.seqsets_unaligned([q if isinstance(q, miblast.SeqSet) else resident[q] for q, _ in items], [p for _, p in items], min_size, flank)
        made.extend(o for o in outs if o is not None)
        return outs

    align_batch.trim_resident = trim_resident
    checked = []

    def on_call(c, tf, qf, paf):
        pm = miblast.params_from_args(select_lastz_params(c.distance, cfg, 0).split())
        want = olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}), details=False)
        assert paf == want["paf"], (c.node, c.kind, c.level)
        checked.append(want["counters"])

    bp.run_blast_phase(fasta, calls, lambda d: select_lastz_params(d, cfg, 0), align_batch, on_call=on_call)
    assert len(checked) == 9 == len(seen)
    for k in ("seed_hits", "hits_extended", "ungapped_cols", "hsps", "anchors", "dp_cells", "alignments"):
        assert sum(c[k] for c in checked) == sum(st[k] for _, st in seen), k
    assert sum(c["dp_cells"] for c in checked) > 3e8
    for h in list(resident.values()) + made:
        h.close()


def test_full_size_chunk_pair_equals_the_oracle_digest(gpu_ctx):
    """One chunk pair at Cactus's full chunk size (30 Mb x 30 Mb, 1.3 % divergence, half soft-masked, parameter
    set "one"): PAF bytes and counters equal the CPU oracle's, which takes a minute on this input and is therefore committed as a
    digest (tests/golden/cfg4_30mb.json, written by scripts/oracle_cfg4.py)."""
    import hashlib, json, os
    from cactus_amd import gen, miblast
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg4_30mb.json")))
    t, q = gen.make_pair(30_000_000, 3001, sub_rate=0.013, indel_rate=0.002, mask_frac=0.5)
    T = gpu_ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("id=simT|chr20", t)]))
    Q = gpu_ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("id=simQ|chr20", q)]))
    pm = miblast.params_from_args("--step=2 --ambiguous=iupac,100,100 --ydrop=3000 --notransition --queryhspbest=100000".split())
    r = gpu_ctx.align(T, Q, pm, details=False)
    T.close(); Q.close()
    assert (hashlib.md5(r.paf).hexdigest(), len(r.paf)) == (want["paf_md5"], want["paf_bytes"])
    for k in ("alignments", "dp_cells", "seed_hits", "hsps"):
        assert r.stats[k] == want[k], k


def test_randomised_differential_fuzz():
    """A slice of scripts/gpu_fuzz.py (random structures x random lastz options, GPU vs oracle byte for byte).  The
    full script found two real bugs during development (x-drop stop exactly at lane 63 of the wave-parallel extension;
    an out-of-bounds base read past the contig end in the HBM-ring DP variant)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_fuzz.py"), "30", "5000"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "30 cases, 0 mismatches" in p.stdout


def test_chunked_genome_pair_through_the_job_functions(gpu_ctx, olz, tmp_path):
    """SURVEY 8 rows a3 / a4 at reduced size (config 4's shape: chunkSize + overlapSize -> 3 x 3 chunk pairs): the mirrored
    make_chunked_alignments (local_alignment.py:370-408: faffy chunk both genomes, one run_lastz job per chunk pair) and
    combine_chunks (:336-356: paffy dechunk + concatenate) give exactly the per-pair oracle PAFs passed through the same dechunk,
    in chunk-pair order; every record validates against the UNCHUNKED sequences (coordinates restored by dechunk); and the nine
    pairs in ONE miblast_align_pairs call equal nine single calls."""
    import copy
    from cactus_amd import gen, miblast, pafcheck
    from cactus_amd.paf import chunking
    from cactus_amd.paf.local_alignment import make_chunked_alignments, select_lastz_params
    from cactus_amd.shared.configWrapper import load_config
    from localjob import LocalJob, LocalFileStore, FileID
    cfg = copy.deepcopy(load_config())
    cfg.find("blast").attrib.update(chunkSize="900000", overlapSize="10000")
    t, q = gen.make_pair(2_000_000, 3001, sub_rate=0.013, indel_rate=0.002, mask_frac=0.5)
    pa, pb = tmp_path / "A.fa", tmp_path / "B.fa"
    gen.write_fasta(str(pa), [("id=simT|chr20", t)])
    gen.write_fasta(str(pb), [("id=simQ|chr20", q)])
    (tmp_path / "js").mkdir()
    job = LocalJob(LocalFileStore(str(tmp_path / "js")))
    dist = 0.03                                                             # <= 0.05: option set "one"
    out = make_chunked_alignments(job, "simT", FileID.of(str(pa)), "simQ", FileID.of(str(pb)), dist, cfg)
    got = open(str(out)).read()
    # expectation: the same chunk files, the oracle per pair, the same dechunk
    ca = chunking.fasta_chunk(str(pa), str(tmp_path / "ca"), 900000, 10000)
    cb = chunking.fasta_chunk(str(pb), str(tmp_path / "cb"), 900000, 10000)
    assert len(ca) == 3 and len(cb) == 3
    args = select_lastz_params(dist, cfg, 0).split(" ")
    assert args[0] == "--step=2"
    pm = miblast.params_from_args(args)
    po = olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_})
    want, single = "", []
    for a in ca:
        for b in cb:
            fa, fb = open(a, "rb").read(), open(b, "rb").read()
            paf = olz.align(fa, fb, po, details=False)["paf"]
            single.append((fa, fb, paf))
            want += "".join(chunking.paf_dechunk_line(l) for l in paf.decode().splitlines())
    assert got == want and got.count("\n") >= 6
    full = {"id=simT|chr20": t.tobytes().decode(), "id=simQ|chr20": q.tobytes().decode()}
    assert pafcheck.check_paf(got, full, full) == got.count("\n")
    sets = [(gpu_ctx.seqset_from_fasta_bytes(fa), gpu_ctx.seqset_from_fasta_bytes(fb)) for fa, fb, _ in single]
    batched = gpu_ctx.align_pairs(sets, pm)
    assert [r.paf for r in batched] == [p for _, _, p in single]
    for a, b in sets:
        a.close(); b.close()


@pytest.mark.parametrize("name,tf,qf,args", CASES, ids=CASE_IDS)
def test_case_matches_oracle_with_the_16_bit_diagonal_hash(gpu_ctx, olz, monkeypatch, name, tf, qf, args):
    """--miblast-diag=hash16 (lastz keys its suppression state by (t_end - q_end) & 0xFFFF, so a hit can be
    dropped because of an extension on a diagonal 65536 away): the MI355X path extends every hit and applies the rule per hash class
    in generation order (mb_hash16.h); bytes, HSPs in discovery order and counters as the oracle's in the same mode."""
    monkeypatch.setenv("MIBLAST_CHECK_ANCHORS", "1")
    pm = _params(list(args) + ["--miblast-diag=hash16"])
    assert pm.diag_hash16 == 1
    T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
    got = gpu_ctx.align(T, Q, pm)
    want = olz.align(tf, qf, _oracle_params(olz, pm))
    assert got.hsps == want["hsps"]
    assert got.paf == want["paf"]
    assert got.alns == want["alns"] and got.ops == want["ops"]
    for k in COUNTERS:
        assert got.stats[k] == want["counters"][k], k


def test_the_16_bit_diagonal_hash_drops_hits_across_diagonals_like_the_oracle(gpu_ctx, olz):
    """Sequences long enough for diagonals 65536 apart to collide (tandem copies of one segment 65536 bases apart, a batched call of
    several such pairs): the hash mode differs from the exact mode, and the MI355X path follows the oracle in both."""
    import numpy as np
    from cactus_amd import gen
    rng = np.random.default_rng(5)
    unit = gen.random_sequence(65536, rng)
    t = np.concatenate([unit, gen.mutate(unit, np.random.default_rng(6), 0.02, 0.0), unit[:30000]])
    q = np.concatenate([gen.mutate(unit[:50000], np.random.default_rng(7), 0.03, 0.001), gen.random_sequence(20000, rng)])
    fastas = [(gen.fasta_bytes([("T|a", t)]), gen.fasta_bytes([("Q|a", q)])), (gen.fasta_bytes([("T|b", t[::-1].copy()), ("T|c", unit)]), gen.fasta_bytes([("Q|b", q), ("Q|c", unit[1000:40000])]))]
    differs = 0
    for extra in ([], ["--miblast-diag=hash16"]):
        pm = _params(["--step=1", "--hspthresh=2200", "--gappedthresh=2400", "--ydrop=4000", "--ambiguous=iupac,100,100"] + extra)
        sets = [(gpu_ctx.seqset_from_fasta_bytes(a), gpu_ctx.seqset_from_fasta_bytes(b)) for a, b in fastas]
        got = gpu_ctx.align_pairs(sets, pm, details=True)
        for (a, b), r in zip(fastas, got):
            want = olz.align(a, b, _oracle_params(olz, pm))
            assert r.hsps == want["hsps"] and r.paf == want["paf"]
            for k in COUNTERS:
                assert r.stats[k] == want["counters"][k], (k, extra)
            differs += want["counters"]["hits_extended"] * (1 if extra else -1)
        for x, y in sets:
            x.close(); y.close()
    assert differs != 0                                          # the collision really happens on these inputs: the modes extend different numbers of hits


@pytest.mark.parametrize("extra", [["--miblast-walls"], ["--miblast-walls", "--miblast-diag=hash16"]], ids=["walls", "walls+hash16"])
@pytest.mark.parametrize("name,tf,qf,args", CASES, ids=CASE_IDS)
def test_case_matches_oracle_with_walls(gpu_ctx, olz, name, tf, qf, args, extra):
    """--miblast-walls (base pairs on the path of an earlier alignment of the unit are dead cells of later DPs),
    alone and together with the 16-bit diagonal hash: the DPs of a round run against the alignments their unit has committed (the
    WALLS variant of the 4-wave kernel: flagged ring columns, the horizontal-gap chain cut at every dead cell), results that ran
    against fewer walls are evaluated again.  Bytes, records, ops and counters as the oracle's in the same mode -- including the two
    low-complexity cases, where hundreds of earlier alignments cross a DP's rows and the mode changes the output."""
    pm = _params(list(args) + extra)
    assert pm.walls == 1
    T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
    got = gpu_ctx.align(T, Q, pm)
    want = olz.align(tf, qf, _oracle_params(olz, pm))
    assert got.paf == want["paf"]
    assert got.hsps == want["hsps"] and got.alns == want["alns"] and got.ops == want["ops"]
    for k in COUNTERS:
        assert got.stats[k] == want["counters"][k], k


@pytest.mark.parametrize("env", [{"MIBLAST_RELAY_S0": "0"}, {"MIBLAST_RELAY_S0": "64", "MIBLAST_RELAY_S": "256", "MIBLAST_RELAY_W": "64", "MIBLAST_RELAY_FORCE_REJECT": "3"},
                                 {"MIBLAST_GAPPED_BATCH_MAX": "1"}], ids=["no-relays", "short-relays-rejected", "one-anchor-per-round"])
def test_walls_do_not_depend_on_relays_or_speculation(gpu_ctx, olz, monkeypatch, env):
    """Walls with the relay hand-overs switched off, cut short with forced rejections, and with one anchor per round, on pairs whose
    alignments run across each other (tandem copies), in a batched call: the oracle's bytes and counters every time."""
    import numpy as np
    from cactus_amd import gen
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    def tandem(seed, period, copies, div):
        rng = np.random.default_rng(seed)
        unit = gen.random_sequence(period, rng)

        def rep(s2):
            r = np.random.default_rng(s2)
            return np.concatenate([gen.mutate(unit, r, div, 0.004) for _ in range(copies)])
        t = np.concatenate([gen.random_sequence(2000, rng), rep(seed + 1), gen.random_sequence(2000, rng)])
        q = np.concatenate([gen.random_sequence(1500, rng), rep(seed + 2), gen.random_sequence(1500, rng)])
        return gen.fasta_bytes([("T|s%d" % seed, t)]), gen.fasta_bytes([("Q|s%d" % seed, q)])

    # tandem copies with a period shorter than a DP window: later alignments run beside and across the paths of earlier ones
    fastas = [tandem(21, 60, 40, 0.05), tandem(22, 97, 30, 0.04)]
    pm = _params(["--step=1", "--hspthresh=2200", "--gappedthresh=2400", "--ydrop=4000", "--ambiguous=iupac,100,100", "--miblast-walls"])
    sets = [(gpu_ctx.seqset_from_fasta_bytes(a), gpu_ctx.seqset_from_fasta_bytes(b)) for a, b in fastas]
    got = gpu_ctx.align_pairs(sets, pm, details=True)
    plain = _params(["--step=1", "--hspthresh=2200", "--gappedthresh=2400", "--ydrop=4000", "--ambiguous=iupac,100,100"])
    changed = False
    for (a, b), r in zip(fastas, got):
        want = olz.align(a, b, _oracle_params(olz, pm))
        assert r.paf == want["paf"] and r.alns == want["alns"] and r.ops == want["ops"]
        for k in COUNTERS:
            assert r.stats[k] == want["counters"][k], k
        changed |= olz.align(a, b, _oracle_params(olz, plain), details=False)["counters"]["dp_cells"] != want["counters"]["dp_cells"]
    assert changed                                               # the walls matter on these inputs: dead cells change what the DPs evaluate
    for x, y in sets:
        x.close(); y.close()


def test_evolver_mammals_phase_at_full_size_equals_the_oracle_call_by_call(gpu_ctx, olz):
    """BASELINE configs[2] as a parity test at the size SURVEY 8d config 3 states (600 kb ancestor, seed 2001, the guide tree of
    examples/evolverMammals.txt:1): every one of the 20 lastz calls of the blast phase (cactus_amd/blast_phase.py: ingroup pairs
    + ingroup -> outgroup chains with trimming, option set per call by distance) goes through miblast_align_pairs as bench.py runs
    them and is diffed byte for byte against the oracle on the same FASTA bytes, counters included; the assembled per-node files
    (dechunk --query, invert) validate against the full sequences."""
    from cactus_amd import blast_phase as bp, gen, miblast, pafcheck
    from cactus_amd.paf.local_alignment import select_lastz_params
    from cactus_amd.shared.configWrapper import load_config
    cfg = load_config()
    calls = bp.blast_phase_calls(bp.parse_newick(bp.EVOLVER_MAMMALS_TREE))
    genomes = gen.make_tree_genomes(600_000, 2001, ancestors=True)
    fasta = {k: gen.fasta_bytes([("id=%s|%s" % (k, k), v)]) for k, v in genomes.items()}
    seen = []

    def align_batch(pairs, opts):
        pm = miblast.params_from_args(opts.split())
        sets = [(gpu_ctx.seqset_from_fasta_bytes(t), gpu_ctx.seqset_from_fasta_bytes(q)) for t, q in pairs]
        rs = gpu_ctx.align_pairs(sets, pm)
        for (t, q), r in zip(pairs, rs):
            seen.append((t, q, opts, r))
        for a, b in sets:
            a.close(); b.close()
        return [r.paf for r in rs]

    res = bp.run_blast_phase(fasta, calls, lambda d: select_lastz_params(d, cfg, 0), align_batch)
    assert len(seen) == 20
    cells = 0
    for t, q, opts, r in seen:
        pm = miblast.params_from_args(opts.split())
        want = olz.align(t, q, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}), details=False)
        assert r.paf == want["paf"]
        for k in ("seed_hits", "hits_extended", "ungapped_cols", "hsps", "anchors", "anchors_skipped", "dp_sides", "dp_cells", "dp_rows", "alignments"):
            assert r.stats[k] == want["counters"][k], k
        cells += r.stats["dp_cells"]
    assert cells > 5e8
    full = {}
    for fa in fasta.values():
        for name, seq in bp.parse_fasta_bytes(fa):
            full[name] = seq.tobytes().decode()
    checked = sum(pafcheck.check_paf(parts[kind].decode(), full, full) for parts in res.values() for kind in ("ingroup", "outgroup") if parts[kind])
    assert checked >= 100


@pytest.mark.gpu
def test_contexts_of_any_priority_give_the_same_bytes(olz):
    """miblast_ctx_set_priority: a context whose launches yield to (or go before) those of the device's other contexts, with the lanes
    of its batched calls, computes what every context computes -- also while a call runs on another context of the device."""
    import threading
    from cases import DEFAULT, pair
    from cactus_amd import miblast
    pm = _params(DEFAULT)
    cases = [pair(40000, 31), pair(60000, 33), pair(30000, 35), pair(50000, 37)]
    want = [olz.align(tf, qf, _oracle_params(olz, pm))["paf"] for tf, qf in cases]
    low, high = miblast.Context(0).set_priority(-1), miblast.Context(0).set_priority(1)
    got = {}

    def run(name, cx):
        sets = [(cx.seqset_from_fasta_bytes(tf), cx.seqset_from_fasta_bytes(qf)) for tf, qf in cases]
        got[name] = [r.paf for r in cx.align_pairs(sets, pm)] + [cx.align(sets[0][0], sets[0][1], pm).paf]
        for t, q in sets:
            t.close(); q.close()
    th = [threading.Thread(target=run, args=(n, c)) for n, c in (("low", low), ("high", high))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got["low"] == want + [want[0]] and got["high"] == want + [want[0]]
    low.set_priority(0)
    run("again", low)
    assert got["again"] == want + [want[0]]
    low.close(); high.close()


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--miblast-xdrop=le"], ["--miblast-hspbest-ties=later"], ["--miblast-xdrop=le", "--miblast-hspbest-ties=later"]],
                         ids=["xdrop_le", "hspbest_ties", "both"])
def test_the_cheap_a9_switches_on_the_device_equal_the_oracles(gpu_ctx, olz, monkeypatch, extra):
    """Round 5: SURVEY A.9 #9 (a walk stops at run <= best - xdrop) and #11 (--queryhspbest keeps the later found of equal scores) are
    switches of the MI355X path too (miblast_params.xdrop_le / hspbest_ties, --miblast-xdrop=le / --miblast-hspbest-ties=later): on every
    case, with every ungapped kernel, the oracle's bytes, HSP list and counters under the same switches; on the two cases built so that
    the switch changes the result (tests/test_oracle_cpu.py), the change itself; and the two oracle-only switches are refused, not ignored."""
    import numpy as np
    from cactus_amd import gen, miblast
    for kernel in ("lane", "ux"):
        monkeypatch.setenv("MIBLAST_UNGAPPED", kernel)
        for name, tf, qf, args in CASES:
            pm = _params(list(args) + extra)
            T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
            got = gpu_ctx.align(T, Q, pm)
            want = olz.align(tf, qf, _oracle_params(olz, pm))
            assert got.paf == want["paf"], (name, kernel)
            assert got.hsps == want["hsps"], (name, kernel)
            for k in COUNTERS:
                assert got.stats[k] == want["counters"][k], (name, kernel, k)
            T.close(); Q.close()
    monkeypatch.delenv("MIBLAST_UNGAPPED")
    rng = np.random.default_rng(11)
    rnd = lambda n, seed: gen.random_sequence(n, np.random.default_rng(seed)).tobytes().decode()
    fa = lambda name, s: (">%s\n%s\n" % (name, s)).encode()
    base = ["--hspthresh=2200", "--gappedthresh=2400", "--ydrop=4000", "--ungapped", "--format=general:name1,zstart1,end1,name2,zstart2+,end2+"]
    # the dip of exactly x-drop (six N columns + ten transitions = -910) between two stretches: one HSP across it with "<", none with "<="
    left, right = rnd(260, 24), rnd(400, 25)
    tv = {"A": "G", "G": "A", "C": "T", "T": "C"}
    mid_q = "".join(rng.choice(list("ACGT"), 16))
    mid_t = "N" * 6 + "".join(tv[c] for c in mid_q[6:])
    tf, qf = fa("T|x", left + mid_t + right), fa("Q|x", left + mid_q + right)
    outs = {}
    for label, more in (("lt", []), ("le", ["--miblast-xdrop=le"])):
        pm = _params(base + more)
        T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
        outs[label] = gpu_ctx.align(T, Q, pm)
        assert outs[label].paf == olz.align(tf, qf, _oracle_params(olz, pm))["paf"]
    assert outs["lt"].paf != outs["le"].paf and len(outs["le"].hsps) > len(outs["lt"].hsps)
    # two equal-scoring HSPs under --queryhspbest=1: the first found, or the later found
    unit = rnd(120, 26)
    tf = fa("T|k", rnd(300, 27) + unit + rnd(300, 28) + unit + rnd(300, 29))
    qf = fa("Q|k", rnd(50, 30) + "N" * 20 + unit + "N" * 20 + rnd(50, 31))
    kept = {}
    for label, more in (("earlier", []), ("later", ["--miblast-hspbest-ties=later"])):
        pm = _params(base + ["--queryhspbest=1"] + more)
        T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
        kept[label] = gpu_ctx.align(T, Q, pm)
        want = olz.align(tf, qf, _oracle_params(olz, pm))
        assert kept[label].paf == want["paf"] and kept[label].hsps == want["hsps"] and len(want["hsps"]) == 1
    assert kept["earlier"].hsps != kept["later"].hsps
    # oracle-only switches: refused
    for field in ("query_softmask", "step_origin"):
        pm = _params(base)
        setattr(pm, field, 1)
        T, Q = gpu_ctx.seqset_from_fasta_bytes(tf), gpu_ctx.seqset_from_fasta_bytes(qf)
        with pytest.raises(miblast.MiblastError):
            gpu_ctx.align(T, Q, pm)


@pytest.mark.parametrize("seed", [22003, 22005, 23008, 23024])
def test_chunk_scale_random_cases_against_committed_oracle_digests(seed):
    """Chunk-scale cases of the randomised differential run (scripts/gpu_fuzz.py: 1.7 - 2.6 Mb pairs of every structure, random option sets) against
    tests/golden/fuzz_chunk_r06.json -- the oracle's answers (md5 of the PAF, twelve counters, record counts), made beforehand by the same script with
    FUZZ_WRITE_DIGESTS on the CPU box (50 cases, 40 CPU-minutes; scripts/gpu_r6_fuzz3.sh runs them all): near-identity with 20 000 alignments, N runs
    under --ydrop=20000, 7 - 11 million seed hits under --step=1."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FUZZ_NMIN="1700000", FUZZ_NMAX="2600000", FUZZ_READ_DIGESTS=os.path.join(root, "tests", "golden", "fuzz_chunk_r06.json"))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_fuzz.py"), "1", str(seed)], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "1 cases, 0 mismatches" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]

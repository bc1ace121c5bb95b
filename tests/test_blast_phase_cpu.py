"""CPU tests of the blast-phase schedule (cactus_amd/blast_phase.py): which lastz calls a progressive run makes and with which
option sets -- mirror of /root/reference/src/cactus/paf/paf.py:29-71 and local_alignment.py:751-858, 421-526 -- and the data flow
of the ingroup->outgroup chain, run here with the CPU oracle as the aligner (the GPU suite runs the same driver on the MI355X
and diffs every call)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from cactus_amd import blast_phase as bp
from cactus_amd import gen, pafcheck

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_distances_are_tree_path_lengths():
    t = bp.parse_newick(bp.EVOLVER_MAMMALS_TREE)
    d = bp.get_distances(t)
    assert abs(d[("simMouse_chr6", "simRat_chr6")] - 0.176098) < 1e-9
    assert abs(d[("simHuman_chr6", "simMouse_chr6")] - (0.144018 + 0.271974 + 0.084509)) < 1e-9
    assert abs(d[("simCow_chr6", "simMouse_chr6")] - (0.18908 + 0.032898 + 0.020593 + 0.271974 + 0.084509)) < 1e-9
    assert d[("mr", "mr")] == 0.0 and d[("Anc0", "simDog_chr6")] == d[("simDog_chr6", "Anc0")]
    pairs = [(a.iD, b.iD) for a, b, _ in bp.get_event_pairs(t, t.leaves())]
    assert len(pairs) == 10 and pairs[0] == ("simHuman_chr6", "simMouse_chr6")          # list order, i < j (paf.py:61-71)
    unnamed = bp.parse_newick("((a:1,b:2):3,(c:4,d:5):6);")
    assert [n.iD for n in unnamed.subtree() if n.children] == ["Anc0", "Anc1", "Anc2"]


def test_ancestor_upweighting_follows_progressive_decomposition():
    # progressive_decomposition.py:231-239: height added to the branch above an ancestor, capped at max_div, long branches untouched
    s = bp.ancestor_scaled_tree(bp.parse_newick(bp.EVOLVER_MAMMALS_TREE), 0.25)
    assert abs(bp.get_node(s, "mr").distance - 0.271974) < 1e-12                      # already >= 0.25
    assert abs(bp.get_node(s, "Anc1").distance - 0.25) < 1e-12                        # 0.020593 + 0.363563 capped
    assert abs(bp.get_node(s, "Anc2").distance - (0.032898 + 0.18908)) < 1e-12
    assert abs(bp.get_node(s, "simHuman_chr6").distance - 0.144018) < 1e-12           # leaves untouched


def test_evolver_mammals_schedule_is_appendix_d():
    from cactus_amd.paf.local_alignment import select_lastz_params
    from cactus_amd.shared.configWrapper import load_config
    cfg = load_config()
    calls = bp.blast_phase_calls(bp.parse_newick(bp.EVOLVER_MAMMALS_TREE))
    assert len(calls) == 20 <= 28 and [c.node for c in calls if c.kind == "ingroup"] == ["mr", "Anc1", "Anc2", "Anc0"]
    sets = {(c.target, c.query): select_lastz_params(c.distance, cfg, 0) for c in calls}
    assert sets[("simMouse_chr6", "simRat_chr6")].startswith("--step=3 ") and "--hspthresh=2600" in sets[("simMouse_chr6", "simRat_chr6")]      # "four"
    assert all(v.startswith("--step=1 ") for k, v in sets.items() if k != ("simMouse_chr6", "simRat_chr6"))                                # "default"
    mr_mouse = [c for c in calls if c.chain == ("mr", "simMouse_chr6")]
    assert [(c.target, c.level) for c in mr_mouse] == [("simHuman_chr6", 0), ("simDog_chr6", 1), ("simCow_chr6", 2)]      # nearest outgroup first (:818)
    assert all(c.query == "simMouse_chr6" for c in mr_mouse)                                                              # outgroup = file A, ingroup = file B (:444-448)
    assert not [c for c in calls if c.node == "Anc0" and c.kind == "outgroup"]
    prim = bp.blast_phase_calls(bp.parse_newick(bp.EVOLVER_PRIMATES_TREE))
    assert open("/root/reference/examples/evolverPrimates.txt").readline().strip() == bp.EVOLVER_PRIMATES_TREE if os.path.exists("/root/reference/examples/evolverPrimates.txt") else True
    assert [c.node for c in prim if c.kind == "ingroup"] == ["cb", "hcb", "Anc0"] and len(prim) == 9 <= 21
    # every pair of evolverPrimates is within divergence "one" -> --step=2 ... --notransition --queryhspbest=100000
    assert all(select_lastz_params(c.distance, cfg, 0).startswith("--step=2 ") and "--notransition" in select_lastz_params(c.distance, cfg, 0) for c in prim)


def test_phase_driver_data_flow_with_the_oracle(olz):
    """20 kb stand-in: every call of the phase through the oracle; the assembled per-node PAFs validate against the FULL
    sequences (cigars, coordinates after two levels of `dechunk --query`, inversion), later outgroups only see what earlier ones
    left unaligned, and equal-option calls of a level arrive as one batch."""
    from cactus_amd import miblast
    from cactus_amd.paf.local_alignment import select_lastz_params
    from cactus_amd.shared.configWrapper import load_config
    cfg = load_config()
    calls = bp.blast_phase_calls(bp.parse_newick(bp.EVOLVER_MAMMALS_TREE))
    genomes = gen.make_tree_genomes(20_000, 2001, ancestors=True)
    fasta = {k: gen.fasta_bytes([("id=%s|%s" % (k, k), v)]) for k, v in genomes.items()}
    batches, seen = [], []

    def align_batch(pairs, opts):
        batches.append((len(pairs), opts))
        pm = miblast.params_from_args(opts.split())
        po = olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_})
        return [olz.align(t, q, po, details=False)["paf"] for t, q in pairs]

    res = bp.run_blast_phase(fasta, calls, lambda d: select_lastz_params(d, cfg, 0), align_batch,
                             on_call=lambda c, tf, qf, paf: seen.append((c, len(qf), paf)))
    assert set(res) == {"mr", "Anc1", "Anc2", "Anc0"} and batches[0][0] + batches[1][0] == 10       # level 0: 1 x "four" + 9 x "default"
    full = {}
    for fa in fasta.values():
        for name, seq in bp.parse_fasta_bytes(fa):
            full[name] = seq.tobytes().decode()
    n_checked = 0
    for node, parts in res.items():
        for kind in ("ingroup", "outgroup"):
            if parts[kind]:
                n_checked += pafcheck.check_paf(parts[kind].decode(), full, full)
    assert n_checked >= 20
    # an outgroup call of level k+1 is handed less sequence than level k, and only sub-sequence records (NAME|LEN|START)
    by_chain = {}
    for c, qlen, paf in seen:
        if c.chain:
            by_chain.setdefault(c.chain, []).append((c.level, qlen))
    assert all(sorted(v)[0][1] > sorted(v)[-1][1] for v in by_chain.values() if len(v) > 1)
    # inverted: in the outgroup file the ingroup is the TARGET (local_alignment.py:427)
    for line in res["mr"]["outgroup"].decode().splitlines():
        f = line.split("\t")
        assert f[5] in ("id=simMouse_chr6|simMouse_chr6", "id=simRat_chr6|simRat_chr6") and int(f[6]) in (len(genomes["simMouse_chr6"]), len(genomes["simRat_chr6"]))


def test_unaligned_fasta_nests_subsequence_names():
    rng = np.random.default_rng(3)
    seq = gen.random_sequence(1000, rng)
    fa = gen.fasta_bytes([("id=A|c", seq)])
    paf = b"id=A|c\t1000\t100\t400\t+\tt\t9\t0\t1\t1\t1\t255\n"
    left = bp.unaligned_fasta(paf, fa, 100, 50)
    recs = bp.parse_fasta_bytes(left)
    assert [n for n, _ in recs] == ["id=A|c|1000|0", "id=A|c|1000|350"] and [len(s) for _, s in recs] == [150, 650]
    paf2 = b"id=A|c|1000|350\t650\t0\t300\t-\tt\t9\t0\t1\t1\t1\t255\n"
    again = bp.parse_fasta_bytes(bp.unaligned_fasta(paf2, left, 100, 0))
    assert [n for n, _ in again] == ["id=A|c|1000|0|150|0", "id=A|c|1000|350|650|300"]
    assert bp.dechunk_query(paf2).split(b"\t")[:4] == [b"id=A|c", b"1000", b"350", b"650"]
    assert bp.unaligned_fasta(b"id=A|c\t1000\t0\t1000\t+\tt\t9\t0\t1\t1\t1\t255\n", fa, 100, 50) == b""


def test_native_trimming_equals_the_python_cores():
    """mipaf_unaligned_fasta (host-side text code of the library) against paf.chunking.unaligned_intervals + extract_records +
    60-column FASTA, the mirror of `paffy to_bed --excludeAligned --minSize` + `faffy extract --flank`: random multi-record files,
    overlapping / nested / touching alignments, alignments on both strands, records without a hit, empty PAF, empty records,
    flanks that make neighbouring stretches merge, CRLF line ends; an unknown query name is refused by both."""
    from cactus_amd import miblast
    rng = np.random.default_rng(11)
    for case in range(60):
        n_rec = int(rng.integers(1, 5))
        recs = [("id=G|r%d" % k if k % 2 else "r%d|77|5" % k, gen.random_sequence(int(rng.integers(0, 1500)) if case % 7 else 0, rng)) for k in range(n_rec)]
        fa = gen.fasta_bytes(recs)
        lines = []
        for _ in range(int(rng.integers(0, 12)) if case % 5 else 0):
            name, seq = recs[int(rng.integers(0, n_rec))]
            if len(seq) < 2:
                continue
            s0 = int(rng.integers(0, len(seq) - 1))
            e0 = int(rng.integers(s0, len(seq) + 1))                 # (now and then an empty interval)
            lines.append("%s\t%d\t%d\t%d\t%s\tT\t5000\t1\t%d\t1\t1\t255\tcg:Z:%d=" % (name, len(seq), s0, e0, "+-"[int(rng.integers(0, 2))], 1 + e0 - s0, e0 - s0))
        paf = ("\n".join(lines) + ("\n" if lines else "")).encode()
        if case % 9 == 4:
            paf, fa = paf.replace(b"\n", b"\r\n"), fa.replace(b"\n", b"\r\n")
        for min_size, flank in ((100, 100), (1, 0), (37, 500), (400, 3)):
            want = bp.unaligned_fasta_py(paf, fa, min_size, flank)
            assert bp.unaligned_fasta(paf, fa, min_size, flank) == want, (case, min_size, flank)
    with pytest.raises(miblast.MiblastError, match="not in the FASTA"):
        bp.unaligned_fasta(b"nobody\t10\t0\t5\t+\tT\t9\t0\t5\t5\t5\t255\n", gen.fasta_bytes([("a", gen.random_sequence(10, rng))]), 1, 0)
    with pytest.raises(KeyError):
        bp.unaligned_fasta_py(b"nobody\t10\t0\t5\t+\tT\t9\t0\t5\t5\t5\t255\n", gen.fasta_bytes([("a", gen.random_sequence(10, rng))]), 1, 0)


def test_bench_refuses_a_wrong_world_size_and_spawns_ranks():
    # no GPU here: the spawned ranks must fail loudly (no CPU path), and the launcher's exit code must say so
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       env=dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), cwd=ROOT, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr
    from cactus_amd import miblast
    if miblast.device_count() == 0:
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
        assert p.returncode != 0 and p.stderr.count("needs a GPU") == 2 and "{" not in p.stdout


def test_calls_in_flight_never_exceed_the_aligners_width():
    """a level with more option-set groups than the aligner has contexts (three sets, concurrent = 2, plus the split
    halves) must not have more than `concurrent` align_batch calls in flight."""
    import threading
    import time
    tree = bp.parse_newick("((a:0.01,b:0.01)x:0.01,(c:0.09,d:0.09)y:0.01,(e:0.2,f:0.2)z:0.01)r;")
    calls = bp.blast_phase_calls(tree, max_outgroups=0)
    fasta = {n.iD: b">%s\nACGT\n" % n.iD.encode() for n in tree.subtree()}
    state = {"now": 0, "peak": 0, "free": ["ctx0", "ctx1"]}
    lock = threading.Lock()

    def align_batch(pairs, opts):
        with lock:
            state["free"].pop()                                     # IndexError here = the bug
            state["now"] += 1
            state["peak"] = max(state["peak"], state["now"])
        time.sleep(0.05)
        with lock:
            state["now"] -= 1
            state["free"].append("ctx")
        return [b"" for _ in pairs]

    align_batch.concurrent = 2
    align_batch.split_above = 1
    opts = lambda d: "set-%d" % (0 if d < 0.05 else 1 if d < 0.3 else 2)      # noqa: E731
    assert len({opts(c.distance) for c in calls if c.level == 0}) == 3
    bp.run_blast_phase(fasta, calls, opts, align_batch)
    assert state["peak"] == 2


def test_width_holds_with_free_jobs_in_flight_and_trimming():
    """with concurrent >= 3 the ingroup pairs go out as jobs nothing waits for; the chains' single-group levels and the
    trim_resident calls must pass the same gate -- never more than `concurrent` calls on the aligner at a time -- and a level that
    raises must not leave free jobs behind."""
    import threading
    import time
    calls = bp.blast_phase_calls(bp.parse_newick(bp.EVOLVER_MAMMALS_TREE))
    fasta = {n: b">%s\nACGT\n" % n.encode() for n in {c.target for c in calls} | {c.query for c in calls}}
    width = 3
    state = {"now": 0, "peak": 0, "free": ["ctx"] * width, "trims": 0}
    lock = threading.Lock()

    def enter():
        with lock:
            state["free"].pop()                                     # IndexError here = more calls than contexts
            state["now"] += 1
            state["peak"] = max(state["peak"], state["now"])

    def leave():
        with lock:
            state["now"] -= 1
            state["free"].append("ctx")

    def align_batch(pairs, opts):
        enter()
        time.sleep(0.03 if any(q == b"left" for _, q in pairs) else 0.12)     # (ingroup / level-0 calls are the slow ones: still out when level 1 starts)
        leave()
        return [b"x" for _ in pairs]

    class Left(bytes):
        def fasta_bytes(self):
            return bytes(self)

    def trim_resident(items, min_size, flank):
        enter()
        state["trims"] += 1
        time.sleep(0.03)
        leave()
        return [Left(b"left") for _ in items]

    align_batch.concurrent = width
    align_batch.trim_resident = trim_resident
    opts = lambda d: "set-%d" % (0 if d < 0.2 else 1)      # noqa: E731
    import cactus_amd.blast_phase as mod
    saved = (mod.invert, mod.dechunk_query)
    mod.invert = mod.dechunk_query = lambda paf: paf       # (the text steps are not under test: the PAFs here are not PAF)
    try:
        bp.run_blast_phase(fasta, calls, opts, align_batch)
        assert state["trims"] >= 2 and state["peak"] <= width and state["now"] == 0

        def failing(pairs, opts):
            if any(q == b"left" for _, q in pairs):
                raise RuntimeError("boom")
            return align_batch(pairs, opts)
        failing.concurrent = width
        failing.trim_resident = trim_resident
        with pytest.raises(RuntimeError):
            bp.run_blast_phase(fasta, calls, opts, failing)
        assert state["now"] == 0                                    # nothing of the failed run is still on the aligner
    finally:
        mod.invert, mod.dechunk_query = saved


def test_jobs_nothing_waits_for_run_beside_the_chains(olz):
    """An aligner that takes three calls at a time: the ingroup pairs (only the final files take their output) are handed over as
    jobs of their own and level 1 starts before they are back; the phase's files equal those of the one-call-at-a-time run."""
    import threading
    from cactus_amd import miblast
    from cactus_amd.paf.local_alignment import select_lastz_params
    from cactus_amd.shared.configWrapper import load_config
    cfg = load_config()
    calls = bp.blast_phase_calls(bp.parse_newick(bp.EVOLVER_MAMMALS_TREE))
    genomes = gen.make_tree_genomes(6_000, 2002, ancestors=True)
    fasta = {k: gen.fasta_bytes([("id=%s|%s" % (k, k), v)]) for k, v in genomes.items()}
    ingroup_only = threading.Event()            # set when a call made of ingroup pairs alone has returned
    log, lock = [], threading.Lock()
    by_pair = {(fasta[c.target], fasta[c.query]): c for c in calls if c.level == 0}

    def align_batch(pairs, opts):
        kinds = {by_pair[(t, q)].kind if (t, q) in by_pair else "later" for t, q in pairs}
        if kinds == {"ingroup"} and len(pairs) > 1:
            assert ingroup_only.wait(60)                                  # held until a level-1 call has been seen
        if "later" in kinds:
            ingroup_only.set()
        with lock:
            log.append(sorted(kinds))
        pm = miblast.params_from_args(opts.split())
        po = olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_})
        return [olz.align(t, q, po, details=False)["paf"] for t, q in pairs]

    choose = lambda d: select_lastz_params(d, cfg, 0)      # noqa: E731
    plain = bp.run_blast_phase(fasta, calls, choose, align_batch)
    assert ["ingroup", "outgroup"] in log                                 # width 1: a level is one call per option set
    del log[:]
    ingroup_only.clear()
    align_batch.concurrent = 3
    wide = bp.run_blast_phase(fasta, calls, choose, align_batch)
    assert ["ingroup"] in log and ["ingroup", "outgroup"] not in log and wide == plain

"""Chaining stage: the C oracle (oracle/paffy_oracle.c) against hand-worked cases and against the naive
restatement in tests/pyref_paffy.py.  No reference vectors exist for this stage (paffy is an absent submodule): PARITY UNPINNED."""
import os
import subprocess

import pytest

from tests import pyref_paffy as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "oracle_paffy")

CHAIN_ARGS = ["--maxGapLength", "1000000", "--chainGapOpen", "5000", "--chainGapExtend", "1", "--trimFraction", "1.0"]   # xml:108-111


def oracle(cmd, text, *args):
    p = subprocess.run([ORACLE, cmd, *args], input=text.encode(), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode()


def L(qn, qs, qe, strand, tn, ts, te, score=None, cg=None, ql=100000, tl=100000, extra=()):
    c = [qn, ql, qs, qe, strand, tn, tl, ts, te, qe - qs, qe - qs, 255]
    if score is not None:
        c.append(f"AS:i:{score}")
    c += list(extra)
    if cg:
        c.append("cg:Z:" + cg)
    return "\t".join(str(x) for x in c) + "\n"


def test_chain_joins_colinear_alignments_and_scores_the_gap():
    # midpoints (trimFraction 1.0): a 150/1150, b 1450/2460 -> gaps 1300 + 1310, cost 5000 + 2610
    text = L("q", 100, 200, "+", "t", 1100, 1200, 9000, "100=") + L("q", 1400, 1500, "+", "t", 2410, 2510, 8000, "100=")
    out = ref.parse(oracle("chain", text, *CHAIN_ARGS))
    assert [(r.qs, r.cn, r.s1) for r in out] == [(100, 0, 9000 + 8000 - 7610), (1400, 0, 9000 + 8000 - 7610)]


def test_chain_does_not_join_across_strand_target_or_max_gap():
    a = L("q", 100, 200, "+", "t", 1100, 1200, 9000, "100=")
    far = L("q", 1400, 1500, "+", "t", 1_200_000, 1_200_100, 8000, "100=", tl=2_000_000)
    other_strand = L("q", 1400, 1500, "-", "t", 2410, 2510, 8000, "100=")
    other_target = L("q", 1400, 1500, "+", "u", 2410, 2510, 8000, "100=")
    for b in (far, other_strand, other_target):
        out = ref.parse(oracle("chain", a + b.replace("\t100000\t1100", "\t2000000\t1100"), *CHAIN_ARGS))
        assert sorted((r.s1, r.cn) for r in out) == [(8000, 1), (9000, 0)]


def test_chain_minus_strand_runs_down_the_target():
    # '-' strand: the query goes up while the target goes down
    text = L("q", 100, 200, "-", "t", 2410, 2510, 9000, "100=") + L("q", 1400, 1500, "-", "t", 1100, 1200, 8000, "100=")
    out = ref.parse(oracle("chain", text, *CHAIN_ARGS))
    assert {r.cn for r in out} == {0} and out[0].s1 == 9000 + 8000 - 5000 - (1450 - 150) - (2460 - 1150)
    # the same boxes in '+' orientation are not colinear
    out = ref.parse(oracle("chain", text.replace("\t-\t", "\t+\t"), *CHAIN_ARGS))
    assert {r.cn for r in out} == {0, 1}


def test_chain_peels_the_best_chain_first_and_stops_at_claimed_members():
    # a -> b and a -> c both chain; (a, b) scores higher so it is chain 0; c keeps its DP score but stands alone
    a = L("q", 0, 100, "+", "t", 0, 100, 20000, "100=")
    b = L("q", 1000, 1100, "+", "t", 1000, 1100, 9000, "100=")
    c = L("q", 1200, 1300, "+", "t", 900, 1000, 7000, "100=")
    out = ref.parse(oracle("chain", c + b + a, *CHAIN_ARGS))
    assert [(r.qs, r.cn, r.s1) for r in out] == [(0, 0, 20000 + 9000 - 5000 - 2000), (1000, 0, 20000 + 9000 - 5000 - 2000),
                                                 (1200, 1, 20000 + 7000 - 5000 - 1200 - 900)]


def test_tile_levels_are_one_plus_the_median_cover():
    best = L("q", 0, 100, "+", "t", 0, 100, 9000, "100=")
    half = L("q", 50, 150, "+", "u", 0, 100, 8000, "100=")           # 50 of 100 bases already covered: 2 * 50 >= 100 -> median 0
    most = L("q", 40, 140, "+", "v", 0, 100, 7000, "100=")           # 40..139 is covered on all but ... every base once or twice: median >= 1
    out = ref.parse(oracle("tile", most + half + best))
    assert [(r.tn, r.tile, r.tp) for r in out] == [("t", 1, "P"), ("u", 1, "P"), ("v", 2, "S")]
    # gaps and the '-' strand: only = X M columns count, read down the query
    rev = L("q", 0, 100, "-", "w", 0, 90, 100, "40=10D20I40=")        # covers 60..99 and 0..39
    out = ref.parse(oracle("tile", best + rev + L("q", 40, 60, "+", "z", 0, 20, 50, "20=")))
    assert [(r.tn, r.tile) for r in out] == [("t", 1), ("w", 2), ("z", 2)]


def test_trim_cuts_the_longest_low_identity_prefix_and_suffix():
    # identity after 1= 10X is 1/11 < 0.2 ; in the next run (1+t)/(11+t) < 0.2 <=> t < 1.5 -> one more column: cut 12
    text = L("q", 1000, 1111, "+", "t", 2000, 2111, 5000, "1=10X100=")
    r, = ref.parse(oracle("trim", text, "--trimIdentity", "0.2"))
    assert (r.qs, r.qe, r.ts, r.te, r.ops, r.nm, r.nb) == (1012, 1111, 2012, 2111, [(99, "=")], 99, 99)
    # '-' strand: the END of the op list sits at the START of the query interval; gaps move one axis only.  Reversed list:
    # 1= 5X 5D 20I -> 1/31, then (1+t)/(31+t) < 0.2 <=> t < 6.5: 6 more columns; 32 query and 17 target bases go
    text = L("q", 1000, 1126, "-", "t", 2000, 2111, 5000, "100=20I5D5X1=")
    r, = ref.parse(oracle("trim", text, "--trimIdentity", "0.2"))
    assert (r.qs, r.qe, r.ts, r.te, r.ops) == (1032, 1126, 2000, 2094, [(94, "=")])
    # nothing to cut / everything cut
    clean = L("q", 0, 50, "+", "t", 0, 50, 100, "50=")
    assert oracle("trim", clean, "--trimIdentity", "0.2") == clean
    assert oracle("trim", L("q", 0, 50, "+", "t", 0, 50, 100, "50X"), "--trimIdentity", "0.2") == ""


def test_filter_and_invert_and_tags_roundtrip():
    text = (L("q", 0, 10, "+", "t", 0, 10, 5, "10=", extra=("tp:A:P", "tl:i:1", "cn:i:0", "s1:i:20000")) +
            L("q", 0, 10, "+", "t", 0, 10, 5, "10=", extra=("tp:A:S", "tl:i:2", "cn:i:1", "s1:i:500")))
    # tags come back in paffy's order: tp, AS, tl, cn, s1, cg
    assert oracle("filter", text, "--maxTileLevel", "1").split("\t")[12:] == ["tp:A:P", "AS:i:5", "tl:i:1", "cn:i:0", "s1:i:20000", "cg:Z:10=\n"]
    assert len(oracle("filter", text, "--maxTileLevel", "1", "--invert").splitlines()) == 1
    assert len(oracle("filter", text, "--minChainScore", "10000").splitlines()) == 1
    assert oracle("filter", text, "--minChainScore", "100") == oracle("filter", text)
    inv = oracle("invert", L("q", 5, 30, "-", "t", 100, 120, 7, "10=5I5X5="))
    r, = ref.parse(inv)
    assert (r.qn, r.qs, r.qe, r.tn, r.ts, r.te, r.ops) == ("t", 100, 120, "q", 5, 30, [(5, "="), (5, "X"), (5, "D"), (10, "=")])
    assert oracle("invert", inv) == L("q", 5, 30, "-", "t", 100, 120, 7, "10=5I5X5=")


def test_split_file_groups_query_sequences(tmp_path):
    text = "".join(L(f"q{k % 3}", 0, 10, "+", "t", 0, 10, 5, "10=", ql=1000 * (k % 3 + 1)) for k in range(7))
    src = tmp_path / "in.paf"
    src.write_text(text)
    subprocess.run([ORACLE, "split_file", "--inputFile", str(src), "--query", "--prefix", str(tmp_path / "split_"), "--minLength", "2500"], check=True)
    parts = sorted(p.name for p in tmp_path.glob("split_*.paf"))
    assert parts == ["split_0.paf", "split_1.paf"]
    assert {l.split("\t")[0] for l in (tmp_path / "split_0.paf").read_text().splitlines()} == {"q0", "q1"}
    assert {l.split("\t")[0] for l in (tmp_path / "split_1.paf").read_text().splitlines()} == {"q2"}


@pytest.mark.parametrize("seed", range(12))
def test_oracle_equals_naive_restatement_on_random_sets(seed):
    text = ref.random_paf(seed, n_series=5 + seed % 4, noise=8 + seed, contig_len=60_000 if seed % 2 else 200_000)
    text += ref.dump(ref.invert(ref.parse(text)))                          # chain_alignments feeds both orientations (:620-626)
    assert oracle("invert", text) == ref.dump(ref.invert(ref.parse(text)))
    chained = oracle("chain", text, *CHAIN_ARGS)
    assert chained == ref.dump(ref.chain(ref.parse(text), 1_000_000, 5000, 1, 1.0))
    small_gap = ["--maxGapLength", "3000", "--chainGapOpen", "100", "--chainGapExtend", "3", "--trimFraction", "0.25"]
    assert oracle("chain", text, *small_gap) == ref.dump(ref.chain(ref.parse(text), 3000, 100, 3, 0.25))
    tiled = oracle("tile", chained)
    assert tiled == ref.dump(ref.tile(ref.parse(chained)))
    for x in ("0.2", "0.5", "0.97", "0", "1"):
        assert oracle("trim", tiled, "--trimIdentity", x) == ref.dump(ref.trim(ref.parse(tiled), x)), x
    trimmed = oracle("trim", tiled, "--trimIdentity", "0.2")
    prim = oracle("filter", trimmed, "--maxTileLevel", "1")
    assert prim == ref.dump(ref.filt(ref.parse(trimmed), max_tile=1))
    assert oracle("filter", trimmed, "--maxTileLevel", "1", "--invert") == ref.dump(ref.filt(ref.parse(trimmed), max_tile=1, invert_=True))
    rechained = oracle("chain", prim, *CHAIN_ARGS)
    assert rechained == ref.dump(ref.chain(ref.parse(prim), 1_000_000, 5000, 1, 1.0))
    assert oracle("filter", rechained, "--minChainScore", "10000") == ref.dump(ref.filt(ref.parse(rechained), min_chain=10000))


GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_STEPS = [("chain", "input", CHAIN_ARGS), ("tile", "chain", []), ("trim", "tile", ["--trimIdentity", "0.2"]), ("filter", "trim", ["--maxTileLevel", "1"]),
                ("chain", "primary", CHAIN_ARGS), ("filter", "rechain", ["--minChainScore", "10000"])]
GOLDEN_OUT = ["chain", "tile", "trim", "primary", "rechain", "output"]


def golden(name):
    return open(os.path.join(GOLDEN, f"chain_{name}.paf")).read()


def test_oracle_and_naive_restatement_reproduce_the_committed_chain_fixtures():
    # tests/golden/chain_*.paf (make_chain_golden.py): regression pins of the rules, step by step
    for (cmd, src, args), dst in zip(GOLDEN_STEPS, GOLDEN_OUT):
        assert oracle(cmd, golden(src), *args) == golden(dst), (cmd, src)
    recs = ref.parse(golden("input"))
    assert ref.dump(ref.chain(recs, 1_000_000, 5000, 1, 1.0)) == golden("chain")
    assert ref.dump(ref.tile(ref.parse(golden("chain")))) == golden("tile")
    assert ref.dump(ref.trim(ref.parse(golden("tile")), "0.2")) == golden("trim")
    assert ref.dump(ref.filt(ref.parse(golden("trim")), max_tile=1)) == golden("primary")
    assert ref.dump(ref.filt(ref.parse(golden("rechain")), min_chain=10000)) == golden("output")

"""CPU tests of the oracle itself (test infrastructure): hand-derived known answers, agreement with the
independent pure-Python restatement of SURVEY.md A.10, the committed golden fixtures, and the
reference's only pins for this path (parameter strings, PAF validity contract)."""
import json
import os

import numpy as np
import pytest

import pyref
from cactus_amd import gen, pafcheck
from cases import CASES, CASE_IDS

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fa(name, s):
    return (">%s\n%s\n" % (name, s)).encode()


def _rand(n, seed):
    return gen.random_sequence(n, np.random.default_rng(seed)).tobytes().decode()


def _match_score(s):
    return sum(91 if c in "AT" else 100 for c in s)


def test_known_answer_identical(olz):
    s = _rand(400, 1)
    r = olz.align(_fa("t", s), _fa("q", s), olz.default_params())
    assert r["paf"].decode() == "q\t400\t0\t400\t+\tt\t400\t0\t400\t400\t400\t255\tAS:i:%d\tcg:Z:400=\n" % _match_score(s)


def test_known_answer_single_substitution(olz):
    s = _rand(400, 2)
    q = s[:200] + {"A": "C", "C": "A", "G": "T", "T": "G"}[s[200]] + s[201:]
    r = olz.align(_fa("t", s), _fa("q", q), olz.default_params())
    f = r["paf"].decode().split("\t")
    assert f[-1].strip() == "cg:Z:200=1X199="
    assert int(f[-2][5:]) == _match_score(s) - _match_score(s[200]) + pafcheck.sub_score(s[200], q[200])


def test_known_answer_deletion_and_insertion(olz):
    s = _rand(600, 3)
    q = s[:300] + s[307:]                    # 7 target-only bases -> one D of length 7 (placement may shift inside ties)
    r = olz.align(_fa("t", s), _fa("q", q), olz.default_params())
    rec = pafcheck.parse_line(r["paf"].decode().splitlines()[0])
    assert (rec["tstart"], rec["tend"], rec["qstart"], rec["qend"]) == (0, 600, 0, 593)
    assert rec["cigar"].count("D") == 1 and "7D" in rec["cigar"] and "I" not in rec["cigar"]
    assert rec["score"] == _match_score(s) - _match_score(s[300:307]) - (400 + 7 * 30)
    r2 = olz.align(_fa("t", q), _fa("q", s), olz.default_params())
    rec2 = pafcheck.parse_line(r2["paf"].decode().splitlines()[0])
    assert "7I" in rec2["cigar"] and rec2["score"] == rec["score"]


def test_known_answer_reverse_strand(olz):
    s = _rand(500, 4)
    r = olz.align(_fa("t", s), _fa("q", pyref.revcomp(s)), olz.default_params())
    assert r["paf"].decode() == "q\t500\t0\t500\t-\tt\t500\t0\t500\t500\t500\t255\tAS:i:%d\tcg:Z:500=\n" % _match_score(s)


def test_soft_masked_bases_do_not_seed_but_score(olz):
    s = _rand(300, 5)
    assert olz.align(_fa("t", s.lower()), _fa("q", s), olz.default_params())["paf"] == b""
    t = s[:100].lower() + s[100:]            # seeds in the uppercase part, extension runs through the masked part
    r = olz.align(_fa("t", t), _fa("q", s), olz.default_params())
    assert r["paf"].decode().split("\t")[-1].strip() == "cg:Z:300="


def test_n_scores_minus_100_and_never_seeds(olz):
    assert olz.load().olz_score(4, 4, 1) == -100 and olz.load().olz_score(4, 0, 1) == -100 and olz.load().olz_score(0, 12, 1) == -100
    s = _rand(400, 6)
    q = s[:200] + "N" + s[201:]
    r = olz.align(_fa("t", s), _fa("q", q), olz.default_params())
    rec = pafcheck.parse_line(r["paf"].decode().splitlines()[0])
    assert rec["cigar"] == "200=1X199=" and rec["score"] == _match_score(s) - _match_score(s[200]) - 100


def test_hoxd70_matrix(olz):
    lib = olz.load()
    want = [[91, -114, -31, -123], [-114, 100, -125, -31], [-31, -125, 100, -114], [-123, -31, -114, 91]]
    for a in range(4):
        for b in range(4):
            assert lib.olz_score(a, b, 1) == want[a][b]
            assert lib.olz_score(a | 8, b, 1) == want[a][b]      # case-insensitive


@pytest.mark.parametrize("seed,n,sub,indel,kw", [
    (11, 700, 0.06, 0.004, {}), (12, 900, 0.10, 0.01, {"step": 2}), (13, 600, 0.03, 0.0, {"transitions": False}),
    (14, 800, 0.12, 0.01, {"K": 2200, "L": 2400, "ydrop": 4000}), (15, 500, 0.05, 0.02, {"ydrop": 1500}),
    (16, 1000, 0.08, 0.006, {"K": 2600, "L": 2800, "ydrop": 3500, "step": 3}), (17, 400, 0.0, 0.0, {"entropy": False}),
])
def test_c_oracle_agrees_with_python_restatement(olz, seed, n, sub, indel, kw):
    rng = np.random.default_rng(seed)
    t = gen.random_sequence(n, rng)
    q = gen.mutate(t, rng, sub, indel)
    if seed % 2:
        q = gen.revcomp(q)
    q = gen.soft_mask(q, rng, 0.1, 40)
    T, Q = t.tobytes().decode(), q.tobytes().decode()
    want, ctr = pyref.align(T, Q, **kw)
    p = olz.default_params(step=kw.get("step", 1), transitions=int(kw.get("transitions", True)), ydrop=kw.get("ydrop", 9400),
                           hspthresh=kw.get("K", 3000), gappedthresh=kw.get("L", -1), entropy=int(kw.get("entropy", True)))
    r = olz.align(_fa("t", T), _fa("q", Q), p)
    got = []
    for line in r["paf"].decode().splitlines():
        rec = pafcheck.parse_line(line)
        got.append((0 if rec["strand"] == "+" else 1, rec["tstart"], rec["tend"], rec["qstart"], rec["qend"], rec["score"],
                    rec["nmatch"], rec["alnlen"], rec["cigar"]))
    assert got == want
    for k in ("seed_hits", "hits_extended", "hsps", "dp_cells"):
        assert r["counters"][k] == ctr[k], k
    assert len(want) >= 1


@pytest.mark.parametrize("name,tf,qf,args", CASES, ids=CASE_IDS)
def test_oracle_output_is_valid_paf_and_matches_golden(olz, name, tf, qf, args):
    """Every parity case: the oracle's PAF passes the caf walk + score re-derivation, and equals the committed
    golden digest (tests/golden/golden.json, made by tests/golden/make_golden.py from this same oracle: a
    regression pin -- the reference has no PAF-level vectors)."""
    import hashlib
    from cactus_amd import miblast
    pm = miblast.params_from_args(args)
    r = olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}))
    if r["paf"]:
        n = pafcheck.check_paf(r["paf"].decode(), pafcheck.read_fasta(tf), pafcheck.read_fasta(qf))
        assert n == r["counters"]["alignments"]
    gold = json.load(open(os.path.join(GOLDEN, "golden.json")))[name]
    assert hashlib.sha256(r["paf"]).hexdigest() == gold["paf_sha256"]
    for k, v in gold["counters"].items():
        assert r["counters"][k] == v, k


def test_golden_small_fixture_files(olz):
    """Two tiny committed input/output fixtures (FASTA + PAF text) that can be read by eye."""
    for stem in ("tiny_plus", "tiny_minus_gap"):
        tf = open(os.path.join(GOLDEN, stem + ".target.fa"), "rb").read()
        qf = open(os.path.join(GOLDEN, stem + ".query.fa"), "rb").read()
        want = open(os.path.join(GOLDEN, stem + ".paf"), "rb").read()
        assert olz.align(tf, qf, olz.default_params(hspthresh=2200, gappedthresh=2400, ydrop=4000))["paf"] == want


def test_reference_parameter_pins():
    """The only pin the reference holds on this path: the literal default lastz option string
    (/root/reference/api/tests/cactusParamsTest.c:16-17) -- and the six sets of the config parse cleanly."""
    from cactus_amd import miblast
    from cactus_amd.shared.configWrapper import load_config
    cfg = load_config()
    la = cfg.find("blast").find("lastzArguments").attrib
    assert la["default"] == "--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=100000"
    p = miblast.params_from_args(la["default"].split())
    assert (p.step, p.ydrop, p.hspthresh, p.gappedthresh, p.queryhspbest, p.transitions) == (1, 4000, 2200, 2400, 100000, 1)
    p1 = miblast.params_from_args(la["one"].split())
    assert (p1.step, p1.ydrop, p1.hspthresh, p1.gappedthresh, p1.transitions) == (2, 3000, 3000, -1, 0)
    for node in ("lastzArguments", "kegalignArguments"):
        for k, v in cfg.find("blast").find(node).attrib.items():
            miblast.params_from_args(v.split())


def test_named_switches_for_the_points_a_real_lastz_may_differ_on(olz):
    """diag=hash16 (lastz's 16-bit diagEnd hash, A.4) and walls (earlier alignments bound later DPs, A.7) are
    default-off modes of the oracle.  Off = A.10 = every committed expectation; on, they change what they are meant to change."""
    import numpy as np
    from cactus_amd import gen, pafcheck
    rng = np.random.default_rng(5)
    p0 = olz.default_params(hspthresh=2200, gappedthresh=2400, ydrop=4000)
    assert (p0.diag_hash16, p0.walls) == (0, 0)
    # (1) hash16: the query matches two target copies 65536 apart -- their diagonals collide in the 16-bit hash, so after the
    # first copy's extension the hits of the second copy at the same query positions look "already extended" and are dropped;
    # with exact diagonals both copies are extended
    a = gen.random_sequence(1500, rng)
    t = np.concatenate([a, gen.random_sequence(65536 - 1500, rng), gen.mutate(a, rng, 0.03, 0.0), gen.random_sequence(300, rng)])
    q = gen.mutate(a, rng, 0.03, 0.0)
    tf, qf = gen.fasta_bytes([("T|h", t)]), gen.fasta_bytes([("Q|h", q)])
    exact = olz.align(tf, qf, p0)
    hashed = olz.align(tf, qf, olz.default_params(hspthresh=2200, gappedthresh=2400, ydrop=4000, diag_hash16=1))
    assert exact["counters"]["seed_hits"] == hashed["counters"]["seed_hits"]
    assert hashed["counters"]["hits_extended"] < exact["counters"]["hits_extended"]            # collisions drop hits (A.4 "silently")
    assert len(hashed["hsps"]) <= len(exact["hsps"])
    # without a collision the two modes agree byte for byte
    tf2, qf2 = gen.fasta_bytes([("T|n", t[:30000])]), gen.fasta_bytes([("Q|n", gen.mutate(t[2000:12000], rng, 0.05, 0.003))])
    assert olz.align(tf2, qf2, p0)["paf"] == olz.align(tf2, qf2, olz.default_params(hspthresh=2200, gappedthresh=2400, ydrop=4000, diag_hash16=1))["paf"]
    # (2) walls: a tandem duplication in the query makes the second alignment run along the first one's path region; with walls no
    # base pair is used twice, without them later alignments may overlap earlier ones
    unit = gen.random_sequence(2500, rng)
    t3 = np.concatenate([gen.random_sequence(800, rng), unit, gen.random_sequence(800, rng)])
    q3 = np.concatenate([gen.random_sequence(500, rng), gen.mutate(unit, rng, 0.04, 0.002), gen.mutate(unit, rng, 0.04, 0.002), gen.random_sequence(500, rng)])
    tf3, qf3 = gen.fasta_bytes([("T|w", t3)]), gen.fasta_bytes([("Q|w", q3)])
    free = olz.align(tf3, qf3, p0)
    walled = olz.align(tf3, qf3, olz.default_params(hspthresh=2200, gappedthresh=2400, ydrop=4000, walls=1))
    full = {"T|w": t3.tobytes().decode(), "Q|w": q3.tobytes().decode()}
    assert pafcheck.check_paf(walled["paf"].decode(), full, full) >= 2

    def pairs(res):
        used = []
        for al, ops in zip(res["alns"], res["ops"]):
            tt, qq = al[3], al[5]
            s = set()
            for o in ops:
                ln, op = o >> 2, o & 3
                if op < 2:
                    s.update((al[0], tt + k, qq + k) for k in range(ln)); tt += ln; qq += ln
                elif op == 2:
                    qq += ln
                else:
                    tt += ln
            used.append(s)
        return used

    w = pairs(walled)
    assert all(not (w[i] & w[j]) for i in range(len(w)) for j in range(i + 1, len(w)))          # no base pair on two paths
    assert walled["alns"][0] == free["alns"][0]                                                # the first alignment has nothing to respect


def test_strand_halves_interleaved_by_query_sequence_are_the_whole(olz):
    """--strand=plus / minus (olz_params.strands; lastz's own option): a strand's search, HSPs and alignments do not depend on the
    other strand's, and the whole PAF is, query sequence by query sequence in file order, the '+' lines then the '-' lines -- so
    (chunk pair, strand) is an exact work unit (cactus_amd.multigpu.merge_strand_pafs; the GPU suite checks the product's halves)."""
    from cases import CASES
    from cactus_amd import miblast
    from cactus_amd.multigpu import fasta_names, merge_strand_pafs
    n = 0
    for name, tf, qf, args in CASES:
        if any(a.startswith("--format=general") for a in args):
            continue
        pm = miblast.params_from_args(args)
        po = lambda **kw: olz.default_params(**{**{f: getattr(pm, f) for f, _ in pm._fields_}, **kw})      # noqa: E731
        both, plus, minus = (olz.align(tf, qf, po(strands=s), details=False) for s in (0, 1, 2))
        assert merge_strand_pafs(plus["paf"], minus["paf"], fasta_names(qf)) == both["paf"], name
        for k in ("seed_lookups", "seed_hits", "hits_extended", "ungapped_cols", "hsps", "anchors", "dp_cells", "dp_rows", "alignments"):
            assert plus["counters"][k] + minus["counters"][k] == both["counters"][k], (name, k)
        n += both["paf"].count(b"\n")
    assert n > 50


def test_the_other_named_switches_of_survey_a9_change_what_they_name(olz):
    """SURVEY A.9 #2, #5, #9, #11 and the bounded traceback (lastz_oracle.h, round 5): each switch is off by default (= A.10 = every
    committed expectation) and, on, changes the output of a case built for it -- so that a difference from a real lastz binary can be
    bisected by switch (tests/test_p1_lastz_binary.py)."""
    rng = np.random.default_rng(11)
    base = dict(hspthresh=2200, gappedthresh=2400, ydrop=4000)
    p0 = olz.default_params(**base)
    assert (p0.query_softmask, p0.step_origin, p0.xdrop_le, p0.hspbest_ties, p0.traceback_cells) == (0, 0, 0, 0, 0)
    # (#2) a query that is soft-masked throughout: no query window seeds under A.10 (rule "apply to both"); with the switch only the
    # target's masking counts and the copy is found
    s = _rand(1200, 21)
    tf, qf = _fa("T|m", s), _fa("Q|m", s.lower())
    a10 = olz.align(tf, qf, p0)
    seeded = olz.align(tf, qf, olz.default_params(query_softmask=1, **base))
    assert a10["counters"]["seed_hits"] == 0 and a10["paf"] == b""
    assert seeded["counters"]["seed_hits"] > 0 and seeded["counters"]["alignments"] == 1
    # ... and a masked TARGET window keeps out of the table either way
    assert olz.align(_fa("T|m", s.lower()), _fa("Q|m", s), olz.default_params(query_softmask=1, **base))["counters"]["seed_hits"] == 0
    # (#5) --step=2 over a [multiple] target whose second sequence starts at an odd position of the concatenation (300 bases + one
    # separator before it): counted from position 0 the indexed positions of that sequence are the even ones, counted from the
    # sequence's own start the odd ones.  The seed hit that makes the copy's HSP tells which lattice the table holds.
    first, second = _rand(300, 22), _rand(900, 23)
    tf = (_fa("T|a", first) + _fa("T|b", second))
    qf = _fa("Q|s", second)
    whole = olz.align(tf, qf, olz.default_params(step=2, gapped=0, **base))
    per_seq = olz.align(tf, qf, olz.default_params(step=2, step_origin=1, gapped=0, **base))
    seed_pos = lambda res: {(h[6] - 19) % 2 for h in res["hsps"] if h[0] == 0}
    assert seed_pos(whole) == {0} and seed_pos(per_seq) == {1}
    assert whole["hsps"][0][7] == per_seq["hsps"][0][7] + 1                  # the query position whose word is the first to hit
    # (#9) a walk whose running score falls EXACTLY x-drop below its best and recovers: strict "<" goes on and joins the two
    # stretches into one HSP, "<=" stops there.  6 columns of N against a base (-100 each) + 10 transitions (-31 each) = -910.
    left, right = _rand(260, 24), _rand(400, 25)
    tv = {"A": "G", "G": "A", "C": "T", "T": "C"}
    mid_q = "".join(rng.choice(list("ACGT"), 16))
    mid_t = "N" * 6 + "".join(tv[c] for c in mid_q[6:])
    tf, qf = _fa("T|x", left + mid_t + right), _fa("Q|x", left + mid_q + right)
    strict = olz.align(tf, qf, olz.default_params(gapped=0, **base))
    le = olz.align(tf, qf, olz.default_params(gapped=0, xdrop_le=1, **base))
    span = lambda res: sorted((h[2], h[2] + h[4]) for h in res["hsps"] if h[0] == 0)
    assert any(a <= 200 and b >= 300 for a, b in span(strict)), span(strict)          # one HSP across the dip
    assert not any(a <= 200 and b >= 300 for a, b in span(le)), span(le)             # ... none with "<="
    assert len(span(le)) > len(span(strict))
    # (#11) --queryhspbest=1 and two HSPs of one score (two identical copies in the target): A.10 keeps the one found first, the
    # switch the one found later
    unit = _rand(120, 26)
    tf, qf = _fa("T|k", _rand(300, 27) + unit + _rand(300, 28) + unit + _rand(300, 29)), _fa("Q|k", _rand(50, 30) + "N" * 20 + unit + "N" * 20 + _rand(50, 31))      # (N against anything: -100 -- neither HSP grows past the unit)
    first_kept = olz.align(tf, qf, olz.default_params(gapped=0, queryhspbest=1, **base))
    later_kept = olz.align(tf, qf, olz.default_params(gapped=0, queryhspbest=1, hspbest_ties=1, **base))
    both = olz.align(tf, qf, olz.default_params(gapped=0, **base))
    assert len(both["hsps"]) == 2 and both["hsps"][0][5] == both["hsps"][1][5]
    assert first_kept["hsps"] == [both["hsps"][0]] and later_kept["hsps"] == [both["hsps"][1]]
    # (bounded traceback) a 6 000-column alignment with room for 150 000 cells per side: the alignment is cut where the memory ends
    # and the rest is found again from a later anchor -- more, shorter alignments, every one of them a valid PAF record
    a = gen.random_sequence(6000, rng)
    t, q = a, gen.mutate(a, rng, 0.04, 0.002)
    tf, qf = gen.fasta_bytes([("T|b", t)]), gen.fasta_bytes([("Q|b", q)])
    whole = olz.align(tf, qf, p0)
    cut = olz.align(tf, qf, olz.default_params(traceback_cells=150000, **base))
    assert whole["counters"]["alignments"] == 1 and cut["counters"]["alignments"] > 1
    assert max(al[4] - al[3] for al in cut["alns"]) < whole["alns"][0][4] - whole["alns"][0][3]
    full = {"T|b": t.tobytes().decode(), "Q|b": q.tobytes().decode()}
    assert pafcheck.check_paf(cut["paf"].decode(), full, full) == cut["counters"]["alignments"]
    # a budget no side reaches changes nothing
    assert olz.align(tf, qf, olz.default_params(traceback_cells=1 << 40, **base))["paf"] == whole["paf"]

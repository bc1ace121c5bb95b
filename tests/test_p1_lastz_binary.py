"""Parity rung P1: oracle == a REAL lastz binary.  No lastz can be built or found in the build container (the
submodule directory is empty), so this test skips there -- and says so; wherever $MIBLAST_LASTZ or a foreign `lastz` on PATH
exists it runs the reference's own command line (local_alignment.py:60-68) on the seeded cases and diffs the sorted PAF records
against the oracle (and reports whether the unsorted order matched too).  The named switches of SURVEY A.9 are tried as well -- alone
and, when no single one explains the binary, in every combination (2^7 oracle runs per case at worst) -- so a mismatch is bisected
to the readings that account for it; $MIBLAST_P1_MODE = A.10 | a "+"-joined list of switch names (diag_hash16, walls, query_softmask,
step_origin, xdrop_le, hspbest_ties, traceback_80M) names the reading the binary is REQUIRED to match (default A.10).  diag_hash16 and
walls are implemented by the MI355X path as well (--miblast-diag=hash16, --miblast-walls; tests/test_parity_gpu.py); the other five
are oracle-side for now (lastz_oracle.h)."""
import itertools
import os
import shutil
import subprocess

import pytest

from cases import CASES, CASE_IDS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _real_lastz():
    cand = os.environ.get("MIBLAST_LASTZ")
    if cand and os.access(cand, os.X_OK):
        return cand
    ours = os.path.realpath(os.path.join(ROOT, "bin", "lastz"))
    for d in os.environ.get("PATH", "").split(os.pathsep):
        p = os.path.join(d, "lastz")
        if os.access(p, os.X_OK) and os.path.realpath(p) != ours:
            v = subprocess.run([p, "--version"], capture_output=True, text=True)
            if "lastz" in (v.stdout + v.stderr).lower() and "miblast" not in (v.stdout + v.stderr).lower():
                return p
    return None


LASTZ = _real_lastz()

# the named switches of SURVEY A.9 (oracle/lastz_oracle.h): name -> parameter override
SWITCHES = {"diag_hash16": {"diag_hash16": 1}, "walls": {"walls": 1}, "query_softmask": {"query_softmask": 1}, "step_origin": {"step_origin": 1},
            "xdrop_le": {"xdrop_le": 1}, "hspbest_ties": {"hspbest_ties": 1}, "traceback_80M": {"traceback_cells": 80 << 20}}


def bisect_switches(olz, tf, qf, base, real, exhaustive=None):
    """{label: (same sorted records, same bytes)} for A.10, every switch alone and -- if none of those reproduces `real` (or when
    `exhaustive`) -- every combination of switches: the labels that come out True are the readings that explain the binary."""
    def run(names):
        over = {}
        for n in names:
            over.update(SWITCHES[n])
        got = olz.align(tf, qf, olz.default_params(**dict(base, **over)), details=False)["paf"]
        return sorted(got.splitlines()) == sorted(real.splitlines()), got == real
    verdicts = {"A.10": run(())}
    for n in SWITCHES:
        verdicts[n] = run((n,))
    if exhaustive or (exhaustive is None and not any(v[0] for v in verdicts.values())):
        for k in range(2, len(SWITCHES) + 1):
            for names in itertools.combinations(SWITCHES, k):
                verdicts["+".join(names)] = run(names)
    return verdicts


@pytest.mark.skipif(LASTZ is None, reason="P1 NOT EXERCISED: no real lastz binary ($MIBLAST_LASTZ / PATH); parity stays unpinned")
@pytest.mark.parametrize("name,tf,qf,args", CASES, ids=CASE_IDS)
def test_oracle_equals_real_lastz(olz, tmp_path, name, tf, qf, args):
    from cactus_amd import miblast
    if not tf or not qf:
        pytest.skip("lastz rejects empty files")
    (tmp_path / "T.fa").write_bytes(tf)
    (tmp_path / "Q.fa").write_bytes(qf)
    p = subprocess.run([LASTZ, "T.fa[multiple][nameparse=darkspace]", "Q.fa[nameparse=darkspace]", "--format=paf:wfmash", *args],
                       cwd=tmp_path, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    real = p.stdout
    pm = miblast.params_from_args(args)
    base = {f: getattr(pm, f) for f, _ in pm._fields_}
    verdicts = bisect_switches(olz, tf, qf, base, real)
    required = os.environ.get("MIBLAST_P1_MODE", "A.10")
    assert required in verdicts, (required, sorted(verdicts))
    assert verdicts[required][0], {k: v for k, v in verdicts.items() if v[0]} or verdicts


@pytest.mark.skipif(LASTZ is None, reason="P1 NOT EXERCISED: no real lastz binary ($MIBLAST_LASTZ / PATH); parity stays unpinned")
def test_a_side_longer_than_lastz_default_traceback_allocation(olz, tmp_path):
    """The third A.9 candidate: lastz bounds one traceback's memory (--allocate:traceback, 80 MiB by default) and
    truncates an alignment whose DP outgrows it; the oracle keeps the whole trace.  A 700 kb pair at 1 % divergence is one alignment whose
    sides evaluate ~1.2 x 10^8 cells: with a real binary this tells which of the two the Cactus command line gets -- and, run again with
    --allocate:traceback=1G, whether that alone accounts for the difference."""
    from cactus_amd import gen, miblast
    t, q = gen.make_pair(700_000, 4242, sub_rate=0.01, indel_rate=0.0005)
    tf, qf = gen.fasta_bytes([("id=simT|chr1", t)]), gen.fasta_bytes([("id=simQ|chr1", q)])
    (tmp_path / "T.fa").write_bytes(tf)
    (tmp_path / "Q.fa").write_bytes(qf)
    args = ["--step=2", "--ambiguous=iupac,100,100", "--ydrop=3000", "--notransition"]
    want = olz.align(tf, qf, olz.default_params(**{f: getattr(miblast.params_from_args(args), f) for f, _ in miblast.params_from_args(args)._fields_}), details=False)["paf"]
    verdict = {}
    for label, extra in (("default allocation", []), ("--allocate:traceback=1G", ["--allocate:traceback=1G"])):
        p = subprocess.run([LASTZ, "T.fa[multiple][nameparse=darkspace]", "Q.fa[nameparse=darkspace]", "--format=paf:wfmash", *args, *extra], cwd=tmp_path, capture_output=True)
        assert p.returncode == 0, p.stderr.decode()
        verdict[label] = sorted(p.stdout.splitlines()) == sorted(want.splitlines())
    assert verdict["--allocate:traceback=1G"], verdict          # with room for the trace the binary must agree with the oracle ...
    assert verdict["default allocation"], verdict               # ... and if only this one fails, the traceback limit is the switch to add


def test_the_bisection_finds_the_switches_that_explain_a_binary(olz):
    """Runs here, without a binary: the oracle with two switches on plays the part of the real lastz -- the bisection must name exactly
    that combination (and no single switch) as the reading that reproduces it."""
    from cactus_amd import miblast
    import numpy as np
    from cactus_amd import gen
    rng = np.random.default_rng(3)
    # two islands between runs of N: a 19-base word at an even offset of the target's SECOND sequence, which starts at an odd position
    # of the concatenation (--step=2 indexes it only when the step counts from the sequence's start), and a 400-base stretch that is
    # lowercase in the query (seeds only if query soft-masking is ignored).  A.10 finds neither.
    word = "ACGGTCATGCTAGCTTGAC"
    island = gen.random_sequence(400, rng).tobytes().decode()
    tf = gen.fasta_bytes([("T|a", gen.random_sequence(300, rng))]) + (">T|b\n" + "N" * 30 + word + "N" * 30 + island + "N" * 30 + "\n").encode()
    qf = (">Q|p\n" + "N" * 30 + word + "N" * 30 + island.lower() + "N" * 30 + "\n").encode()
    args = ["--step=2", "--ambiguous=iupac,100,100", "--hspthresh=1500", "--ungapped", "--format=general:name1,zstart1,end1,name2,zstart2+,end2+"]
    pm = miblast.params_from_args(args)
    base = {f: getattr(pm, f) for f, _ in pm._fields_}
    pretend = olz.align(tf, qf, olz.default_params(**dict(base, query_softmask=1, step_origin=1)), details=False)["paf"]
    records = lambda out: [l for l in out.splitlines() if not l.startswith(b"#")]
    assert len(records(pretend)) == 2 and records(olz.align(tf, qf, olz.default_params(**base), details=False)["paf"]) == []
    verdicts = bisect_switches(olz, tf, qf, base, pretend)
    explained = sorted(k for k, v in verdicts.items() if v[0])
    assert "query_softmask+step_origin" in explained and "A.10" not in explained
    assert all("query_softmask" in k and "step_origin" in k for k in explained), explained


def test_the_skip_is_reported_not_silent():
    if LASTZ is None:
        assert shutil.which("lastz") in (None, os.path.join(ROOT, "bin", "lastz")) or True
        pytest.skip("P1 NOT EXERCISED: no real lastz binary in this environment")

"""What CAN be pinned to the reference itself (SURVEY 8 row f3's host half): the repeat masker's three helper tools exist in the
reference tree as runnable sources -- cactus_fasta_fragments.py and cactus_fasta_softmask_intervals.py (python3, no dependencies)
and cactus_covered_intervals.c (built unmodified by oracle/Makefile into oracle/_ref/ against a five-function sonLib stand-in).
The product's mirrors (cactus_amd.preprocessor.lastz_repeat_mask) are diffed byte for byte against
  (1) fixtures the reference tools wrote (tests/golden/repeatmask/, generator: tests/golden/make_repeatmask_golden.py) -- these
      travel to boxes without /root/reference;
  (2) the live tools on fresh random inputs, wherever /root/reference (and the built oracle/_ref binary) is present.
The lastz and paffy arithmetic cannot be pinned this way: their submodule directories are empty."""
import os
import subprocess
import sys

import numpy as np
import pytest

from cactus_amd.preprocessor.lastz_repeat_mask import covered_intervals, fasta_fragments, softmask_intervals

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "repeatmask")
REF = "/root/reference/preprocessor/lastzRepeatMasking"
COVERED = os.path.join(ROOT, "oracle", "_ref", "cactus_covered_intervals")
live = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present on this box (fixtures cover it)")


def _read(name):
    return open(os.path.join(G, name)).read()


@pytest.mark.parametrize("frag,step,origin", [(200, 100, "zero"), (64, 16, "one"), (100, 50, "one")])
def test_fasta_fragments_equals_reference_written_fixture(frag, step, origin):
    assert fasta_fragments(_read("input.fa"), frag, step, origin) == _read("fragments_%d_%d_%s.fa" % (frag, step, origin))


@pytest.mark.parametrize("M,origin", [(1, "zero"), (3, "one"), (7, "zero")])
def test_covered_intervals_equals_reference_written_fixture(M, origin):
    want = _read("covered_M%d_%s.txt" % (M, origin))
    assert want.count("\n") >= (3 if M < 7 else 1)
    assert covered_intervals(_read("general.txt").splitlines(), M, origin == "one", True) == want


@pytest.mark.parametrize("origin,unmask", [("zero", False), ("one", False), ("zero", True)])
def test_softmask_intervals_equals_reference_written_fixture(origin, unmask):
    want = _read("softmask_%s%s.fa" % (origin, "_unmask" if unmask else ""))
    assert softmask_intervals(_read("input.fa"), _read("intervals.txt").splitlines(), origin == "one", unmask) == want


def _run(cmd, text):
    p = subprocess.run(cmd, input=text.encode(), capture_output=True)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def _random_fasta(rng):
    recs = []
    for k in range(int(rng.integers(1, 6))):
        n = int(rng.integers(0, 900))
        s = np.frombuffer(b"ACGTacgtNn", dtype=np.uint8)[rng.integers(0, 10, size=n)].copy()
        if n > 300 and rng.random() < 0.7:
            a = int(rng.integers(0, n - 250)); s[a:a + 250] = ord("N")
        text = s.tobytes().decode()
        w = int(rng.choice([50, 60, 100, 2000]))
        recs.append(">r%d extra words\n" % k + "".join(text[i:i + w] + "\n" for i in range(0, len(text), w)))
    return "".join(recs)


@live
@pytest.mark.parametrize("seed", range(6))
def test_mirrors_equal_the_live_reference_tools_on_random_input(seed, tmp_path):
    rng = np.random.default_rng(100 + seed)
    fa = _random_fasta(rng)
    frag, step, origin = int(rng.choice([50, 100, 200])), int(rng.choice([25, 50, 100])), str(rng.choice(["zero", "one"]))
    rc, out, err = _run([sys.executable, os.path.join(REF, "cactus_fasta_fragments.py"), "--fragment=%d" % frag, "--step=%d" % step, "--origin=%s" % origin], fa)
    assert rc == 0, err
    frags = fasta_fragments(fa, frag, step, origin)
    assert frags == out
    # masker-style HSP lines over those fragments (fragment order, as lastz writes them) -> covered intervals
    lines = ["#name1\tzstart1\tend1\tname2\tzstart2+\tend2+\n"]
    for h in [l[1:] for l in frags.splitlines() if l.startswith(">")]:
        for _ in range(int(rng.integers(0, 6))):
            s = int(rng.integers(0, frag - 10)); e = min(frag, s + int(rng.integers(5, 60)))
            lines.append("tgt\t%d\t%d\t%s\t%d\t%d\n" % (rng.integers(0, 900), rng.integers(900, 999), h, s, e))
    text = "".join(lines)
    if os.access(COVERED, os.X_OK) and origin == "zero":       # (--queryoffsets adds the fragment's origin-zero offset)
        M = int(rng.integers(1, 5))
        for out_origin in ("zero", "one"):
            rc, want, err = _run([COVERED, "--queryoffsets", "M=%d" % M, "--origin=%s" % out_origin], text)
            assert rc == 0, err
            assert covered_intervals(text.splitlines(), M, out_origin == "one", True) == want
    ivals = covered_intervals(text.splitlines(), 1, False, True) if origin == "zero" else ""
    path = tmp_path / "iv.txt"
    path.write_text(ivals)
    for unmask in (False, True):
        cmd = [sys.executable, os.path.join(REF, "cactus_fasta_softmask_intervals.py"), "--origin=zero", str(path)] + (["--unmask"] if unmask else [])
        rc, want, err = _run(cmd, fa)
        assert rc == 0, err
        assert softmask_intervals(fa, ivals.splitlines(), False, unmask) == want


@live
def test_error_behaviour_matches_the_reference_script(tmp_path):
    fa = ">a\nACGT\n>b\nAC\n"
    for bad in ("zzz\t1\t3\n", "a\t3\t3\n", "a\t-1\t2\n", "a\t1\n"):
        path = tmp_path / "iv.txt"
        path.write_text(bad)
        rc, _, err = _run([sys.executable, os.path.join(REF, "cactus_fasta_softmask_intervals.py"), str(path)], fa)
        assert rc != 0 and "AssertionError" in err
        with pytest.raises(AssertionError):
            softmask_intervals(fa, bad.splitlines())
    with pytest.raises(AssertionError):
        softmask_intervals(">a\nAC\n>a\nGT\n", [])


def test_ref_recipe_is_committed_and_outputs_are_not():
    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    assert "cactus_covered_intervals.c" in mk and "_ref" in mk
    assert "oracle/_ref/" in open(os.path.join(ROOT, ".gitignore")).read()
    ign = open(os.path.join(ROOT, ".gpurunignore")).read()
    assert "_ref" not in ign                                                 # built checkers travel to the GPU box
    tracked = subprocess.run(["git", "ls-files", "oracle/_ref"], capture_output=True, text=True, cwd=ROOT).stdout
    assert tracked.strip() == ""

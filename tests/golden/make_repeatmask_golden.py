#!/usr/bin/env python3
"""Reference-executed fixtures for the repeat-masker helpers: runs the reference's OWN tools, unmodified, from
/root/reference/preprocessor/lastzRepeatMasking -- cactus_fasta_fragments.py and cactus_fasta_softmask_intervals.py under this
image's python3, cactus_covered_intervals.c built by oracle/Makefile into oracle/_ref/ -- on seeded inputs and writes inputs and
outputs under tests/golden/repeatmask/.  The fixtures travel to boxes without /root/reference; tests/test_reference_pins_cpu.py
diffs the product's helpers against them (and against the live tools where the reference tree is present)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/preprocessor/lastzRepeatMasking"
OUT = os.path.join(ROOT, "tests", "golden", "repeatmask")
COVERED = os.path.join(ROOT, "oracle", "_ref", "cactus_covered_intervals")


def fasta(seed):
    """ragged multi-record FASTA: mixed case, N runs (one fragment-aligned all-N stretch), wrapped lines, descriptions"""
    rng = np.random.default_rng(seed)
    recs = []
    for k, n in enumerate([977, 0, 150, 1203, 50, 431]):
        s = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)[rng.choice(9, size=n, p=[.2, .2, .2, .2, .04, .04, .04, .04, .04])].copy()
        if n > 900:
            s[200:420] = ord("N"); s[600:650] = ord("n")
        text = s.tobytes().decode()
        w = [60, 80, 1000][k % 3]
        recs.append(">seq%d some description %d\n" % (k, k) + "".join(text[i:i + w] + "\n" for i in range(0, len(text), w)))
    return "".join(recs)


def general_lines(seed):
    """lastz --format=general output of the masker's call: per query fragment (name_offset, origin as given), in fragment order,
    a few HSPs each; plus a self alignment and a header line"""
    rng = np.random.default_rng(seed)
    lines = ["#name1\tzstart1\tend1\tname2\tzstart2+\tend2+\n"]
    for chrom, length in (("seq0", 977), ("seq3", 1203), ("chrX", 5000)):
        for off in range(0, length, 100):
            for _ in range(int(rng.integers(0, 9))):
                s = int(rng.integers(0, 160)); e = s + int(rng.integers(15, 40))
                lines.append("t%d\t%d\t%d\t%s_%d\t%d\t%d\n" % (rng.integers(0, 3), rng.integers(0, 4000), rng.integers(4000, 4100), chrom, off, s, min(e, 200)))
            if off == 300:
                lines.append("%s\t%d\t%d\t%s_%d\t%d\t%d\n" % (chrom, off + 7, off + 50, chrom, off, 7, 50))     # trivial self-alignment: ignored
    lines.append("# lastz end-of-file\n")
    return "".join(lines)


def run(cmd, text):
    p = subprocess.run(cmd, input=text.encode(), capture_output=True)
    assert p.returncode == 0, (cmd, p.stderr.decode())
    return p.stdout.decode()


def main():
    os.makedirs(OUT, exist_ok=True)
    fa = fasta(7)
    open(os.path.join(OUT, "input.fa"), "w").write(fa)
    for frag, step, origin in ((200, 100, "zero"), (64, 16, "one"), (100, 50, "one")):
        out = run([sys.executable, os.path.join(REF, "cactus_fasta_fragments.py"), "--fragment=%d" % frag, "--step=%d" % step, "--origin=%s" % origin], fa)
        open(os.path.join(OUT, "fragments_%d_%d_%s.fa" % (frag, step, origin)), "w").write(out)
    gl = general_lines(11)
    open(os.path.join(OUT, "general.txt"), "w").write(gl)
    for M, origin in ((1, "zero"), (3, "one"), (7, "zero")):
        out = run([COVERED, "--queryoffsets", "M=%d" % M, "--origin=%s" % origin], gl)
        open(os.path.join(OUT, "covered_M%d_%s.txt" % (M, origin)), "w").write(out)
    ivals = "seq0\t10\t60\nseq0\t50\t300\nseq3\t1\t5\nseq3\t1100\t1203\nseq5\t400\t431\nseq2\t140\t9999\n"
    open(os.path.join(OUT, "intervals.txt"), "w").write(ivals)
    for origin, unmask in (("zero", False), ("one", False), ("zero", True)):
        args = [sys.executable, os.path.join(REF, "cactus_fasta_softmask_intervals.py"), "--origin=%s" % origin, os.path.join(OUT, "intervals.txt")]
        if unmask:
            args.insert(2, "--unmask")
        open(os.path.join(OUT, "softmask_%s%s.fa" % (origin, "_unmask" if unmask else "")), "w").write(run(args, fa))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()

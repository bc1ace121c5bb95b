"""CPU tests of the repeat-masker helpers and of the oracle's general-format / hsplimit output."""
import numpy as np

from cactus_amd import gen
from cactus_amd.preprocessor.lastz_repeat_mask import covered_intervals, fasta_fragments, softmask_intervals


def test_fasta_fragments_matches_reference_script_semantics():
    fa = ">chr1 desc\nACGTACGTAC\nGTNNNNNNNN\nNNNN\n>c2\nacgt\n"
    out = fasta_fragments(fa, fragment=8, step=4, origin="zero")
    recs = out.strip().split("\n")
    # chr1 = ACGTACGTACGTNNNNNNNNNNNN (24): fragments at 0,4,8 (12.. are all-N and dropped except partial tail rules), c2 upper-cased
    assert recs[0] == ">chr1_0" and recs[1] == "ACGTACGT"
    assert ">chr1_8" in recs and "ACGTNNNN" in recs
    assert ">chr1_16" not in recs                       # NNNNNNNN == all-N fragment -> skipped
    assert ">chr1_20" in recs                           # shorter tail fragment "NNNN" != "N"*8 -> kept, like the reference
    assert recs[-2:] == [">c2_0", "ACGT"]
    assert ">chr1_1" in fasta_fragments(fa, 8, 4, "one")


def _naive_covered(lines, M, origin_one, query_offsets):
    depth = {}
    order = []
    for line in lines:
        if line.startswith("#") or not line.strip():
            continue
        f = line.split()
        qc, qs, qe = f[3], int(f[4]), int(f[5])
        if query_offsets:
            qc, off = qc.rsplit("_", 1); qs += int(off); qe += int(off)
        if qc == f[0] and qs == int(f[1]) and qe == int(f[2]):
            continue
        if qc not in depth:
            depth[qc] = {}; order.append(qc)
        for p in range(qs, qe):
            depth[qc][p] = min(255, depth[qc].get(p, 0) + 1)
    out = []
    for qc in order:
        pos = sorted(p for p, d in depth[qc].items() if d >= M)
        run = []
        for p in pos:
            if run and p == run[-1] + 1:
                run.append(p)
            else:
                if run:
                    out.append("%s\t%d\t%d\n" % (qc, run[0] + (1 if origin_one else 0), run[-1] + 1))
                run = [p]
        if run:
            out.append("%s\t%d\t%d\n" % (qc, run[0] + (1 if origin_one else 0), run[-1] + 1))
    return "".join(out)


def test_covered_intervals_equals_naive_depth_counting():
    rng = np.random.default_rng(1)
    lines = ["#name1\tzstart1\tend1\tname2\tzstart2+\tend2+"]
    for chrom in ("q1", "q2"):
        for off in range(0, 2000, 100):
            for _ in range(int(rng.integers(0, 12))):
                s = int(rng.integers(0, 150)); e = s + int(rng.integers(20, 50))
                lines.append("t\t%d\t%d\t%s_%d\t%d\t%d" % (rng.integers(0, 9000), rng.integers(9000, 9999), chrom, off, s, e))
    lines.append("q1\t5\t9\tq1_0\t5\t9")              # self alignment, ignored
    lines.append("# lastz end-of-file")
    for M in (1, 3, 6):
        for origin_one in (False, True):
            assert covered_intervals(lines, M, origin_one, True) == _naive_covered(lines, M, origin_one, True)
    assert covered_intervals(lines, 2, True, True, markend=True).endswith("# covered_intervals end-of-file\n")


def test_softmask_intervals():
    fa = ">a\nACGTACGTAC\n>b\nGGGGGGGGGG\n"
    out = softmask_intervals(fa, ["a\t2\t4", "b\t10\t10", "# covered_intervals end-of-file"], origin_one=True)
    assert out == ">a\nAcgtACGTAC\n>b\nGGGGGGGGGg\n"
    assert softmask_intervals(">a\nacGT\n", ["a\t0\t1"], origin_one=False, unmask=True) == ">a\naCGT\n"


def test_oracle_general_format_and_hsplimit(olz):
    """--ungapped --format=general...: header, one line per HSP ('+' strand coordinates), --markend, and
    --queryhsplimit=keep,nowarn:N keeps the first N HSPs found per query record and strand."""
    rng = np.random.default_rng(3)
    t = gen.random_sequence(4000, rng)
    rep = t[1000:1400]
    q = np.concatenate([gen.random_sequence(300, rng), rep, gen.random_sequence(200, rng), gen.revcomp(rep), gen.random_sequence(100, rng)])
    tf = gen.fasta_bytes([("id=T|c1", t)]); qf = gen.fasta_bytes([("id=Q|c1", q)])
    p = olz.default_params(gapped=0, format=1, markend=1, hspthresh=2200)
    r = olz.align(tf, qf, p)
    lines = r["paf"].decode().splitlines()
    assert lines[0] == "#name1\tzstart1\tend1\tname2\tzstart2+\tend2+" and lines[-1] == "# lastz end-of-file"
    body = [l.split("\t") for l in lines[1:-1]]
    assert len(body) == len(r["hsps"]) >= 2
    assert ["id=T|c1", "1000", "1400", "id=Q|c1", "300", "700"] in body
    # the reverse-complemented copy is reported with '+' strand query coordinates (chance matches may extend it a little)
    assert any(b[0] == "id=T|c1" and abs(int(b[2]) - 1400) <= 30 and abs(int(b[4]) - 900) <= 30 and abs(int(b[5]) - 1300) <= 30 for b in body)
    for b in body:
        assert int(b[2]) - int(b[1]) == int(b[5]) - int(b[4])                   # ungapped: equal lengths
    r1 = olz.align(tf, qf, olz.default_params(gapped=0, format=1, markend=1, hspthresh=2200, queryhsplimit=1))
    assert [h[0] for h in r1["hsps"]] == [0, 1]                               # one per strand: the first found

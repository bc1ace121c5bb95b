"""PAF structural validator mirroring the contract cactus_consolidated enforces on blast output.

Walks every record exactly as caf does (/root/reference/caf/impl/pinchIterator.c:59-121): op lengths
>= 1, '='/'X'/'M' consume both sequences, 'I' consumes the query only, 'D' the target only, and the
walk must land exactly on query_end (same strand) / query_start (opposite strand) and target_end.
Additionally re-derives nmatch, alnlen and the AS:i score from the sequences (HOXD70, O=400, E=30,
SURVEY.md A.2) so a cigar cannot drift from its score.
"""
from __future__ import annotations

import re

_HOX = {("A", "A"): 91, ("C", "C"): 100, ("G", "G"): 100, ("T", "T"): 91,
        ("A", "C"): -114, ("A", "G"): -31, ("A", "T"): -123, ("C", "G"): -125, ("C", "T"): -31, ("G", "T"): -114}
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
_OP = re.compile(r"(\d+)([=XIDM])")


def sub_score(a: str, b: str) -> int:
    a, b = a.upper(), b.upper()
    if a not in "ACGT" or b not in "ACGT":
        return -100
    return _HOX.get((a, b), _HOX.get((b, a)))


def read_fasta(path_or_bytes):
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    seqs, name, parts = {}, None, []
    for line in data.decode().splitlines():
        if line.startswith(">"):
            if name is not None:
                seqs[name] = "".join(parts)
            name, parts = line[1:].split()[0] if line[1:].split() else "", []
        else:
            parts.append(line.strip())
    if name is not None:
        seqs[name] = "".join(parts)
    return seqs


def revcomp(s: str) -> str:
    return "".join(_COMP.get(c.upper(), "N") if c.isupper() else _COMP.get(c.upper(), "N").lower() for c in reversed(s))


def parse_line(line: str):
    f = line.rstrip("\n").split("\t")
    rec = dict(qname=f[0], qlen=int(f[1]), qstart=int(f[2]), qend=int(f[3]), strand=f[4], tname=f[5], tlen=int(f[6]),
               tstart=int(f[7]), tend=int(f[8]), nmatch=int(f[9]), alnlen=int(f[10]), mapq=int(f[11]), tags=f[12:])
    for t in f[12:]:
        if t.startswith("AS:i:"):
            rec["score"] = int(t[5:])
        if t.startswith("cg:Z:"):
            rec["cigar"] = t[5:]
    return rec


def check_record(rec, tseqs=None, qseqs=None, gap_open=400, gap_extend=30):
    """Raises AssertionError on any violation; returns number of aligned columns."""
    assert rec["mapq"] == 255
    assert rec["tags"][-1].startswith("cg:Z:") and rec["tags"][-2].startswith("AS:i:"), "AS:i then cg:Z last (local_alignment.py:312-313)"
    assert 0 <= rec["qstart"] < rec["qend"] <= rec["qlen"]
    assert 0 <= rec["tstart"] < rec["tend"] <= rec["tlen"]
    ops = _OP.findall(rec["cigar"])
    assert "".join(n + o for n, o in ops) == rec["cigar"], "cigar has junk"
    same = rec["strand"] == "+"
    x = rec["qstart"] if same else rec["qend"]
    y = rec["tstart"]
    nmatch = alnlen = 0
    score = 0
    q = t = None
    if tseqs is not None:
        t = tseqs[rec["tname"]]
        q = qseqs[rec["qname"]]
        assert len(t) == rec["tlen"] and len(q) == rec["qlen"]
        if not same:
            q = revcomp(q)
    qi = rec["qstart"] if same else rec["qlen"] - rec["qend"]      # position on the aligned strand
    prev = None
    for n, o in ops:
        n = int(n)
        assert n >= 1
        assert o != prev, "adjacent equal ops must be merged"
        prev = o
        alnlen += n
        if o in "=XM":
            if t is not None:
                for k in range(n):
                    a, b = t[y + k], q[qi + k]
                    eq = a.upper() == b.upper() and a.upper() in "ACGT"
                    assert eq == (o == "="), f"op {o} disagrees with bases {a}/{b}"
                    score += sub_score(a, b)
            if o == "=":
                nmatch += n
            x += n if same else -n
            y += n
            qi += n
        elif o == "I":
            x += n if same else -n
            qi += n
            score -= gap_open + n * gap_extend
        else:
            y += n
            score -= gap_open + n * gap_extend
    assert x == (rec["qend"] if same else rec["qstart"])
    assert y == rec["tend"]
    assert nmatch == rec["nmatch"] and alnlen == rec["alnlen"]
    if t is not None:
        assert score == rec["score"], f"AS {rec['score']} != recomputed {score}"
    return alnlen


def check_paf(text: str, tseqs=None, qseqs=None) -> int:
    n = 0
    for line in text.splitlines():
        if line:
            check_record(parse_line(line), tseqs, qseqs)
            n += 1
    return n

"""CPU tests of the text steps either side of the blast call against oracle/paffy_text_oracle.c --
a third, independently built restatement (per-base counters instead of interval merging; written from the reference's call sites,
/root/reference/src/cactus/paf/local_alignment.py:352, 378-387, 460-488): the Python cores of cactus_amd/paf/chunking.py and the
native text code of libmiblast (mipaf_dechunk_text, mipaf_unaligned_fasta) must write the oracle's bytes.  (The device
implementation, miblast_seqsets_unaligned, is diffed against the same oracle in tests/test_parity_gpu.py.)"""
import os
import subprocess

import numpy as np
import pytest

from cactus_amd import gen
from cactus_amd.paf import chunking

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "oracle_paffy_text")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle_paffy_text"], check=True, capture_output=True)


def oracle(*args):
    p = subprocess.run([ORACLE, *[str(a) for a in args]], capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout


def random_case(seed, n_contigs=4, max_len=3000, n_aln=12, nested=False):
    """A FASTA file (ragged contigs, soft-masked and N stretches; NAME|LEN|START names when nested) and PAF lines whose query
    intervals overlap, touch, repeat, reach the contig ends or leave a contig untouched."""
    rng = np.random.default_rng(seed)
    recs = []
    for k in range(n_contigs):
        n = int(rng.integers(1, max_len))
        s = gen.random_sequence(n, rng)
        if n > 40:                                                   # a soft-masked stretch and a few N (Cactus's alphabet is ACGTNacgtn)
            a = int(rng.integers(0, n - 20))
            s = s.copy()
            s[a:a + 20] = np.frombuffer(bytes(s[a:a + 20]).lower(), dtype=np.uint8)
            s[int(rng.integers(0, n))] = ord("N")
        name = f"id=Q|c{k}" + (f"|{n + 500}|{int(rng.integers(0, 400))}" if nested else "")
        recs.append((name, s))
    lines = []
    for _ in range(n_aln):
        k = int(rng.integers(0, n_contigs - 1)) if n_contigs > 1 else 0          # the last contig stays unaligned
        n = len(recs[k][1])
        a = int(rng.integers(0, n))
        b = int(min(n, a + rng.integers(0, max(2, n // 3))))
        if rng.random() < 0.2:
            a = 0
        if rng.random() < 0.2:
            b = n
        lines.append(f"{recs[k][0]}\t{n}\t{a}\t{b}\t{'+-'[int(rng.integers(0, 2))]}\tid=T|x\t9999\t5\t{5 + b - a}\t{b - a}\t{b - a}\t255\tAS:i:7\tcg:Z:{max(1, b - a)}=\n")
    return gen.fasta_bytes(recs), "".join(lines).encode(), recs


@pytest.mark.parametrize("seed", range(12))
def test_unaligned_parts_text_implementations_write_the_oracles_bytes(tmp_path, seed):
    from cactus_amd import blast_phase as bp, mipaf
    fa, paf, _ = random_case(seed, n_contigs=1 + seed % 5, nested=seed % 3 == 2)
    (tmp_path / "q.fa").write_bytes(fa)
    (tmp_path / "a.paf").write_bytes(paf)
    for min_size, flank in ((100, 100), (1, 0), (37, 5), (5000, 10), (10, 3000)):
        want = oracle("to_bed_extract", tmp_path / "a.paf", tmp_path / "q.fa", min_size, flank)
        assert mipaf.unaligned_fasta(paf, fa, min_size, flank) == want, (seed, min_size, flank)           # native text code
        assert bp.unaligned_fasta_py(paf, fa, min_size, flank) == want, (seed, min_size, flank)           # Python cores
    empty = tmp_path / "none.paf"
    empty.write_bytes(b"")
    assert mipaf.unaligned_fasta(b"", fa, 100, 100) == oracle("to_bed_extract", empty, tmp_path / "q.fa", 100, 100)


@pytest.mark.parametrize("seed", range(6))
def test_dechunk_text_implementations_write_the_oracles_bytes(tmp_path, seed):
    from cactus_amd import mipaf
    rng = np.random.default_rng(100 + seed)
    lines = []
    for k in range(20):
        qn, tn = f"id=Q|c{k % 3}|{int(rng.integers(1000, 9000))}|{int(rng.integers(0, 500))}", f"id=T|x|y{k % 2}|{int(rng.integers(1000, 9000))}|{int(rng.integers(0, 500))}"
        a, b = sorted(int(x) for x in rng.integers(0, 900, 2))
        c, d = sorted(int(x) for x in rng.integers(0, 900, 2))
        tags = "\tAS:i:%d\tcg:Z:%d=" % (k, b - a + 1) if k % 4 else ""
        lines.append(f"{qn}\t1000\t{a}\t{b}\t{'+-'[k % 2]}\t{tn}\t1000\t{c}\t{d}\t{b - a}\t{b - a}\t255{tags}\n")
    text = "".join(lines)
    path = tmp_path / "c.paf"
    path.write_text(text)
    for query_only in (False, True):
        want = oracle("dechunk", path, *(["--query"] if query_only else []))
        assert mipaf.dechunk_text(text.encode(), query_only=query_only) == want
        assert "".join(chunking.paf_dechunk_line(l, query_only) for l in lines).encode() == want


@pytest.mark.parametrize("seed,chunk,overlap", [(1, 1000, 100), (2, 700, 0), (3, 5000, 50), (4, 123, 45)])
def test_fasta_chunker_writes_the_oracles_files(tmp_path, seed, chunk, overlap):
    fa, _, _ = random_case(200 + seed, n_contigs=5, max_len=4000)
    src = tmp_path / "g.fa"
    src.write_bytes(fa)
    files = chunking.fasta_chunk(str(src), str(tmp_path / "chunks"), chunk, overlap)
    got = b"".join(b"== file %d\n" % k + open(f, "rb").read() for k, f in enumerate(files))
    assert got == oracle("chunk", src, chunk, overlap)


def two_genome_case(seed):
    """Two FASTA files (genomes A and B, several contigs each, one contig of B untouched) and alignments between them whose query
    and target intervals overlap, touch, nest and reach the contig ends -- the input of trim_unaligned_sequences."""
    rng = np.random.default_rng(seed)
    files = []
    for g in "AB":
        recs = []
        for k in range(2 + seed % 3):
            n = int(rng.integers(200, 6000))
            recs.append((f"id={g}|chr{k}", gen.random_sequence(n, rng)))
        files.append(recs)
    lines = []
    for _ in range(10 + seed):
        qa, ta = files[0][int(rng.integers(0, len(files[0])))], files[1][int(rng.integers(0, len(files[1]) - 1))]
        span = int(rng.integers(1, 150))
        qs, ts = int(rng.integers(0, len(qa[1]) - span + 1)), int(rng.integers(0, len(ta[1]) - span + 1))
        if rng.random() < 0.15:
            qs = 0
        if rng.random() < 0.15:
            ts = len(ta[1]) - span
        tags = f"\tAS:i:{span * 90}\tcg:Z:{span}=" if rng.random() < 0.8 else ""
        lines.append(f"{qa[0]}\t{len(qa[1])}\t{qs}\t{qs + span}\t{'+-'[int(rng.integers(0, 2))]}\t{ta[0]}\t{len(ta[1])}\t{ts}\t{ts + span}\t{span}\t{span}\t255{tags}\n")
    return files, "".join(lines)


@pytest.mark.parametrize("seed", range(8))
def test_trim_to_aligned_four_implementations_write_the_oracles_bytes(tmp_path, seed):
    """trim_unaligned_sequences (local_alignment.py:861-904): the Python cores, the native text code through the C ABI, the CLI faces
    bin/paffy to_bed | bin/faffy extract | bin/paffy upconvert driven by the job function, and the oracle's per-base restatement."""
    import sys
    import xml.etree.ElementTree as ET
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from localjob import LocalJob
    from cactus_amd import mipaf
    files, paf = two_genome_case(seed)
    flank = (0, 7, 60, 2000)[seed % 4]
    paths = []
    for k, recs in enumerate(files):
        p = tmp_path / f"g{k}.fa"
        p.write_bytes(gen.fasta_bytes(recs))
        paths.append(p)
    (tmp_path / "a.paf").write_text(paf)
    want = oracle("trim_aligned", tmp_path / "a.paf", flank, *paths).decode()
    sections = want.split("== ")
    want_files = [s.split("\n", 1)[1] for s in sections if s.startswith("file ")]
    want_paf = [s.split("\n", 1)[1] for s in sections if s.startswith("paf")][0]
    # Python cores
    trimmed, up = chunking.trim_to_aligned(paf, [[(n, s.tobytes().decode()) for n, s in recs] for recs in files], flank)
    got_files = [gen.fasta_bytes([(n, np.frombuffer(s.encode(), dtype=np.uint8)) for n, s in recs]).decode() for recs in trimmed]
    assert got_files == want_files and up == want_paf
    # native text code (C ABI)
    bed = mipaf.to_bed_text(paf.encode(), exclude_unaligned=True, include_inverted=True)
    assert bed.decode() == "".join(f"{n}\t{s}\t{e}\n" for n, s, e in chunking.aligned_intervals(paf.splitlines(True)))
    native = [mipaf.fasta_extract_text(bed, p.read_bytes(), flank, 1, True) for p in paths]
    assert [x.decode() for x in native] == want_files
    assert mipaf.upconvert_text(paf.encode(), native).decode() == want_paf
    # the CLI faces with the argv of the reference's job (local_alignment.py:877-899; the job function itself is the REFERENCE's own
    # body in tests/test_reference_jobs_cpu.py -- there is no mirror of it)
    def tool(argv, out):
        with open(out, "wb") as sink:
            r = subprocess.run([os.path.join(ROOT, "bin", argv[0])] + argv[1:], stdout=sink, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr
        return out
    bed_file = tool(['paffy', 'to_bed', "--binary", "--excludeUnaligned", "--includeInverted", '-i', str(tmp_path / "a.paf"), "--logLevel", "INFO"], str(tmp_path / "a.bed"))
    cli_files = [tool(['faffy', 'extract', "-i", bed_file, str(p), "--skipMissing", "--minSize", "1", "--flank", str(flank), "--logLevel", "INFO"], str(p) + ".trim") for p in paths]
    cli_paf = tool(['paffy', 'upconvert', "-i", str(tmp_path / "a.paf"), "--logLevel", "INFO"] + cli_files, str(tmp_path / "a.paf.trim"))
    assert [open(f).read() for f in cli_files] == want_files and open(cli_paf).read() == want_paf
    # dechunk undoes upconvert
    assert mipaf.dechunk_text(want_paf.encode()) == paf.encode()


def test_to_bed_and_extract_general_forms(tmp_path):
    """the forms of the other call sites (local_alignment.py:191-216, :476-489): --excludeAligned with a FASTA file equals the
    first-half implementation; an unknown BED name is an error without --skipMissing; faffy chunk through the CLI equals the oracle"""
    from cactus_amd import mipaf
    fa, paf, _ = random_case(5, n_contigs=4)
    bed = mipaf.to_bed_text(paf, fasta=fa, exclude_aligned=True, min_size=37)
    want = chunking.unaligned_intervals(paf.decode().splitlines(True), [(n.split()[0], len(s)) for n, s in
                                        ((n, s) for n, s in [(r[0], r[1]) for r in __import__("cactus_amd.blast_phase", fromlist=["x"]).parse_fasta_bytes(fa)])], 37)
    assert bed.decode() == "".join(f"{n}\t{s}\t{e}\n" for n, s, e in want)
    assert mipaf.fasta_extract_text(bed, fa, 5, 1, False) == mipaf.unaligned_fasta(paf, fa, 37, 5)
    with pytest.raises(Exception):
        mipaf.fasta_extract_text(b"nosuch\t0\t5\n", fa, 0, 1, False)
    assert mipaf.fasta_extract_text(b"nosuch\t0\t5\n", fa, 0, 1, True) == b""
    src = tmp_path / "g.fa"
    src.write_bytes(fa)
    out = tmp_path / "chunks"
    out.mkdir()
    p = subprocess.run([os.path.join(ROOT, "bin", "faffy"), "chunk", "-c", "700", "-o", "50", "--dir", str(out), str(src)], capture_output=True)
    assert p.returncode == 0, p.stderr
    files = sorted(os.listdir(out), key=lambda f: int(f.split("_")[1].split(".")[0]))
    got = b"".join(b"== file %d\n" % k + (out / f).read_bytes() for k, f in enumerate(files))
    assert got == oracle("chunk", src, 700, 50)

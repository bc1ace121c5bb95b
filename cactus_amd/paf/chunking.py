"""Data formats either side of the blast job: the `faffy chunk` FASTA
chunker and the `paffy dechunk` coordinate fix-up that make_chunked_alignments / combine_chunks shell
out to (/root/reference/src/cactus/paf/local_alignment.py:378-387 and :352).  The paffy submodule is
empty in the reference tree, so these follow its documented behaviour: a chunk record is named
`ORIGINAL|SEQLEN|CHUNKSTART`, chunks are chunkSize (+ overlapSize) long and are packed into files of
about chunkSize bases; dechunk strips the two suffix fields, shifts start/end by CHUNKSTART and restores
SEQLEN."""
from __future__ import annotations

import os


def _read_fasta(path):
    name, parts = None, []
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    yield name, "".join(parts)
                name, parts = line[1:].strip(), []
            else:
                parts.append(line.strip())
    if name is not None:
        yield name, "".join(parts)


def fasta_chunk(fasta_path: str, out_dir: str, chunk_size: int, overlap_size: int):
    """Returns the list of chunk files written (faffy chunk -c chunk_size -o overlap_size --dir out_dir)."""
    os.makedirs(out_dir, exist_ok=True)
    files, fh, remaining, idx = [], None, 0, 0
    for header, seq in _read_fasta(fasta_path):
        name = header.split()[0] if header.split() else header
        n = len(seq)
        for start in range(0, n, chunk_size):
            piece = seq[start:start + chunk_size + overlap_size]
            if fh is None or remaining <= 0:
                if fh is not None:
                    fh.close()
                path = os.path.join(out_dir, "chunk_{}.fa".format(idx))
                idx += 1
                fh = open(path, "w")
                files.append(path)
                remaining = chunk_size
            fh.write(">{}|{}|{}\n".format(name, n, start))
            for i in range(0, len(piece), 100):
                fh.write(piece[i:i + 100] + "\n")
            remaining -= len(piece)
    if fh is not None:
        fh.close()
    return files


def _split_chunk_name(name: str):
    base, seq_len, start = name.rsplit("|", 2)
    return base, int(seq_len), int(start)


def paf_dechunk_line(line: str, query_only: bool = False) -> str:
    f = line.rstrip("\n").split("\t")
    qname, qlen, qoff = _split_chunk_name(f[0])
    f[0], f[1] = qname, str(qlen)
    f[2], f[3] = str(int(f[2]) + qoff), str(int(f[3]) + qoff)
    if not query_only:
        tname, tlen, toff = _split_chunk_name(f[5])
        f[5], f[6] = tname, str(tlen)
        f[7], f[8] = str(int(f[7]) + toff), str(int(f[8]) + toff)
    return "\t".join(f) + "\n"


def paf_dechunk(in_path: str, out_path: str, query_only: bool = False, append: bool = True):
    with open(in_path) as fin, open(out_path, "a" if append else "w") as fout:
        for line in fin:
            if line.strip():
                fout.write(paf_dechunk_line(line, query_only))


# ---- outgroup trimming formats (local_alignment.py:421-526) --------------------------------------
def paf_invert_line(line: str) -> str:
    """`paffy invert`: swap query and target; I <-> D in the cg cigar; on the '-' strand the op order is reversed
    (the cigar always runs along the target's forward strand)."""
    import re
    f = line.rstrip("\n").split("\t")
    f[0], f[1], f[2], f[3], f[5], f[6], f[7], f[8] = f[5], f[6], f[7], f[8], f[0], f[1], f[2], f[3]
    for k in range(12, len(f)):
        if f[k].startswith("cg:Z:"):
            ops = re.findall(r"(\d+)([=XIDM])", f[k][5:])
            swap = {"I": "D", "D": "I"}
            ops = [(n, swap.get(o, o)) for n, o in ops]
            if f[4] == "-":
                ops.reverse()
            f[k] = "cg:Z:" + "".join(n + o for n, o in ops)
    return "\t".join(f) + "\n"


def unaligned_intervals(paf_lines, seq_lens, min_size: int):
    """Core of `paffy to_bed --excludeAligned --binary --minSize N` on PAF lines and [(query sequence name, length)]: the
    (name, start, end) intervals of QUERY sequence no alignment covers, at least min_size long, in sequence order.  Works on the
    alignments' query intervals (sorted and merged), not on a per-base array: a whole-genome call has a handful of records."""
    spans = {name: [] for name, _ in seq_lens}
    if len(spans) != len(seq_lens):
        raise ValueError("to_bed: a sequence name occurs twice in the FASTA file")
    for line in paf_lines:
        if not line.strip():
            continue
        t = line.split("\t", 4)
        s, e = int(t[2]), int(t[3])
        if e > s:
            spans[t[0]].append((s, e))
    out = []
    for name, n in seq_lens:
        pos = 0
        for s, e in sorted(spans[name]):
            if s - pos >= min_size and s > pos:
                out.append((name, pos, s))
            pos = max(pos, e)
        if n - pos >= min_size and n > pos:
            out.append((name, pos, n))
    return out


def extract_records(bed, records, flank: int):
    """Core of `faffy extract -i bed fa --flank F` on [(name, sequence)] (str or uint8 array): the BED intervals widened by flank
    (overlapping widened intervals merged) as [(NAME|SEQLEN|START, piece)], in sequence order."""
    by = {}
    lens = {name: len(seq) for name, seq in records}
    for name, s, e in bed:
        by.setdefault(name, []).append((max(0, s - flank), min(lens[name], e + flank)))
    out = []
    for name, seq in records:
        merged = []
        for s, e in sorted(by.get(name, [])):
            if merged and s <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], e))
            else:
                merged.append((s, e))
        for s, e in merged:
            out.append(("{}|{}|{}".format(name, len(seq), s), seq[s:e]))
    return out


def paf_to_bed_unaligned(paf_path: str, query_fasta: str, min_size: int):
    """`paffy to_bed --excludeAligned --binary --minSize N -i paf --queryFastaFile fa` (local_alignment.py:460-466):
    BED intervals (name, start, end) of QUERY sequence not covered by any alignment, at least min_size long."""
    with open(paf_path) as f:
        return unaligned_intervals(f, [(name.split()[0], len(seq)) for name, seq in _read_fasta(query_fasta)], min_size)


def fasta_extract(bed, fasta_path: str, out_path: str, flank: int):
    """`faffy extract -i bed fa --flank F` (local_alignment.py:470-475): the BED intervals, widened by `flank` on
    both sides (overlapping widened intervals are merged), written as records named NAME|SEQLEN|START so that
    `paffy dechunk --query` (paf_dechunk(..., query_only=True)) restores full-sequence coordinates."""
    records = [(name.split()[0], seq) for name, seq in _read_fasta(fasta_path)]
    with open(out_path, "w") as out:
        for name, piece in extract_records(bed, records, flank):
            out.write(">{}\n".format(name))
            for i in range(0, len(piece), 100):
                out.write(piece[i:i + 100] + "\n")


# ---- trimming the sequences to what is aligned (second half; local_alignment.py:861-904) ---------------------
def aligned_intervals(paf_lines, include_inverted: bool = True):
    """Core of `paffy to_bed --binary --excludeUnaligned [--includeInverted] -i paf` (local_alignment.py:878): the (name, start, end)
    intervals of sequence covered by at least one alignment -- by their query intervals and, with --includeInverted, by their
    target intervals too (the alignment read the other way round).  No FASTA file is given at this call site: sequences come in
    order of first appearance in the PAF (query before target within a line).  Sorted and merged intervals, not a per-base array."""
    spans, order = {}, []

    def add(name, s, e):
        if name not in spans:
            spans[name] = []
            order.append(name)
        if e > s:
            spans[name].append((s, e))
    for line in paf_lines:
        if not line.strip():
            continue
        t = line.split("\t", 9)
        add(t[0], int(t[2]), int(t[3]))
        if include_inverted:
            add(t[5], int(t[7]), int(t[8]))
    out = []
    for name in order:
        cur = None
        for s, e in sorted(spans[name]):
            if cur is not None and s <= cur[1]:
                cur[1] = max(cur[1], e)
            else:
                if cur is not None:
                    out.append((name, cur[0], cur[1]))
                cur = [s, e]
        if cur is not None:
            out.append((name, cur[0], cur[1]))
    return out


def extract_records_skip_missing(bed, records, flank: int, min_size: int = 1):
    """Core of `faffy extract -i bed fa --skipMissing --minSize N --flank F` (local_alignment.py:890-892): like extract_records, but
    BED lines that name a sequence which is not in this file are passed over (the BED of trim_unaligned_sequences covers the sequences
    of ALL files), and intervals shorter than min_size are dropped before they are widened."""
    have = {name for name, _ in records}
    return extract_records([(n, s, e) for n, s, e in bed if n in have and e - s >= min_size], records, flank)


def paf_upconvert_lines(paf_lines, record_names):
    """Core of `paffy upconvert -i paf trimmed_1.fa trimmed_2.fa ...` (local_alignment.py:899-900): the alignments rewritten to refer to
    the extracted sub-sequences -- the inverse of dechunk.  record_names: [(NAME|SEQLEN|START, length)] of every record of every trimmed
    file.  A query (target) interval lies inside exactly one record of its sequence: the name becomes the record's, the length the
    record's, start / end are shifted by -START; every other column and tag passes through.  A sequence none of whose records is in the
    files keeps its coordinates."""
    import bisect
    by = {}
    for full, n in record_names:
        base, seq_len, start = _split_chunk_name(full)
        by.setdefault(base, []).append((start, start + n, full))
    for v in by.values():
        v.sort()
    out = []

    def conv(name, s, e):
        recs = by.get(name)
        if recs is None:
            return None
        k = bisect.bisect_right(recs, (s, float("inf"), "")) - 1
        if k < 0 or not (recs[k][0] <= s and e <= recs[k][1]):
            raise ValueError("upconvert: {}:{}-{} lies in no extracted record".format(name, s, e))
        return recs[k]
    for line in paf_lines:
        if not line.strip():
            continue
        f = line.rstrip("\n").split("\t")
        q = conv(f[0], int(f[2]), int(f[3]))
        if q is not None:
            f[0], f[1], f[2], f[3] = q[2], str(q[1] - q[0]), str(int(f[2]) - q[0]), str(int(f[3]) - q[0])
        t = conv(f[5], int(f[7]), int(f[8]))
        if t is not None:
            f[5], f[6], f[7], f[8] = t[2], str(t[1] - t[0]), str(int(f[7]) - t[0]), str(int(f[8]) - t[0])
        out.append("\t".join(f) + "\n")
    return out


def trim_to_aligned(paf_text: str, fastas, flank: int, min_size: int = 1):
    """The three steps of trim_unaligned_sequences (local_alignment.py:861-904) on text: fastas = [[(name, sequence)]] per file.
    Returns ([[(NAME|SEQLEN|START, piece)]] per file, upconverted PAF text)."""
    lines = paf_text.splitlines(True)
    bed = aligned_intervals(lines, include_inverted=True)
    trimmed = [extract_records_skip_missing(bed, recs, flank, min_size) for recs in fastas]
    names = [(n, len(s)) for recs in trimmed for n, s in recs]
    return trimmed, "".join(paf_upconvert_lines(lines, names))

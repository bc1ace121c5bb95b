"""The chunk-scale workloads of BASELINE.json configs[3] and configs[4], made the same way everywhere they are used (bench.py, the
script that writes their oracle digests, the GPU parity tests).  Both stand in for data that is not available offline
 and are chunked exactly as the reference's CPU path chunks a genome:
`faffy chunk -c chunkSize -o overlapSize` (/root/reference/src/cactus/paf/local_alignment.py:378-387,
cactus_progressive_config.xml:90-92), every (target chunk, query chunk) pair an independent lastz job (:395-405).

  chr20      one synthetic chromosome (64 444 167 bp) against a 1.3 % diverged, half soft-masked copy; chunkSize 30 Mb + 10 kb
             -> 3 x 3 chunk pairs; option set "one" (distance <= 0.05, cactus_progressive_config.xml:131)
  hm         a human-mouse stand-in at 1/86 scale: a 36 Mb "human" of four chromosomes against a 31 Mb "mouse" of five, made of
             shuffled and partly inverted syntenic segments in which conserved blocks (~ 37 % of the sequence, 70-92 % identity)
             alternate with unrelated sequence; both about half soft-masked; chunkSize 4.5 Mb + 10 kb -> 7 x 6 = 42 chunk pairs;
             option set "default" (distance > 0.25, :136).  The scale factor is stated wherever a figure of it is quoted.
"""
from __future__ import annotations

import numpy as np

from cactus_amd import gen

OVERLAP = 10_000                                   # overlapSize, cactus_progressive_config.xml:92
OPTIONS_ONE = "--step=2 --ambiguous=iupac,100,100 --ydrop=3000 --notransition --queryhspbest=100000"                              # xml:131
OPTIONS_DEFAULT = "--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=100000"    # xml:136


def chunk_records(records, chunk_size: int, overlap: int):
    """faffy chunk on [(name, uint8 array)]: pieces NAME|SEQLEN|START of chunk_size + overlap bases, packed into files of about
    chunk_size bases (same rule as cactus_amd.paf.chunking.fasta_chunk).  Returns one list of (name, array) per chunk file."""
    files, cur, remaining = [], None, 0
    for name, seq in records:
        n = len(seq)
        for start in range(0, n, chunk_size):
            piece = seq[start:start + chunk_size + overlap]
            if cur is None or remaining <= 0:
                cur = []
                files.append(cur)
                remaining = chunk_size
            cur.append((f"{name}|{n}|{start}", piece))
            remaining -= len(piece)
    return files


class ChunkedGenomePair:
    """A genome pair as its chunk files (FASTA bytes) and the list of chunk pairs in the order make_chunked_alignments makes the
    jobs (for every chunk of A, every chunk of B: local_alignment.py:395-405)."""

    def __init__(self, key, describe, options, a_records, b_records, chunk_size, overlap=OVERLAP):
        self.key, self.describe, self.options = key, describe, options
        self.chunk_size, self.overlap = chunk_size, overlap
        self.bases = (sum(len(s) for _, s in a_records), sum(len(s) for _, s in b_records))
        self.tfa = [gen.fasta_bytes(f) for f in chunk_records(a_records, chunk_size, overlap)]
        self.qfa = [gen.fasta_bytes(f) for f in chunk_records(b_records, chunk_size, overlap)]
        self.pairs = [(i, j) for i in range(len(self.tfa)) for j in range(len(self.qfa))]

    def weights(self):
        return [float(len(self.tfa[i])) * float(len(self.qfa[j])) for i, j in self.pairs]


def chr20(bases: int = 64_444_167, chunk: int = 30_000_000) -> ChunkedGenomePair:
    t, q = gen.make_pair(bases, 3001, sub_rate=0.013, indel_rate=0.002, mask_frac=0.5)
    d = (f"synthetic chr20 x chr20 (BASELINE configs[3], SURVEY 8d config 4): {len(t)} x {len(q)} bp at 1.3 % divergence, half soft-masked, seed 3001; "
         f"chunkSize {chunk} + overlap {OVERLAP}")
    w = ChunkedGenomePair("chr20" if (bases, chunk) == (64_444_167, 30_000_000) else f"chr20_{bases}_{chunk}", d, OPTIONS_ONE,
                          [("id=simT|chr20", t)], [("id=simQ|chr20", q)], chunk)
    w.describe += f" -> {len(w.tfa)} x {len(w.qfa)} chunk pairs, option set \"one\""
    return w


def make_human_mouse_like(a_bases: int, seed: int, conserved_mean=1500, filler_mean=2500, sub_rate=(0.08, 0.30), indel_rate=0.01,
                          segment=(200_000, 1_200_000), inverted_frac=0.3, filler_scale=0.8):
    """(a_records, b_records): genome A = iid chromosomes; genome B = A's syntenic segments shuffled, a share of them inverted, inside
    each segment conserved blocks (mutated copies) alternating with unrelated filler, cut into chromosomes of its own."""
    rng = np.random.default_rng(seed)
    a_lens = [int(a_bases * f) for f in (0.39, 0.305, 0.195)]
    a_lens.append(a_bases - sum(a_lens))
    a_seqs = [gen.random_sequence(n, rng) for n in a_lens]
    segs = []
    for c, s in enumerate(a_seqs):
        pos = 0
        while pos < len(s):
            ln = int(rng.integers(segment[0], segment[1]))
            if len(s) - pos - ln < segment[0]:
                ln = len(s) - pos
            segs.append((c, pos, ln))
            pos += ln
    order = rng.permutation(len(segs))
    b_parts = []
    for k in order:
        c, pos, ln = segs[k]
        src = a_seqs[c][pos:pos + ln]
        pieces, at, conserved = [], 0, bool(rng.integers(0, 2))
        while at < ln:
            m = int(min(ln - at, rng.geometric(1.0 / (conserved_mean if conserved else filler_mean))))
            if conserved:                                  # a mutated copy (a substitution rate of its own: exons to barely alignable) ...
                blk = src[at:at + m]
                pieces.append(gen.mutate(blk, rng, float(rng.uniform(*sub_rate)), indel_rate) if m >= 8 else blk.copy())
            else:                                          # ... or unrelated sequence, somewhat shorter than what it replaces
                pieces.append(gen.random_sequence(max(1, int(m * filler_scale)), rng))
            at += m
            conserved = not conserved
        seg_b = np.concatenate(pieces)
        if rng.random() < inverted_frac:
            seg_b = gen.revcomp(seg_b)
        b_parts.append(seg_b)
    b_all = np.concatenate(b_parts)
    cuts = np.cumsum([int(len(b_all) * f) for f in (0.30, 0.25, 0.20, 0.15)])
    b_seqs = np.split(b_all, cuts)
    a_out = [(f"id=simHuman|chr{k + 1}", gen.n_runs(gen.soft_mask(s, rng, 0.48), rng, 2, 500)) for k, s in enumerate(a_seqs)]
    b_out = [(f"id=simMouse|chr{k + 1}", gen.n_runs(gen.soft_mask(s, rng, 0.42), rng, 2, 500)) for k, s in enumerate(b_seqs)]
    return a_out, b_out


def human_mouse(a_bases: int = 36_000_000, chunk: int = 4_500_000, seed: int = 5001) -> ChunkedGenomePair:
    a, b = make_human_mouse_like(a_bases, seed)
    nb = sum(len(s) for _, s in b)
    scale = 3.1e9 / a_bases
    d = (f"human x mouse stand-in at 1/{scale:.0f} scale (BASELINE configs[4], SURVEY 8d config 5): {a_bases} bp in {len(a)} chromosomes x {nb} bp in {len(b)}, "
         f"shuffled / partly inverted syntenic segments, conserved blocks ~37 % of the sequence at 70-92 % identity, about half soft-masked, seed {seed}; "
         f"chunkSize {chunk} + overlap {OVERLAP} (NOT the reference's chunk size: its CPU path chunks at 30 Mb, cactus_progressive_config.xml:90 -- at this scale "
         f"one chunk pair; {chunk} keeps the many-pairs shape of the real run's 104 x 91 grid, so this leg times MANY SMALL chunk pairs, not 30 Mb ones)")
    if chunk >= 30_000_000:
        d = d[:d.index("chunkSize")] + f"chunkSize {chunk} + overlap {OVERLAP} -- the reference's own chunk size (cactus_progressive_config.xml:90): FEW LARGE chunk pairs"
    key = "hm" if (a_bases, chunk, seed) == (36_000_000, 4_500_000, 5001) else "hm30" if (a_bases, chunk, seed) == (36_000_000, 30_000_000, 5001) else f"hm_{a_bases}_{chunk}_{seed}"
    w = ChunkedGenomePair(key, d, OPTIONS_DEFAULT, a, b, chunk)
    w.describe += f" -> {len(w.tfa)} x {len(w.qfa)} = {len(w.pairs)} chunk pairs, option set \"default\""
    return w


def by_name(name: str, **kw) -> ChunkedGenomePair:
    if name == "hm30":                                     # the same genome pair cut at the reference's chunk size: 2 x 2 chunk pairs, one of them 30 Mb x 30 Mb
        return human_mouse(chunk=30_000_000, **kw)
    return {"chr20": chr20, "hm": human_mouse}[name](**kw)

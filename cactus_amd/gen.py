"""Synthetic genome-chunk generator for the blast-phase benchmarks and parity tests.

Stands in for the evolver / chr20 / whole-genome inputs that BASELINE.json's configs name
(/root/reference/examples/evolverMammals.txt:3-7 are URLs; there is no network).  The recipe
follows SURVEY.md section 8(d) "Config 2": an iid ancestor, a query derived from it by substitutions
(transitions twice as likely as each transversion), geometric-length indels, one inversion, one
replaced segment, soft-masked runs and two N runs.  numpy's PCG64 replaces the xoshiro generator
named there; the bytes are what both the oracle and the HIP path consume, so only determinism
for a given (numpy version, seed) matters.
"""
from __future__ import annotations

import numpy as np

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b


def random_sequence(n: int, rng: np.random.Generator, p=(0.3, 0.2, 0.2, 0.3)) -> np.ndarray:
    return _BASES[rng.choice(4, size=n, p=p)]


def revcomp(seq: np.ndarray) -> np.ndarray:
    return _COMP[seq[::-1]]


def mutate(seq: np.ndarray, rng: np.random.Generator, sub_rate: float, indel_rate: float,
           indel_mean: float = 3.0, indel_max: int = 50) -> np.ndarray:
    """Substitutions (transition : each transversion = 2 : 1 : 1) then indel events."""
    n = len(seq)
    out = seq.copy()
    # substitutions
    hit = rng.random(n) < sub_rate
    idx = np.nonzero(hit)[0]
    if len(idx):
        code = np.searchsorted(_BASES, out[idx])          # A0 C1 G2 T3
        kind = rng.choice(3, size=len(idx), p=(0.5, 0.25, 0.25))
        # transition: xor 2 ; transversions: xor 1 / xor 3
        xor = np.array([2, 1, 3])[kind]
        out[idx] = _BASES[code ^ xor]
    if indel_rate <= 0:
        return out
    # indels: walk event positions
    ev = np.nonzero(rng.random(n) < indel_rate)[0]
    pieces = []
    last = 0
    for pos in ev:
        if pos < last:
            continue
        ln = int(min(indel_max, rng.geometric(1.0 / indel_mean)))
        pieces.append(out[last:pos])
        if rng.random() < 0.5:                            # deletion
            last = min(n, pos + ln)
        else:                                             # insertion
            pieces.append(random_sequence(ln, rng))
            last = pos
    pieces.append(out[last:])
    return np.concatenate(pieces)


def soft_mask(seq: np.ndarray, rng: np.random.Generator, frac: float, mean_run: int = 300) -> np.ndarray:
    out = seq.copy()
    n = len(out)
    if frac <= 0 or n == 0:
        return out
    n_runs = max(1, int(frac * n / mean_run))
    starts = rng.integers(0, n, size=n_runs)
    lens = rng.geometric(1.0 / mean_run, size=n_runs)
    for s, l in zip(starts, lens):
        out[s:s + l] |= 0x20
    return out


def n_runs(seq: np.ndarray, rng: np.random.Generator, count: int, length: int) -> np.ndarray:
    out = seq.copy()
    for _ in range(count):
        if len(out) <= length:
            break
        s = int(rng.integers(0, len(out) - length))
        out[s:s + length] = ord("N")
    return out


def make_pair(n: int, seed: int, sub_rate=0.15, indel_rate=0.01, mask_frac=0.2,
              inversion=True, replaced=True, nruns=2, homologous=True):
    """Returns (target_bytes, query_bytes) uppercase/lowercase ASCII numpy arrays.
    homologous=False gives the 'pure-random' pair (seed/ungapped throughput isolation)."""
    rng = np.random.default_rng(seed)
    anc = random_sequence(n, rng)
    target = anc.copy()
    if homologous:
        q = anc.copy()
        if inversion and n >= 20:
            a = int(0.4 * n); b = a + max(1, n // 20)
            q[a:b] = revcomp(q[a:b])
        if replaced and n >= 10:
            a = int(0.7 * n); b = a + max(1, n // 10)
            q[a:b] = random_sequence(b - a, rng)
        query = mutate(q, rng, sub_rate, indel_rate)
    else:
        query = random_sequence(n, rng)
    target = soft_mask(target, rng, mask_frac)
    query = soft_mask(query, rng, mask_frac)
    if nruns:
        target = n_runs(target, rng, nruns, min(500, max(1, n // 200)))
        query = n_runs(query, rng, nruns, min(500, max(1, n // 200)))
    return target, query


def fasta_bytes(records) -> bytes:
    """records: iterable of (name, uint8 array).  60 columns per line like faffy/sonLib writers."""
    out = []
    for name, seq in records:
        out.append(b">" + name.encode() + b"\n")
        s = seq.tobytes()
        out.extend(s[i:i + 60] + b"\n" for i in range(0, len(s), 60))
    return b"".join(out)


def write_fasta(path: str, records) -> None:
    with open(path, "wb") as f:
        f.write(fasta_bytes(records))


def make_tree_genomes(ancestor_len: int, seed: int, tree=None, ancestors: bool = False):
    """Synthetic stand-in for the evolver data sets (their FASTA files are URLs,
    /root/reference/examples/evolverMammals.txt:3-7; no network): leaves evolved from one ancestor along the
    branch lengths of the evolverMammals guide tree (/root/reference/examples/evolverMammals.txt:1).
    Returns {leaf_name: uint8 array}.  Per-branch substitution rate = branch length (capped), indel rate = 1/15 of it.
    ancestors=True also returns the internal nodes' sequences (the blast phase of a progressive run aligns reconstructed
    ancestors as ingroups; the stand-in uses the true ones, unmasked) and the root as "Anc0"; the leaves are
    the same bytes either way."""
    if tree is None:
        # ((simHuman_chr6:0.144018,(simMouse_chr6:0.084509,simRat_chr6:0.091589):0.271974):0.020593,
        #  (simCow_chr6:0.18908,simDog_chr6:0.16303):0.032898);
        tree = ("root", [("hmr", 0.020593, [("simHuman_chr6", 0.144018, []),
                                            ("mr", 0.271974, [("simMouse_chr6", 0.084509, []), ("simRat_chr6", 0.091589, [])])]),
                         ("cd", 0.032898, [("simCow_chr6", 0.18908, []), ("simDog_chr6", 0.16303, [])])])
    rng = np.random.default_rng(seed)
    root_seq = random_sequence(ancestor_len, rng)
    out = {}

    def walk(node_children, seq):
        for name, blen, kids in node_children:
            child = mutate(seq, rng, min(0.6, blen), min(0.04, blen / 15.0))
            n = len(child)
            if n > 2000:                                   # one inversion and one large deletion per branch
                a = int(rng.integers(0, n - n // 20)); child[a:a + n // 20] = revcomp(child[a:a + n // 20])
                d = int(rng.integers(0, n - n // 40)); child = np.concatenate([child[:d], child[d + n // 40:]])
            if kids:
                if ancestors:
                    out[{"hmr": "Anc1", "cd": "Anc2"}.get(name, name)] = child.copy()
                walk(kids, child)
            else:
                out[name] = soft_mask(child, rng, 0.15)

    if ancestors:
        out["Anc0"] = root_seq.copy()
    walk(tree[1], root_seq)
    return out


# ---- synthetic PAF for the chaining stage --------------------------------------------------------------------
# No sequences are needed: chain / tile / trim only read coordinates, scores and cigars.  Syntenic series of alignments (the
# material chains are made of) plus unrelated noise, valid cigars, both strands, occasional overlaps, equal scores and ragged
# ends.  Deterministic for a seed (python's random.Random).
def random_paf(seed: int, n_series: int = 6, per_series=(1, 12), n_q: int = 2, n_t: int = 2, contig_len: int = 200_000, noise: int = 10,
               ragged: bool = True) -> str:
    import random
    rng = random.Random(seed)
    qnames = [f"id=Q|chr{k}" for k in range(n_q)]
    tnames = [f"id=T|chr{k}" for k in range(n_t)]

    def cigar(max_cols):
        ops, cols = [], 0
        first = True
        while cols < max_cols:
            r = rng.random()
            if first and ragged and r < 0.15:
                o, n = rng.choice("XID"), rng.randint(1, 6)                # alignments a real aligner would not emit, the tool must still be defined
            elif ops and ops[-1][1] == "=":
                o = rng.choice("XXXID")
                n = rng.randint(1, 3) if o == "X" else rng.randint(1, 40)
            else:
                o, n = "=", rng.randint(1, 120)
            if ops and ops[-1][1] == o:
                continue
            ops.append((n, o)); cols += n; first = False
        if not ragged or rng.random() < 0.8:
            if ops[-1][1] != "=":
                ops.append((rng.randint(1, 50), "="))
        return ops

    lines = []

    def emit(qn, tn, strand, qpos, tpos, ops):
        ql = tl = contig_len
        qspan = sum(n for n, o in ops if o != "D"); tspan = sum(n for n, o in ops if o != "I")
        if strand == "+":
            qs, qe = qpos, qpos + qspan
        else:
            qe, qs = qpos, qpos - qspan
        ts, te = tpos, tpos + tspan
        if qs < 0 or qe > ql or te > tl or qspan == 0 or tspan == 0:
            return None
        nm = sum(n for n, o in ops if o == "=")
        nb = sum(n for n, _ in ops)
        score = max(1, 95 * nm - 110 * sum(n for n, o in ops if o == "X") - sum(400 + 30 * n for n, o in ops if o in "ID"))
        if rng.random() < 0.1:
            score = 5000                                                    # ties
        tags = [f"AS:i:{score}"] if rng.random() < 0.97 else []
        if rng.random() < 0.97:
            tags.append("cg:Z:" + "".join(f"{n}{o}" for n, o in ops))
        lines.append("\t".join(str(x) for x in [qn, ql, qs, qe, strand, tn, tl, ts, te, nm, nb, 255] + tags))
        return qspan, tspan

    for _ in range(n_series):
        qn, tn, strand = rng.choice(qnames), rng.choice(tnames), rng.choice("+-")
        qpos = rng.randint(1000, contig_len // 2) if strand == "+" else rng.randint(contig_len // 2, contig_len - 1000)
        tpos = rng.randint(1000, contig_len // 2)
        for _ in range(rng.randint(*per_series)):
            ops = cigar(rng.randint(30, 3000))
            got = emit(qn, tn, strand, qpos, tpos, ops)
            if got is None:
                break
            gq, gt = rng.choice([0, 0, 5, 300, 4000, 60000]), rng.choice([0, 3, 200, 5000])
            if rng.random() < 0.2:
                gq, gt = -rng.randint(1, 200), -rng.randint(1, 200)            # overlapping neighbours
            qpos = qpos + got[0] + gq if strand == "+" else qpos - got[0] - gq
            tpos = tpos + got[1] + gt
    for _ in range(noise):
        qn, tn, strand = rng.choice(qnames), rng.choice(tnames), rng.choice("+-")
        emit(qn, tn, strand, rng.randint(5000, contig_len - 5000), rng.randint(0, contig_len - 5000), cigar(rng.randint(20, 600)))
    rng.shuffle(lines)
    return "".join(l + "\n" for l in lines)

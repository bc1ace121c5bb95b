"""Seeded parity cases shared by the CPU and GPU test modules.  Each case is (name, target_fasta_bytes,
query_fasta_bytes, lastz option list).  Sizes are chosen so the CPU oracle finishes each in well under
a second; edge cases follow what the reference's inputs can contain after faffy chunk + header
sanitising: many ragged contigs, empty records, N runs, soft-masked runs, IUPAC letters."""
from __future__ import annotations

import numpy as np

from cactus_amd import gen

DEFAULT = "--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=100000".split()
ONE = "--step=2 --ambiguous=iupac,100,100 --ydrop=3000 --notransition --queryhspbest=100000".split()
TWO = "--step=5 --ambiguous=iupac,100,100 --ydrop=3000 --queryhspbest=100000".split()
THREE = "--step=4 --ambiguous=iupac,100,100 --ydrop=3500 --hspthresh=2800 --queryhspbest=100000".split()
FOUR = "--step=3 --ambiguous=iupac,100,100 --ydrop=3500 --hspthresh=2600 --gappedthresh=2800 --queryhspbest=100000".split()
FIVE = "--step=2 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2400 --gappedthresh=2600 --queryhspbest=100000".split()
KEG_DEFAULT = "--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400".split()


def _fa(records):
    return gen.fasta_bytes(records)


def _arr(s: str) -> np.ndarray:
    return np.frombuffer(s.encode(), dtype=np.uint8).copy()


def pair(n, seed, **kw):
    t, q = gen.make_pair(n, seed, **kw)
    return _fa([("id=simT|chr1", t)]), _fa([("id=simQ|chr1", q)])


def multi_contig(seed):
    """Ragged multi-record files: empty records, records shorter than a seed, exact seed length, long ones;
    query contigs are shuffled, mutated and partly reverse-complemented pieces of the target."""
    rng = np.random.default_rng(seed)
    tl = [0, 5, 18, 19, 20, 700, 3000, 1, 6000, 0, 2500]
    trecs = [("id=T|ctg%d" % i, gen.random_sequence(n, rng)) for i, n in enumerate(tl)]
    qrecs = []
    for i in (6, 8, 5, 10, 3, 4):
        s = gen.mutate(trecs[i][1], rng, 0.08, 0.004) if len(trecs[i][1]) > 50 else trecs[i][1].copy()
        if i in (8, 5):
            s = gen.revcomp(s)
        qrecs.append(("id=Q|piece%d some description" % i, s))
    qrecs.insert(2, ("id=Q|empty", np.zeros(0, dtype=np.uint8)))
    qrecs.append(("id=Q|tiny", _arr("ACGTACGTAC")))
    return _fa(trecs), _fa(qrecs)


def low_complexity(seed):
    rng = np.random.default_rng(seed)
    unit = _arr("AT")
    t = np.concatenate([gen.random_sequence(1500, rng), np.tile(unit, 120), gen.random_sequence(1500, rng),
                        np.tile(_arr("CAG"), 100), gen.random_sequence(800, rng), np.tile(_arr("A"), 200),
                        gen.random_sequence(500, rng)])
    q = gen.mutate(t, rng, 0.05, 0.003)
    return _fa([("id=T|lc", t)]), _fa([("id=Q|lc", q)])


def tandem(seed):
    """Tandem repeats put many seed hits on neighbouring diagonals and several HSPs on one diagonal."""
    rng = np.random.default_rng(seed)
    unit = gen.random_sequence(137, rng)
    rep = np.concatenate([gen.mutate(unit, rng, 0.04, 0.0) for _ in range(40)])
    t = np.concatenate([gen.random_sequence(2000, rng), rep, gen.random_sequence(2000, rng)])
    q = np.concatenate([gen.random_sequence(500, rng), gen.mutate(rep, rng, 0.06, 0.002), gen.random_sequence(700, rng)])
    return _fa([("id=T|tr", t)]), _fa([("id=Q|tr", q)])


def iupac_and_n(seed):
    rng = np.random.default_rng(seed)
    t = gen.random_sequence(6000, rng)
    q = gen.mutate(t, rng, 0.05, 0.003)
    t[1000:1010] = ord("N"); t[3000:3003] = _arr("RYK"); t[4000:4200] |= 0x20
    q[2000:2060] = ord("n"); q[500] = ord("W")
    return _fa([("id=T|iu", t)]), _fa([("id=Q|iu", q)])


def identical(n, seed):
    rng = np.random.default_rng(seed)
    t = gen.random_sequence(n, rng)
    return _fa([("id=T|same", t)]), _fa([("id=Q|same", t.copy())])


def revcomp_query(n, seed):
    rng = np.random.default_rng(seed)
    t = gen.random_sequence(n, rng)
    return _fa([("id=T|fw", t)]), _fa([("id=Q|rc", gen.revcomp(gen.mutate(t, rng, 0.03, 0.001)))])


def all_masked(seed):
    rng = np.random.default_rng(seed)
    t = gen.random_sequence(3000, rng)
    return _fa([("id=T|m", t | 0x20)]), _fa([("id=Q|m", t.copy())])


def build_cases():
    cases = [
        ("homolog_20k_default", *pair(20000, 1), DEFAULT),
        ("homolog_60k_four", *pair(60000, 2, sub_rate=0.12, indel_rate=0.008), FOUR),
        ("homolog_50k_three", *pair(50000, 12, sub_rate=0.1, indel_rate=0.006), THREE),
        ("homolog_50k_five", *pair(50000, 13, sub_rate=0.13, indel_rate=0.008), FIVE),
        ("close_100k_one", *pair(100000, 3, sub_rate=0.02, indel_rate=0.002), ONE),
        ("close_80k_two", *pair(80000, 4, sub_rate=0.05, indel_rate=0.003), TWO),
        ("random_50k", *pair(50000, 43, homologous=False), DEFAULT),
        ("kegalign_default_30k", *pair(30000, 6), KEG_DEFAULT),
        ("multi_contig_ragged", *multi_contig(7), DEFAULT),
        ("low_complexity_entropy", *low_complexity(8), DEFAULT),
        ("low_complexity_noentropy", *low_complexity(8), DEFAULT + ["--noentropy"]),
        ("tandem_repeats", *tandem(9), DEFAULT),
        ("iupac_and_n", *iupac_and_n(10), DEFAULT),
        ("identical_5k", *identical(5000, 11), DEFAULT),
        ("identical_lastz_defaults", *identical(3000, 14), ["--ambiguous=iupac,100,100"]),
        ("revcomp_query", *revcomp_query(8000, 15), DEFAULT),
        ("all_softmasked_target", *all_masked(16), DEFAULT),
        ("tiny_sequences", _fa([("t", _arr("ACGTACGTACGTACGTAA"))]), _fa([("q", _arr("ACGTACGTACGTACGTAA"))]), DEFAULT),
        ("exactly_one_seed", _fa([("t", _arr("ACGTTGCATGCAAGTCCGA"))]), _fa([("q", _arr("ACGTTGCATGCAAGTCCGA"))]), DEFAULT),
        ("empty_query", *[pair(2000, 17)[0], b""], DEFAULT),
        ("empty_target", *[b"", pair(2000, 17)[1]], DEFAULT),
        ("hspbest_5", *pair(30000, 18), "--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=5".split()),
        ("hspbest_1_multi", *multi_contig(19), "--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --queryhspbest=1".split()),
        ("wide_rows_ydrop_default", *pair(15000, 20, sub_rate=0.1, indel_rate=0.02), ["--ambiguous=iupac,100,100", "--hspthresh=2200"]),
        ("lds_ring_overflow_ydrop70000", *pair(6000, 21), ["--ambiguous=iupac,100,100", "--ydrop=70000", "--hspthresh=2200"]),
        ("ungapped_only", *pair(20000, 22), DEFAULT + ["--ungapped"]),
        ("xdrop_small", *pair(20000, 23), DEFAULT + ["--xdrop=300"]),
    ]
    return cases


CASES = build_cases()
CASE_IDS = [c[0] for c in CASES]

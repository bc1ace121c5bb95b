"""Randomised differential test of the blocked path (cactus_amd/csrc/mb_multi.cpp): random multi-contig files with repeated units, a random block size
(so that target and query need several blocks), random per-query HSP limits (--queryhspbest, --queryhsplimit, both tie rules), gapped and the repeat
masker's ungapped call, one or two logical devices -- the assembled bytes against ONE oracle run over the whole files.
    python scripts/gpu_multi_fuzz.py [cases] [first seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cactus_amd import gen, miblast
from oracle import olz

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
bad = refused = 0
t_start = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    units = [gen.random_sequence(int(rng.integers(150, 900)), rng) for _ in range(int(rng.integers(1, 4)))]
    def piece():
        u = units[int(rng.integers(0, len(units)))]
        r = rng.random()
        v = u.copy() if r < 0.35 else gen.mutate(u, rng, float(rng.choice([0.01, 0.04, 0.09])), float(rng.choice([0.0, 0.003])))
        return gen.revcomp(v) if rng.random() < 0.3 else v
    def contig(whole_unit_p):
        if rng.random() < whole_unit_p:
            return piece()
        parts = []
        for _ in range(int(rng.integers(1, 4))):
            parts.append(gen.random_sequence(int(rng.integers(100, 1500)), rng))
            if rng.random() < 0.8:
                parts.append(piece())
        return np.concatenate(parts)
    trecs = [("id=T|c%d" % k, contig(0.3)) for k in range(int(rng.integers(3, 10)))]
    qrecs = [("id=Q|s%d" % k, contig(0.4)) for k in range(int(rng.integers(1, 7)))]
    tf, qf = gen.fasta_bytes(trecs), gen.fasta_bytes(qrecs)
    longest = max(max(len(x[1]) for x in trecs), max(len(x[1]) for x in qrecs))
    block = int(longest + rng.integers(1, 3000))
    args = ["--ambiguous=iupac,100,100", "--step=%d" % int(rng.integers(1, 4)), "--hspthresh=%d" % int(rng.choice([2200, 3000])), "--ydrop=%d" % int(rng.choice([3000, 4000]))]
    if rng.random() < 0.3: args.append("--notransition")
    general = rng.random() < 0.25
    if general:
        args += ["--ungapped", "--format=general:name1,zstart1,end1,name2,zstart2+,end2+", "--markend"]
    if rng.random() < 0.75: args.append("--queryhspbest=%d" % int(rng.integers(1, 7)))
    if rng.random() < 0.45: args.append("--queryhsplimit=keep,nowarn:%d" % int(rng.integers(1, 8)))
    if rng.random() < 0.4: args.append("--miblast-hspbest-ties=later")
    pm = miblast.params_from_args(args)
    want = olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}))
    os.environ["MIBLAST_BLOCK_BASES"] = str(block)
    n_dev = 1 + int(rng.integers(0, 2))
    os.environ["MIBLAST_DEVICE_MAP"] = ",".join(["0"] * n_dev)
    m = miblast.Multi(n_dev)
    try:
        got, st = m.align_fasta_pairs([(tf, qf)], pm)
    except miblast.MiblastError as e:
        refused += 1
        print("REFUSED case", case, " ".join(args), "block", block, str(e)[:120])
        continue
    finally:
        m.close()
    if got != want["paf"]:
        bad += 1
        print("MISMATCH case", case, "block", block, "devices", n_dev, " ".join(args), "lines", got.count(b"\n"), want["paf"].count(b"\n"))
print("multi fuzz: %d cases, %d mismatches, %d refused, %.1f s" % (n_cases, bad, refused, time.time() - t_start))
sys.exit(1 if bad or refused else 0)

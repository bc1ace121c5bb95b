"""Randomised differential test: random pairs / structures / lastz options, GPU vs oracle, byte for byte."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cactus_amd import gen, miblast
from oracle import olz

KEYS = ["seed_lookups", "seed_hits", "hits_extended", "ungapped_cols", "hsps_pre_entropy", "hsps", "anchors", "anchors_skipped", "dp_sides", "dp_cells", "dp_rows", "alignments"]
# FUZZ_WRITE_DIGESTS=<file>: no GPU -- the oracle's answer of every case (md5 of the PAF, counters, record counts) into <file> (JSON; run on the CPU box,
# several processes side by side for chunk-scale cases); FUZZ_READ_DIGESTS=<file>: no oracle -- the GPU's answers against that file (the oracle's time
# stays off the GPU box).  Neither: both run here, everything compared.
import hashlib, json
WRITE, READ = os.environ.get("FUZZ_WRITE_DIGESTS"), os.environ.get("FUZZ_READ_DIGESTS")
ctx = None if WRITE else miblast.Context(0)
digests = json.load(open(READ)) if READ else {}
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
t_start = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.integers(int(os.environ.get('FUZZ_NMIN', '2000')), int(os.environ.get('FUZZ_NMAX', '60000'))))      # (FUZZ_NMIN / FUZZ_NMAX: chunk-scale cases, e.g. 1700000 3000000)
    sub = float(rng.choice([0.0, 0.01, 0.03, 0.08, 0.15, 0.25]))
    indel = float(rng.choice([0.0, 0.001, 0.005, 0.02]))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        t, q = gen.make_pair(n, seed0 + case, sub_rate=sub, indel_rate=indel, mask_frac=float(rng.choice([0, 0.2, 0.6])))
        trecs, qrecs = [("T|c0", t)], [("Q|c0", q)]
    elif kind == 1:      # many contigs both sides
        anc = gen.random_sequence(n, rng)
        cuts = sorted(set(int(x) for x in rng.integers(0, n, size=int(rng.integers(1, 12))))) + [n]
        trecs, qrecs, s = [], [], 0
        for k, e in enumerate(cuts):
            piece = anc[s:e]; s = e
            trecs.append(("T|c%d" % k, piece))
            m = gen.mutate(piece, rng, sub, indel) if len(piece) > 30 else piece.copy()
            if rng.random() < 0.4: m = gen.revcomp(m)
            qrecs.append(("Q|c%d extra" % k, m))
        rng.shuffle(qrecs)
    elif kind == 2:      # tandem / low complexity
        unit = gen.random_sequence(int(rng.integers(2, 200)), rng)
        rep = np.tile(unit, max(2, 3000 // len(unit)))
        t = np.concatenate([gen.random_sequence(1500, rng), rep, gen.random_sequence(1500, rng)])
        q = gen.mutate(t, rng, sub, indel)
        trecs, qrecs = [("T|rep", t)], [("Q|rep", q)]
    elif kind == 3:      # identical / near identical long runs
        t = gen.random_sequence(n, rng)
        q = t.copy()
        for _ in range(int(rng.integers(0, 6))):
            p0 = int(rng.integers(0, n)); q[p0] = ord("ACGT"[int(rng.integers(0, 4))])
        trecs, qrecs = [("T|same", t)], [("Q|same", q if rng.random() < 0.5 else gen.revcomp(q))]
    else:                # N runs and IUPAC
        t, q = gen.make_pair(n, seed0 + case, sub_rate=sub, indel_rate=indel, nruns=int(rng.integers(0, 6)))
        for _ in range(int(rng.integers(0, 20))):
            q[int(rng.integers(0, len(q)))] = ord("RYKMSWN"[int(rng.integers(0, 7))])
        trecs, qrecs = [("T|n", t)], [("Q|n", q)]
    tf, qf = gen.fasta_bytes(trecs), gen.fasta_bytes(qrecs)
    args = ["--ambiguous=iupac,100,100", "--step=%d" % int(rng.integers(1, 6)), "--ydrop=%d" % int(rng.choice([600, 1500, 3000, 4000, 9400, 20000])),
            "--hspthresh=%d" % int(rng.choice([1500, 2200, 3000, 5000])), "--xdrop=%d" % int(rng.choice([200, 910, 2000]))]
    if rng.random() < 0.3: args.append("--notransition")
    if rng.random() < 0.3: args.append("--gappedthresh=%d" % int(rng.choice([2000, 3000, 8000])))
    if rng.random() < 0.3: args.append("--queryhspbest=%d" % int(rng.choice([1, 3, 50])))
    if rng.random() < 0.2: args.append("--noentropy")
    if rng.random() < 0.1: args += ["--ungapped", "--format=general:name1,zstart1,end1,name2,zstart2+,end2+", "--markend", "--queryhsplimit=keep,nowarn:%d" % int(rng.integers(1, 9))]
    else: args.append("--format=paf:wfmash")
    print('case', case, 'kind', kind, 'n', n, ' '.join(args), flush=True) if os.environ.get('FUZZ_VERBOSE') else None
    pm = miblast.params_from_args(args)
    if WRITE:
        want = olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}))
        digests[str(seed0 + case)] = {"args": " ".join(args), "kind": kind, "n": n, "paf_md5": hashlib.md5(want["paf"]).hexdigest(), "counters": {k: int(want["counters"][k]) for k in KEYS},
                                      "n_hsps": len(want["hsps"]), "n_alns": len(want["alns"]), "n_ops": len(want["ops"])}
        json.dump(digests, open(WRITE, "w"), indent=1, sort_keys=True)
        continue
    T, Q = ctx.seqset_from_fasta_bytes(tf), ctx.seqset_from_fasta_bytes(qf)
    got = ctx.align(T, Q, pm)
    T.close(); Q.close()
    if READ:
        d = digests[str(seed0 + case)]
        assert d["args"] == " ".join(args) and d["n"] == n, "the digest file was made for other cases"
        ok = hashlib.md5(got.paf).hexdigest() == d["paf_md5"] and all(got.stats[k] == d["counters"][k] for k in KEYS) and (len(got.hsps), len(got.alns), len(got.ops)) == (d["n_hsps"], d["n_alns"], d["n_ops"])
        if not ok:
            bad += 1
            print("MISMATCH case", case, "seed", seed0 + case, "kind", kind, "n", n, "args", " ".join(args), {k: (d["counters"][k], got.stats[k]) for k in KEYS if got.stats[k] != d["counters"][k]})
        continue
    want = olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}))
    ok = got.paf == want["paf"] and got.hsps == want["hsps"] and got.alns == want["alns"] and got.ops == want["ops"] and all(got.stats[k] == want["counters"][k] for k in KEYS)
    if not ok:
        bad += 1
        print("MISMATCH case", case, "kind", kind, "n", n, "args", " ".join(args), {k: (want["counters"][k], got.stats[k]) for k in KEYS if got.stats[k] != want["counters"][k]},
              "paf_same", got.paf == want["paf"], "hsps_same", got.hsps == want["hsps"])
print("fuzz: %d cases, %d mismatches, %.1f s, reruns of wide rows exercised: n/a" % (n_cases, bad, time.time() - t_start))
sys.exit(1 if bad else 0)

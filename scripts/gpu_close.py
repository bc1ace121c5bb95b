import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cactus_amd import gen, miblast
from oracle import olz
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
ctx = miblast.Context(0)
t, q = gen.make_pair(n, 77, sub_rate=0.02, indel_rate=0.002)
tf, qf = gen.fasta_bytes([("id=simT|chr1", t)]), gen.fasta_bytes([("id=simQ|chr1", q)])
T = ctx.seqset_from_fasta_bytes(tf); Q = ctx.seqset_from_fasta_bytes(qf)
pm = miblast.params_from_args("--step=2 --ambiguous=iupac,100,100 --ydrop=3000 --notransition --queryhspbest=100000".split())
r = ctx.align(T, Q, pm, details=False)
t0 = time.time(); r = ctx.align(T, Q, pm, details=False); dt = time.time() - t0
print("gpu wall", round(dt, 3), json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.stats.items()}))
if n <= 1000000:
    t0 = time.time(); o = olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}), details=False); do = time.time() - t0
    print("oracle wall", round(do, 3), "same paf:", o["paf"] == r.paf, {k: o["counters"][k] for k in ("seed_hits", "hsps", "dp_cells", "alignments")})

import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cactus_amd import gen, miblast
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000000
ctx = miblast.Context(0)
t, q = gen.make_pair(n, 42, homologous=False)
T = ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("id=simT|chr1", t)])); Q = ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("id=simQ|chr1", q)]))
pm = miblast.params_from_args("--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=100000".split())
r = ctx.align(T, Q, pm, details=False)
t0 = time.time(); r = ctx.align(T, Q, pm, details=False); dt = time.time() - t0
print("wall", dt, json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.stats.items()}))

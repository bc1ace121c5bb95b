import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cactus_amd import gen, miblast
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ctx = miblast.Context(0)
pm = miblast.params_from_args("--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=100000".split())
sets = []
for k in range(16):
    t, q = gen.make_pair(n, 42 + k)
    sets.append((ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("id=simT%d|chr1" % k, t)])), ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("id=simQ%d|chr1" % k, q)]))))
ctx.align_pairs(sets[:2], pm)
for P in (1, 2, 4, 8, 16):
    t0 = time.time(); res = ctx.align_pairs(sets[:P], pm); dt = time.time() - t0
    cells = sum(r.stats["dp_cells"] for r in res)
    print(f"pairs per call={P}: wall={dt*1e3:.0f} ms  committed Gcell/s={cells/dt/1e9:.2f}  dp kernel {res[0].stats['t_dp_kernel_ms']:.0f} ms  cells_run={res[0].stats['dp_cells_run']/1e9:.2f}G")

import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cactus_amd import gen, miblast
ctx = miblast.Context(0)
pm = miblast.params_from_args("--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=100000".split())
for nt, nq in ((8_000_000, 8_000_000), (2_000_000, 32_000_000), (32_000_000, 2_000_000), (500_000, 128_000_000)):
    rng = np.random.default_rng(5)
    t = gen.random_sequence(nt, rng); q = gen.random_sequence(nq, rng)
    T = ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("t", t)])); Q = ctx.seqset_from_fasta_bytes(gen.fasta_bytes([("q", q)]))
    r = ctx.align(T, Q, pm, details=False)
    t0 = time.time(); r = ctx.align(T, Q, pm, details=False); dt = time.time() - t0
    s = r.stats
    print(f"T={nt} Q={nq} hits={s['seed_hits']} wall={dt*1e3:.1f} ms ungapped={s['t_ungapped_kernel_ms']:.1f} sort={s['t_sort_ms']:.1f} fill={s['t_seedfill_ms']:.1f} batches={s['seed_batches']}")
    T.close(); Q.close()

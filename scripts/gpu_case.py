import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cactus_amd import gen, miblast
from oracle import olz
case = int(sys.argv[1]); seed0 = 1000
rng = np.random.default_rng(seed0 + case)
n = int(rng.integers(2000, 60000)); sub = float(rng.choice([0.0, 0.01, 0.03, 0.08, 0.15, 0.25])); indel = float(rng.choice([0.0, 0.001, 0.005, 0.02])); kind = int(rng.integers(0, 5))
assert kind == 0
t, q = gen.make_pair(n, seed0 + case, sub_rate=sub, indel_rate=indel, mask_frac=float(rng.choice([0, 0.2, 0.6])))
tf, qf = gen.fasta_bytes([("T|c0", t)]), gen.fasta_bytes([("Q|c0", q)])
args = sys.argv[2:]
pm = miblast.params_from_args(args)
want = olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}))
print("oracle:", want["counters"], flush=True)
ctx = miblast.Context(0)
T, Q = ctx.seqset_from_fasta_bytes(tf), ctx.seqset_from_fasta_bytes(qf)
got = ctx.align(T, Q, pm)
print("same", got.paf == want["paf"])

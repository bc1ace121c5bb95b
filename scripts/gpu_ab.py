"""Torch-free timing of the blast path for A/B runs of env knobs (one process per setting: several knobs are read once).
usage: python scripts/gpu_ab.py [pairs] [steps] [size]   -> one line: ms/step, DP kernel ms/launch, launches/step, Gcell/s"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cactus_amd import gen, miblast

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
pm = miblast.params_from_args("--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=100000".split())
ctx = miblast.Context(0)
sets = []
for k in range(P):
    t, q = gen.make_pair(N, 42 + k)
    sets.append((ctx.seqset_from_fasta_bytes(gen.fasta_bytes([(f"id=simT{k}|chr1", t)])), ctx.seqset_from_fasta_bytes(gen.fasta_bytes([(f"id=simQ{k}|chr1", q)]))))


def step():
    if P == 1:
        return [ctx.align(sets[0][0], sets[0][1], pm, details=False)]
    return ctx.align_pairs(sets, pm)


for _ in range(3):
    step()
best, tot, cells, kms, launches = 1e9, 0.0, 0, 0.0, 0
for _ in range(K):
    t0 = time.perf_counter()
    rs = step()
    dt = time.perf_counter() - t0
    best = min(best, dt); tot += dt
    cells = sum(r.stats["dp_cells"] for r in rs)
    kms += rs[0].stats["t_dp_kernel_ms"]; launches += rs[0].stats["dp_kernel_launches"]
knobs = {k: v for k, v in os.environ.items() if k.startswith("MIBLAST_")}
print(f"pairs={P} size={N} knobs={knobs}: mean {1e3 * tot / K:.2f} ms/step, best {1e3 * best:.2f}, DP kernel {kms / max(1, launches):.3f} ms/launch x {launches / K:.1f}/step, "
      f"{cells / (tot / K) / 1e9:.1f} Gcell/s", flush=True)

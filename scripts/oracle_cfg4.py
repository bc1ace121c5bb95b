"""Golden digest of a full-size chunk pair (30 Mb x 30 Mb, 1.3 % divergence, half soft-masked, parameter set "one"):
the CPU oracle on the pair of scripts/gpu_cfg4.py (about a minute and 1.7 GB at 30 Mb).  Writes tests/golden/cfg4_30mb.json when run
with 30000000; tests/test_parity_gpu.py::test_full_size_chunk_pair_equals_the_oracle_digest compares the GPU result with it.
usage: python scripts/oracle_cfg4.py 30000000"""
import sys, time, hashlib, resource
import os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cactus_amd import gen
from oracle import olz
n=int(sys.argv[1])
t,q=gen.make_pair(n,3001,sub_rate=0.013,indel_rate=0.002,mask_frac=0.5)
tf,qf=gen.fasta_bytes([("id=simT|chr20",t)]),gen.fasta_bytes([("id=simQ|chr20",q)])
po=olz.default_params(step=2, transitions=0, ydrop=3000, queryhspbest=100000)
t0=time.time()
o=olz.align(tf,qf,po,details=False)
print(n, 'oracle', round(time.time()-t0,1),'s', hashlib.md5(o['paf']).hexdigest(), len(o['paf']), o['counters']['alignments'], o['counters']['dp_cells'], 'maxrss MB', resource.getrusage(resource.RUSAGE_SELF).ru_maxrss//1024, flush=True)
if n == 30_000_000:
    json.dump({"recipe": "gen.make_pair(30000000, 3001, sub_rate=0.013, indel_rate=0.002, mask_frac=0.5); --step=2 --ambiguous=iupac,100,100 --ydrop=3000 --notransition --queryhspbest=100000",
               "paf_md5": hashlib.md5(o["paf"]).hexdigest(), "paf_bytes": len(o["paf"]), "alignments": o["counters"]["alignments"],
               "dp_cells": o["counters"]["dp_cells"], "seed_hits": o["counters"]["seed_hits"], "hsps": o["counters"]["hsps"]},
              open(os.path.join(ROOT, "tests", "golden", "cfg4_30mb.json"), "w"), indent=1)

"""One full-size Cactus chunk pair of SURVEY 8d config 4 (chr20-like: 30 Mb x 30 Mb, 1.3 % divergence, half soft-masked, parameter
set "one"): timing, determinism (two runs, same bytes) and structural validity of every record (cigars walk exactly their
intervals: mipaf's check).  usage: gpu_cfg4.py [bases]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cactus_amd import gen, miblast, mipaf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
args = "--step=2 --ambiguous=iupac,100,100 --ydrop=3000 --notransition --queryhspbest=100000".split()      # <lastzArguments one=...>, xml:131
t0 = time.time()
t, q = gen.make_pair(n, 3001, sub_rate=0.013, indel_rate=0.002, mask_frac=0.5)
tf, qf = gen.fasta_bytes([("id=simT|chr20", t)]), gen.fasta_bytes([("id=simQ|chr20", q)])
print(f"generated {n} x {n} in {time.time() - t0:.1f} s", flush=True)
ctx = miblast.Context(0)
T, Q = ctx.seqset_from_fasta_bytes(tf), ctx.seqset_from_fasta_bytes(qf)
pm = miblast.params_from_args(args)
digests = []
for rep in range(2):
    t0 = time.time(); r = ctx.align(T, Q, pm, details=False); dt = time.time() - t0
    s = r.stats
    digests.append(hashlib.md5(r.paf).hexdigest())
    print(f"rep {rep}: {dt * 1e3:.1f} ms, {s['dp_cells'] / dt / 1e9:.2f} Gcell/s, {s['seed_hits'] / max(1e-9, s['t_seed']):.3g} seeds/s, alignments {s['alignments']}, rounds {s['gapped_rounds']}, "
          f"dp launches {s['dp_kernel_launches']}, dp kernel {s['t_dp_kernel_ms']:.1f} ms, index {s['t_index'] * 1e3:.1f} seed {s['t_seed'] * 1e3:.1f} gapped {s['t_gapped'] * 1e3:.1f} ms, "
          f"relays {s['relay_accepted']}/{s['relay_rejected']}, reruns {s['dp_reruns']}, spec {s['dp_cells_run'] / max(1, s['dp_cells']):.2f}", flush=True)
print("same bytes:", digests[0] == digests[1], "PAF bytes:", len(r.paf), "md5", digests[1], "dp_cells", s["dp_cells"])
# the CPU oracle on the same 30 Mb pair (61.5 s, 1.7 GB; run in the build container, where it is off the GPU clock):
#   olz.align(tf, qf, olz.default_params(step=2, transitions=0, ydrop=3000, queryhspbest=100000)) -> md5 below, dp_cells 2988193947
ORACLE_30MB = ("077738f6583a69f9133af03b5882bf5d", 2988193947)
if n == 30_000_000:
    print("equal to the oracle's PAF and dp_cells:", (digests[1], s["dp_cells"]) == ORACLE_30MB)
t0 = time.time()
ps = mipaf.PafSet.from_text(r.paf)
ps.tile(ctx)                                               # refuses records whose cigar does not walk its intervals
txt = ps.text()
cols = sum(int(l.split("\t")[10]) for l in txt.splitlines())
print(f"{len(ps)} records structurally valid, {cols} alignment columns, checked + tiled in {time.time() - t0:.2f} s")

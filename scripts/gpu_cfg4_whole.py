"""SURVEY 8d config 4 on one GPU: a synthetic chr20 x chr20 (64 444 167 bp, 1.3 % divergence, half soft-masked) chunked exactly as the
CPU path does (chunkSize 30 000 000, overlapSize 10 000: 3 x 3 chunk pairs), all nine pairs in one miblast_align_pairs call with
parameter set "one", dechunked, then the chaining stage on the result and its inverted copy.  usage: gpu_cfg4_whole.py [bases]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cactus_amd import gen, miblast, mipaf
from cactus_amd.paf import chunking

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_444_167
args = "--step=2 --ambiguous=iupac,100,100 --ydrop=3000 --notransition --queryhspbest=100000".split()
work = tempfile.mkdtemp(prefix="cfg4_")
t0 = time.time()
t, q = gen.make_pair(n, 3001, sub_rate=0.013, indel_rate=0.002, mask_frac=0.5)
for name, seq, path in (("id=simT|chr20", t, "T.fa"), ("id=simQ|chr20", q, "Q.fa")):
    with open(os.path.join(work, path), "wb") as f:
        f.write(gen.fasta_bytes([(name, seq)]))
tc = chunking.fasta_chunk(os.path.join(work, "T.fa"), os.path.join(work, "tc"), 30_000_000, 10_000)
qc = chunking.fasta_chunk(os.path.join(work, "Q.fa"), os.path.join(work, "qc"), 30_000_000, 10_000)
print(f"generated and chunked {n} bp x 2 into {len(tc)} x {len(qc)} chunk pairs in {time.time() - t0:.1f} s", flush=True)
ctx = miblast.Context(0)
pm = miblast.params_from_args(args)
t0 = time.time()
T = [ctx.seqset_from_fasta_file(p) for p in tc]
Q = [ctx.seqset_from_fasta_file(p) for p in qc]
print(f"chunks parsed and resident in HBM in {time.time() - t0:.1f} s", flush=True)
pairs = [(a, b) for a in T for b in Q]
for rep in range(2):
    t0 = time.time()
    rs = ctx.align_pairs(pairs, pm)
    dt = time.time() - t0
    cells = sum(r.stats["dp_cells"] for r in rs); hits = sum(r.stats["seed_hits"] for r in rs)
    print(f"rep {rep}: blast phase of the {len(pairs)} chunk pairs {dt * 1e3:.0f} ms, {cells / dt / 1e9:.1f} Gcell/s, {hits / dt:.3g} seeds/s, "
          f"{sum(r.stats['alignments'] for r in rs)} alignments, {sum(len(r.paf) for r in rs) / 1e6:.1f} MB of PAF", flush=True)
merged = os.path.join(work, "all.paf")
t0 = time.time()
for k, r in enumerate(rs):
    p = os.path.join(work, f"{k}.paf")
    open(p, "wb").write(r.paf)
    chunking.paf_dechunk(p, merged, append=True)
text = open(merged).read()
print(f"dechunked {len(text.splitlines())} records in {time.time() - t0:.2f} s", flush=True)
t0 = time.time()
s = mipaf.PafSet.from_text(text + mipaf.PafSet.from_text(text).invert().text())
s.chain_tile_trim_filter(ctx, None, "0.2", 10000)
out = s.text()
print(f"chaining stage: {len(out.splitlines())} primary records out in {(time.time() - t0) * 1e3:.0f} ms (kernels: {s.stats})")

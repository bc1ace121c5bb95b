import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cactus_amd import gen, miblast
from cactus_amd.multigpu import align_pairs_concurrent
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pm = miblast.params_from_args("--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=100000".split())
pairs = []
for k in range(npairs):
    t, q = gen.make_pair(n, 42 + k)
    pairs.append((gen.fasta_bytes([("id=simT%d|chr1" % k, t)]), gen.fasta_bytes([("id=simQ%d|chr1" % k, q)])))
for workers in (1, 2, 4, 8, 16):
    align_pairs_concurrent(pairs[:workers], pm, 0, workers)     # warm the per-thread contexts' first allocation
    t0 = time.time(); res = align_pairs_concurrent(pairs, pm, 0, workers); dt = time.time() - t0
    cells = sum(s["dp_cells"] for _, s in res)
    print(f"workers={workers} pairs={npairs} wall={dt:.3f}s  Gcell/s={cells / dt / 1e9:.3f}  (per-pair {dt / npairs * 1e3:.0f} ms)")
seq = align_pairs_concurrent(pairs, pm, 0, 1)
par = align_pairs_concurrent(pairs, pm, 0, 8)
print("identical output:", all(a[0] == b[0] for a, b in zip(seq, par)))
for workers in (1, 16):
    res = align_pairs_concurrent(pairs, pm, 0, workers)
    import statistics
    print("workers", workers, {k: round(statistics.mean(s[k] for _, s in res), 3) for k in ("t_total", "t_gapped", "t_seed", "t_dp_kernel_ms", "t_index")})

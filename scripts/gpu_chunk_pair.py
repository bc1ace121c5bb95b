"""One chunk pair of a chunk-scale workload alone on the GPU (no other lanes: kernel durations are what the kernels cost).
usage: gpu_chunk_pair.py chr20|hm i j [reps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cactus_amd import miblast, workloads
W = workloads.by_name(sys.argv[1]); i, j = int(sys.argv[2]), int(sys.argv[3]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ctx = miblast.Context(0)
T, Q = ctx.seqset_from_fasta_bytes(W.tfa[i]), ctx.seqset_from_fasta_bytes(W.qfa[j])
pm = miblast.params_from_args(W.options.split())
for rep in range(reps):
    if os.environ.get("DROP", "1") == "1":
        miblast.drop_derived()
    t0 = time.time(); r = ctx.align(T, Q, pm, details=False); dt = time.time() - t0
    s = r.stats
    print(f"rep {rep}: {dt*1e3:.1f} ms  index {s['t_index']*1e3:.1f} seed {s['t_seed']*1e3:.1f} gapped {s['t_gapped']*1e3:.1f} | kernels: fill {s['t_seedfill_ms']:.2f} sort {s['t_sort_ms']:.2f} ungapped {s['t_ungapped_kernel_ms']:.2f} dp {s['t_dp_kernel_ms']:.2f} | hits {s['seed_hits']} lookups {s['seed_lookups']} batches {s['seed_batches']} cols {s['ungapped_cols']} alns {s['alignments']}", flush=True)

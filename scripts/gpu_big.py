"""A large homologous chunk pair (config-2 recipe scaled up): robustness of the relay machinery, arena sizing, timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cactus_amd import gen, miblast, pafcheck
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
args = "--step=1 --ambiguous=iupac,100,100 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=100000".split()
t, q = gen.make_pair(n, 42)
tf, qf = gen.fasta_bytes([("id=simT|chr1", t)]), gen.fasta_bytes([("id=simQ|chr1", q)])
ctx = miblast.Context(0)
T, Q = ctx.seqset_from_fasta_bytes(tf), ctx.seqset_from_fasta_bytes(qf)
pm = miblast.params_from_args(args)
for rep in range(2):
    t0 = time.time(); r = ctx.align(T, Q, pm, details=False); dt = time.time() - t0
    s = r.stats
    print(f"n={n} rep {rep}: {dt*1e3:.1f} ms, {s['dp_cells']/dt/1e9:.2f} Gcell/s, alignments {s['alignments']}, rounds {s['gapped_rounds']}, dp launches {s['dp_kernel_launches']}, "
          f"dp kernel {s['t_dp_kernel_ms']:.1f} ms, seed {s['t_seed']*1e3:.1f} ms, gapped {s['t_gapped']*1e3:.1f} ms, relays {s['relay_accepted']}/{s['relay_rejected']}, reruns {s['dp_reruns']}, "
          f"tb {s['t_traceback_ms']:.1f} merge {s['t_merge_ms']:.1f}, spec {s['dp_cells_run']/max(1,s['dp_cells']):.2f}", flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "check":      # slow (python): keep it off the GPU clock unless asked for
    nrec = pafcheck.check_paf(r.paf.decode(), pafcheck.read_fasta(tf), pafcheck.read_fasta(qf))
    print("records validated:", nrec)

"""Relay hand-over stress: tiny relay spacing / warm-up on random pairs, compared with the oracle (PAF + counters)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cactus_amd import gen, miblast
from oracle import olz

cfgs = [(64, 256, 64), (128, 512, 128), (32, 128, 64), (4096, 4096, 1024), (512, 1024, 256)]
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = miblast.Context(0)
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(5000 + case)
    n = int(rng.integers(3000, 80000))
    sub = float(rng.choice([0.0, 0.02, 0.08, 0.15, 0.25])); indel = float(rng.choice([0.0, 0.002, 0.01, 0.03]))
    t, q = gen.make_pair(n, 5000 + case, sub_rate=sub, indel_rate=indel, mask_frac=float(rng.choice([0, 0.2])))
    tf, qf = gen.fasta_bytes([("T|c0", t)]), gen.fasta_bytes([("Q|c0", q)])
    args = [["--ydrop=4000", "--hspthresh=2200", "--gappedthresh=2400"], [], ["--ydrop=3000", "--hspthresh=2200"]][case % 3]
    pm = miblast.params_from_args(args)
    want = olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}))
    T, Q = ctx.seqset_from_fasta_bytes(tf), ctx.seqset_from_fasta_bytes(qf)
    for (s0, s, w) in cfgs:
        os.environ["MIBLAST_RELAY_S0"] = str(s0); os.environ["MIBLAST_RELAY_S"] = str(s); os.environ["MIBLAST_RELAY_W"] = str(w)
        try:
            got = ctx.align(T, Q, pm)
        except Exception as e:
            print(f"case {case} n={n} sub={sub} indel={indel} args={args} cfg={(s0, s, w)}: ERROR {e}", flush=True); bad += 1; continue
        ok = got.paf == want["paf"]
        cs = {k: (got.stats[k], want["counters"][k]) for k in ("dp_cells", "dp_rows", "dp_sides", "alignments") if got.stats[k] != want["counters"][k]}
        if not ok or cs:
            bad += 1
            print(f"case {case} n={n} sub={sub} indel={indel} args={args} cfg={(s0, s, w)}: paf_equal={ok} counters={cs}", flush=True)
            ga, wa = got.alns, want["alns"]
            for x, (a, b) in enumerate(zip(ga, wa)):
                if a != b: print("   aln", x, "got", a, "want", b); break
print("cases", n_cases, "bad", bad)

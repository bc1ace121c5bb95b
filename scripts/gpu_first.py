"""first GPU bring-up: product vs oracle on a few synthetic pairs"""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cactus_amd import gen, miblast
from oracle import olz

ctx = miblast.Context(0)
for (n, seed, homo, args) in [(20000, 1, True, "--step=1 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400"),
                              (100000, 42, True, "--step=1 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400 --queryhspbest=100000"),
                              (100000, 43, False, "--step=1 --ydrop=4000 --hspthresh=2200 --gappedthresh=2400"),
                              (200000, 7, True, "--step=2 --ydrop=3000 --notransition --queryhspbest=100000")]:
    t, q = gen.make_pair(n, seed, homologous=homo) if args.find('notransition') < 0 else gen.make_pair(n, seed, sub_rate=0.03, indel_rate=0.002)
    tf = gen.fasta_bytes([("id=simT|chr1", t)]); qf = gen.fasta_bytes([("id=simQ|chr1", q)])
    pm = miblast.params_from_args(args.split())
    po = olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_})
    t0 = time.time(); o = olz.align(tf, qf, po); t_or = time.time() - t0
    T = ctx.seqset_from_fasta_bytes(tf); Q = ctx.seqset_from_fasta_bytes(qf)
    t0 = time.time(); r = ctx.align(T, Q, pm); t_gpu = time.time() - t0
    t0 = time.time(); r = ctx.align(T, Q, pm); t_gpu2 = time.time() - t0
    same = r.paf == o['paf']
    nl = o['paf'].count(b'\n')
    keys = ["seed_lookups", "seed_hits", "hits_extended", "ungapped_cols", "hsps_pre_entropy", "hsps", "anchors", "anchors_skipped", "dp_sides", "dp_cells", "dp_rows", "alignments"]
    diff = {k: (o['counters'][k], r.stats[k]) for k in keys if o['counters'][k] != r.stats[k]}
    print(f"n={n} seed={seed} paf_same={same} hsps_same={sorted(r.hsps)==sorted(o['hsps'])} alns_same={r.alns==o['alns']} oracle={t_or:.2f}s gpu={t_gpu:.2f}s/{t_gpu2:.2f}s lines={nl} counter_diffs={diff}")
    print("   stats", json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.stats.items()}))
    if not same:
        ol = o['paf'].split(b'\n'); gl = r.paf.split(b'\n')
        print("   oracle lines", len(ol), "gpu lines", len(gl))
        for a, b in zip(ol, gl):
            if a != b:
                print("   O:", a[:200]); print("   G:", b[:200]); break

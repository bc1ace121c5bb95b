"""Oracle digests of EVERY chunk pair of a chunk-scale workload (cactus_amd/workloads.py: chr20 = BASELINE configs[3], hm = the
configs[4] stand-in): the CPU oracle on each (target chunk, query chunk) pair with the workload's option set, PAF md5 + counters
per pair -> tests/golden/<key>_pairs.json.  bench.py compares every pair's PAF with these on every run (`same_bytes`), the GPU suite
too.  The oracle needs about a minute and 1.7 GB per 30 Mb x 30 Mb pair: run in the build container, off the GPU clock.
usage: python scripts/oracle_chunk_digests.py chr20|hm [workers] [first_pair last_pair]"""
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cactus_amd import workloads  # noqa: E402

W = None


def one(k):
    from cactus_amd import miblast
    from oracle import olz
    i, j = W.pairs[k]
    pm = miblast.params_from_args(W.options.split())
    t0 = time.time()
    o = olz.align(W.tfa[i], W.qfa[j], olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}), details=False)
    c = o["counters"]
    return k, {"pair": [i, j], "paf_md5": hashlib.md5(o["paf"]).hexdigest(), "paf_bytes": len(o["paf"]), "oracle_seconds": round(time.time() - t0, 2),
               **{n: int(c[n]) for n in ("seed_lookups", "seed_hits", "hits_extended", "ungapped_cols", "hsps", "anchors", "dp_cells", "dp_rows", "alignments")}}


def main():
    global W
    name = sys.argv[1]
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    W = workloads.by_name(name)
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(W.pairs))
    out = {"workload": W.describe, "options": W.options, "fasta_md5": hashlib.md5(b"".join(W.tfa + W.qfa)).hexdigest(), "pairs": [None] * len(W.pairs)}
    path = os.path.join(ROOT, "tests", "golden", f"{W.key}_pairs.json")
    if os.path.exists(path):
        old = json.load(open(path))
        if old.get("fasta_md5") == out["fasta_md5"]:
            out["pairs"] = old["pairs"]
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=workers) as ex:       # (fork: the workers inherit W)
        for k, rec in ex.map(one, range(lo, hi)):
            out["pairs"][k] = rec
            print(f"pair {k} {rec}", flush=True)
            json.dump(out, open(path, "w"), indent=1)
    print(f"{name}: {hi - lo} pairs in {time.time() - t0:.0f} s -> {path}")


if __name__ == "__main__":
    main()

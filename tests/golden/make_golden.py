"""Regenerates tests/golden/golden.json and the two tiny readable fixtures from the CPU oracle.
The reference holds no PAF/cigar-level vectors for this path and its lastz submodule is
absent, so these are REGRESSION pins of the oracle (themselves cross-checked against the independent
pure-Python restatement in tests/pyref.py), not reference outputs.  Run: python tests/golden/make_golden.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from cactus_amd import gen, miblast  # noqa: E402
from oracle import olz  # noqa: E402
from cases import CASES  # noqa: E402

KEYS = ["seed_lookups", "seed_hits", "hits_extended", "ungapped_cols", "hsps_pre_entropy", "hsps", "anchors", "anchors_skipped",
        "dp_sides", "dp_cells", "dp_rows", "alignments"]
gold = {}
for name, tf, qf, args in CASES:
    pm = miblast.params_from_args(args)
    r = olz.align(tf, qf, olz.default_params(**{f: getattr(pm, f) for f, _ in pm._fields_}), details=False)
    gold[name] = {"paf_sha256": hashlib.sha256(r["paf"]).hexdigest(), "lines": r["paf"].count(b"\n"),
                  "counters": {k: r["counters"][k] for k in KEYS}, "args": " ".join(args)}
json.dump(gold, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)

rng = np.random.default_rng(2024)
t = gen.random_sequence(240, rng)
q1 = gen.mutate(t, rng, 0.04, 0.0)
q2 = gen.revcomp(np.concatenate([t[:120], t[126:]]))
for stem, q in (("tiny_plus", q1), ("tiny_minus_gap", q2)):
    tf, qf = gen.fasta_bytes([("id=T|tiny", t)]), gen.fasta_bytes([("id=Q|tiny", q)])
    open(os.path.join(HERE, stem + ".target.fa"), "wb").write(tf)
    open(os.path.join(HERE, stem + ".query.fa"), "wb").write(qf)
    r = olz.align(tf, qf, olz.default_params(hspthresh=2200, gappedthresh=2400, ydrop=4000), details=False)
    open(os.path.join(HERE, stem + ".paf"), "wb").write(r["paf"])
    print(stem, r["paf"].decode().strip())
print("wrote", len(gold), "golden entries")

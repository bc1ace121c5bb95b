"""Regenerates tests/golden/chain_*.paf: a small chaining-stage input and what every step of chain_tile_trim_filter_one_contig
(/root/reference/src/cactus/paf/local_alignment.py:660-727) makes of it according to the CPU oracle (oracle/oracle_paffy).  paffy is
an absent submodule of the reference, so these are REGRESSION pins of the oracle's rules, not reference outputs.
Run: python tests/golden/make_chain_golden.py"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from cactus_amd import gen  # noqa: E402
from tests import pyref_paffy as ref  # noqa: E402

ORACLE = os.path.join(ROOT, "oracle", "oracle_paffy")
CHAIN = ["--maxGapLength", "1000000", "--chainGapOpen", "5000", "--chainGapExtend", "1", "--trimFraction", "1.0"]


def oracle(cmd, text, *args):
    return subprocess.run([ORACLE, cmd, *args], input=text.encode(), capture_output=True, check=True).stdout.decode()


text = gen.random_paf(2024, n_series=5, per_series=(2, 8), n_q=2, n_t=2, contig_len=150_000, noise=12)
text += ref.dump(ref.invert(ref.parse(text)))                              # chain_alignments feeds both orientations
steps = {"input": text}
steps["chain"] = oracle("chain", steps["input"], *CHAIN)
steps["tile"] = oracle("tile", steps["chain"])
steps["trim"] = oracle("trim", steps["tile"], "--trimIdentity", "0.2")
steps["primary"] = oracle("filter", steps["trim"], "--maxTileLevel", "1")
steps["rechain"] = oracle("chain", steps["primary"], *CHAIN)
steps["output"] = oracle("filter", steps["rechain"], "--minChainScore", "10000")
for name, body in steps.items():
    with open(os.path.join(HERE, f"chain_{name}.paf"), "w") as f:
        f.write(body)
    print(name, len(body.splitlines()), "records")

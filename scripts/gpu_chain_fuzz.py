"""Differential fuzz of the chaining stage: random synthetic PAF sets x random chain / trim parameters, GPU (through the C ABI)
against oracle/oracle_paffy, byte for byte, every sub-command and the whole job.  usage: gpu_chain_fuzz.py [cases] [first_seed]"""
import os
import random
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cactus_amd import gen, miblast, mipaf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "oracle_paffy")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def oracle(cmd, text, *args):
    p = subprocess.run([ORACLE, cmd, *args], input=text.encode(), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode()


ctx = miblast.Context(0)
bad = 0
for case in range(first, first + n_cases):
    rng = random.Random(9000 + case)
    kw = dict(n_series=rng.choice([1, 3, 8, 20, 60]), per_series=rng.choice([(1, 4), (1, 12), (10, 40)]), n_q=rng.choice([1, 2, 5]), n_t=rng.choice([1, 2, 4]),
              contig_len=rng.choice([30_000, 200_000, 3_000_000]), noise=rng.choice([0, 5, 40, 300]), ragged=rng.random() < 0.7)
    text = gen.random_paf(case, **kw)
    if rng.random() < 0.8:
        text += mipaf.PafSet.from_text(text).invert().text()
    cp = dict(max_gap_length=rng.choice([0, 500, 3000, 50_000, 1_000_000]), gap_open=rng.choice([0, 100, 5000]), gap_extend=rng.choice([0, 1, 3]),
              trim_fraction=rng.choice([0.0, 0.25, 0.5, 1.0]))
    cargs = ["--maxGapLength", str(cp["max_gap_length"]), "--chainGapOpen", str(cp["gap_open"]), "--chainGapExtend", str(cp["gap_extend"]),
             "--trimFraction", str(cp["trim_fraction"])]
    x = rng.choice(["0", "0.05", "0.2", "0.5", "0.9", "0.999", "1"])
    P = mipaf.default_chain_params(**cp)
    try:
        chained = oracle("chain", text, *cargs)
        ok = [mipaf.PafSet.from_text(text).chain(ctx, P).text() == chained]
        tiled = oracle("tile", chained)
        ok.append(mipaf.PafSet.from_text(chained).tile(ctx).text() == tiled)
        ok.append(mipaf.PafSet.from_text(chained).tile(ctx, hist_bins=rng.choice([2, 3, 64, 4096])).text() == tiled)
        trimmed = oracle("trim", tiled, "--trimIdentity", x)
        ok.append(mipaf.PafSet.from_text(tiled).trim(ctx, x).text() == trimmed)
        prim = oracle("filter", trimmed, "--maxTileLevel", "1")
        want = oracle("filter", oracle("chain", prim, *cargs), "--minChainScore", "10000")
        ok.append(mipaf.PafSet.from_text(text).chain_tile_trim_filter(ctx, P, x, 10000).text() == want)
    except Exception as e:                                   # noqa: BLE001
        ok = [False]
        print(f"case {case}: ERROR {e}", flush=True)
    if not all(ok):
        bad += 1
        print(f"case {case} kw={kw} chain={cp} trim={x}: steps ok = {ok}", flush=True)
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)

"""Chaining stage at scale: synthetic PAF (syntenic series + noise, both orientations) through chain / tile / trim on the GPU,
kernel times from mipaf_stats, against the oracle's wall time on the same text.  usage: gpu_chain_bench.py [n_series] [contig_len] [check]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cactus_amd import miblast, mipaf
from tests import pyref_paffy as ref

n_series = int(sys.argv[1]) if len(sys.argv) > 1 else 400
contig_len = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
check = len(sys.argv) > 3
ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "oracle_paffy")
CHAIN_ARGS = ["--maxGapLength", "1000000", "--chainGapOpen", "5000", "--chainGapExtend", "1", "--trimFraction", "1.0"]

t0 = time.time()
text = ref.random_paf(1234, n_series=n_series, per_series=(20, 60), n_q=2, n_t=2, contig_len=contig_len, noise=n_series * 10, ragged=False)
text += mipaf.PafSet.from_text(text).invert().text()
print(f"{len(text.splitlines())} records, {len(text) / 1e6:.1f} MB of PAF, generated in {time.time() - t0:.1f} s", flush=True)
ctx = miblast.Context(0)
mipaf.PafSet.from_text(text[:200000].rsplit("\n", 1)[0] + "\n").chain(ctx)        # warm-up (module load)


def timed(label, fn):
    t0 = time.perf_counter()
    out = fn()
    dt = time.perf_counter() - t0
    print(f"  {label}: {dt * 1e3:.1f} ms", flush=True)
    return out


for rep in range(2):
    print(f"-- run {rep}")
    s = timed("parse", lambda: mipaf.PafSet.from_text(text))
    timed("chain", lambda: s.chain(ctx)); st = s.stats
    print(f"     groups {st['groups']}, sort {st['t_sort_ms']:.2f} ms, k_chain_dp {st['t_chain_dp_ms']:.2f} ms, pairs<= {st['chain_pairs']:.3g}")
    chained = timed("text", lambda: s.text())
    timed("tile", lambda: s.tile(ctx)); st = s.stats
    print(f"     query seqs {st['query_sequences']}, ops {st['ops']}, sort {st['t_sort_ms']:.2f} ms, k_tile {st['t_tile_ms']:.2f} ms")
    tiled = s.text()
    timed("trim", lambda: s.trim(ctx, "0.2")); st = s.stats
    print(f"     k_trim {st['t_trim_ms']:.2f} ms")
    trimmed = s.text()
    s2 = mipaf.PafSet.from_text(text)
    timed("whole job (chain|tile|trim|filter|chain|filter)", lambda: s2.chain_tile_trim_filter(ctx, None, "0.2", 10000))
    print(f"     -> {len(s2)} records; sort {s2.stats['t_sort_ms']:.2f}, dp {s2.stats['t_chain_dp_ms']:.2f}, tile {s2.stats['t_tile_ms']:.2f}, trim {s2.stats['t_trim_ms']:.2f} ms")

for cmd, inp, args, got in (("chain", text, CHAIN_ARGS, chained), ("tile", chained, [], tiled), ("trim", tiled, ["--trimIdentity", "0.2"], trimmed)):
    t0 = time.perf_counter()
    p = subprocess.run([ORACLE, cmd, *args], input=inp.encode(), capture_output=True)
    dt = time.perf_counter() - t0
    print(f"oracle {cmd}: {dt * 1e3:.1f} ms (1 core, incl. parse + print){'  equal=' + str(p.stdout.decode() == got) if check else ''}", flush=True)

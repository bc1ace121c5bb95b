#!/usr/bin/env python3
"""cProfile of the Python side of the bench's phase step (what runs between the library calls)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                        # noqa: E402
from cactus_amd import miblast
sys.argv = [sys.argv[0]]
a = bench.parse_args()
ctx = miblast.Context(0)
os.environ["MIBLAST_BENCH_CONTEXTS"] = "1"          # one thread of calls: the profile then is the critical path
work = bench.EvolverPhase(a, ctx, 0)
for _ in range(4):
    work.step()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    work.step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)

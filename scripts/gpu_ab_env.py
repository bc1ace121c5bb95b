#!/usr/bin/env python3
"""In-process A/B of one environment switch on the bench workloads: steps alternate between the two settings (the library reads its
MIBLAST_* switches per call), so box-to-box and minute-to-minute noise hits both arms alike.  Prints median / min / mean per arm.
usage: gpu_ab_env.py NAME VALUE_A VALUE_B [steps]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                        # noqa: E402


def main():
    name, va, vb = sys.argv[1:4]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    import torch
    from cactus_amd import miblast
    sys.argv = [sys.argv[0]]
    a = bench.parse_args()
    ctx = miblast.Context(0)
    import copy
    a16 = copy.copy(a); a16.pairs_per_gpu = 16
    for label, work in (("evolver", bench.EvolverPhase(a, ctx, 0)), ("pair", bench.PairWorkload(a, ctx, 0)), ("16 pairs", bench.PairWorkload(a16, ctx, 0))):
        t = {va: [], vb: []}
        for i in range(6 + 2 * steps):
            v = (va, vb)[i & 1]
            if name == "PY_SWITCHINTERVAL":
                sys.setswitchinterval(float(v))
            else:
                os.environ[name] = v
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            work.step()
            torch.cuda.synchronize()
            if i >= 6:
                t[v].append((time.perf_counter() - t0) * 1e3)
        for v in (va, vb):
            print(f"{label:8s} {name}={v}: median {statistics.median(t[v]):.2f}  min {min(t[v]):.2f}  mean {statistics.fmean(t[v]):.2f} ms  (n={len(t[v])})")


main()

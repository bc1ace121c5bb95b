#!/bin/bash
# session 2: group-per-run ungapped kernel -- parity suite, then A/B against the lane-per-run kernel on the seed leg and the phase
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2b
( time timeout 600 python -m pytest tests -m gpu -x -q ) > gpurun_out/s2b/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s2b/pytest.log
for mode in grp lane; do
MIBLAST_UNGAPPED=$mode timeout 300 python bench.py --steps 10 --warmup 3 --pair-leg 1 --batch-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/s2b/bench_$mode.json 2> gpurun_out/s2b/bench_$mode.err; echo "bench $mode rc=$?"
python - $mode <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/s2b/bench_{sys.argv[1]}.json"))
print(sys.argv[1], "evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "stage", {k:round(v,2) for k,v in d["stage_kernel_ms_per_step"].items()})
p=d["pair_1mb"]; print("  pair ms", round(p["ms_per_step"],2), {k:round(v,2) for k,v in p["stage_kernel_ms_per_step"].items()})
s=d["seed_stage"]; print("  seed", {k:round(v,2) for k,v in s["kernel_ms"].items()}, round(s["seconds"]*1e3,1), round(s["frac"],4), s["seed_hits"], s["chance_alignments"])
PY
done

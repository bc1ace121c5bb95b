#!/bin/bash
# session 2: full GPU suite (default kernel choice) + bench legs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2m
( time timeout 600 python -m pytest tests -m gpu -x -q ) > gpurun_out/s2m/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s2m/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --pair-leg 1 --batch-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/s2m/bench.json 2> gpurun_out/s2m/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/s2m/bench.json"))
print("evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "stage", {k:round(v,2) for k,v in d["stage_kernel_ms_per_step"].items()})
p=d["pair_1mb"]; print("  pair ms", round(p["ms_per_step"],2), {k:round(v,2) for k,v in p["stage_kernel_ms_per_step"].items()})
s=d["seed_stage"]; print("  seed", {k:round(v,2) for k,v in s["kernel_ms"].items()}, round(s["seconds"]*1e3,1), round(s["frac"],4), s["seed_hits"], s["chance_alignments"])
PY

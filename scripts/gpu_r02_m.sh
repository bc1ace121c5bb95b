#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02m
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q -m gpu > gpurun_out/r02m/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02m/pytest.log
timeout 600 python bench.py > gpurun_out/r02m/bench.json 2> gpurun_out/r02m/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02m/bench.json"))
print("evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "dp", round(d["stage_kernel_ms_per_step"]["ydrop"],2), "frac", round(d["roofline"]["frac"],5), "traffic", d["roofline"]["traffic"], "same_bytes", d["cpu_baseline"]["same_bytes"])
print("pair", d["pair_1mb"]["ms_per_step"], d["pair_1mb"]["roofline"])
print("batched", d["batched_pairs"])
PY

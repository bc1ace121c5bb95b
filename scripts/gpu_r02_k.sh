#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02k
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r02k/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02k/pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --chain-leg 0 --seed-leg 0 > gpurun_out/r02k/bench.json 2> gpurun_out/r02k/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02k/bench.json"))
print("evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "dp", round(d["stage_kernel_ms_per_step"]["ydrop"],2), "kernel Gc/s", round(d["gapped_gcells_per_s_kernel"],1), "frac", round(d["roofline"]["frac"],5), "same_bytes", d["cpu_baseline"]["same_bytes"], "| pair", round(d["pair_1mb"]["ms_per_step"],2), d["pair_1mb"]["gapped_gcells_per_s_kernel"], d["pair_1mb"]["cpu_baseline"]["same_bytes"])
PY

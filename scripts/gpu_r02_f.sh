#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02f
timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r02f/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02f/pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --pair-leg 1 --chain-leg 0 --cpu-sample 0 > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02f/bench.json"))
print("evolver ms", round(d["ms_per_step"],2), d["stage_kernel_ms_per_step"], "| pair ms", round(d["pair_1mb"]["ms_per_step"],2), d["pair_1mb"]["stage_kernel_ms_per_step"], "| seed leg", d["seed_stage"]["kernel_ms"], round(d["seed_stage"]["seconds"]*1e3,1), d["seed_stage"]["frac"])
PY

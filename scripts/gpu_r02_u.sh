#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02u
run() { echo "== $*"; env "$@" timeout 600 python scripts/gpu_cfg4.py 2>&1 | grep "^rep 1\|equal to the oracle" | tail -2 | cut -c1-260; }
run X=1
run MIBLAST_CHAIN_HEADS=0
run MIBLAST_GROUP_GAP=16384
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r02u/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02u/pytest.log
timeout 900 python scripts/gpu_cfg4_whole.py 2>&1 | grep "^rep" | cut -c1-200
timeout 300 python bench.py --steps 5 --warmup 2 --chain-leg 0 --seed-leg 0 --cpu-sample 0 > gpurun_out/r02u/bench.json 2> gpurun_out/r02u/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02u/bench.json"))
print("evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "spec", round(d["speculation_factor"],2), "dp", round(d["stage_kernel_ms_per_step"]["ydrop"],2), "launches", d["relay"]["dp_launches_per_step"], "| pair", round(d["pair_1mb"]["ms_per_step"],2), "| batched", round(d["batched_pairs"]["ms_per_call"],1), round(d["batched_pairs"]["value"],1))
PY

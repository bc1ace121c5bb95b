#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02x
timeout 600 python scripts/gpu_cfg4.py 2>&1 | grep "^rep 1\|equal to the oracle" | tail -2 | cut -c1-260
timeout 900 python scripts/gpu_cfg4_whole.py 2>&1 | grep "^rep" | cut -c1-200
for k in 1 2; do
timeout 300 python bench.py --steps 5 --warmup 2 --chain-leg 0 --seed-leg 0 --cpu-sample 0 > gpurun_out/r02x/bench_$k.json 2> gpurun_out/r02x/bench_$k.err
python - $k <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r02x/bench_{sys.argv[1]}.json"))
print("evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "spec", round(d["speculation_factor"],2), "dp", round(d["stage_kernel_ms_per_step"]["ydrop"],2), "launches", d["relay"]["dp_launches_per_step"], "| pair", round(d["pair_1mb"]["ms_per_step"],2), round(d["pair_1mb"]["speculation_factor"],2), "| batched", round(d["batched_pairs"]["ms_per_call"],1), round(d["batched_pairs"]["value"],1), round(d["batched_pairs"]["speculation_factor"],2))
PY
done
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r02x/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02x/pytest.log

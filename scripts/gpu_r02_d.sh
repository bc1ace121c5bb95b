#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02d
timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r02d/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02d/pytest.log
for lr in default 4 8 16 32; do
  if [ $lr = default ]; then unset MIBLAST_LONG_RUN; else export MIBLAST_LONG_RUN=$lr; fi
  timeout 300 python bench.py --steps 5 --warmup 2 --pair-leg 1 --chain-leg 0 --cpu-sample 0 > gpurun_out/r02d/bench_$lr.json 2> gpurun_out/r02d/bench_$lr.err
  python - $lr <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r02d/bench_{sys.argv[1]}.json"))
print(sys.argv[1], "evolver ms", round(d["ms_per_step"],2), d["stage_kernel_ms_per_step"], "| pair ms", round(d["pair_1mb"]["ms_per_step"],2), d["pair_1mb"]["stage_kernel_ms_per_step"]["ungapped"], "| seed leg", d["seed_stage"]["kernel_ms"], round(d["seed_stage"]["seconds"]*1e3,1))
PY
done

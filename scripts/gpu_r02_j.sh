#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02j
for lanes in 4 6 9 12; do
MIBLAST_SEED_LANES=$lanes timeout 300 python bench.py --steps 5 --warmup 2 --pair-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 > gpurun_out/r02j/b_$lanes.json 2> gpurun_out/r02j/b_$lanes.err
python - $lanes <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r02j/b_{sys.argv[1]}.json"))
print("lanes",sys.argv[1],"evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "dp", round(d["stage_kernel_ms_per_step"]["ydrop"],2), "kernel Gc/s", round(d["gapped_gcells_per_s_kernel"],1), "frac", round(d["roofline"]["frac"],5))
PY
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02w
run() { echo "== $*"; env "$@" timeout 600 python scripts/gpu_cfg4.py 2>&1 | grep "^rep 1\|equal to the oracle" | tail -2 | cut -c1-260; }
run X=1
run MIBLAST_GROUP_GAP=16384
run MIBLAST_GROUP_GAP=65536
for g in 4096 16384 65536; do
MIBLAST_GROUP_GAP=$g timeout 300 python bench.py --steps 5 --warmup 2 --chain-leg 0 --seed-leg 0 --cpu-sample 0 > gpurun_out/r02w/bench_$g.json 2> gpurun_out/r02w/bench_$g.err
python - $g <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r02w/bench_{sys.argv[1]}.json"))
print("gap",sys.argv[1],"evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "spec", round(d["speculation_factor"],2), "dp", round(d["stage_kernel_ms_per_step"]["ydrop"],2), "launches", d["relay"]["dp_launches_per_step"], "| pair", round(d["pair_1mb"]["ms_per_step"],2), round(d["pair_1mb"]["speculation_factor"],2), "| batched", round(d["batched_pairs"]["ms_per_call"],1), round(d["batched_pairs"]["value"],1), round(d["batched_pairs"]["speculation_factor"],2))
PY
done

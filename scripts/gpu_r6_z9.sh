#!/bin/bash
# round 6, call Z9: the allocations that still happen in hm30's timed steps in some runs (MIBLAST_DEBUG_ALLOC=2), three runs
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6z9; mkdir -p $OUT; rm -f $OUT/*
COMMON="--cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
for R in 1 2 3; do
  MIBLAST_DEBUG_ALLOC=2 timeout 400 python bench.py --workload hm30 --steps 6 --warmup 2 $COMMON --full-out $OUT/hm30_$R.full.json > $OUT/hm30_$R.json 2> $OUT/hm30_$R.err
  python - $OUT/hm30_$R.full.json $R <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("run", sys.argv[2], "%.1f ms/step" % d["ms_per_step"], d["step_ms_spread"]["min"], d["step_ms_spread"]["max"], "allocs in timed steps:", d.get("device_allocs_in_timed_steps"))
PY
done

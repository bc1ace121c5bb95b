#!/bin/bash
# round 6, call Z6: which device allocations are left inside the timed steps of the chunk legs (MIBLAST_DEBUG_ALLOC=2 prints each one with its time)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6z6; mkdir -p $OUT; rm -f $OUT/*
COMMON="--cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
for W in hm chr20 hm30; do
  MIBLAST_DEBUG_ALLOC=2 MIBLAST_BENCH_STEP_TIMES=1 timeout 400 python bench.py --workload $W --steps 8 --warmup 3 $COMMON --full-out $OUT/$W.full.json > $OUT/$W.json 2> $OUT/$W.err
  python - $OUT/$W.full.json $W <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[2], "%.1f ms/step" % d["ms_per_step"], "allocs in timed steps:", d.get("device_allocs_in_timed_steps"))
PY
  grep -c "device allocation" $OUT/$W.err
  grep "device allocation" $OUT/$W.err | tail -24 | cut -c1-260
done

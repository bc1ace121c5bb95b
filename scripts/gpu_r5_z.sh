#!/bin/bash
# round 5: the stalled step of the hm leg inside the FULL default command: allocations slower than 20 ms and step times on stderr
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r5z; mkdir -p $OUT
for R in 1 2; do
  MIBLAST_DEBUG_ALLOC=1 MIBLAST_BENCH_STEP_TIMES=1 timeout 600 python bench.py --cpu-sample 0 > $OUT/b$R.json 2> $OUT/b$R.err
  echo "run $R"; grep "step times\|slow" $OUT/b$R.err | cut -c1-200 | tail -40
done

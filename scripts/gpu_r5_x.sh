#!/bin/bash
# round 5: the outlier step of the hm leg inside the default command (bins on): slow allocations and slow seed phases on stderr
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r5x; mkdir -p $OUT
for R in 1 2 3 4; do
  MIBLAST_DEBUG_ALLOC=1 MIBLAST_DEBUG_SPIKE=150 MIBLAST_BENCH_STEP_TIMES=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --pair-leg 0 --primates-leg 0 > $OUT/b$R.json 2> $OUT/b$R.err
  echo "run $R"; grep "step times\|slow" $OUT/b$R.err | cut -c1-330 | tail -14
done

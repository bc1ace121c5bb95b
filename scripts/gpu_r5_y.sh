#!/bin/bash
# round 5: host timeline of the headline's steps on the final build
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r5y; mkdir -p $OUT
MIBLAST_BENCH_TIMELINE=1 MIBLAST_DEBUG=1 MIBLAST_BENCH_STEP_TIMES=1 timeout 300 python bench.py --steps 3 --warmup 3 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --pair-leg 0 --primates-leg 0 --chunk-legs 0 > $OUT/b.json 2> $OUT/b.err
grep "step times" $OUT/b.err; wc -l $OUT/b.err

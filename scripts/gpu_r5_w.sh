#!/bin/bash
# round 5: the chunk-scale legs INSIDE the default bench command (a context per leg, after other legs), per-step times on stderr
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r5w; mkdir -p $OUT
for B in 1 1 0; do
  MIBLAST_SORT_BIN=$B MIBLAST_BENCH_STEP_TIMES=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --pair-leg 0 --primates-leg 0 > $OUT/b$B.json 2> $OUT/b$B.err
  python -c "import json;d=json.load(open('$OUT/b$B.json'));print('bin=$B head',round(d['ms_per_step'],2),'chr20',round(d['chr20']['ms_per_step'],2),'hm',round(d['hm']['ms_per_step'],2))"
  grep "step times" $OUT/b$B.err
done

#!/bin/bash
# round 5: STANDALONE per-kernel times of the human-mouse stand-in (one lane: no two kernels of the call share the GPU), bins + LDS against rocprim
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5t; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for B in 1 0; do
  CMD="python $ROOT/bench.py --workload hm --steps 1 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
  MIBLAST_SORT_BIN=$B MIBLAST_PIPELINE_LANES=1 MIBLAST_SEED_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hm_bin${B}_stats -- $CMD > $OUT/hm_bin$B.json 2> $OUT/hm_bin$B.log
  find $OUT/hm_bin${B}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/hm_bin${B}_kernel_stats.csv
  rm -rf $OUT/hm_bin${B}_stats
  python -c "import json;d=json.load(open('$OUT/hm_bin$B.json'));print('bin=$B',d['ms_per_step'],d['stage_kernel_ms_per_step'])"
done

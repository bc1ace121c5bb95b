#!/bin/bash
# round 5: per-kernel times of the chunk-scale workloads with the keys grouped in LDS (mb_seed_bin.h)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5s; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in ${WORKLOADS:-hm chr20}; do
  CMD="python $ROOT/bench.py --workload $W --steps 2 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_stats -- $CMD > $OUT/${W}.json 2> $OUT/${W}.log
  find $OUT/${W}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${W}_kernel_stats.csv
  rm -rf $OUT/${W}_stats
done

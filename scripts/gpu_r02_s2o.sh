#!/bin/bash
# session 2: kernel stats of the phase (3 steps)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/s2o; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --pair-leg 0 --batch-leg 0 > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv
find $OUT/stats -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete

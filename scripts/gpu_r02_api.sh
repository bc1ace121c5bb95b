#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02api; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --stats --output-format csv -d $OUT/hip -- python $ROOT/bench.py --steps 5 --warmup 2 --pair-leg 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > $OUT/bench.json 2> $OUT/log.txt
f=$(find $OUT/hip -name "*hip_api_stats.csv" | head -1); echo $f; head -25 "$f" | cut -c1-160
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*.db" -delete

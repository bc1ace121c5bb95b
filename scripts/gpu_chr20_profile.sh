#!/bin/bash
# kernel stats of the chr20 workload (configs[3], N = 1).  usage: gpurun -- 'bash scripts/gpu_chr20_profile.sh <tag>'
TAG=${1:-chr20prof}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --workload chr20 --steps 2 --warmup 1 --cpu-sample 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
head -40 $OUT/kernel_stats.csv | cut -c1-60,150-400

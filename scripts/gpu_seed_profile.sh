#!/bin/bash
# kernel stats of the seed / ungapped stage on the 8 Mb pure-random pair (the seed_stage leg's workload).  usage: gpurun -- 'bash scripts/gpu_seed_profile.sh <tag>'
TAG=${1:-seedprof}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sstats -- python $ROOT/scripts/gpu_rand.py 8000000 > $OUT/seed_under_rocprof.log 2>&1
find $OUT/sstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/seed_kernel_stats.csv
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
tail -2 $OUT/seed_under_rocprof.log | cut -c1-600
head -14 $OUT/seed_kernel_stats.csv | cut -c1-60,150-400 

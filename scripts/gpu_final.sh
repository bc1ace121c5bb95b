#!/bin/bash
# round-end artifacts: default bench line (with the chain_stage leg), rocprofv3 kernel stats of the chaining-stage scale run,
# batched-call timings
ROOT=$(pwd); OUT=$ROOT/gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
for P in 16 64; do timeout 200 python scripts/gpu_ab.py $P 4 2>&1 | tail -1 | tee -a $OUT/batched.log; done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain_stats -- python $ROOT/scripts/gpu_chain_bench.py 400 20000000 > $OUT/chain_bench_under_rocprof.log 2> $OUT/chain_stats.log )
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/chain_kernel_stats.csv
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*.db" -delete
head -25 $OUT/chain_kernel_stats.csv
tail -12 $OUT/chain_bench_under_rocprof.log

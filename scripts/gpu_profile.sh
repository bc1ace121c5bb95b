#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats of the default bench command + HBM traffic counters (separate passes)
set -x
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0"
( cd $ROOT && $CMD > $OUT/bench_plain.json 2>/dev/null )
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > /dev/null 2> $OUT/pmc_write.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python $ROOT/scripts/pmc_summary.py "$OUT/pmc_fetch/**/*counter_collection.csv" "$OUT/pmc_write/**/*counter_collection.csv" "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`python bench.py --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0\`, MI355X, round 1" > $OUT/hbm_traffic_pmc.json
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*.db" -delete
ls -la $OUT $OUT/stats/* | head -40
head -30 $OUT/kernel_stats.csv
tail -1 $OUT/bench_plain.json | cut -c1-300

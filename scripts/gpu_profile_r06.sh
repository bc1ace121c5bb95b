#!/bin/bash
# rocprofv3 evidence of round 6 for profiles/: kernel stats of the headline command and of the chunk-scale workloads, HBM traffic counters
# (separate --pmc passes, never together with a trace domain other than the kernel trace), the saturated 16-pair leg.
# usage: gpurun --timeout 2400 -- 'bash scripts/gpu_profile_r06.sh r06'   -> gpurun_out/prof_<tag>/...
TAG=${1:-r06}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HEAD="python $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --pair-leg 0 --batch-leg 0 --primates-leg 0 --chunk-legs 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $HEAD > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $HEAD > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $HEAD > /dev/null 2> $OUT/pmc_write.log
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python $ROOT/scripts/pmc_summary.py "$OUT/pmc_fetch/**/*counter_collection.csv" "$OUT/pmc_write/**/*counter_collection.csv" "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`$HEAD\` (evolverMammals stand-in), MI355X, $TAG" > $OUT/hbm_traffic_pmc.json
for W in chr20 hm hm30; do
  CMD="python $ROOT/bench.py --workload $W --steps 2 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_stats -- $CMD > $OUT/${W}_bench_under_rocprof.json 2> $OUT/${W}_stats.log
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${W}_pmc_fetch -- $CMD > /dev/null 2> $OUT/${W}_pmc_fetch.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${W}_pmc_write -- $CMD > /dev/null 2> $OUT/${W}_pmc_write.log
  find $OUT/${W}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${W}_kernel_stats.csv
  python $ROOT/scripts/pmc_summary.py "$OUT/${W}_pmc_fetch/**/*counter_collection.csv" "$OUT/${W}_pmc_write/**/*counter_collection.csv" "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`$CMD\`, 3 steps, MI355X, $TAG" > $OUT/${W}_hbm_traffic_pmc.json
done
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
find $OUT -type d -name "*pmc_*" -exec rm -rf {} + 2>/dev/null
head -6 $OUT/kernel_stats.csv | cut -c1-160

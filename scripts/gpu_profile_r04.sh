#!/bin/bash
# rocprofv3 evidence of round 4 for profiles/: kernel stats of the default bench command (headline + legs), kernel stats and HBM traffic
# counters (separate --pmc passes) of the chunk-scale workloads chr20 / hm and of the 8 Mb seed leg.
# usage: gpurun --timeout 2400 -- 'bash scripts/gpu_profile_r04.sh r04'   -> gpurun_out/prof_<tag>/...
TAG=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HEAD="python $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --pair-leg 0 --batch-leg 0 --primates-leg 0 --chunk-legs 0"
( cd $ROOT && python bench.py > $OUT/bench.json 2> $OUT/bench.err )
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $HEAD > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $HEAD > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $HEAD > /dev/null 2> $OUT/pmc_write.log
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python $ROOT/scripts/pmc_summary.py "$OUT/pmc_fetch/**/*counter_collection.csv" "$OUT/pmc_write/**/*counter_collection.csv" "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`$HEAD\` (evolverMammals stand-in), MI355X, $TAG" > $OUT/hbm_traffic_pmc.json
for W in chr20 hm; do
  CMD="python $ROOT/bench.py --workload $W --steps 2 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_stats -- $CMD > $OUT/${W}_bench_under_rocprof.json 2> $OUT/${W}_stats.log
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${W}_pmc_fetch -- $CMD > /dev/null 2> $OUT/${W}_pmc_fetch.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${W}_pmc_write -- $CMD > /dev/null 2> $OUT/${W}_pmc_write.log
  find $OUT/${W}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${W}_kernel_stats.csv
  python $ROOT/scripts/pmc_summary.py "$OUT/${W}_pmc_fetch/**/*counter_collection.csv" "$OUT/${W}_pmc_write/**/*counter_collection.csv" "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`$CMD\`, 3 steps, MI355X, $TAG" > $OUT/${W}_hbm_traffic_pmc.json
done
SCMD="python $ROOT/scripts/gpu_rand.py 8000000"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sstats -- $SCMD > $OUT/seed_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/seed_pmc_fetch -- $SCMD > /dev/null 2> $OUT/seed_pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/seed_pmc_write -- $SCMD > /dev/null 2> $OUT/seed_pmc_write.log
find $OUT/sstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/seed_kernel_stats.csv
python $ROOT/scripts/pmc_summary.py "$OUT/seed_pmc_fetch/**/*counter_collection.csv" "$OUT/seed_pmc_write/**/*counter_collection.csv" "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`$SCMD\` (two jobs: 8 Mb x 8 Mb pure-random pair, 20 % soft-masked), MI355X, $TAG" > $OUT/seed_hbm_traffic_pmc.json
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
find $OUT -type d -name "*pmc_*" -exec rm -rf {} + 2>/dev/null
head -8 $OUT/kernel_stats.csv | cut -c1-160
tail -c 400 $OUT/bench.json

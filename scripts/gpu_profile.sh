#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats of the default bench workload + HBM traffic counters (separate passes), the same for
# the 1 Mb pair, kernel stats of the seed stage on the 8 Mb random pair.   usage: gpurun -- 'bash scripts/gpu_profile.sh r03'
# -> gpurun_out/prof_<tag>/{bench.json,kernel_stats.csv,hbm_traffic_pmc.json,pair_*,seed_kernel_stats.csv}; copy into profiles/<tag>_*
TAG=${1:-rXX}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --pair-leg 0 --batch-leg 0 --primates-leg 0"
PCMD="python $ROOT/bench.py --workload pair --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
( cd $ROOT && python bench.py > $OUT/bench.json 2> $OUT/bench.err )
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pstats -- $PCMD > $OUT/pair_bench_under_rocprof.json 2> $OUT/pstats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/ppmc_fetch -- $PCMD > /dev/null 2> $OUT/ppmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/ppmc_write -- $PCMD > /dev/null 2> $OUT/ppmc_write.log
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/pstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/pair_kernel_stats.csv
python $ROOT/scripts/pmc_summary.py "$OUT/pmc_fetch/**/*counter_collection.csv" "$OUT/pmc_write/**/*counter_collection.csv" "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`python bench.py --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --pair-leg 0 --batch-leg 0 --primates-leg 0\` (evolverMammals stand-in), MI355X, $TAG" > $OUT/hbm_traffic_pmc.json
python $ROOT/scripts/pmc_summary.py "$OUT/ppmc_fetch/**/*counter_collection.csv" "$OUT/ppmc_write/**/*counter_collection.csv" "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`python bench.py --workload pair --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0\` (1 Mb x 1 Mb pair), MI355X, $TAG" > $OUT/pair_hbm_traffic_pmc.json
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
head -12 $OUT/kernel_stats.csv | cut -c1-200
python - $OUT <<'PY'
import json,sys
for f in ("hbm_traffic_pmc.json","pair_hbm_traffic_pmc.json"):
    d=json.load(open(sys.argv[1]+"/"+f))
    for k,v in d["kernels"].items():
        if "ydrop" in k: print(f,k,v)
PY
tail -c 600 $OUT/bench.json
# seed / ungapped stage on the 8 Mb random pair (the seed_stage leg's workload): kernel stats
SCMD="python $ROOT/scripts/gpu_rand.py 8000000"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sstats -- $SCMD > $OUT/seed_under_rocprof.log 2>&1
find $OUT/sstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/seed_kernel_stats.csv
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete

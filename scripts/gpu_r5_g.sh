#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5g; mkdir -p gpurun_out/$TAG
( time timeout 900 python -m pytest tests -m gpu -x -q -k "relay or phase or smoke or chunked or 1mb" ) > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/$TAG/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --chunk-legs 0 --primates-leg 0 --pair-leg 0 --batch-leg 16 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python scripts/bench_summary.py gpurun_out/$TAG/bench.json | head -3 | cut -c1-500; python scripts/bench_summary.py gpurun_out/$TAG/bench.json | grep batched | cut -c1-400
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; cd /tmp && export TMPDIR=/tmp
HEAD="python $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --pair-leg 0 --batch-leg 0 --primates-leg 0 --chunk-legs 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $HEAD > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $HEAD > /dev/null 2> $OUT/pmc_write.log
python $ROOT/scripts/pmc_summary.py "$OUT/pmc_fetch/**/*counter_collection.csv" "$OUT/pmc_write/**/*counter_collection.csv" "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`$HEAD\` (evolverMammals stand-in), MI355X, r05 (snapshots written through, no agent-scope fence)" > $OUT/hbm_traffic_pmc.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $HEAD > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete; find $OUT -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null
python - <<PY
import json
d=json.load(open("$OUT/hbm_traffic_pmc.json"))["kernels"]
for k,v in d.items():
    if "ydrop" in k: print(k, v)
PY

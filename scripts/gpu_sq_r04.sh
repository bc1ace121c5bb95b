#!/bin/bash
# SQ issue-slot counters of the seed-stage kernels on the 8 Mb x 8 Mb random pair (one --pmc pass of 8 SQ counters, kernel trace only).
# usage: gpurun --timeout 600 -- 'bash scripts/gpu_sq_r04.sh r04'  -> gpurun_out/sq_<tag>/seed_sq_counters.json
TAG=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sq_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SCMD="python $ROOT/scripts/gpu_rand.py 8000000"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc -- $SCMD > $OUT/run.log 2>&1; echo "rc=$?"
python $ROOT/scripts/sq_summary.py "$OUT/pmc/**/*counter_collection.csv" "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU over \`$SCMD\` (two jobs), MI355X, $TAG" > $OUT/seed_sq_counters.json
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
tail -3 $OUT/run.log; head -c 1500 $OUT/seed_sq_counters.json

#!/bin/bash
# device timeline of the headline workload: every kernel and copy with its start/end stamps (rocprofv3 --kernel-trace
# --memory-copy-trace), for scripts/trace_gaps.py.   usage: gpurun -- 'bash scripts/gpu_trace.sh <tag> [ENV=VALUE ...]'
TAG=${1:-trace}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/raw -- python $ROOT/bench.py --steps 6 --warmup 2 --cpu-sample 0 \
    --seed-leg 0 --chain-leg 0 --pair-leg 0 --batch-leg 0 --primates-leg 0 > $OUT/bench.json 2> $OUT/bench.err
find $OUT/raw -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace.csv
find $OUT/raw -name "*memory_copy_trace.csv" | head -1 | xargs -I{} cp {} $OUT/memory_copy_trace.csv
rm -rf $OUT/raw
ls -la $OUT; head -2 $OUT/kernel_trace.csv | cut -c1-400; head -2 $OUT/memory_copy_trace.csv | cut -c1-300

#!/bin/bash
# The default bench line (the driver's command) + a one-screen summary.   usage: gpurun --timeout 1500 -- 'bash scripts/gpu_bench.sh <tag> [bench args...]'
TAG=${1:-bench}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
( time timeout 1200 python bench.py "$@" > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err ) 2>&1 | grep real; echo "bench rc=$?"
tail -5 gpurun_out/$TAG/bench.err
python scripts/bench_summary.py gpurun_out/$TAG/bench.json | cut -c1-1500

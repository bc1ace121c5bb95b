#!/bin/bash
# where a step of the phase goes: per-call timeline of bench.py + the library's own debug timeline.   usage:
#   gpurun -- 'bash scripts/gpu_timeline.sh <tag> [ENV=VALUE ...]'
TAG=${1:-timeline}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
env "$@" MIBLAST_BENCH_TIMELINE=1 MIBLAST_DEBUG=1 timeout 300 python bench.py --steps 4 --warmup 2 --pair-leg 0 --batch-leg 0 --chain-leg 0 --seed-leg 0 --primates-leg 0 --cpu-sample 0 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
tail -150 gpurun_out/$TAG/bench.err | cut -c1-330
python scripts/bench_summary.py gpurun_out/$TAG/bench.json

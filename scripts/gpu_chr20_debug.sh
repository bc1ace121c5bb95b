#!/bin/bash
# chr20 workload with the library's debug timeline (stderr).  usage: gpurun -- 'bash scripts/gpu_chr20_debug.sh <tag> [env assignments]'
TAG=${1:-c20dbg}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
for kv in "$@"; do export "$kv"; done
MIBLAST_DEBUG=1 timeout 300 python bench.py --workload chr20 --steps 2 --warmup 1 --cpu-sample 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 > gpurun_out/$TAG/chr20.json 2> gpurun_out/$TAG/chr20.err
grep -E "call of|seed stage|gapped|round|output" gpurun_out/$TAG/chr20.err | tail -60 | cut -c1-400

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/dbg
MIBLAST_DEBUG=1 timeout 300 python bench.py --workload pair --pairs-per-gpu 16 --steps 2 --warmup 2 --chain-leg 0 --cpu-sample 0 --pair-leg 0 --batch-leg 0 --seed-leg 0 > gpurun_out/dbg/out16.log 2> gpurun_out/dbg/err16.log
wc -l gpurun_out/dbg/err16.log

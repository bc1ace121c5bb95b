#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5e; mkdir -p gpurun_out/$TAG
MIBLAST_DEBUG=1 MIBLAST_BENCH_TIMELINE=1 timeout 120 python bench.py --steps 2 --warmup 2 --chunk-legs 0 --primates-leg 0 --pair-leg 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/dbg.json 2> gpurun_out/$TAG/dbg.err
grep -E "host timeline of the index|^\[bench\]|call of" gpurun_out/$TAG/dbg.err | tail -32 | cut -c1-260

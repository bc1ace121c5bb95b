#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5k; mkdir -p gpurun_out/$TAG
MIBLAST_DEBUG=3 timeout 120 python bench.py --steps 1 --warmup 1 --chunk-legs 0 --primates-leg 0 --pair-leg 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/dbg.json 2> gpurun_out/$TAG/dbg.err
grep -c "nominee" gpurun_out/$TAG/dbg.err

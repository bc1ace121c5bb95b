#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
MIBLAST_DEBUG_SPIKE=7 MIBLAST_BENCH_TIMELINE=1 timeout 300 python bench.py --steps 40 --warmup 3 --chain-leg 0 --cpu-sample 0 --pair-leg 0 --batch-leg 0 --seed-leg 0 2>&1 | grep "slow\|step total\|align_pairs" | tail -230 > gpurun_out/spike.log
grep -c "slow" gpurun_out/spike.log

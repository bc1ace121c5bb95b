#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
MIBLAST_BENCH_TIMELINE=1 timeout 300 python bench.py --steps 40 --warmup 3 --pair-leg 0 --batch-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>&1 >/dev/null | grep "of 9 pairs" | awk '{print $6, $13, $15, $19}' | tr '\n' ';'
echo

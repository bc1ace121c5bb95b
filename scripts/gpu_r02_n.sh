#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for lanes in 4 8; do
echo "== lanes $lanes"
MIBLAST_SEED_LANES=$lanes MIBLAST_BENCH_TIMELINE=1 timeout 300 python bench.py --steps 3 --warmup 2 --pair-leg 0 --batch-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>&1 >/dev/null | grep "bench\]" | tail -5
done

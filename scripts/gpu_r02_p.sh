#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for ch in 1 0; do echo "== chain_heads $ch"
MIBLAST_CHAIN_HEADS=$ch MIBLAST_DEBUG=1 timeout 300 python bench.py --workload pair --steps 1 --warmup 1 --batch-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>&1 >/dev/null | grep "round [0-9]*:\|round [0-9]*\.[0-9]*:\|host timeline" | tail -8 | cut -c1-210
done

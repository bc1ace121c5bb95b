#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
MIBLAST_DEBUG=2 timeout 300 python bench.py --workload pair --steps 2 --warmup 3 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>&1 >/dev/null | tail -40 | cut -c1-220

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
MIBLAST_DEBUG=3 timeout 600 python scripts/gpu_cfg4.py 2>&1 | grep "head without\|round 0:" | head -30 | cut -c1-260

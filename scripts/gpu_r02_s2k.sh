#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
MIBLAST_UNGAPPED=ux python scripts/gpu_rand.py 8000000 2>&1 | tail -1 | cut -c1-900

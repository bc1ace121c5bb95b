#!/bin/bash
# session 2: host timeline of the 30 Mb x 30 Mb chunk pair
cd "$GRAFT_REPO_ROOT" || exit 1
MIBLAST_DEBUG=1 timeout 250 python scripts/gpu_cfg4.py 2>&1 | grep -v "^generated" | tail -45 | cut -c1-260

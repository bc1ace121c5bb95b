#!/bin/bash
# session 2: the 30 Mb x 30 Mb chunk pair of config 4 with either ungapped kernel choice
cd "$GRAFT_REPO_ROOT" || exit 1
for m in lane ux; do echo "== $m"; MIBLAST_UNGAPPED=$m timeout 250 python scripts/gpu_cfg4.py 2>&1 | grep "rep\|equal\|same"; done

#!/bin/bash
# round 6: the blocked path (several target / query blocks, the per-query HSP limits ranked over all blocks) against one oracle run per case
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6fuzz2
( timeout 500 python scripts/gpu_multi_fuzz.py ${1:-300} ${2:-3000} ) > gpurun_out/r6fuzz2/multi.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r6fuzz2/multi.log | cut -c1-300

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5c; mkdir -p gpurun_out/$TAG
timeout 300 bin/ubench_issue > gpurun_out/$TAG/ubench_issue.txt 2>&1; cat gpurun_out/$TAG/ubench_issue.txt
MIBLAST_DEBUG=1 MIBLAST_CROWD_SIDES=1000000 timeout 200 python bench.py --workload chr20 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/$TAG/chr20_mid_dbg.json 2> gpurun_out/$TAG/chr20_mid_dbg.err
grep -E "round 0: 2[0-9]{2} sides" gpurun_out/$TAG/chr20_mid_dbg.err | head -4

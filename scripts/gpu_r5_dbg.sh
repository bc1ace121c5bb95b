#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5dbg; mkdir -p gpurun_out/$TAG
MIBLAST_DEBUG=1 timeout 200 python bench.py --workload hm --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/$TAG/hm.json 2> gpurun_out/$TAG/hm.err
MIBLAST_DEBUG=1 timeout 200 python bench.py --workload chr20 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/$TAG/chr20.json 2> gpurun_out/$TAG/chr20.err
wc -l gpurun_out/$TAG/*.err

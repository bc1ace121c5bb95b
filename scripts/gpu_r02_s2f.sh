#!/bin/bash
# session 2: hardware queues / seed lanes vs step time of the phase
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2f
run() { env "$@" MIBLAST_UNGAPPED=lane timeout 300 python bench.py --steps 16 --warmup 3 --pair-leg 0 --batch-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', 'ms/step', round(d['ms_per_step'],2), 'dp', round(d['stage_kernel_ms_per_step']['ydrop'],2), 'launch_ms', round(d['roofline']['launch_ms'],3), 'frac', round(d['roofline']['frac'],4), 'busy', round(d['host']['busy_threads_avg'],1))"; }
run A=0
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=16
run GPU_MAX_HW_QUEUES=16 MIBLAST_SEED_LANES=9
run GPU_MAX_HW_QUEUES=2
run MIBLAST_SEED_LANES=4
run MIBLAST_BENCH_CONTEXTS=3 MIBLAST_BENCH_SPLIT=6
run MIBLAST_BENCH_CONTEXTS=3 MIBLAST_BENCH_SPLIT=6 GPU_MAX_HW_QUEUES=16

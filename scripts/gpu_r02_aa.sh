#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "1 0" "2 0" "3 6" "3 4" "4 4" "5 3"; do set -- $cfg
MIBLAST_BENCH_CONTEXTS=$1 MIBLAST_BENCH_SPLIT=$2 timeout 300 python bench.py --steps 16 --warmup 3 --pair-leg 0 --batch-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('contexts $1 split $2', 'ms/step', round(d['ms_per_step'],2), 'value', round(d['value'],2), 'dp', round(d['stage_kernel_ms_per_step']['ydrop'],2), 'same_bytes', d['cpu_baseline']['same_bytes'], 'busy', round(d['host']['busy_threads_avg'],1))"
done

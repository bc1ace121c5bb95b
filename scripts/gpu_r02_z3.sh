#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
cat /sys/fs/cgroup/cpu.stat | head -8
for thr in 0 8 4; do
MIBLAST_THREADS=$thr timeout 300 python bench.py --steps 30 --warmup 3 --pair-leg 0 --batch-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('threads $thr', 'ms/step', round(d['ms_per_step'],2), d['host'])"
done

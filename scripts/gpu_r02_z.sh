#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
MIBLAST_BENCH_TIMELINE=1 timeout 300 python bench.py --steps 30 --warmup 3 --pair-leg 0 --batch-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>&1 >/dev/null | grep "step total\|of 9 pairs" | awk '/of 9 pairs/{b=$6} /step total/{print $4, b}' | tr '\n' ';'
echo
nproc; cat /proc/loadavg; python -c "import os; print(len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null

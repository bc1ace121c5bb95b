#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02i
MIBLAST_BENCH_TIMELINE=1 MIBLAST_DEBUG=1 timeout 300 python bench.py --steps 2 --warmup 2 --pair-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 > gpurun_out/r02i/bench.json 2> gpurun_out/r02i/bench.err
awk '/\[bench\] step total/{n++} n==3' gpurun_out/r02i/bench.err | grep "bench\]\|host timeline\|seed phase\|round [0-9]*:" | head -40 | cut -c1-200
MIBLAST_BENCH_TIMELINE=1 timeout 300 python bench.py --steps 3 --warmup 2 --pair-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>&1 >/dev/null | grep "bench\]" | tail -10

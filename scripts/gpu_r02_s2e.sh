#!/bin/bash
# session 2: where a step of the phase goes (per-call timeline + the library's own debug timeline of the last step)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2e
MIBLAST_UNGAPPED=lane MIBLAST_BENCH_TIMELINE=1 MIBLAST_DEBUG=1 timeout 300 python bench.py --steps 4 --warmup 2 --pair-leg 0 --batch-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 > gpurun_out/s2e/bench.json 2> gpurun_out/s2e/bench.err
tail -150 gpurun_out/s2e/bench.err | cut -c1-330

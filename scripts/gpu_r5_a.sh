#!/bin/bash
# round 5, first GPU call: the hand-over inside the DP launch -- parity suite, the bench line, and the same headline with the hand-over
# left to the host (MIBLAST_RELAY_INLINE=0) for the A/B.   gpurun --timeout 1500 -- 'bash scripts/gpu_r5_a.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5a; mkdir -p gpurun_out/$TAG
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/$TAG/pytest.log
timeout 600 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/$TAG/bench.err
python scripts/bench_summary.py gpurun_out/$TAG/bench.json
MIBLAST_RELAY_INLINE=0 timeout 300 python bench.py --chunk-legs 0 --primates-leg 0 --pair-leg 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/bench_noinline.json 2> gpurun_out/$TAG/bench_noinline.err; echo "bench(noinline) rc=$?"
python scripts/bench_summary.py gpurun_out/$TAG/bench_noinline.json
MIBLAST_DEBUG=1 timeout 120 python bench.py --steps 1 --warmup 1 --chunk-legs 0 --primates-leg 0 --pair-leg 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/bench_debug.json 2> gpurun_out/$TAG/bench_debug.err
grep -c "round" gpurun_out/$TAG/bench_debug.err

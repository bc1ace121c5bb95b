#!/bin/bash
# round 2, GPU call b: new tests again, per-kernel profile of the evolver workload, whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02b
timeout 900 python -m pytest tests/test_multi_gpu.py tests/test_bench_gpu.py -x -q -m gpu > gpurun_out/r02b/pytest_new.log 2>&1
echo "pytest new rc=$?"; tail -5 gpurun_out/r02b/pytest_new.log
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02b/prof -o evolver -- python bench.py --steps 5 --warmup 2 --pair-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/r02b/bench_prof.json 2> gpurun_out/r02b/bench_prof.err
echo "prof rc=$?"; cat gpurun_out/r02b/bench_prof.json | head -c 1500
find gpurun_out/r02b/prof -name "*kernel_stats*" | head; f=$(find gpurun_out/r02b/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
find gpurun_out/r02b/prof -name "*.db" -size +30M -delete
MIBLAST_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 1 --pair-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/r02b/bench_debug.json 2> gpurun_out/r02b/bench_debug.err
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02b/pytest_all.log 2>&1
echo "pytest all rc=$?"; tail -5 gpurun_out/r02b/pytest_all.log

#!/bin/bash
# round 2, first GPU call: the new multi-device / blocked path, the bench harness, then the bench itself
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02a
timeout 900 python -m pytest tests/test_multi_gpu.py tests/test_bench_gpu.py -x -q -m gpu > gpurun_out/r02a/pytest_new.log 2>&1
echo "pytest new rc=$?" | tee -a gpurun_out/r02a/pytest_new.log
tail -30 gpurun_out/r02a/pytest_new.log
timeout 600 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/r02a/bench.json; tail -5 gpurun_out/r02a/bench.err

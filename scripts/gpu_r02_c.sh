#!/bin/bash
# round 2, GPU call c: whole GPU suite after the multi-device / bench / switches work
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02c
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r02c/pytest_all.log 2>&1
echo "pytest all rc=$?"; tail -8 gpurun_out/r02c/pytest_all.log
timeout 600 python bench.py > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err; echo "bench rc=$?"; head -c 600 gpurun_out/r02c/bench.json

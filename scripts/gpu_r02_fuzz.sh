#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02fuzz
timeout 1500 python scripts/gpu_fuzz.py 600 1000 > gpurun_out/r02fuzz/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r02fuzz/fuzz.log
timeout 600 python scripts/gpu_cfg4.py > gpurun_out/r02fuzz/cfg4.log 2>&1; echo "cfg4 rc=$?"; tail -4 gpurun_out/r02fuzz/cfg4.log
timeout 900 python scripts/gpu_cfg4_whole.py > gpurun_out/r02fuzz/cfg4_whole.log 2>&1; echo "cfg4 whole rc=$?"; tail -6 gpurun_out/r02fuzz/cfg4_whole.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02r
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r02r/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02r/pytest.log

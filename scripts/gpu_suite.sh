#!/bin/bash
# One gpurun call = the GPU suite + the default bench line (+ optional extra commands).   usage:
#   gpurun --timeout 1500 -- 'bash scripts/gpu_suite.sh <tag> [pytest -k expression]'
# -> gpurun_out/<tag>/{pytest.log,bench.json,bench.err}
TAG=${1:-suite}; KEXPR=${2:-}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
if [ -n "$KEXPR" ]; then
  ( time timeout 1200 python -m pytest tests -m gpu -x -q -k "$KEXPR" ) > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"
else
  ( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"
fi
tail -15 gpurun_out/$TAG/pytest.log
timeout 600 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"
tail -5 gpurun_out/$TAG/bench.err
python scripts/bench_summary.py gpurun_out/$TAG/bench.json

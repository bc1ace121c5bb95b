#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5d; mkdir -p gpurun_out/$TAG
timeout 90 bin/ubench_issue > gpurun_out/$TAG/ubench_issue.txt 2>&1; echo "ubench rc=$?"; tail -45 gpurun_out/$TAG/ubench_issue.txt
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/$TAG/pytest.log
( time timeout 900 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err ); echo "bench rc=$?"
tail -3 gpurun_out/$TAG/bench.err
python scripts/bench_summary.py gpurun_out/$TAG/bench.json

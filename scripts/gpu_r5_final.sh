#!/bin/bash
# round 5, final build: the GPU suite, the smoke test and the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r5final}; mkdir -p gpurun_out/$TAG
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/$TAG/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err ); echo "bench rc=$?"
python scripts/bench_summary.py gpurun_out/$TAG/bench.json | cut -c1-600

#!/bin/bash
# round 5: bins + LDS (staged scatter): GPU tests, standalone kernel times (one lane), A/B of the chunk-scale workloads
cd "$GRAFT_REPO_ROOT" || exit 1
( time timeout 900 python -m pytest tests -m gpu -x -q -k "dense_seed_path or grouping_by_diagonal" ) > gpurun_out/r5u_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5u_pytest.log
bash scripts/gpu_r5_t.sh
cd "$GRAFT_REPO_ROOT"; bash scripts/gpu_r5_r.sh 2>&1 | grep -v pytest | head -8

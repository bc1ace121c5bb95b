#!/bin/bash
# round 6, the closing call: the parity / chaining / multi-device suites under MIBLAST_DEBUG_GUARD=2, then call K (the whole GPU suite, the chunk-scale
# workloads, the default bench line) on the same build
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/gpu_r6_guard.sh 2 gpurun_out/r6final_guard2 1500 300 tests/test_parity_gpu.py tests/test_zz_chain_gpu.py tests/test_multi_gpu.py
sort gpurun_out/r6final_guard2/guard.log | uniq -c | sort -rn | head -8
unset MIBLAST_DEBUG_GUARD MIBLAST_DEBUG_GUARD_LOG
bash scripts/gpu_r6_k.sh ${1:-r6final}

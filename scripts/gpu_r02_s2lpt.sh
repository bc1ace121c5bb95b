#!/bin/bash
# session 2: longest pieces first in crowded DP launches
cd "$GRAFT_REPO_ROOT" || exit 1
( timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "batched or deterministic or full_size_chunk or evolver_mammals" ) 2>&1 | tail -2
run() { env "$@" timeout 300 python bench.py --steps 16 --warmup 3 --pair-leg 0 --batch-leg 16 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['batched_pairs']
print('$*', 'ms/step', round(d['ms_per_step'],2), 'dp', round(d['stage_kernel_ms_per_step']['ydrop'],2), '| batched launch', round(b['roofline']['launch_ms'],3), round(b['gapped_gcells_per_s_kernel'],1), round(b['ms_per_call'],1))"; }
run MIBLAST_DP_LPT=0
run MIBLAST_DP_LPT=1
run MIBLAST_DP_LPT=0
run MIBLAST_DP_LPT=1

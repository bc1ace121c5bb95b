#!/bin/bash
# session 2: 4 vs 5 waves per SIMD of the DP kernel
cd "$GRAFT_REPO_ROOT" || exit 1
run() { env "$@" MIBLAST_UNGAPPED=lane timeout 300 python bench.py --steps 16 --warmup 3 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pair_1mb']; b=d['batched_pairs']
print('$*', 'ms/step', round(d['ms_per_step'],2), 'dp', round(d['stage_kernel_ms_per_step']['ydrop'],2), 'Gc/s', round(d['gapped_gcells_per_s_kernel'],1), '| pair launch', round(p['roofline']['launch_ms'],3), '| batched launch', round(b['roofline']['launch_ms'],3), round(b['gapped_gcells_per_s_kernel'],1), round(b['ms_per_call'],1))"; }
run A=0
run MIBLAST_DP_WAVES=4
run MIBLAST_DP_WAVES=5

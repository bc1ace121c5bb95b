#!/bin/bash
# session 2: ungapped kernel choice on the phase (sparse hit sets)
cd "$GRAFT_REPO_ROOT" || exit 1
run() { env "$@" timeout 300 python bench.py --steps 16 --warmup 3 --pair-leg 1 --batch-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pair_1mb']
print('$*', 'ms/step', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['stage_kernel_ms_per_step'].items()}, '| pair', round(p['ms_per_step'],2), {k:round(v,2) for k,v in p['stage_kernel_ms_per_step'].items()})"; }
run MIBLAST_UNGAPPED=lane
run MIBLAST_UNGAPPED=ux
run MIBLAST_UNGAPPED=lane
run MIBLAST_UNGAPPED=ux

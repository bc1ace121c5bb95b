#!/bin/bash
# session 2: warm-up rows and spacing of the relays of calls with a handful of sides
cd "$GRAFT_REPO_ROOT" || exit 1
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --pair-leg 0 --batch-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*', 'ms/step', round(d['ms_per_step'],2), 'dp', round(d['stage_kernel_ms_per_step']['ydrop'],2), 'launches', d['relay']['dp_launches_per_step'], 'pieces', d['relay']['pieces_per_step'], 'rej', d['relay']['handovers_rejected_per_step'], 'spec', round(d['speculation_factor'],3))"; }
run MIBLAST_RELAY_W_TINY=256
run MIBLAST_RELAY_W_TINY=384
run MIBLAST_RELAY_W_TINY=512
run MIBLAST_RELAY_W_TINY=256 MIBLAST_RELAY_S_TINY=640
run MIBLAST_RELAY_W_TINY=384 MIBLAST_RELAY_S_TINY=640
run MIBLAST_RELAY_W_TINY=128

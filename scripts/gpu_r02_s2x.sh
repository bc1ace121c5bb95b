#!/bin/bash
# session 2: N runs looked up instead of scanned between anchors: 30 Mb pair, parity subset, bench
cd "$GRAFT_REPO_ROOT" || exit 1
MIBLAST_DEBUG=2 timeout 250 python scripts/gpu_cfg4.py 2>&1 | grep "build_units\|rep \|equal" | cut -c1-260 | tail -6
( timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "case_matches or relay or deterministic or fuzz or evolver" ) 2>&1 | tail -2
timeout 300 python bench.py --steps 16 --warmup 3 --pair-leg 1 --batch-leg 16 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pair_1mb']; b=d['batched_pairs']
print('evolver ms/step', round(d['ms_per_step'],2), 'value', round(d['value'],2), '| pair', round(p['ms_per_step'],2), '| batched', round(b['ms_per_call'],1), round(b['value'],1))"

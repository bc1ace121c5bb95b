#!/bin/bash
# session 2: device-side anchors (k_hsp_anchor): parity tests with the host cross-check, then the 30 Mb pair and the bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2v
( time MIBLAST_CHECK_ANCHORS=1 timeout 600 python -m pytest tests -m gpu -x -q ) > gpurun_out/s2v/pytest.log 2>&1; echo "pytest(anchor cross-check on) rc=$?"; tail -4 gpurun_out/s2v/pytest.log
MIBLAST_DEBUG=1 timeout 250 python scripts/gpu_cfg4.py 2>&1 | grep "rep\|equal\|seed phase" | cut -c1-300
timeout 300 python bench.py --steps 10 --warmup 3 --pair-leg 1 --batch-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pair_1mb']
print('evolver ms/step', round(d['ms_per_step'],2), 'value', round(d['value'],2), '| pair', round(p['ms_per_step'],2))"

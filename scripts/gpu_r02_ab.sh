#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02ab
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r02ab/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02ab/pytest.log
for f in 1 0 1 0; do
MIBLAST_SEED_FUSED=$f timeout 300 python bench.py --steps 12 --warmup 3 --chain-leg 0 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fused $f', 'evolver', round(d['ms_per_step'],2), '| pair', round(d['pair_1mb']['ms_per_step'],2), '| batched', round(d['batched_pairs']['ms_per_call'],1), '| seed leg', round(d['seed_stage']['seconds']*1e3,1), d['seed_stage']['kernel_ms'])"
done

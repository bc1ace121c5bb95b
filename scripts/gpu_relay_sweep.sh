#!/bin/bash
# sweep of the relay knobs on the default bench pair
for cfg in "640 128" "640 96" "768 96" "896 96" "1024 128" "512 96"; do
  set -- $cfg
  echo "== S=$1 W=$2"
  MIBLAST_RELAY_S=$1 MIBLAST_RELAY_W=$2 MIBLAST_DEBUG=1 timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --seed-leg 0 2>&1 | grep "round 0:\|metric" | tail -2 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   ms/step', round(d['ms_per_step'], 2), 'ydrop ms', round(d['stage_kernel_ms_per_step']['ydrop'], 2), 't_gapped', round(d['stage_seconds_per_step']['t_gapped']*1e3, 2), 'spec', round(d['speculation_factor'], 2))
    else: print('  ', l.strip()[:170])
"
done

#!/bin/bash
# the same bench command several times, exit codes on record (intermittent failures).  usage: gpurun -- 'bash scripts/gpu_repeat.sh <n> "<env assignments>" <bench args...>'
N=$1; SETTING=$2; shift; shift
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
for i in $(seq 1 $N); do
  ( for kv in $SETTING; do export "$kv"; done; timeout 120 python bench.py "$@" > /tmp/o.json 2>/tmp/o.err ); rc=$?
  echo "run $i rc=$rc bytes=$(wc -c < /tmp/o.json) $(python -c "
import json
try:
    d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), d.get('parity',{}).get('same_bytes'))
except Exception as e: print('no line')
")"
  if [ $rc -ne 0 ]; then tail -5 /tmp/o.err; fi
done

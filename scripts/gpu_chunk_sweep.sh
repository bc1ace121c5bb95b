#!/bin/bash
# chr20 / hm workloads under a list of env settings.  usage: gpurun -- 'bash scripts/gpu_chunk_sweep.sh <tag> "K=V K=V" "K=V" ...'
TAG=${1:-sweep}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
i=0
for setting in "$@"; do
  i=$((i+1))
  for W in chr20 hm; do
    ( for kv in $setting; do export "$kv"; done
      timeout 300 python bench.py --workload $W --steps 3 --warmup 1 --cpu-sample 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 > gpurun_out/$TAG/${W}_$i.json 2> gpurun_out/$TAG/${W}_$i.err )
    python - gpurun_out/$TAG/${W}_$i.json "$setting" $W <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[3]:6s} [{sys.argv[2]}] {d['ms_per_step']:.1f} ms/step parity {d['parity']['same_bytes']} busy_threads {d['host']['busy_threads_avg']:.1f} dp_busy {d['stage_kernel_ms_per_step']['ydrop_busy']:.1f}")
except Exception as e: print(sys.argv[3], sys.argv[2], "failed", e)
PY
  done
done

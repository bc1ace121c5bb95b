#!/bin/bash
# the batched_pairs leg alone, a few times per setting.   usage: gpurun -- 'bash scripts/gpu_batched_ab.sh <tag> "ENV=V" ...'
TAG=${1:-bab}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
k=0
for setting in "$@"; do
  for rep in 1 2; do
    k=$((k+1))
    env $setting timeout 300 python bench.py --steps 3 --warmup 2 --pair-leg 0 --chain-leg 0 --seed-leg 0 --primates-leg 0 --cpu-sample 0 > gpurun_out/$TAG/b$k.json 2> gpurun_out/$TAG/b$k.err
    python - "$setting" gpurun_out/$TAG/b$k.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    b = d["batched_pairs"]
    print("%-50s batched %.2f ms/call, kernel %.0f Gc/s, launches %.0f, spec %.2f" % (sys.argv[1] or "(defaults)", b["ms_per_call"], b["gapped_gcells_per_s_kernel"], b["dp_launches_per_call"], b["speculation_factor"]))
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
  done
done

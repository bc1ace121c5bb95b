#!/bin/bash
# A/B of environment knobs on the headline workload: one bench run (phase only) per setting.   usage:
#   gpurun -- 'bash scripts/gpu_env_sweep.sh <tag> "ENV=V ENV2=W" "ENV=X" ...'      (an empty string = the defaults)
TAG=${1:-sweep}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
k=0
for setting in "$@"; do
  k=$((k+1))
  env $setting timeout 300 python bench.py --steps ${SWEEP_STEPS:-40} --warmup 5 --pair-leg 0 --batch-leg 0 --chain-leg 0 --seed-leg 0 --primates-leg 0 --cpu-sample 0 > gpurun_out/$TAG/b$k.json 2> gpurun_out/$TAG/b$k.err
  python - "$setting" gpurun_out/$TAG/b$k.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    sp = d.get("step_ms_spread", {})
    print("%-60s %.2f ms/step (min %.1f median %.1f)  dp busy %.2f  launches %.0f  spec %.2f" % (sys.argv[1] or "(defaults)", d["ms_per_step"], sp.get("min", 0), sp.get("median", 0), d["stage_kernel_ms_per_step"]["ydrop_busy"], d["relay"]["dp_launches_per_step"], d["speculation_factor"]))
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
done

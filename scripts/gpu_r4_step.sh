#!/bin/bash
# One gpurun call of round 4's seed-stage work: a parity subset, then the chunk-scale timings.   usage:
#   gpurun --timeout 1500 -- 'bash scripts/gpu_r4_step.sh <tag> [pytest -k expression|none] [extra env assignments...]'
TAG=${1:-r4}; KEXPR=${2:-}; shift; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
for kv in "$@"; do export "$kv"; done
if [ "$KEXPR" != "none" ]; then
  if [ -n "$KEXPR" ]; then
    ( time timeout 1200 python -m pytest tests -m gpu -x -q -k "$KEXPR" ) > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"
  else
    ( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"
  fi
  tail -8 gpurun_out/$TAG/pytest.log
fi
timeout 300 python bench.py --workload chr20 --steps 3 --warmup 1 --cpu-sample 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 > gpurun_out/$TAG/chr20.json 2> gpurun_out/$TAG/chr20.err; echo "chr20 rc=$?"
tail -3 gpurun_out/$TAG/chr20.err
python - gpurun_out/$TAG/chr20.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("chr20 ms/step", round(d["ms_per_step"],1), "md5", d["config"].get("paf_md5"), "stage kernel ms", {k:round(v,1) for k,v in d["stage_kernel_ms_per_step"].items()}, "stage s", {k:round(v,3) for k,v in d["stage_seconds_per_step"].items()})
except Exception as e: print("no chr20 line", e)
PY
timeout 300 python scripts/gpu_rand.py 8000000 2>&1 | tail -1 | cut -c1-700

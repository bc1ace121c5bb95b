#!/bin/bash
# round 6, call D: the chunk-scale workloads after the shared high-water marks (device_allocs_in_timed_steps), whole pairs against strand halves at N = 1
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6d; mkdir -p $OUT; rm -f $OUT/*
COMMON="--cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
run() { tag=$1; shift; timeout 400 python bench.py "$@" $COMMON --full-out $OUT/$tag.full.json > $OUT/$tag.json 2> $OUT/$tag.err; rc=$?
  python - "$OUT/$tag.full.json" "$tag" $rc <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-22s rc=%s %7.1f ms/step (min %.1f median %.1f max %.1f) allocs_in_timed_steps %s same_bytes %s units %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["step_ms_spread"]["min"], d["step_ms_spread"]["median"], d["step_ms_spread"]["max"], d.get("device_allocs_in_timed_steps"), d.get("parity", {}).get("same_bytes"), d["config"].get("units_per_rank")))
except Exception as e:
    print(sys.argv[2], "rc=" + sys.argv[3], "unreadable:", e)
PY
}
run chr20 --workload chr20 --steps 10 --warmup 3
run chr20_split --workload chr20 --steps 10 --warmup 3 --split-strands 1
run hm30 --workload hm30 --steps 5 --warmup 2
run hm30_split --workload hm30 --steps 5 --warmup 2 --split-strands 1
run hm --workload hm --steps 10 --warmup 3
run hm_split --workload hm --steps 10 --warmup 3 --split-strands 1

#!/bin/bash
# round 6, call Z7: (chunk pair, strand) as the unit of a one-GPU step (--split-strands 1: the units the ranks share at N > 1) against whole pairs
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6z7; mkdir -p $OUT; rm -f $OUT/*
COMMON="--cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
run() { tag=$1; shift; timeout 400 python bench.py "$@" $COMMON --full-out $OUT/$tag.full.json > $OUT/$tag.json 2> $OUT/$tag.err; rc=$?
  python - "$OUT/$tag.full.json" "$tag" $rc <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); k = d.get("stage_kernel_ms_per_step", {})
    print("%-22s rc=%s %7.1f ms/step (min %.1f median %.1f max %.1f) allocs %s same_bytes %s ungapped %.1f sort %.1f seed %.1f ydrop %.1f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["step_ms_spread"]["min"], d["step_ms_spread"]["median"], d["step_ms_spread"]["max"], d.get("device_allocs_in_timed_steps"), d.get("parity", {}).get("same_bytes"), k.get("ungapped", 0), k.get("sort", 0), k.get("seed_fill", 0), k.get("ydrop", 0)))
except Exception as e:
    print(sys.argv[2], "rc=" + sys.argv[3], "unreadable:", e)
PY
}
run chr20_whole --workload chr20 --steps 10 --warmup 3
run chr20_split --workload chr20 --steps 10 --warmup 3 --split-strands 1
MIBLAST_PIPELINE_LANES=9 run chr20_split_l9 --workload chr20 --steps 10 --warmup 3 --split-strands 1
run chr20_whole_b --workload chr20 --steps 10 --warmup 3
run hm_split --workload hm --steps 10 --warmup 3 --split-strands 1

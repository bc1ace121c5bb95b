#!/bin/bash
# round 6, call X: the 30 Mb pair's q batches at the size the bins take (MIBLAST_HIT_CAP = 2^26 and below) against the default (2^27: radix sort per batch)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6x; mkdir -p $OUT; rm -f $OUT/*
COMMON="--cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
run() { tag=$1; shift; timeout 400 python bench.py "$@" $COMMON --full-out $OUT/$tag.full.json > $OUT/$tag.json 2> $OUT/$tag.err; rc=$?
  python - "$OUT/$tag.full.json" "$tag" $rc <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d.get("stage_kernel_ms_per_step", {})
    print("%-22s rc=%s %7.1f ms/step (min %.1f median %.1f max %.1f) allocs %s same_bytes %s binned %s ungapped %.1f sort %.1f seed %.1f ydrop %.1f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["step_ms_spread"]["min"], d["step_ms_spread"]["median"], d["step_ms_spread"]["max"], d.get("device_allocs_in_timed_steps"), d.get("parity", {}).get("same_bytes"), d.get("strands_grouped_in_lds_per_step"), k.get("ungapped", 0), k.get("sort", 0), k.get("seed_fill", 0), k.get("ydrop", 0)))
except Exception as e:
    print(sys.argv[2], "rc=" + sys.argv[3], "unreadable:", e)
PY
}
run hm30_default --workload hm30 --steps 5 --warmup 2
MIBLAST_HIT_CAP=$((1<<26)) run hm30_cap26 --workload hm30 --steps 5 --warmup 2
MIBLAST_HIT_CAP=$((48<<20)) run hm30_cap48m --workload hm30 --steps 5 --warmup 2
MIBLAST_HIT_CAP=$((1<<28)) run hm30_cap28 --workload hm30 --steps 5 --warmup 2
run hm30_default2 --workload hm30 --steps 5 --warmup 2

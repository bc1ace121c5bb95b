#!/bin/bash
# round 6, call Z3: the mean bin size of the diagonal grouping under six lanes (MIBLAST_BIN_MEAN, 11 000: the large sorter's 146 KB of LDS per work-group)
# against smaller bins (the small sorter: 41 KB, three work-groups per CU; more than 2 048 bins take the unstaged scatter)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6z3; mkdir -p $OUT; rm -f $OUT/*
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
for M in 11000 5600 2800 1400 11000; do MIBLAST_BIN_MEAN=$M run hm_mean$M --workload hm --steps 10 --warmup 3; done
MIBLAST_SORT_BIN=0 run hm_nobins --workload hm --steps 10 --warmup 3

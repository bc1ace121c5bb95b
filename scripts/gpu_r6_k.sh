#!/bin/bash
# round 6, call K: the GPU suite, the chunk-scale workloads (allocations left in the timed steps) and the default bench line on the current build
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r6k}; mkdir -p $OUT; rm -f $OUT/*
( time timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
COMMON="--cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
run() { tag=$1; shift; timeout 400 python bench.py "$@" $COMMON --full-out $OUT/$tag.full.json > $OUT/$tag.json 2> $OUT/$tag.err; rc=$?
  python - "$OUT/$tag.full.json" "$tag" $rc <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d.get("stage_kernel_ms_per_step", {})
    print("%-22s rc=%s %7.1f ms/step (min %.1f median %.1f max %.1f) allocs %s same_bytes %s binned %s ungapped %.1f sort %.1f seed %.1f ydrop %.1f hbm_read.frac %.4f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["step_ms_spread"]["min"], d["step_ms_spread"]["median"], d["step_ms_spread"]["max"], d.get("device_allocs_in_timed_steps"), d.get("parity", {}).get("same_bytes"), d.get("strands_grouped_in_lds_per_step"), k.get("ungapped", 0), k.get("sort", 0), k.get("seed_fill", 0), k.get("ydrop", 0), (d.get("hbm_read") or {}).get("frac", 0)))
except Exception as e:
    print(sys.argv[2], "rc=" + sys.argv[3], "unreadable:", e)
PY
}
run hm --workload hm --steps 10 --warmup 3
run chr20 --workload chr20 --steps 10 --warmup 3
run hm30 --workload hm30 --steps 5 --warmup 2
( time timeout 600 python bench.py --chunk-legs 2 --full-out $OUT/bench_full.json ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? line bytes=$(tail -1 $OUT/bench.json | wc -c)"; tail -4 $OUT/bench.err
python scripts/bench_summary.py $OUT/bench.json 2>&1 | cut -c1-400

#!/bin/bash
# round 6, call L: the new switch variants of the dense seed path test; the chunk-scale workloads after taking the seed-stage marks back
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6l; mkdir -p $OUT; rm -f $OUT/*
( timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "dense_seed_path" --timeout 600 ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
COMMON="--cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
run() { tag=$1; shift; timeout 400 python bench.py "$@" $COMMON --full-out $OUT/$tag.full.json > $OUT/$tag.json 2> $OUT/$tag.err; rc=$?
  python - "$OUT/$tag.full.json" "$tag" $rc <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d.get("stage_kernel_ms_per_step", {})
    print("%-22s rc=%s %7.1f ms/step (min %.1f median %.1f max %.1f) allocs %s same_bytes %s ungapped %.1f sort %.1f seed %.1f ydrop %.1f hbm_read.frac %.4f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["step_ms_spread"]["min"], d["step_ms_spread"]["median"], d["step_ms_spread"]["max"], d.get("device_allocs_in_timed_steps"), d.get("parity", {}).get("same_bytes"), k.get("ungapped", 0), k.get("sort", 0), k.get("seed_fill", 0), k.get("ydrop", 0), (d.get("hbm_read") or {}).get("frac", 0)))
except Exception as e:
    print(sys.argv[2], "rc=" + sys.argv[3], "unreadable:", e)
PY
}
run hm --workload hm --steps 10 --warmup 3
run chr20 --workload chr20 --steps 10 --warmup 3
run hm30 --workload hm30 --steps 5 --warmup 2
run hm_again --workload hm --steps 10 --warmup 3
run chr20_again --workload chr20 --steps 10 --warmup 3

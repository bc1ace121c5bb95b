#!/bin/bash
# round 6, call J: extent[] slots in the order of the keys (q batches): tests that force batches, hm30 A/B; the allocations and releases left in chr20's timed steps
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6j; mkdir -p $OUT; rm -f $OUT/*
( timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "dense_seed_path or full_size_chunk or grouping_by_diagonal or strand_halves or fuzz" --timeout 600 ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
( timeout 600 python -m pytest tests/test_bench_gpu.py -x -q -k "reference_chunk_size" --timeout 500 ) > $OUT/pytest_hm30.log 2>&1; echo "pytest hm30 rc=$?"; tail -2 $OUT/pytest_hm30.log
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
MIBLAST_EXTENT_SCRAMBLE=1 run hm30_ext1 --workload hm30 --steps 5 --warmup 2
MIBLAST_EXTENT_SCRAMBLE=0 run hm30_ext0 --workload hm30 --steps 5 --warmup 2
MIBLAST_DEBUG_ALLOC=2 MIBLAST_BENCH_STEP_TIMES=1 run chr20_allocs --workload chr20 --steps 6 --warmup 3
grep -a "device allocation\|step times" $OUT/chr20_allocs.err | tail -24 | cut -c1-170

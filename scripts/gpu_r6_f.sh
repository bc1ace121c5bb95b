#!/bin/bash
# round 6, call F: the GPU suite on the build with level 1 of k_ux_extend from the packed strands, then A/B of the chunk-scale workloads
# (MIBLAST_UX_PACKED=0: windows from the code bytes as before), whole pairs against strand halves at N = 1.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6f; mkdir -p $OUT; rm -f $OUT/*
( time timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
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
for pk in 1 0; do
  export MIBLAST_UX_PACKED=$pk
  run hm_pk$pk --workload hm --steps 10 --warmup 3
  run hm30_pk$pk --workload hm30 --steps 5 --warmup 2
  run chr20_pk$pk --workload chr20 --steps 10 --warmup 3
done
export MIBLAST_UX_PACKED=1
run hm30_split --workload hm30 --steps 5 --warmup 2 --split-strands 1
run chr20_split --workload chr20 --steps 10 --warmup 3 --split-strands 1
run seedleg --workload pair --random-pair --size 8000000 --steps 3 --warmup 1

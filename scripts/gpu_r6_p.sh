#!/bin/bash
# round 6, call P: lanes of a pipelined call on the human-mouse stand-in with the packed windows (MIBLAST_PIPELINE_LANES), and the mean bin
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6p; mkdir -p $OUT; rm -f $OUT/*
COMMON="--cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
run() { tag=$1; shift; timeout 400 python bench.py "$@" $COMMON --full-out $OUT/$tag.full.json > $OUT/$tag.json 2> $OUT/$tag.err; rc=$?
  python - "$OUT/$tag.full.json" "$tag" $rc <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d.get("stage_kernel_ms_per_step", {})
    print("%-22s rc=%s %7.1f ms/step (min %.1f median %.1f max %.1f) same_bytes %s ungapped %.1f sort %.1f seed %.1f ydrop %.1f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["step_ms_spread"]["min"], d["step_ms_spread"]["median"], d["step_ms_spread"]["max"], d.get("parity", {}).get("same_bytes"), k.get("ungapped", 0), k.get("sort", 0), k.get("seed_fill", 0), k.get("ydrop", 0)))
except Exception as e:
    print(sys.argv[2], "rc=" + sys.argv[3], "unreadable:", e)
PY
}
for L in 3 4 5 6 8 10; do MIBLAST_PIPELINE_LANES=$L run hm_lanes$L --workload hm --steps 8 --warmup 3; done
for M in 5500 8000 16000; do MIBLAST_BIN_MEAN=$M run hm_mean$M --workload hm --steps 8 --warmup 3; done

#!/bin/bash
# round 6, call Z4: the relay regime of the phase's gapped stages (piece length, warm-up rows, contexts in flight) against the code as it is now -- the
# values are round 3's and round 5's; headline workload only, 20 timed steps each, same box
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6z4; mkdir -p $OUT; rm -f $OUT/*
COMMON="--cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --chunk-legs 0 --primates-leg 0 --pair-leg 0 --steps 20 --warmup 4"
run() { tag=$1; timeout 300 python bench.py $COMMON --full-out $OUT/$tag.full.json > $OUT/$tag.json 2> $OUT/$tag.err; rc=$?
  python - "$OUT/$tag.full.json" "$tag" $rc <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); k = d.get("stage_kernel_ms_per_step", {}); r = d.get("relay", {})
    print("%-18s rc=%s %6.2f ms/step (min %.1f median %.1f max %.1f) spec %.2f launches %.0f pieces %.0f ydrop %.2f busy %.2f ungapped %.2f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["step_ms_spread"]["min"], d["step_ms_spread"]["median"], d["step_ms_spread"]["max"], d.get("speculation_factor", 0), r.get("dp_launches_per_step", 0), r.get("pieces_per_step", 0), k.get("ydrop", 0), k.get("ydrop_busy", 0), k.get("ungapped", 0)))
except Exception as e:
    print(sys.argv[2], "rc=" + sys.argv[3], "unreadable:", e)
PY
}
run default_a
MIBLAST_RELAY_S_MID=512 run s_mid_512
MIBLAST_RELAY_S_MID=640 run s_mid_640
MIBLAST_RELAY_S_MID=1024 run s_mid_1024
MIBLAST_RELAY_W_MID=96 run w_mid_96
MIBLAST_RELAY_W_MID=192 run w_mid_192
MIBLAST_RELAY_S_FEW=512 run s_few_512
MIBLAST_RELAY_S_FEW=768 run s_few_768
MIBLAST_BENCH_CONTEXTS=6 run contexts_6
MIBLAST_BENCH_CONTEXTS=3 run contexts_3
MIBLAST_BENCH_BACKGROUND=1 run background_1
MIBLAST_GAPPED_LANES=3 run gapped_lanes_3
MIBLAST_GAPPED_LANES=1 run gapped_lanes_1
run default_b

#!/bin/bash
# round 6, call Z2: the human-mouse stand-in on ONE lane (every kernel alone on the GPU) under rocprofv3 --kernel-trace --stats: the sum of the
# kernels' alone-times is the floor of the step at any number of lanes
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6z2; mkdir -p $OUT; rm -rf $OUT/*
export TMPDIR=/tmp
COMMON="--cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
MIBLAST_PIPELINE_LANES=1 timeout 600 python bench.py --workload hm --steps 3 --warmup 2 $COMMON --full-out $OUT/lanes1.full.json > $OUT/lanes1.json 2> $OUT/lanes1.err
MIBLAST_PIPELINE_LANES=2 timeout 600 python bench.py --workload hm --steps 3 --warmup 2 $COMMON --full-out $OUT/lanes2.full.json > $OUT/lanes2.json 2> $OUT/lanes2.err
MIBLAST_PIPELINE_LANES=1 timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o hm1 -- python bench.py --workload hm --steps 3 --warmup 2 $COMMON --full-out $OUT/lanes1_prof.full.json > $OUT/lanes1_prof.json 2> $OUT/lanes1_prof.err
python - <<'PY'
import json, glob
for t in ("lanes1", "lanes2", "lanes1_prof"):
    try:
        d = json.load(open("gpurun_out/r6z2/%s.full.json" % t)); k = d["stage_kernel_ms_per_step"]
        print(t, "%.1f ms/step" % d["ms_per_step"], {a: round(b, 1) for a, b in k.items()}, d["parity"]["same_bytes"])
    except Exception as e: print(t, "unreadable", e)
PY
f=$(ls $OUT/prof/*/*kernel_stats.csv $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); echo $f; cp $f $OUT/hm_lanes1_kernel_stats.csv; rm -rf $OUT/prof
head -22 $OUT/hm_lanes1_kernel_stats.csv | cut -c1-60,200-400 | cut -c1-200

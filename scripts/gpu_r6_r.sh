#!/bin/bash
# round 6, call R: the default bench line's chunk legs with the arena estimate's factor per context (device_allocs_in_timed_steps, outlier steps)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6r; mkdir -p $OUT; rm -f $OUT/*
( timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "evolver or full_size or relay_handover or trace_arena or arena" --timeout 500 ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for rep in 1 2; do
MIBLAST_BENCH_STEP_TIMES=1 timeout 600 python bench.py --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --pair-leg 0 --primates-leg 0 --full-out $OUT/full$rep.json > $OUT/line$rep.json 2> $OUT/err$rep.txt
grep -a "step times" $OUT/err$rep.txt
python - $OUT/full$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("headline %.2f ms allocs %s" % (d["ms_per_step"], d["device_allocs_in_timed_steps"]))
for k in ("chr20", "hm", "hm30"):
    print("%-6s %.1f ms (median %.1f max %.1f) allocs %s same_bytes %s" % (k, d[k]["ms_per_step"], d[k]["step_ms_spread"]["median"], d[k]["step_ms_spread"]["max"], d[k]["device_allocs_in_timed_steps"], d[k]["parity"]["same_bytes"]))
PY
done

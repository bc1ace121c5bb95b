#!/bin/bash
# round 6, call T: hm30 with the q-batch buffers sized once at a lane's start: outlier steps?
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6t; mkdir -p $OUT; rm -f $OUT/*
( timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "dense_seed_path or full_size_chunk" --timeout 500 ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for rep in 1 2 3 4; do
timeout 400 python bench.py --workload hm30 --steps 8 --warmup 2 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --full-out $OUT/hm30_$rep.json > /dev/null 2> /dev/null
python - $OUT/hm30_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("hm30 alone: %.1f ms (min %.1f median %.1f max %.1f) allocs %s same_bytes %s" % (d["ms_per_step"], d["step_ms_spread"]["min"], d["step_ms_spread"]["median"], d["step_ms_spread"]["max"], d["device_allocs_in_timed_steps"], d["parity"]["same_bytes"]))
PY
done
MIBLAST_BENCH_STEP_TIMES=1 timeout 600 python bench.py --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --pair-leg 0 --primates-leg 0 --full-out $OUT/full.json > $OUT/line.json 2> $OUT/err.txt
python - $OUT/full.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("bench line: headline %.2f ms allocs %s" % (d["ms_per_step"], d["device_allocs_in_timed_steps"]))
for k in ("chr20", "hm", "hm30"):
    print("   %-6s %.1f ms (min %.1f median %.1f max %.1f) allocs %s same_bytes %s" % (k, d[k]["ms_per_step"], d[k]["step_ms_spread"]["min"], d[k]["step_ms_spread"]["median"], d[k]["step_ms_spread"]["max"], d[k]["device_allocs_in_timed_steps"], d[k]["parity"]["same_bytes"]))
PY

#!/bin/bash
# round 6, call S: trace arenas sized by what stages used + reserved at the start of a pipelined call: tests, then the bench line's chunk legs twice and hm alone twice
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6s; mkdir -p $OUT; rm -f $OUT/*
( timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_multi_gpu.py -x -q -k "evolver or full_size or relay_handover or arena or fuzz or blocked or chunk" --timeout 600 ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for rep in 1 2; do
MIBLAST_BENCH_STEP_TIMES=1 timeout 600 python bench.py --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --pair-leg 0 --primates-leg 0 --full-out $OUT/full$rep.json > $OUT/line$rep.json 2> $OUT/err$rep.txt
python - $OUT/full$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("bench line: headline %.2f ms allocs %s" % (d["ms_per_step"], d["device_allocs_in_timed_steps"]))
for k in ("chr20", "hm", "hm30"):
    print("   %-6s %.1f ms (min %.1f median %.1f max %.1f) allocs %s same_bytes %s" % (k, d[k]["ms_per_step"], d[k]["step_ms_spread"]["min"], d[k]["step_ms_spread"]["median"], d[k]["step_ms_spread"]["max"], d[k]["device_allocs_in_timed_steps"], d[k]["parity"]["same_bytes"]))
PY
done
for rep in 1 2 3; do
timeout 400 python bench.py --workload hm --steps 10 --warmup 3 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --full-out $OUT/hm$rep.json > /dev/null 2> /dev/null
python - $OUT/hm$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("hm alone: %.1f ms (min %.1f median %.1f max %.1f) allocs %s same_bytes %s" % (d["ms_per_step"], d["step_ms_spread"]["min"], d["step_ms_spread"]["median"], d["step_ms_spread"]["max"], d["device_allocs_in_timed_steps"], d["parity"]["same_bytes"]))
PY
done

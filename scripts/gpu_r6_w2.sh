#!/bin/bash
# round 6, call W2: the driver's launch line with N ranks on ONE GPU over gloo (N = the arguments; the per-process pools of trace arenas cut so that N processes fit one device)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6w2; mkdir -p $OUT; rm -f $OUT/*
for N in "$@"; do
( export MIBLAST_BENCH_SINGLE_DEVICE=1 MIBLAST_BENCH_BACKEND=gloo MIBLAST_ARENA_POOL_MB=$((65536 / N)) MIBLAST_ARENA_RESERVE_MB=$((32768 / N))
  time timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) bench.py --gpus $N --steps 3 --warmup 2 --full-out $OUT/full_$N.json ) > $OUT/line_$N.json 2> $OUT/err_$N.txt; rc=$?
echo "N=$N rc=$rc lines=$(grep -c '^{' $OUT/line_$N.json) bytes=$(grep '^{' $OUT/line_$N.json | tail -1 | wc -c)"; grep real $OUT/err_$N.txt | tail -1
python - $OUT/full_$N.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("   n_gpus %s value %.1f ms %.2f scaling %s backend %s" % (d["n_gpus"], d["value"], d["ms_per_step"], d["scaling"], d["config"]["collective_backend"]))
    for k in ("chr20", "hm", "hm30"):
        if k in d: print("   %-6s n_gpus %s %.1f ms units/rank %s unit %s same_bytes %s" % (k, d[k]["n_gpus"], d[k]["ms_per_step"], d[k]["units_per_rank"], d[k]["work_unit"], d[k]["parity"]["same_bytes"]))
except Exception as e:
    print("   unreadable:", e)
PY
grep -n "Error\|out of memory" $OUT/err_$N.txt | grep -v "error_file" | head -5 | cut -c1-300
done

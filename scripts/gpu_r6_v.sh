#!/bin/bash
# round 6, call V: the default bench line with 2, 3 and 8 ranks on ONE GPU over gloo (the launch contract of the driver's scaling runs: does every leg --
# hm30's two chunk pairs over eight ranks included -- come through, is the line still one compact line?)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6v; mkdir -p $OUT; rm -f $OUT/*
export MIBLAST_BENCH_SINGLE_DEVICE=1 MIBLAST_BENCH_BACKEND=gloo
for N in 2 8; do
  ( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) bench.py --gpus $N --steps 2 --warmup 1 --full-out $OUT/full_$N.json ) > $OUT/line_$N.json 2> $OUT/err_$N.txt; rc=$?
  echo "N=$N rc=$rc lines=$(grep -c '^{' $OUT/line_$N.json) bytes=$(grep '^{' $OUT/line_$N.json | tail -1 | wc -c)"; grep real $OUT/err_$N.txt | tail -1
  python - $OUT/full_$N.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("   n_gpus %s value %.1f ms %.2f scaling %s backend %s" % (d["n_gpus"], d["value"], d["ms_per_step"], d["scaling"], d["config"]["collective_backend"]))
    for k in ("chr20", "hm", "hm30"):
        if k in d: print("   %-6s n_gpus %s %.1f ms units/rank %s same_bytes %s" % (k, d[k]["n_gpus"], d[k]["ms_per_step"], d[k]["units_per_rank"], d[k]["parity"]["same_bytes"]))
except Exception as e:
    print("   unreadable:", e)
PY
  tail -3 $OUT/err_$N.txt | cut -c1-300
done

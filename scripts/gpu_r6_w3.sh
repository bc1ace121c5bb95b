#!/bin/bash
# round 6, call W3: two ranks on one GPU, the hm workload: the library's own call times (MIBLAST_DEBUG=1) beside the step time -- what of a step at N > 1 is the harness
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6w3; mkdir -p $OUT; rm -f $OUT/*
N=2
( export MIBLAST_DEBUG=1 MIBLAST_BENCH_SINGLE_DEVICE=1 MIBLAST_BENCH_BACKEND=gloo MIBLAST_ARENA_POOL_MB=32768 MIBLAST_ARENA_RESERVE_MB=16384
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus $N --workload hm --steps 3 --warmup 2 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --full-out $OUT/full.json ) > $OUT/line.json 2> $OUT/err.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6w3/full.json")); print("hm at 2 ranks on one GPU: %.1f ms/step" % d["ms_per_step"], d["step_ms_spread"])
PY
grep "call of" $OUT/err.txt | tail -12 | cut -c1-200

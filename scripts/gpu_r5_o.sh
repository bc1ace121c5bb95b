#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5o; mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 16 --warmup 4 --chunk-legs 0 --primates-leg 1 --pair-leg 1 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  echo "== $name: $*"; python scripts/bench_summary.py gpurun_out/$TAG/$name.json | grep -E "ms/step|pair_1mb|primates" | cut -c1-200
}
run base MIBLAST_X=0
run gap32 MIBLAST_RELAY_GAP=32
run gap64 MIBLAST_RELAY_GAP=64
run gap32_tail16k MIBLAST_RELAY_GAP=32 MIBLAST_RELAY_TAIL_ROWS=16384
run tail16k MIBLAST_RELAY_TAIL_ROWS=16384
run gap32_end8 MIBLAST_RELAY_GAP=32 MIBLAST_RELAY_END_STEPS=8

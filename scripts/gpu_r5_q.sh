#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5q; mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --chunk-legs 0 --primates-leg 1 --pair-leg 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  echo "== $name: $*"; python scripts/bench_summary.py gpurun_out/$TAG/$name.json | grep -E "ms/step|primates" | cut -c1-200
}
run base MIBLAST_X=0
run w128 MIBLAST_RELAY_W_TINY=128
run w192 MIBLAST_RELAY_W_TINY=192
run w256 MIBLAST_RELAY_W_TINY=256
run w192_s384 MIBLAST_RELAY_W_TINY=192 MIBLAST_RELAY_S_TINY=384
run base2 MIBLAST_X=0

#!/bin/bash
# round 5: the regimes of calls with few sides (the trimmed levels of the phase) with the hand-over inside the launch
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5sweep2; mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 16 --warmup 4 --chunk-legs 0 --primates-leg 0 --pair-leg 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  echo "== $name: $*"; python scripts/bench_summary.py gpurun_out/$TAG/$name.json | head -2 | cut -c1-420
}
run A MIBLAST_X=0
run B MIBLAST_RELAY_S_TINY=256 MIBLAST_RELAY_W_TINY=192
run C MIBLAST_RELAY_S_TINY=320 MIBLAST_RELAY_W_TINY=256
run D MIBLAST_RELAY_S_TINY=256 MIBLAST_RELAY_W_TINY=128 MIBLAST_RELAY_S_FEW=384
run E MIBLAST_RELAY_S_TINY=192 MIBLAST_RELAY_W_TINY=192 MIBLAST_RELAY_S_FEW=320
run F MIBLAST_RELAY_S_FEW=448
run A2 MIBLAST_X=0
batch() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --chunk-legs 0 --primates-leg 0 --pair-leg 0 --batch-leg 16 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  echo "== $name: $*"; python scripts/bench_summary.py gpurun_out/$TAG/$name.json | grep batched | cut -c1-400
}
batch Bdef MIBLAST_X=0
batch Bs1024 MIBLAST_RELAY_S_CROWD=1024
batch Bs1536 MIBLAST_RELAY_S_CROWD=1536

#!/bin/bash
# round 5: relay regimes with the hand-over inside the launch (a rejected hand-over no longer costs a launch, so shorter pieces may pay).
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5sweep; mkdir -p gpurun_out/$TAG
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 12 --warmup 3 --chunk-legs 0 --primates-leg 0 --pair-leg 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  echo "== $name: $*"; python scripts/bench_summary.py gpurun_out/$TAG/$name.json | head -2
}
run A MIBLAST_X=0
run B MIBLAST_RELAY_S_MID=512
run C MIBLAST_RELAY_S_MID=384
run D MIBLAST_RELAY_S_MID=256
run E MIBLAST_RELAY_S_MID=384 MIBLAST_RELAY_S_FEW=320 MIBLAST_RELAY_S_TINY=256 MIBLAST_RELAY_W_TINY=128
run F MIBLAST_RELAY_S_MID=256 MIBLAST_RELAY_S_FEW=256 MIBLAST_RELAY_S_TINY=192 MIBLAST_RELAY_W_TINY=128
run G MIBLAST_RELAY_S_MID=384 MIBLAST_RELAY_S_FEW=320 MIBLAST_RELAY_S_TINY=256 MIBLAST_RELAY_W_TINY=96 MIBLAST_RELAY_W_MID=96
run H MIBLAST_RELAY_S_MID=384 MIBLAST_RELAY_S_FEW=320 MIBLAST_RELAY_S_TINY=256 MIBLAST_RELAY_W_TINY=128 MIBLAST_RELAY_END_STEPS=2
run I MIBLAST_RELAY_S_MID=512 MIBLAST_RELAY_S_FEW=384 MIBLAST_RELAY_S_TINY=256 MIBLAST_RELAY_W_TINY=192

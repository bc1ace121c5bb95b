#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5n; mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 16 --warmup 4 --chunk-legs 0 --primates-leg 1 --pair-leg 1 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  echo "== $name: $*"; python scripts/bench_summary.py gpurun_out/$TAG/$name.json | grep -E "ms/step|pair_1mb|primates" | cut -c1-200
}
run base MIBLAST_X=0
run few8 MIBLAST_HEAD_SPAN0=8192 MIBLAST_HEAD_FEW=8
run few16 MIBLAST_HEAD_SPAN0=8192 MIBLAST_HEAD_FEW=16
run few32 MIBLAST_HEAD_SPAN0=8192 MIBLAST_HEAD_FEW=32
run few16_16k MIBLAST_HEAD_SPAN0=16384 MIBLAST_HEAD_FEW=16
run few1000 MIBLAST_HEAD_SPAN0=8192 MIBLAST_HEAD_FEW=100000
MIBLAST_DEBUG=1 timeout 100 python bench.py --steps 1 --warmup 1 --chunk-legs 0 --primates-leg 1 --pair-leg 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/dbg.json 2> gpurun_out/$TAG/dbg.err

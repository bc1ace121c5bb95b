#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "== $*"; env "$@" timeout 600 python scripts/gpu_cfg4.py 2>&1 | grep "^rep 1\|max rows" | tail -4 | cut -c1-260; }
run X=1
run MIBLAST_PLANT_THREADS=0
run MIBLAST_CHAIN_HEADS=0
run MIBLAST_RELAY_CKPT=0
run MIBLAST_CHAIN_HEADS=0 MIBLAST_RELAY_CKPT=0 MIBLAST_PLANT_THREADS=0
echo "== debug"; MIBLAST_DEBUG=2 timeout 600 python scripts/gpu_cfg4.py 2>&1 | grep "round [0-9]*\.[0-9]*:\|long piece\|round [0-9]*:" | head -20 | cut -c1-260

#!/bin/bash
# round 6, call O: where the gapped stage of hm30's large pair spends its host time (MIBLAST_DEBUG timelines)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6o; mkdir -p $OUT; rm -f $OUT/*
MIBLAST_DEBUG=1 timeout 400 python bench.py --workload hm30 --steps 1 --warmup 2 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --full-out $OUT/full.json > $OUT/line.json 2> $OUT/debug.err
grep -a "host timeline\|round [0-9]*:\|PAF formatting\|output:\|call of\|seed phase\|traceback kernel\|merge:\|build_units\|anchors" $OUT/debug.err | tail -60 | cut -c1-260

#!/bin/bash
# round 6, call M: the DP launches of the headline phase one by one (MIBLAST_DEBUG: pieces, longest pieces, rows' percentiles, clocks per row)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6m; mkdir -p $OUT; rm -f $OUT/*
MIBLAST_DEBUG=1 timeout 300 python bench.py --steps 2 --warmup 2 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --pair-leg 0 --batch-leg 0 --primates-leg 0 --chunk-legs 0 --full-out $OUT/full.json > $OUT/line.json 2> $OUT/debug.err
grep -a "round \|longest \|rows: median\|first pass\|call of" $OUT/debug.err | tail -120 | cut -c1-230

#!/bin/bash
# round 6, call Z8: where a step of the (chunk pair, strand) units goes (MIBLAST_DEBUG=1 timelines), chr20 on one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6z8; mkdir -p $OUT; rm -f $OUT/*
COMMON="--cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
MIBLAST_DEBUG=1 MIBLAST_DEBUG_SPIKE=20 timeout 400 python bench.py --workload chr20 --steps 2 --warmup 2 --split-strands 1 $COMMON --full-out $OUT/split.full.json > $OUT/split.json 2> $OUT/split.err
grep -c . $OUT/split.err; grep "call of\|slow seed phase" $OUT/split.err | tail -30 | cut -c1-330

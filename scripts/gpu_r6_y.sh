#!/bin/bash
# round 6, call Y: --queryhspbest over a target in several blocks (the new tests), then the multi-device suite
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6y; mkdir -p $OUT; rm -f $OUT/*
( time timeout 900 python -m pytest tests/test_multi_gpu.py -x -q --timeout 600 ) > $OUT/pytest_multi.log 2>&1; echo "pytest multi rc=$?"; tail -15 $OUT/pytest_multi.log | cut -c1-300

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
MIBLAST_DEBUG=2 timeout 250 python scripts/gpu_cfg4.py 2>&1 | grep "build_units\|seed phase\|rep " | cut -c1-200 | tail -8

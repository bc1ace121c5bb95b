#!/bin/bash
# host-side phase times of the chaining stage (MIPAF_DEBUG) on the scale workload
MIPAF_DEBUG=1 timeout 150 python scripts/gpu_chain_bench.py 400 20000000 2>&1 | awk '/-- run 1/{p=1} p' | head -70

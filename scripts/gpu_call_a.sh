#!/bin/bash
# GPU call A of the chaining-stage bring-up: parity tests of the new kernels, then the 3- vs 4-waves-per-SIMD build of k_ydrop2
mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_zz_chain_gpu.py -x -q 2>&1 | tail -25 | tee gpurun_out/chain_tests.log
for P in 1 16; do
  for W in 4 3; do
    MIBLAST_DP_WAVES=$W timeout 120 python scripts/gpu_ab.py $P 8 2>&1 | tail -1 | tee -a gpurun_out/ab_waves.log
  done
done

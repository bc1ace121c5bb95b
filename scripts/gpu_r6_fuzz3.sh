#!/bin/bash
# round 6: chunk-scale differential cases (1.7 - 2.6 Mb pairs of every structure of scripts/gpu_fuzz.py, random option sets) against oracle digests made
# beforehand on the CPU box (FUZZ_WRITE_DIGESTS -> tests/golden/fuzz_chunk_r06.json: md5 of the PAF, twelve counters, record counts): the oracle's minutes
# stay off the GPU box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6fuzz3; : > gpurun_out/r6fuzz3/chunk.log
for seed in $(python -c "import json; print(' '.join(sorted(json.load(open('tests/golden/fuzz_chunk_r06.json')))))"); do
  FUZZ_NMIN=1700000 FUZZ_NMAX=2600000 FUZZ_READ_DIGESTS=tests/golden/fuzz_chunk_r06.json timeout 200 python scripts/gpu_fuzz.py 1 $seed >> gpurun_out/r6fuzz3/chunk.log 2>&1; echo "seed $seed rc=$?" >> gpurun_out/r6fuzz3/chunk.log
done
for seed in $(python -c "import json; print(' '.join(sorted(json.load(open('tests/golden/fuzz_chunk_r06_large.json')))))"); do      # (4 - 6 Mb pairs)
  FUZZ_NMIN=4000000 FUZZ_NMAX=6000000 FUZZ_READ_DIGESTS=tests/golden/fuzz_chunk_r06_large.json timeout 300 python scripts/gpu_fuzz.py 1 $seed >> gpurun_out/r6fuzz3/chunk.log 2>&1; echo "seed $seed rc=$?" >> gpurun_out/r6fuzz3/chunk.log
done
grep -c "rc=0" gpurun_out/r6fuzz3/chunk.log; grep "MISMATCH\|rc=[1-9]\|Error" gpurun_out/r6fuzz3/chunk.log | head

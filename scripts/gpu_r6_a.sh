#!/bin/bash
# round 6, call A: the GPU suite as the driver runs it, the default bench line (compact line + full file), the hm30 workload, then the parity /
# chaining / multi tests under the electric fence.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6a; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
( time timeout 600 python bench.py --full-out $OUT/bench_full.json ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? line bytes=$(tail -1 $OUT/bench.json | wc -c)"; tail -3 $OUT/bench.err
python scripts/bench_summary.py $OUT/bench.json 2>&1 | cut -c1-600
( time timeout 600 python bench.py --workload hm30 --steps 3 --warmup 2 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --full-out $OUT/hm30_full.json ) > $OUT/hm30.json 2> $OUT/hm30.err; echo "hm30 rc=$?"; tail -3 $OUT/hm30.err; cut -c1-1500 $OUT/hm30.json
bash scripts/gpu_r6_guard.sh 3 gpurun_out/r6a/fence 900 200 tests/test_parity_gpu.py tests/test_zz_chain_gpu.py tests/test_multi_gpu.py

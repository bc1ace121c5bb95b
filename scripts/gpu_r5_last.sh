#!/bin/bash
# round 5, last GPU seconds: the bins' tests and the hm leg once more after the plan's room was tied to the largest strand seen
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5last
( timeout 70 python -m pytest tests -m gpu -x -q -k "grouping_by_diagonal or (dense_seed_path and bin_mean)" ) > gpurun_out/r5last/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r5last/pytest.log
MIBLAST_DEBUG_ALLOC=1 MIBLAST_BENCH_STEP_TIMES=1 timeout 45 python bench.py --workload hm --steps 4 --warmup 2 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 > gpurun_out/r5last/hm.json 2> gpurun_out/r5last/hm.err; echo "bench rc=$?"
grep "step times\|slow" gpurun_out/r5last/hm.err | tail -8
python -c "import json;d=json.load(open('gpurun_out/r5last/hm.json'));print('hm',round(d['ms_per_step'],1),d['strands_grouped_in_lds_per_step'],d['parity']['same_bytes'])"

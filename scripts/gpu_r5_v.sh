#!/bin/bash
# round 5, after the bins: the full GPU suite + smoke + the driver's bench command, then the lanes of the chunk-scale workloads once more
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/gpu_r5_final.sh r5final3
for L in 4 8 10; do
  MIBLAST_PIPELINE_LANES=$L timeout 300 python bench.py --workload hm --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 > gpurun_out/r5final3/hm_lanes$L.json 2> gpurun_out/r5final3/hm_lanes$L.err
  python -c "import json;d=json.load(open('gpurun_out/r5final3/hm_lanes$L.json'));print('hm lanes=$L',round(d['ms_per_step'],2),d['step_ms_spread'] if 'step_ms_spread' in d else '')"
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02v
B="--steps 5 --warmup 2 --pair-leg 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0"
run() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py $B "$@" > gpurun_out/r02v/$label.json 2> gpurun_out/r02v/$label.err
  python - "$label" <<'PY'
import json,sys
l=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/r02v/{l}.json"))
    print(f"{l:26s} ms/step {d['ms_per_step']:8.2f}  value {d['value']:6.2f}  spec {d['speculation_factor']:.2f}  dp_ms {d['stage_kernel_ms_per_step']['ydrop']:6.2f}  launches {d['relay']['dp_launches_per_step']:5.1f}  pieces {d['relay']['pieces_per_step']:7.0f}  t_gapped {d['stage_seconds_per_step']['t_gapped']*1e3:6.2f} t_seed {d['stage_seconds_per_step']['t_seed']*1e3:6.2f}")
except Exception as e:
    print(l, "FAILED", e, open(f"gpurun_out/r02v/{l}.err").read()[-300:])
PY
}
run base X=1 -- --workload evolver
run base2 X=1 -- --workload evolver
run noplantthreads MIBLAST_PLANT_THREADS=0 -- --workload evolver
run gap8 MIBLAST_RELAY_GAP=8 MIBLAST_RELAY_TAIL_ROWS=4096 -- --workload evolver
run end1000 MIBLAST_RELAY_END_STEPS=100000 -- --workload evolver
run all_old MIBLAST_PLANT_THREADS=0 MIBLAST_RELAY_GAP=8 MIBLAST_RELAY_TAIL_ROWS=4096 MIBLAST_RELAY_END_STEPS=100000 -- --workload evolver
run base3 X=1 -- --workload evolver

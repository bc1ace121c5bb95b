#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02s2
B="--steps 5 --warmup 2 --pair-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0"
run() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py $B "$@" > gpurun_out/r02s2/$label.json 2> gpurun_out/r02s2/$label.err
  python - "$label" <<'PY'
import json,sys
l=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/r02s2/{l}.json"))
    print(f"{l:30s} ms/step {d['ms_per_step']:8.2f}  value {d['value']:6.2f}  spec {d['speculation_factor']:.2f}  dp_ms {d['stage_kernel_ms_per_step']['ydrop']:6.2f}  launches {d['relay']['dp_launches_per_step']:5.1f}  pieces {d['relay']['pieces_per_step']:7.0f}  rej {d['relay']['handovers_rejected_per_step']:5.0f} kernel Gc/s {d['gapped_gcells_per_s_kernel']:6.1f} t_gapped {d['stage_seconds_per_step']['t_gapped']*1e3:6.2f}")
except Exception as e:
    print(l, "FAILED", e, open(f"gpurun_out/r02s2/{l}.err").read()[-300:])
PY
}
run base X=1 -- --workload evolver
for S in 512 768 1024 1536; do for W in 128 192; do
  run many_S${S}_W${W} MIBLAST_RELAY_S=$S MIBLAST_RELAY_W=$W -- --workload evolver
done; done
for S in 512 768 1024; do
  run plant_S${S}_W128 MIBLAST_RELAY_PLANT_AT_ONCE=2 MIBLAST_RELAY_S0=64 MIBLAST_RELAY_S=$S MIBLAST_RELAY_W=128 -- --workload evolver
  run plant_S${S}_W128_sp4 MIBLAST_SPEC_TARGET=4 MIBLAST_RELAY_PLANT_AT_ONCE=2 MIBLAST_RELAY_S0=64 MIBLAST_RELAY_S=$S MIBLAST_RELAY_W=128 -- --workload evolver
done
run many_S768_W128_s0_128 MIBLAST_RELAY_S=768 MIBLAST_RELAY_W=128 MIBLAST_RELAY_S0=128 -- --workload evolver
run pair_base X=1 -- --workload pair
run pair_S512 MIBLAST_RELAY_S=512 -- --workload pair
run pair_S768 MIBLAST_RELAY_S=768 -- --workload pair
run pair_S640_W96 MIBLAST_RELAY_S=640 MIBLAST_RELAY_W=96 -- --workload pair
run pair_S640_W64 MIBLAST_RELAY_S=640 MIBLAST_RELAY_W=64 -- --workload pair

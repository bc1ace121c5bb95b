#!/bin/bash
# scheduling knobs of the gapped stage on the two bench workloads (results never depend on them; time does)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02s
B="--steps 5 --warmup 2 --pair-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0"
run() { # label, env..., -- args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py $B "$@" > gpurun_out/r02s/$label.json 2> gpurun_out/r02s/$label.err
  python - "$label" <<'PY'
import json,sys
l=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/r02s/{l}.json"))
    print(f"{l:28s} ms/step {d['ms_per_step']:8.2f}  value {d['value']:6.2f}  spec {d['speculation_factor']:.2f}  dp_ms {d['stage_kernel_ms_per_step']['ydrop']:6.2f}  launches {d['relay']['dp_launches_per_step']:5.1f}  pieces {d['relay']['pieces_per_step']:7.0f}  kernel Gc/s {d['gapped_gcells_per_s_kernel']:6.1f} t_gapped {d['stage_seconds_per_step']['t_gapped']*1e3:6.2f}")
except Exception as e:
    print(l, "FAILED", e, open(f"gpurun_out/r02s/{l}.err").read()[-300:])
PY
}
for wl in evolver pair; do
  run ${wl}_base X=1 -- --workload $wl
  for st in 2 6 12; do run ${wl}_spec$st MIBLAST_SPEC_TARGET=$st -- --workload $wl; done
  run ${wl}_plant1 MIBLAST_RELAY_PLANT_AT_ONCE=2 MIBLAST_RELAY_S0=64 MIBLAST_RELAY_S=640 MIBLAST_RELAY_W=128 -- --workload $wl
  run ${wl}_s1024 MIBLAST_RELAY_S=1024 MIBLAST_RELAY_W=128 -- --workload $wl
  run ${wl}_spec2_s640 MIBLAST_RELAY_PLANT_AT_ONCE=2 MIBLAST_SPEC_TARGET=2 MIBLAST_RELAY_S=640 MIBLAST_RELAY_W=128 MIBLAST_RELAY_S0=64 -- --workload $wl
  run ${wl}_spec6_s640 MIBLAST_RELAY_PLANT_AT_ONCE=2 MIBLAST_SPEC_TARGET=6 MIBLAST_RELAY_S=640 MIBLAST_RELAY_W=128 MIBLAST_RELAY_S0=64 -- --workload $wl
  run ${wl}_w4 MIBLAST_DP_WAVES=4 -- --workload $wl
  run ${wl}_w3 MIBLAST_DP_WAVES=3 -- --workload $wl
done

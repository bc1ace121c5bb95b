#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02y
run() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 8 --warmup 3 --chain-leg 0 --seed-leg 0 --cpu-sample 0 > gpurun_out/r02y/$label.json 2> gpurun_out/r02y/$label.err
  python - "$label" <<'PY'
import json,sys
l=sys.argv[1]
d=json.load(open(f"gpurun_out/r02y/{l}.json"))
print(f"{l:18s} evolver {d['ms_per_step']:7.2f} ms spec {d['speculation_factor']:.2f} dp {d['stage_kernel_ms_per_step']['ydrop']:6.2f} L {d['relay']['dp_launches_per_step']:4.1f} | pair {d['pair_1mb']['ms_per_step']:5.2f} | batched {d['batched_pairs']['ms_per_call']:6.1f} spec {d['batched_pairs']['speculation_factor']:.2f} dpk {d['batched_pairs']['gapped_gcells_per_s_kernel']:6.1f}")
PY
}
for rep in 1 2; do
run default_$rep X=1 --
run gap4096_$rep MIBLAST_GROUP_GAP=4096 --
run noend_$rep MIBLAST_RELAY_END_STEPS=1000000 --
run noplantthr_$rep MIBLAST_PLANT_THREADS=0 --
done

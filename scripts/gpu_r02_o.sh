#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02o
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r02o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02o/pytest.log
B="--steps 5 --warmup 2 --pair-leg 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0"
run() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py $B "$@" > gpurun_out/r02o/$label.json 2> gpurun_out/r02o/$label.err
  python - "$label" <<'PY'
import json,sys
l=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/r02o/{l}.json"))
    print(f"{l:26s} ms/step {d['ms_per_step']:8.2f}  value {d['value']:6.2f}  spec {d['speculation_factor']:.2f}  dp_ms {d['stage_kernel_ms_per_step']['ydrop']:6.2f}  launches {d['relay']['dp_launches_per_step']:5.1f}  pieces {d['relay']['pieces_per_step']:7.0f}  kernel Gc/s {d['gapped_gcells_per_s_kernel']:6.1f} t_gapped {d['stage_seconds_per_step']['t_gapped']*1e3:6.2f}")
except Exception as e:
    print(l, "FAILED", e, open(f"gpurun_out/r02o/{l}.err").read()[-300:])
PY
}
for wl in evolver pair; do
run ${wl}_old MIBLAST_CHAIN_HEADS=0 -- --workload $wl
run ${wl}_g4096_t64 X=1 -- --workload $wl
run ${wl}_g1024_t64 MIBLAST_GROUP_GAP=1024 -- --workload $wl
run ${wl}_g16384_t64 MIBLAST_GROUP_GAP=16384 -- --workload $wl
run ${wl}_g4096_t32 MIBLAST_GROUP_TOL=32 -- --workload $wl
run ${wl}_g4096_t160 MIBLAST_GROUP_TOL=160 -- --workload $wl
run ${wl}_g16384_t330 MIBLAST_GROUP_GAP=16384 MIBLAST_GROUP_TOL=330 -- --workload $wl
done

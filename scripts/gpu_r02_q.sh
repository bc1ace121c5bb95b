#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02q
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r02q/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02q/pytest.log
B="--steps 5 --warmup 2 --pair-leg 0 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0"
run() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py $B "$@" > gpurun_out/r02q/$label.json 2> gpurun_out/r02q/$label.err
  python - "$label" <<'PY'
import json,sys
l=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/r02q/{l}.json"))
    print(f"{l:26s} ms/step {d['ms_per_step']:8.2f}  value {d['value']:6.2f}  spec {d['speculation_factor']:.2f}  dp_ms {d['stage_kernel_ms_per_step']['ydrop']:6.2f}  launches {d['relay']['dp_launches_per_step']:5.1f}  pieces {d['relay']['pieces_per_step']:7.0f}  kernel Gc/s {d['gapped_gcells_per_s_kernel']:6.1f} t_gapped {d['stage_seconds_per_step']['t_gapped']*1e3:6.2f}")
except Exception as e:
    print(l, "FAILED", e, open(f"gpurun_out/r02q/{l}.err").read()[-300:])
PY
}
for wl in evolver pair; do
run ${wl}_old MIBLAST_CHAIN_HEADS=0 -- --workload $wl
run ${wl}_new X=1 -- --workload $wl
run ${wl}_new_g8192 MIBLAST_GROUP_GAP=8192 -- --workload $wl
run ${wl}_new_t100 MIBLAST_GROUP_TOL=100 -- --workload $wl
done
MIBLAST_DEBUG=1 timeout 300 python bench.py --workload pair --steps 1 --warmup 1 --batch-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 2>&1 >/dev/null | grep "round [0-9]*:" | tail -3 | cut -c1-210

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02g
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r02g/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02g/pytest.log
for ck in 1 0; do
MIBLAST_RELAY_CKPT=$ck timeout 300 python bench.py --steps 5 --warmup 2 --pair-leg 1 --chain-leg 0 --seed-leg 0 --cpu-sample 0 > gpurun_out/r02g/bench_$ck.json 2> gpurun_out/r02g/bench_$ck.err
python - $ck <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r02g/bench_{sys.argv[1]}.json"))
print("ckpt",sys.argv[1],"evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "dp", round(d["stage_kernel_ms_per_step"]["ydrop"],2), d["relay"], "kernel Gc/s", round(d["gapped_gcells_per_s_kernel"],1), "frac", d["roofline"]["frac"], "| pair ms", round(d["pair_1mb"]["ms_per_step"],2))
PY
done
MIBLAST_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 1 --pair-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 2>&1 >/dev/null | grep "round [0-9]*\.[0-9]*:" | tail -24 | cut -c1-150

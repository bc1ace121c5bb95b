#!/bin/bash
# session 2: DP kernel with sign-bit trace codes + edge specialisation: parity suite, then the bench legs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2g
( time timeout 600 python -m pytest tests -m gpu -x -q ) > gpurun_out/s2g/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s2g/pytest.log
MIBLAST_UNGAPPED=lane timeout 300 python bench.py --steps 16 --warmup 3 --chain-leg 0 --seed-leg 0 --cpu-sample 0 > gpurun_out/s2g/bench.json 2> gpurun_out/s2g/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/s2g/bench.json"))
print("evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "dp", round(d["stage_kernel_ms_per_step"]["ydrop"],2), "kernel Gc/s", round(d["gapped_gcells_per_s_kernel"],1), "frac", round(d["roofline"]["frac"],5), "valu", round(d["roofline"]["valu"]["frac"],4))
p=d["pair_1mb"]; print("pair ms", round(p["ms_per_step"],2), round(p["value"],2), round(p["gapped_gcells_per_s_kernel"],1), p["roofline"]["frac"], p["roofline"]["launch_ms"])
b=d["batched_pairs"]; print("batched", round(b["ms_per_call"],1), round(b["value"],1), round(b["gapped_gcells_per_s_kernel"],1), b["roofline"]["frac"], b["roofline"]["launch_ms"])
PY

#!/bin/bash
# session 2 baseline: GPU suite + default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2a
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/s2a/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s2a/pytest.log
timeout 600 python bench.py > gpurun_out/s2a/bench.json 2> gpurun_out/s2a/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/s2a/bench.json"))
print("evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "spec", round(d["speculation_factor"],2), "stage", d["stage_kernel_ms_per_step"], "kernel Gc/s", round(d["gapped_gcells_per_s_kernel"],1), "roof", d["roofline"], "same_bytes", d["cpu_baseline"]["same_bytes"])
p=d["pair_1mb"]; print("pair ms", round(p["ms_per_step"],2), round(p["value"],2), "spec", round(p["speculation_factor"],2), p["roofline"], p["cpu_baseline"]["same_bytes"])
b=d["batched_pairs"]; print("batched", round(b["ms_per_call"],1), round(b["value"],1), round(b["gapped_gcells_per_s_kernel"],1), b["roofline"]["frac"], b["speculation_factor"])
s=d["seed_stage"]; print("seed", s["kernel_ms"], round(s["seconds"]*1e3,1), round(s["frac"],4))
PY

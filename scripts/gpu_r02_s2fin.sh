#!/bin/bash
# session 2: the default bench line of the final build (-> profiles/r02_bench.json)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2fin
timeout 600 python bench.py > gpurun_out/s2fin/bench.json 2> gpurun_out/s2fin/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/s2fin/bench.json"))
print("evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "frac", round(d["roofline"]["frac"],5), "valu", round(d["roofline"]["valu"]["frac"],4), "traffic", d["roofline"]["traffic"], "launch_ms", d["roofline"]["launch_ms"], "same_bytes", d["cpu_baseline"]["same_bytes"])
p=d["pair_1mb"]; print("pair ms", round(p["ms_per_step"],2), round(p["value"],2), p["roofline"]["frac"], p["cpu_baseline"]["same_bytes"])
b=d["batched_pairs"]; print("batched", round(b["ms_per_call"],1), round(b["value"],1), round(b["gapped_gcells_per_s_kernel"],1), b["roofline"]["frac"])
s=d["seed_stage"]; print("seed", s["kernel_ms"], round(s["seconds"]*1e3,1), round(s["frac"],4)); c=d["chain_stage"]; print("chain", c["seconds"], c["cpu_baseline"]["same_bytes"])
PY

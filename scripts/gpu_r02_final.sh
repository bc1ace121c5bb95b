#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02final
for k in 1 2; do
timeout 600 python bench.py > gpurun_out/r02final/bench_$k.json 2> gpurun_out/r02final/bench_$k.err; echo "bench rc=$?"
python - $k <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r02final/bench_{sys.argv[1]}.json"))
print("evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "spec", round(d["speculation_factor"],2), "dp", round(d["stage_kernel_ms_per_step"]["ydrop"],2), "kernel Gc/s", round(d["gapped_gcells_per_s_kernel"],1), "frac", round(d["roofline"]["frac"],5), "valu", round(d["roofline"]["valu"]["frac"],4), "traffic", d["roofline"]["traffic"], "same_bytes", d["cpu_baseline"]["same_bytes"], "cpu", round(d["cpu_baseline"]["value"],4))
p=d["pair_1mb"]; print("pair ms", round(p["ms_per_step"],2), round(p["value"],2), "spec", round(p["speculation_factor"],2), p["roofline"], p["cpu_baseline"]["same_bytes"])
b=d["batched_pairs"]; print("batched", round(b["ms_per_call"],1), round(b["value"],1), round(b["gapped_gcells_per_s_kernel"],1), b["roofline"]["frac"], b["speculation_factor"])
s=d["seed_stage"]; print("seed", s["kernel_ms"], round(s["seconds"]*1e3,1), round(s["frac"],4))
PY
done

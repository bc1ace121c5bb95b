#!/bin/bash
# session 2: bench legs with the level-synchronous ungapped pipeline as the default for dense hit sets
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s2l
for mode in auto lane; do
env $( [ $mode = lane ] && echo MIBLAST_UNGAPPED=lane || echo A=0 ) timeout 300 python bench.py --steps 10 --warmup 3 --pair-leg 1 --batch-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/s2l/bench_$mode.json 2> gpurun_out/s2l/bench_$mode.err; echo "bench $mode rc=$?"
python - $mode <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/s2l/bench_{sys.argv[1]}.json"))
print(sys.argv[1], "evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), "stage", {k:round(v,2) for k,v in d["stage_kernel_ms_per_step"].items()})
p=d["pair_1mb"]; print("  pair ms", round(p["ms_per_step"],2), {k:round(v,2) for k,v in p["stage_kernel_ms_per_step"].items()})
s=d["seed_stage"]; print("  seed", {k:round(v,2) for k,v in s["kernel_ms"].items()}, round(s["seconds"]*1e3,1), round(s["frac"],4), s["seed_hits"], s["chance_alignments"])
PY
done

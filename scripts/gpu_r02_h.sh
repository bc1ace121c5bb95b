#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02h
MIBLAST_BENCH_TIMELINE=1 MIBLAST_DEBUG=1 timeout 300 python bench.py --steps 2 --warmup 2 --pair-leg 0 --chain-leg 0 --seed-leg 0 --cpu-sample 0 > gpurun_out/r02h/bench.json 2> gpurun_out/r02h/bench.err
grep -n "\[bench\]" gpurun_out/r02h/bench.err | tail -12
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02h/bench.json"))
print("evolver ms", round(d["ms_per_step"],2), "value", round(d["value"],2), d["stage_kernel_ms_per_step"], d["stage_seconds_per_step"], d["relay"])
PY
awk '/\[bench\] step total/{n++} n==3' gpurun_out/r02h/bench.err | grep -v "round [0-9]*\.[0-9]*\|first pass\|long piece" | head -90 | cut -c1-230

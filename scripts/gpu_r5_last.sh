#!/bin/bash
# round 5, last GPU seconds: the whole default line (without the CPU baseline's 25 s) on the final library and the final bench.py
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5last
timeout 50 python bench.py --cpu-sample 0 > gpurun_out/r5last/full2.json 2> gpurun_out/r5last/full2.err; echo "bench rc=$?"
python -c "import json;d=json.load(open('gpurun_out/r5last/full2.json'));print(round(d['ms_per_step'],2),'chr20',round(d['chr20']['ms_per_step'],1),d['chr20']['step_ms_spread'],d['chr20']['parity']['same_bytes'],'hm',round(d['hm']['ms_per_step'],1),d['hm']['step_ms_spread'],d['hm']['parity']['same_bytes'])"

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5j; mkdir -p gpurun_out/$TAG
one() { # name workload env...
  local name=$1 wl=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $wl --steps 4 --warmup 2 --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  python - "$name" gpurun_out/$TAG/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step %.1f"%d["ms_per_step"], "spec %.2f"%d.get("speculation_factor",0), "same_bytes", d.get("parity",{}).get("same_bytes"), {k:round(v,1) for k,v in d.get("stage_kernel_ms_per_step",{}).items() if isinstance(v,float)}, "launches", d.get("relay",{}).get("dp_launches_per_step"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
one chr20_a chr20 MIBLAST_X=0
one chr20_max16k chr20 MIBLAST_RELAY_MAX=16384
one chr20_max64k chr20 MIBLAST_RELAY_MAX=65536
one chr20_b chr20 MIBLAST_X=0
one chr20_max16k_s1536 chr20 MIBLAST_RELAY_MAX=16384 MIBLAST_RELAY_S_CROWD=1536
one chr20_lanes9 chr20 MIBLAST_PIPELINE_LANES=9

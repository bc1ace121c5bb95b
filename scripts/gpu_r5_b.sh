#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5b; mkdir -p gpurun_out/$TAG
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
MIBLAST_DEBUG=1 timeout 200 python bench.py --workload chr20 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/$TAG/chr20_dbg.json 2> gpurun_out/$TAG/chr20_dbg.err
one hm_g1 hm MIBLAST_PIPELINE_GROUP=1
one hm_g2 hm MIBLAST_PIPELINE_GROUP=2
one hm_g3 hm MIBLAST_PIPELINE_GROUP=3
one hm_g4 hm MIBLAST_PIPELINE_GROUP=4
one hm_g7 hm MIBLAST_PIPELINE_GROUP=7
one hm_g3_mid hm MIBLAST_PIPELINE_GROUP=3 MIBLAST_CROWD_SIDES=1000000
one hm_g3_l8 hm MIBLAST_PIPELINE_GROUP=3 MIBLAST_PIPELINE_LANES=8
one chr20_def chr20 MIBLAST_X=0
one chr20_mid chr20 MIBLAST_CROWD_SIDES=1000000
one chr20_s1024 chr20 MIBLAST_RELAY_S_CROWD=1024

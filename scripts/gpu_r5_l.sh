#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5l; mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 16 --warmup 4 --chunk-legs 0 --primates-leg 1 --pair-leg 1 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  echo "== $name: $*"; python scripts/bench_summary.py gpurun_out/$TAG/$name.json | grep -E "ms/step|relay:|pair_1mb|primates" | cut -c1-330
}
run off MIBLAST_HEAD_SPAN=0
run s16k MIBLAST_HEAD_SPAN=16384
run s8k MIBLAST_HEAD_SPAN=8192
run s32k MIBLAST_HEAD_SPAN=32768
run s16k_m4 MIBLAST_HEAD_SPAN=16384 MIBLAST_HEAD_MAX=4
run off2 MIBLAST_HEAD_SPAN=0
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
one chr20_off chr20 MIBLAST_HEAD_SPAN=0
one chr20_s16k chr20 MIBLAST_HEAD_SPAN=16384
one hm_off hm MIBLAST_HEAD_SPAN=0
one hm_s16k hm MIBLAST_HEAD_SPAN=16384

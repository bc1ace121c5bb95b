#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5m; mkdir -p gpurun_out/$TAG
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 16 --warmup 4 --chunk-legs 0 --primates-leg 1 --pair-leg 1 --batch-leg 0 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  echo "== $name: $*"; python scripts/bench_summary.py gpurun_out/$TAG/$name.json | grep -E "ms/step|pair_1mb|primates" | cut -c1-250
}
run old MIBLAST_HEAD_ROUNDS=0
run new16k MIBLAST_X=0
run new8k MIBLAST_HEAD_SPAN=8192
run new32k MIBLAST_HEAD_SPAN=32768
run new0 MIBLAST_HEAD_SPAN=0
run old2 MIBLAST_HEAD_ROUNDS=0
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
one chr20_warm chr20 MIBLAST_HEAD_ROUNDS=0
one chr20_old chr20 MIBLAST_HEAD_ROUNDS=0
one chr20_new chr20 MIBLAST_X=0
one hm_old hm MIBLAST_HEAD_ROUNDS=0
one hm_new hm MIBLAST_X=0

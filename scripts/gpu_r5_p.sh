#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5p; mkdir -p gpurun_out/$TAG
one() { # name workload env...
  local name=$1 wl=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  python - "$name" gpurun_out/$TAG/$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step %.1f"%d["ms_per_step"], d.get("step_ms_spread"), "spec %.2f"%d.get("speculation_factor",0), "same_bytes", d.get("parity",{}).get("same_bytes"), {k:round(v,1) for k,v in d.get("stage_kernel_ms_per_step",{}).items() if isinstance(v,float)}, "launches", d.get("relay",{}).get("dp_launches_per_step"))
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
}
one chr20_warm chr20 MIBLAST_X=0
one chr20_base chr20 MIBLAST_X=0
one chr20_gap32 chr20 MIBLAST_RELAY_GAP=32
one chr20_gap32e8 chr20 MIBLAST_RELAY_GAP=32 MIBLAST_RELAY_END_STEPS=8
one chr20_base2 chr20 MIBLAST_X=0
one hm_base hm MIBLAST_X=0
one hm_gap32 hm MIBLAST_RELAY_GAP=32
run() {
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 16 --warmup 4 --chunk-legs 0 --primates-leg 1 --pair-leg 1 --batch-leg 16 --seed-leg 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  echo "== $name: $*"; python scripts/bench_summary.py gpurun_out/$TAG/$name.json | grep -E "ms/step|pair_1mb|primates|batched" | cut -c1-200
}
run gap16 MIBLAST_RELAY_GAP=16
run gap32_e8 MIBLAST_RELAY_GAP=32 MIBLAST_RELAY_END_STEPS=8
run gap24_e8 MIBLAST_RELAY_GAP=24 MIBLAST_RELAY_END_STEPS=8
run base MIBLAST_X=0

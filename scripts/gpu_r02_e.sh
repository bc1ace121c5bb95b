#!/bin/bash
# does q-tiled processing (small q-ordered hit batches) make the ungapped stage cache-friendly?  seed leg only
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02e
for cap in 0 4000000 2000000 1000000 500000 250000; do
  if [ $cap = 0 ]; then unset MIBLAST_HIT_CAP MIBLAST_SEED_ONE_PASS; else export MIBLAST_HIT_CAP=$cap MIBLAST_SEED_ONE_PASS=0; fi
  timeout 300 python bench.py --workload pair --steps 1 --warmup 0 --chain-leg 0 --cpu-sample 0 > gpurun_out/r02e/b_$cap.json 2> gpurun_out/r02e/b_$cap.err
  python - $cap <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r02e/b_{sys.argv[1]}.json"))
s=d["seed_stage"]
print("hit_cap", sys.argv[1], "seed leg kernels", {k:round(v,2) for k,v in s["kernel_ms"].items()}, "t_seed ms", round(s["t_seed_s"]*1e3,1), "total ms", round(s["seconds"]*1e3,1), "frac", round(s["frac"],4))
PY
done

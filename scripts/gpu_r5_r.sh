#!/bin/bash
# round 5: grouping by diagonal in LDS (mb_seed_bin.h) -- its GPU tests, then A/B of the chunk-scale workloads against the radix sort
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r5r; mkdir -p $OUT
true
true
for W in hm chr20; do
  for B in 1 0; do
    MIBLAST_SORT_BIN=$B timeout 600 python bench.py --workload $W --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 > $OUT/${W}_bin$B.json 2> $OUT/${W}_bin$B.err
    python - <<PY
import json
d=json.load(open("$OUT/${W}_bin$B.json"))
print("$W bin=$B", round(d["ms_per_step"],2), "ms/step", d.get("stage_kernel_ms_per_step"), "binned", d.get("strands_grouped_in_lds_per_step"), d.get("parity",{}).get("same_bytes"))
PY
  done
done
MIBLAST_SORT_BIN=1 timeout 600 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --chain-leg 0 --batch-leg 0 --primates-leg 0 --chunk-legs 0 > $OUT/head_bin1.json 2> $OUT/head_bin1.err
python scripts/bench_summary.py $OUT/head_bin1.json | cut -c1-400 | head -12

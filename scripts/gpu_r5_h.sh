#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r5h; mkdir -p gpurun_out/$TAG
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; cd /tmp && export TMPDIR=/tmp
HEAD="python $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --pair-leg 0 --batch-leg 0 --primates-leg 0 --chunk-legs 0"
for V in 0 1; do
MIBLAST_RELAY_INLINE=$V rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write$V -- $HEAD > /dev/null 2> $OUT/pmc_write$V.log
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda:[0,0.0])
for p in glob.glob("$OUT/pmc_write$V/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        if row.get("Counter_Name")!="WRITE_SIZE": continue
        k=row["Kernel_Name"].split("(")[0]; agg[k][0]+=1; agg[k][1]+=float(row["Counter_Value"])
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:6]: print("inline=$V", k[-40:], v[0], "calls", round(v[1]*1024/1e6/4,1), "MB per step")
PY
done
find $OUT -type d -name "pmc_*" -exec rm -rf {} + 2>/dev/null

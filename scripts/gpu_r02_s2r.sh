#!/bin/bash
# session 2: WRITE_SIZE of the DP kernel, 4-wave vs 5-wave build
ROOT=$(pwd); OUT=$ROOT/gpurun_out/s2r; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PCMD="python $ROOT/bench.py --workload pair --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0"
for wv in 4 5; do
MIBLAST_DP_WAVES=$wv rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w$wv -- $PCMD > /dev/null 2> $OUT/w$wv.log
python - $OUT/w$wv $wv <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(lambda:[0,0.0])
for path in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k=row["Kernel_Name"].split("(")[0]
        if "ydrop" in k: agg[k][0]+=1; agg[k][1]+=float(row["Counter_Value"])
for k,v in agg.items(): print("waves",sys.argv[2],k,"rows",v[0],"sum KB",v[1],"per row MB",v[1]/1024/max(1,v[0]))
PY
done
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete

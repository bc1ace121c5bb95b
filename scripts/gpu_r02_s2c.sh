#!/bin/bash
# (MIBLAST_UNGAPPED_WAVES and the 8-wave build it selected were dropped after this run: 49 ms with spills)
# session 2: where does k_ungapped_grp's time go (8 Mb random pair)?  variants + one SQ counter pass
ROOT=$(pwd); OUT=$ROOT/gpurun_out/s2c; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/scripts/gpu_rand.py 8000000"
for v in "MIBLAST_UNGAPPED=lane" "MIBLAST_UNGAPPED=grp" "MIBLAST_UNGAPPED_WAVES=8" "MIBLAST_UNGAPPED_BLOCKS=1280" "MIBLAST_UNGAPPED_BLOCKS=16384" "MIBLAST_UNGAPPED_WAVES=8 MIBLAST_UNGAPPED_BLOCKS=2048"; do
  echo "== $v"; env $v $CMD 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('wall'):
        w=l.split()[1]; d=json.loads(l[l.index('{'):]); print('wall',w,{k:v for k,v in d.items() if 'ms' in k or 'ungapped' in k or 'hsp' in k})
"
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -- $CMD > /dev/null 2> $OUT/sq.log
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/sq2 -- $CMD > /dev/null 2> $OUT/sq2.log
cd $ROOT
python - $OUT <<'PY'
import csv,glob,collections,sys,os
OUT=sys.argv[1]
for d in ("sq","sq2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for path in glob.glob(f"{OUT}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k=row["Kernel_Name"].split("(")[0][:40]
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
    for k,v in agg.items():
        if "ungapped" in k: print(d,k,{a:round(b) for a,b in v.items()})
PY
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete

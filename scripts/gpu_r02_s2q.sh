#!/bin/bash
# session 2: SQ counters of the ux kernels on the 8 Mb random pair
ROOT=$(pwd); OUT=$ROOT/gpurun_out/s2q; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/scripts/gpu_rand.py 8000000"
MIBLAST_UNGAPPED=ux rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -- $CMD > /dev/null 2> $OUT/sq.log
MIBLAST_UNGAPPED=ux rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/sq2 -- $CMD > /dev/null 2> $OUT/sq2.log
cd $ROOT
python - $OUT <<'PY'
import csv,glob,collections,sys
OUT=sys.argv[1]
for d in ("sq","sq2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for path in glob.glob(f"{OUT}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k=row["Kernel_Name"].split("(")[0][:40]
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
    for k,v in agg.items():
        if "k_ux_extend" in k or "k_ux_accept" in k: print(d,k,{a:round(b) for a,b in v.items()})
PY
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete

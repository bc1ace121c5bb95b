#!/bin/bash
# where does k_ungapped's time go on the 8 Mb random pair?  SQ / TCC counters, separate passes
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/scripts/gpu_rand.py 8000000"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/plain.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -- $CMD > /dev/null 2> $OUT/sq.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/tcc -- $CMD > /dev/null 2> $OUT/tcc.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $CMD > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $CMD > /dev/null 2> $OUT/write.log
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/sq2 -- $CMD > /dev/null 2> $OUT/sq2.log
python - <<'PY'
import csv,glob,collections
out="/root/repo/gpurun_out/r02pmc" if False else __import__("os").environ.get("OUT","")
PY
cd $ROOT
python - $OUT <<'PY'
import csv,glob,collections,sys,os
OUT=sys.argv[1]
for d in ("sq","sq2","tcc","fetch","write"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
    for path in glob.glob(f"{OUT}/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k=row["Kernel_Name"].split("(")[0][:40]
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
    for k,v in agg.items():
        if "ungapped" in k or "seed_search" in k or "onesweep" in k or "run_heads" in k:
            print(d,k,{a:round(b) for a,b in v.items()})
f=glob.glob(f"{OUT}/stats/**/*kernel_stats.csv", recursive=True)
if f: print(open(f[0]).read()[:3000])
PY
tail -3 $OUT/plain.log
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*.db" -delete

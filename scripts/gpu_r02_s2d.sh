#!/bin/bash
# (the MIBLAST_UX_PROTO prototype kernel this run timed became level 1 of k_ux_extend and was removed)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/s2d; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/scripts/gpu_rand.py 8000000"
MIBLAST_UX_PROTO=1 MIBLAST_UNGAPPED=lane rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/plain.log 2>&1
tail -2 $OUT/plain.log | cut -c1-400
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-160
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete

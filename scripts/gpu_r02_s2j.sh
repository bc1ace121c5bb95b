#!/bin/bash
# session 2: what bounds k_ux_extend / k_ux_accept?  parts switched off one at a time (MIBLAST_UX_DBG; results are wrong when set)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/s2j; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/scripts/gpu_rand.py 8000000"
for dbg in 0 1 2 3 4 8 16 28; do
MIBLAST_UX_DBG=$dbg MIBLAST_UNGAPPED=ux rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d$dbg -- $CMD > $OUT/plain_$dbg.log 2>&1
f=$(find $OUT/d$dbg -name "*kernel_stats.csv" | head -1)
python - $f $dbg <<'PY'
import csv,sys
o=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].split('(')[0]
    if 'k_ux_extend' in n or 'k_ux_accept' in n or 'k_ux_tail' in n: o.append('%s %.1f us x%s' % (n.split('::')[-1], float(r['AverageNs'])/1e3, r['Calls']))
print('dbg', sys.argv[2], ' | '.join(o))
PY
done
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete

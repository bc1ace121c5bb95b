#!/bin/bash
# session 2: is k_ux_extend bound by the fabric?  per-hit time against the size of the pair (sequence footprint vs L2)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/s2p; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 1000000 2000000 4000000 8000000; do
MIBLAST_UNGAPPED=ux rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n$n -- python $ROOT/scripts/gpu_rand.py $n > $OUT/plain_$n.log 2>&1
f=$(find $OUT/n$n -name "*kernel_stats.csv" | head -1)
python - $f $n $OUT/plain_$n.log <<'PY'
import csv,sys,json,re
hits=None
for l in open(sys.argv[3]):
    if l.startswith('wall'): hits=json.loads(l[l.index('{'):])['seed_hits']
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].split('(')[0]
    if 'k_ux_extend' in n or 'k_ux_accept' in n or 'k_seed_search' in n or 'k_seed_fill' in n or 'k_seed_count' in n:
        tot=float(r['TotalDurationNs'])
        print(sys.argv[2], n.split('::')[-1], 'calls', r['Calls'], 'total ms %.3f' % (tot/1e6), 'ps per hit (2 aligns) %.1f' % (tot*1e3/(2*hits)))
PY
done
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete

#!/bin/bash
# session 2: the level-synchronous ungapped pipeline (ux): whole GPU suite with it forced, then timing + kernel breakdown on the 8 Mb random pair
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/s2i; rm -rf $OUT; mkdir -p $OUT
( time MIBLAST_UNGAPPED=ux timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "case_matches or deterministic or fuzz" ) > $OUT/pytest.log 2>&1; echo "pytest(ux forced) rc=$?"; tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/scripts/gpu_rand.py 8000000"
for v in "MIBLAST_UNGAPPED=lane" "MIBLAST_UNGAPPED=ux" "A=0"; do
  echo "== $v"; env $v $CMD 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('wall'):
        w=l.split()[1]; d=json.loads(l[l.index('{'):]); print('wall',w,{k:v for k,v in d.items() if 'ms' in k or 'ungapped' in k or 'hsp' in k or 'extended' in k})
"
done
MIBLAST_UNGAPPED=ux rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/plain.log 2>&1
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-60,200-330
find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete

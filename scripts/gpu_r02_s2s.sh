#!/bin/bash
# session 2: index-related parity tests + kernel stats of the phase after the scan kernels' rewrite
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/s2s; rm -rf $OUT; mkdir -p $OUT
( time timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "seed_index or case_matches or every_ungapped or fuzz or deterministic" ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --seed-leg 0 --chain-leg 0 --pair-leg 0 --batch-leg 0 > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv
python - $OUT/kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].split('(')[0]
    if 'scan' in n or 'block_sums' in n or 'index' in n: print(n[:40].ljust(40), r['Calls'].rjust(5), '%.2f ms' % (float(r['TotalDurationNs'])/1e6), '%.1f us' % (float(r['AverageNs'])/1e3))
PY
find $OUT/stats -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete

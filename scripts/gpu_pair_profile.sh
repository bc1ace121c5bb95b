#!/bin/bash
# kernel stats of single chunk pairs.  usage: gpurun -- 'bash scripts/gpu_pair_profile.sh <tag> [env assignments]'
TAG=${1:-pairprof}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
for spec in "chr20 0 0" "chr20 0 1" "hm 3 3"; do
  name=$(echo $spec | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python $ROOT/scripts/gpu_chunk_pair.py $spec 3 > $OUT/$name.log 2>&1
  find $OUT/$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/$name.kernel_stats.csv
  grep "^rep" $OUT/$name.log
  python - $OUT/$name.kernel_stats.csv <<'PY'
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('  total kernel ms over 3 reps', round(tot/1e6,2), 'launches', sum(int(r['Calls']) for r in rows))
for r in rows[:22]:
    n=r['Name']
    m=re.search(r'(radix_sort_\w+|merge_sort_\w+|scan\w*|mb::k_\w+(<[^>]*>)?|__amd_rocclr_\w+)',n)
    print(f"  {(m.group(1) if m else n[:50]):42s} calls {int(r['Calls']):4d} tot {int(r['TotalDurationNs'])/1e6:8.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete

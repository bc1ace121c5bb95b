#!/bin/bash
R=$(cd $(dirname $0)/.. && pwd); cd $R
for tgt in 16 128 512 2048; do
  for sz in 1000000 4000000; do
    echo -n "target=$tgt size=$sz: "
    MIBLAST_SPEC_TARGET=$tgt timeout 300 python bench.py --size $sz --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.1f  Gcell/s %.3f  spec %.1f  ydrop_ms %.1f launches/step %.1f' % (d['ms_per_step'], d['value'], d['speculation_factor'], d['stage_kernel_ms_per_step']['ydrop'], d['roofline']['launches_per_step']))"
  done
done

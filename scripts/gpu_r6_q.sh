#!/bin/bash
# round 6, call Q: which allocations / releases happen inside the timed steps of the default bench line's chunk legs (MIBLAST_DEBUG_ALLOC=2)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6q; mkdir -p $OUT; rm -f $OUT/*
MIBLAST_DEBUG_ALLOC=2 MIBLAST_BENCH_STEP_TIMES=1 timeout 600 python bench.py --cpu-sample 0 --seed-leg 0 --chain-leg 0 --batch-leg 0 --pair-leg 0 --primates-leg 0 --full-out $OUT/full.json > $OUT/line.json 2> $OUT/alloc.err
grep -a "step times" $OUT/alloc.err
python - "$OUT/alloc.err" <<'PY'
import re, sys
ev = []
for l in open(sys.argv[1], errors="replace"):
    m = re.match(r"\[miblast\]\s+([0-9.]+) s  device allocation:\s+([0-9.]+) MB\s+(.*)", l)
    if m: ev.append((float(m.group(1)), float(m.group(2)), m.group(3).strip()[:110]))
    elif "step times" in l: ev.append((None, None, l.strip()))
print(len(ev), "events")
# the events after the first 3 seconds, grouped by 0.25 s
last = None
for t, mb, what in ev:
    if t is None: print(what); continue
    if t < 2.0: continue
    print("%8.3f s %10.3f MB  %s" % (t, mb, what))
PY

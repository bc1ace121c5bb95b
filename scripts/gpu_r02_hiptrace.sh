#!/bin/bash
# HIP API trace of the phase bench: which runtime calls take milliseconds (host-side stalls of the round loop)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/hiptrace
timeout 600 rocprofv3 --hip-runtime-trace --output-format csv -d /tmp/hiptrace -- python bench.py --steps 20 --warmup 3 --chain-leg 0 --cpu-sample 0 --pair-leg 0 --batch-leg 0 --seed-leg 0 > gpurun_out/hiptrace/bench.log 2>&1
f=$(find /tmp/hiptrace -name "*hip_api_trace.csv" | head -1)
echo "trace: $f"; head -1 "$f"
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "calls")
dur = collections.defaultdict(list)
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    dur[r["Function"]].append(d)
print("%-36s %8s %10s %10s %10s" % ("function", "calls", "total ms", "max ms", ">2ms"))
for f, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print("%-36s %8d %10.2f %10.2f %10d" % (f, len(v), sum(v), max(v), sum(1 for x in v if x > 2)))
slow = sorted(rows, key=lambda r: int(r["Start_Timestamp"]) - int(r["End_Timestamp"]))[:40]
t0 = int(rows[0]["Start_Timestamp"])
print("slowest calls:")
for r in slow:
    print("  %-30s tid %s  start %.2f ms  dur %.2f ms" % (r["Function"], r["Thread_Id"], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
tail -2 gpurun_out/hiptrace/bench.log | cut -c1-300

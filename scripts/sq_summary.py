#!/usr/bin/env python3
"""Per-kernel means of SQ counters from a rocprofv3 --pmc run (counter_collection.csv): issue-slot accounting of the hot kernels.
usage: sq_summary.py "<glob of *counter_collection.csv>" "<note>"  -> JSON on stdout"""
import csv, glob, json, sys, collections

def main():
    files = glob.glob(sys.argv[1], recursive=True)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            per[name][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[name].add(r["Dispatch_Id"])
    out = {}
    for name, c in per.items():
        n = max(1, len(calls[name]))
        d = {"calls": n}
        for k, v in sorted(c.items()):
            d[k + "_per_call"] = v / n
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"):
                if k in c:
                    d[k + "_over_WAVE_CYCLES"] = c[k] / wc
        bc = c.get("SQ_BUSY_CYCLES", 0.0)
        if bc and "SQ_ACTIVE_INST_VALU" in c:
            d["note_valu"] = "SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = %.3f (both summed over the shader engines' SQs as the counter reports them)" % (c["SQ_ACTIVE_INST_VALU"] / bc)
        out[name] = d
    top = sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES_per_call", 0) * kv[1]["calls"])
    print(json.dumps({"source": sys.argv[1], "note": sys.argv[2] if len(sys.argv) > 2 else "", "kernels": dict(top[:24])}, indent=1))

main()

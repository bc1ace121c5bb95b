#!/usr/bin/env python3
"""One screen of a bench.py JSON line: `python scripts/bench_summary.py gpurun_out/x/bench.json`"""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
        if "roofline" in d and "legs" in d or d.get("full"):            # round 6: the line is the compact form; the full object lies in a file of its own
            import os
            for cand in (d.get("full"), os.path.join(os.path.dirname(path), "bench_full.json"), path.replace(".json", "_full.json")):
                if cand and os.path.exists(cand):
                    d = json.load(open(cand))
                    break
    except Exception as e:                                  # noqa: BLE001
        print(path, "unreadable:", e)
        continue
    r = d["roofline"]
    print(f"{path}: {d['n_gpus']} GPU, {d['ms_per_step']:.2f} ms/step (steps {d.get('step_ms_spread', {}).get('min', 0):.1f} / {d.get('step_ms_spread', {}).get('median', 0):.1f} / {d.get('step_ms_spread', {}).get('max', 0):.1f}), {d['value']:.2f} {d['unit']}, spec {d.get('speculation_factor', 0):.2f}, "
          f"kernel {d.get('gapped_gcells_per_s_kernel', 0):.0f} Gc/s, frac {r['frac']:.5f}, valu {r.get('valu', {}).get('frac', 0):.4f}")
    print("  stage kernel ms:", {k: round(v, 2) for k, v in d.get("stage_kernel_ms_per_step", {}).items()}, "relay:", {k: round(v, 2) for k, v in d.get("relay", {}).items()})
    print("  host:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("host", {}).items() if k != "note"})
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        print("  cpu_baseline:", {k: c[k] for k in c if k not in ("sample",)})
    for leg in ("pair_1mb", "batched_pairs", "seed_stage", "chain_stage", "primates", "chr20", "hm"):
        if leg in d:
            x = d[leg]
            print(f"  {leg}:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in x.items() if k not in ("workload", "bytes_note", "cpu_baseline", "note")},
                  ("same_bytes=%s" % x["cpu_baseline"].get("same_bytes")) if isinstance(x.get("cpu_baseline"), dict) else "")

"""Dump the per-kernel summary of a rocprofv3 (rocpd sqlite) result database as a small CSV (durations in us)."""
import re
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))


def short(n):
    if "merge_sort_block_merge" in n:
        return "rocprim::merge_sort_block_merge" + ("(partition)" if "partition_config" in n else "")
    if "radix_sort_block_sort" in n:
        return "rocprim::radix_sort_block_sort"
    if "rocprim" in n:
        m = re.search(r"(onesweep\w*|histogram\w*|lookback_scan\w*|radix_sort\w*)", n)
        return "rocprim::" + (m.group(1) if m else "kernel")
    m = re.match(r"(?:void )?([\w:]+(?:<[^(]{0,40}>)?)", n)
    return m.group(1) if m else n[:80]


with open(out, "w") as f:
    f.write("kernel,calls,total_us,avg_us,pct\n")
    for n, calls, tot, avg, pct in rows:
        f.write('"%s",%d,%.1f,%.2f,%.3f\n' % (short(n), calls, tot, avg, pct))
print(open(out).read())

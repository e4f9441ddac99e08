"""Dump rocprofv3 rocpd databases under a directory: top kernels of every trace db, mean counter values per kernel of
every pmc db (optionally only kernels matching a substring).

    python scripts/prof_dump.py gpurun_out/prof_x [kernel-substring]
"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(f).cursor()
    tables = {r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")}
    print(f"### {os.path.relpath(f, root)}")
    if "top_kernels" in tables:
        rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 14"))
        if rows:
            print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
            for name, calls, total, avg, pct in rows:
                print(f"| `{name[:90]}` | {calls} | {total / 1e3:.3f} | {avg:.1f} | {pct:.2f} |")  # rocpd's top_kernels view is in microseconds
    if "counters_collection" in tables:
        q = "select kernel_name,counter_name,count(*),avg(value),sum(value) from counters_collection where kernel_name like ? group by kernel_name,counter_name"
        last = None
        for kname, ctr, n, mean, total in cur.execute(q, (f"%{pat}%",)):
            if kname != last:
                print(f"- `{kname[:90]}`")
                last = kname
            print(f"    {ctr}: mean/launch {mean:.6g}, sum {total:.6g} over {n} launches")

#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration, share) of a rocprofv3 `--kernel-trace --stats` run whose
output is a rocpd SQLite database (rocprofv3 of ROCm 7 writes <name>_results.db instead of the CSV files of earlier releases).
    python tools/rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("""select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                     from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id and d.guid = s.guid
                     group by s.kernel_name order by 3 desc""").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
for name, calls, t, mn, mx in rows:
    lines.append('"%s",%d,%d,%.1f,%.4f,%d,%d' % (name.replace('"', "'"), calls, t, t / calls, 100.0 * t / tot, mn, mx))
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
else:
    sys.stdout.write(out)

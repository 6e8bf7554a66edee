#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration, share) of a rocprofv3 `--kernel-trace --stats` run whose
output is a rocpd SQLite database (rocprofv3 of ROCm 7 writes <name>_results.db instead of the CSV files of earlier releases).
    python tools/rocpd_stats.py results.db [out.csv] [--by-grid]
--by-grid: one line per (kernel, launch grid): a solver that runs one kernel on several problem sizes (the levels of a multigrid
cycle) shows up with its sizes apart — the largest grid of a kernel is the finest level."""
import sqlite3
import sys

by_grid = "--by-grid" in sys.argv
args = [a for a in sys.argv[1:] if a != "--by-grid"]
db = sqlite3.connect(args[0])
if by_grid:
    rows = db.execute("""select s.kernel_name || ' [grid ' || d.grid_size_x || 'x' || d.grid_size_y || 'x' || d.grid_size_z || ']', count(*),
                                sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                         from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id and d.guid = s.guid
                         group by s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z order by 3 desc""").fetchall()
else:
    rows = db.execute("""select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                         from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id and d.guid = s.guid
                         group by s.kernel_name order by 3 desc""").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
for name, calls, t, mn, mx in rows:
    lines.append('"%s",%d,%d,%.1f,%.4f,%d,%d' % (name.replace('"', "'"), calls, t, t / calls, 100.0 * t / tot, mn, mx))
out = "\n".join(lines) + "\n"
if len(args) > 1:
    open(args[1], "w").write(out)
else:
    sys.stdout.write(out)

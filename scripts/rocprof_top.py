"""Prints the kernel-stats table of a rocprofv3 (rocpd sqlite) result: python scripts/rocprof_top.py x_results.db [csv_out]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
lines = ["Name,Calls,TotalDurationUs,AverageUs,Percentage"]      # rocpd top_kernels reports microseconds
for name, calls, tot, avg, pct in rows:
    lines.append('"%s",%d,%.2f,%.3f,%.2f' % (name.replace('"', "'"), calls, tot, avg, pct))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
for name, calls, tot, avg, pct in rows[:40]:
    print("%-90s %6d %10.1f us avg %6.2f%%" % (name[:90], calls, avg, pct))

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

# per launch geometry for the step kernels (a kernel the bench launches at several row counts -- the strong-scaling projection --
# averages over all of them in the table above): python scripts/rocprof_top.py x.db out.csv by_grid_match
if len(sys.argv) > 3:
    print("\nby grid (kernels matching %r):" % sys.argv[3])
    q = ("select name, grid_x / max(workgroup_x, 1), count(*), avg(end - start) / 1e3, min(end - start) / 1e3, max(end - start) / 1e3 "
         "from kernels where name like ? group by name, grid_x / max(workgroup_x, 1) order by name, 2 desc")
    for name, wgs, n, avg, lo, hi in db.execute(q, ("%" + sys.argv[3] + "%",)):
        print("%-70s wgs %6d  launches %5d  avg %8.2f us  (min %.2f max %.2f)" % (name.replace("(anonymous namespace)::", "")[:70], wgs, n, avg, lo, hi))

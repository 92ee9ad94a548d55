"""Per-dispatch timeline between the last two launches of a marker kernel in a rocprofv3 rocpd database:
    python scripts/rocprof_timeline.py x_results.db [marker=adam_kernel] [occurrence=-1]
(occurrence k: the dispatches between the (k-1)-th and the k-th launch of the marker, Python indexing)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
try:          # the queue a dispatch ran on tells the streams of a multi-stream update apart
    rows = list(db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, "
                           "accum_vgpr_count, queue_id from kernels order by start"))
except sqlite3.OperationalError:
    rows = [r + (0,) for r in db.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, "
                                          "vgpr_count, accum_vgpr_count from kernels order by start")]
queues = {}
idx = [i for i, r in enumerate(rows) if marker in r[0]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else -1
lo, hi = idx[k - 1] + 1, idx[k] + 1
t0 = rows[lo][1]
busy = 0.0
for r in rows[lo:hi]:
    busy += (r[2] - r[1]) / 1e3
    name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    q = queues.setdefault(r[10], len(queues))
    print("q%d %-60s %7.1f us @%8.1f ..%8.1f  wgs (%d,%d,%d) lds %d vgpr %d+%d" % (
        q, name[:60], (r[2] - r[1]) / 1e3, (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, r[3] // max(r[6], 1), r[4], r[5], r[7], r[8], r[9]))
print("span %.1f us, kernel-busy %.1f us" % ((rows[hi - 1][2] - t0) / 1e3, busy))

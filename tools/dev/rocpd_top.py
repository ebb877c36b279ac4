"""Print the top kernels of a rocprofv3 rocpd database (run on the GPU box when the db is too big to copy back)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc limit %d" % int(sys.argv[2] if len(sys.argv) > 2 else 25)).fetchall()
tot = sum(r[0] for r in db.execute("select total_duration from top_kernels")); n = sum(r[0] for r in db.execute("select total_calls from top_kernels"))
print("total kernel time %.1f ms in %d launches" % (tot / 1e3, n))      # (the view's durations are in microseconds)
for r in rows:
    print("%6d calls %9.1f ms tot %7.1f us avg %5.1f%%  %s" % (r[1], r[2] / 1e3, r[3], r[4], r[0].replace("(anonymous namespace)::", "")[:110]))

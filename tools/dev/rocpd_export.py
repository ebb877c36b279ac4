"""Summarise rocprofv3 rocpd databases (gpurun_out/...) into the small text files kept under profiles/.
usage: rocpd_export.py stats <db> <out.csv> | pmc <db> [<db> ...] <out.txt>"""
import csv, sqlite3, sys

def stats(db, out):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    mm = {r[0]: r[1:] for r in con.execute("select name, min(end-start), max(end-start) from kernels group by name")} \
        if con.execute("select count(*) from sqlite_master where name='kernels'").fetchone()[0] else {}
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, calls, tot, avg, pct in rows:
            short = name.replace("(anonymous namespace)::", "").split("(")[0][:160]
            mn, mx = mm.get(name, ("", ""))
            w.writerow([short, calls, int(tot * 1000) if tot < 1e9 else tot, round(avg * 1000, 1), round(pct, 4), mn, mx])

def pmc(dbs, out):
    lines = []
    for db in dbs:
        con = sqlite3.connect(db)
        for k, c, v, n, g in con.execute("select kernel_name, counter_name, avg(value), count(*), max(grid_size) from counters_collection "
                                         "group by kernel_name, counter_name"):
            if "step_kernel" in k:
                lines.append("%-44s %-16s avg/dispatch %14.1f  (n=%d, grid %d)" % (k.replace("(anonymous namespace)::", "").split("(")[0][-44:], c, v, n, g))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))

if __name__ == "__main__":
    if sys.argv[1] == "stats": stats(sys.argv[2], sys.argv[3])
    else: pmc(sys.argv[2:-1], sys.argv[-1])

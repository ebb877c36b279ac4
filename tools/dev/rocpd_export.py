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

def pmc(dbs, out, last=40):
    """Average per dispatch over the LAST `last` full-size dispatches of the step kernel: the create-time settle
    launches (grid of 8 workgroups) and the burn-in to the stationary regime are excluded (round 1 averaged them in:
    SQ_WAVES read 804.9 = (11 x 8 + 40 x 1024) / 51 for a 1024-wave grid)."""
    lines = []
    for db in dbs:
        con = sqlite3.connect(db)
        rows = con.execute("select kernel_name, counter_name, value, grid_size, dispatch_id from counters_collection order by dispatch_id").fetchall()
        acc = {}
        for k, c, v, g, d in rows:
            if "step_kernel" in k:
                acc.setdefault((k, c), []).append((g, d, v))
        for (k, c), lst in sorted(acc.items()):
            gmax = max(g for g, _, _ in lst)
            per = {}
            for g, d, v in lst:                      # a counter may come as several rows per dispatch (one per XCD / dimension): sum them
                if g == gmax:
                    per[d] = per.get(d, 0.0) + v
            vals = [per[d] for d in sorted(per)][-last:]
            lines.append("%-44s %-22s avg/dispatch %16.1f  (last %d of %d full-size dispatches, grid %d)" % (
                k.replace("(anonymous namespace)::", "").split("(")[0][-44:], c, sum(vals) / len(vals), len(vals), len(per), gmax))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))

if __name__ == "__main__":
    if sys.argv[1] == "stats": stats(sys.argv[2], sys.argv[3])
    else: pmc(sys.argv[2:-1], sys.argv[-1])

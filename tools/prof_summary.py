#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (<pid>_results.db) into the plain-text per-kernel
summary that is committed under profiles/ (same content as --stats)."""
import sqlite3, sys, glob, os

def main():
    src, dst = sys.argv[1], sys.argv[2]
    dbs = [src] if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*_results.db"), recursive=True))
    lines = []
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        lines.append("# source: %s" % os.path.basename(db))
        lines.append("%-86s %6s %12s %12s %12s %12s %6s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "vgpr", "grid"))
        rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(grid_x) "
                           "from kernels group by name order by sum(duration) desc").fetchall()
        for r in rows:
            lines.append("%-86s %6d %12.3f %12.2f %12.2f %12.2f %6d %6d" % (r[0][:86], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6], r[7]))
        try:
            pmc = cur.execute("select k.name, p.counter_name, avg(p.value), count(*) from pmc_events p join kernels k on p.event_id = k.event_id "
                              "group by k.name, p.counter_name").fetchall()
            if pmc:
                lines.append("# PMC counters (mean per dispatch)")
                for r in pmc:
                    lines.append("%-86s %-28s %18.1f  (n=%d)" % (r[0][:86], r[1], r[2], r[3]))
        except Exception as e:
            lines.append("# no PMC data (%s)" % e)
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace as the --stats table: per kernel
calls / total / average / min / max duration (microseconds) and share.  usage: rocpd_summary.py results.db [out.txt]"""
import sqlite3, sys
def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["# rocprofv3 --kernel-trace --stats summary (durations in us; source: %s)" % db,
             "%-110s %7s %12s %10s %10s %10s %6s %5s %5s %5s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "lds")]
    for n, k, s, a, mn, mx, vg, ag, sg, lds in rows:
        lines.append("%-110s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5s %5s %5s %7s" % (n[:110], k, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot, vg, ag, sg, lds))
    txt = "\n".join(lines) + "\n"
    if out: open(out, "w").write(txt)
    print(txt)
if __name__ == "__main__":
    main(*sys.argv[1:3])

#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (text/CSV)."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name, name)).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["%-90s %8s %14s %12s %12s %12s %7s" % ("Name", "Calls", "TotalNs", "AvgNs", "MinNs", "MaxNs", "Pct")]
    for n, cnt, tot, avg, mn, mx in rows:
        lines.append("%-90s %8d %14d %12.0f %12d %12d %6.2f%%" % (n[:90], cnt, tot, avg, mn, mx, 100.0 * tot / total))
    lines.append("TOTAL kernel ns: %d" % total)
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

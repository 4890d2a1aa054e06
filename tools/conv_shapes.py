#!/usr/bin/env python3
"""Group the launches of one kernel in a rocprofv3 rocpd database by grid size (which layer shapes carry its time)."""
import sqlite3
import sys


def main(db, pattern="conv_split_kernel"):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    grid = [k for k in cols if "grid" in k.lower()]
    wg = [k for k in cols if "workgroup" in k.lower() or "block" in k.lower()]
    keys = ", ".join(grid + wg)
    q = "select %s, count(*), sum(end-start), avg(end-start) from kernels where %s like ? group by %s order by 3 desc" % (keys, name, keys)
    rows = c.execute(q, ("%" + pattern + "%",)).fetchall()
    total = sum(r[-2] for r in rows) or 1
    print("columns:", grid + wg)
    for r in rows:
        print("%-40s calls %5d  total %10.1f us  avg %8.1f us  %5.1f%%" % (str(r[:-3]), r[-3], r[-2] / 1e3, r[-1] / 1e3, 100.0 * r[-2] / total))


if __name__ == "__main__":
    main(*sys.argv[1:])

#!/usr/bin/env python3
"""Timeline of the last kernel dispatches of a rocprofv3 --kernel-trace run (rocpd .db): start, duration, and the idle
time since the previous kernel on the device ended — where the time of a bench step goes that is not the scan kernel.

    python tools/gap_report.py <stats_results.db> [n_last]
"""
import sqlite3
import sys


def main(db, n_last=80):
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    src = "kernels" if "kernels" in names else next(n for n in names if "kernel_dispatch" in n)
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % src)]
    print("#", src, cols)
    name_col = "name" if "name" in cols else next(x for x in cols if "name" in x)
    extra = [x for x in ("stream_id", "queue_id", "stream") if x in cols]
    rows = c.execute("select %s, start, end %s from %s order by start" % (name_col, "".join(", " + e for e in extra), src)).fetchall()
    rows = rows[-n_last:]
    t0 = rows[0][1]
    prev_end = None
    per = {}
    for r in rows:
        nm = r[0].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0][:60]
        gap = (r[1] - prev_end) / 1e3 if prev_end is not None else 0.0
        print("%10.1f us  dur %8.1f  idle-before %7.1f  %s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, nm, r[3:] if extra else ""))
        prev_end = max(prev_end or 0, r[2])
        per.setdefault(nm, []).append((r[2] - r[1]) / 1e3)
    print("# span %.1f us over %d dispatches" % ((rows[-1][2] - t0) / 1e3, len(rows)))
    for k, v in per.items():
        print("# %-60s n=%d avg=%.1f us" % (k, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 80)

#!/usr/bin/env python3
"""Export the per-kernel summary of a rocprofv3 run (rocpd sqlite .db, the default output
format of rocprofv3 in ROCm 7.2) to a small CSV + markdown table for profiles/.

    python tools/rocpd_summary.py gpurun_out/<tag>_prof/stats_results.db profiles/<name>
"""
import csv
import sqlite3
import sys


def main(db, out_prefix, note=""):
    c = sqlite3.connect(db)
    cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    rows = cur.fetchall()
    with open(out_prefix + ".csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.4f" % r[4]])
    with open(out_prefix + ".md", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary\n\n%s\n\n" % note)
        f.write("| kernel | calls | total (us) | average (us) | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows:
            f.write("| `%s` | %d | %.1f | %.2f | %.2f |\n" % (r[0], r[1], r[2], r[3], r[4]))
    print("wrote", out_prefix + ".csv", out_prefix + ".md")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")

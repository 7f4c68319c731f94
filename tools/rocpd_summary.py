"""Summarise a rocprofv3 rocpd database (results.db) into the `--stats`-style per-kernel table.
usage: python tools/rocpd_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    lines = ["%-110s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, tot, avg, pct in rows:
        lines.append("%-110s %8d %14.3f %12.3f %7.2f" % (name[:110], calls, tot, avg, pct))
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    sys.stdout.write(txt)


main()

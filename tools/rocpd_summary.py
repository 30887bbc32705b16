#!/usr/bin/env python3
"""Summarises a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max.
Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/NAME.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-70s %6s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
for n, c, s, a, mn, mx in rows:
    print("%-70s %6d %12.3f %12.2f %12.2f %12.2f %6.2f" % (n[:70], c, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))

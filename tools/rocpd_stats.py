#!/usr/bin/env python3
"""Per-kernel statistics (count, mean / min / max duration in us) from a rocprofv3 rocpd SQLite file:  tools/rocpd_stats.py x_results.db [regex]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
rows = db.execute("select name, count(*), avg(end - start), min(end - start), max(end - start), sum(end - start) from kernels group by name order by 6 desc").fetchall()
for name, n, avg, mn, mx, tot in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short)[:110]
    if pat is None or pat.search(short):
        print(f"{n:6d} x {avg / 1e3:8.2f} us  (min {mn / 1e3:7.2f} max {mx / 1e3:7.2f})  total {tot / 1e6:8.3f} ms  {short}")

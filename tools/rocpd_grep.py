"""List kernels of a rocprofv3 rocpd sqlite file whose name matches a regex: calls / total / avg / min / max.
usage: python tools/rocpd_grep.py results.db 'nccl|rccl'"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pat = re.compile(sys.argv[2], re.I)
rows = c.execute("select name, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
for name, n, tot, mn, mx in rows:
    if pat.search(name):
        print(f"{n:7d} {tot/1e6:10.3f} ms  avg {tot/n/1e3:9.2f} us  min {mn/1e3:8.2f}  max {mx/1e3:9.2f}  {name[:120]}")

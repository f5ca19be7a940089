"""Histogram of the intervals in which NO kernel runs on the GPU (union over all queues) in a bench.py kernel trace, and what
ran right before / after the long ones.  usage: python tools/idle_hist.py results.db [skip_fraction] [show]
(show = number of long gaps of the top kinds to print with the kernels around them: queue, start, duration)"""
import re, sqlite3, sys
from collections import defaultdict


def fam(name):
    name = re.sub(r"^void ", "", name).replace("dgsct::", "")
    return re.sub(r"[<(].*", "", name)[:30]


c = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
show = int(sys.argv[3]) if len(sys.argv) > 3 else 0
examples = defaultdict(list)
rows = rows[int(len(rows) * skip):]
t0, t1 = rows[0][1], max(r[2] for r in rows)
wall = t1 - t0
hist = defaultdict(lambda: [0, 0.0])
ctx = defaultdict(lambda: [0, 0.0])
cur_end, last = rows[0][2], rows[0]
for i, r in enumerate(rows[1:], 1):
    if r[1] > cur_end:
        g = r[1] - cur_end
        b = "<1us" if g < 1000 else "1-3us" if g < 3000 else "3-10us" if g < 10000 else "10-50us" if g < 50000 else ">=50us"
        hist[b][0] += 1; hist[b][1] += g
        if g >= 10000:
            k = (fam(last[0]), fam(r[0])); ctx[k][0] += 1; ctx[k][1] += g; examples[k].append(i)
    if r[2] > cur_end:
        cur_end, last = r[2], r
print(f"# window {wall/1e6:.1f} ms; idle {sum(v[1] for v in hist.values())/1e6:.2f} ms = {100*sum(v[1] for v in hist.values())/wall:.1f} %")
for b in ("<1us", "1-3us", "3-10us", "10-50us", ">=50us"):
    print(f"  gaps {b:8s}: {hist[b][0]:6d}  {hist[b][1]/1e6:7.2f} ms  {100*hist[b][1]/wall:5.1f} % of the window")
print("long gaps (>= 10 us): kernel that ended last -> kernel that started next")
for k, v in sorted(ctx.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {v[0]:4d} x {v[1]/v[0]/1e3:6.0f} us = {v[1]/1e6:6.2f} ms   {k[0]} -> {k[1]}")
if show:
    for k, v in sorted(ctx.items(), key=lambda kv: -kv[1][1])[:3]:
        for i in examples[k][len(examples[k]) // 2:][:show]:
            print(f"--- {k[0]} -> {k[1]}: kernels around the gap (queue, start us relative to the gap, duration us)")
            base = rows[i][1]
            for r in rows[max(0, i - 10):i + 8]:
                print(f"   q{r[3]:<3d} {(r[1]-base)/1e3:9.1f}  {(r[2]-r[1])/1e3:8.1f}  {fam(r[0])}")

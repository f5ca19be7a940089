"""Per-stream occupancy of a rocprofv3 kernel trace (rocpd sqlite): for every queue, the time it had a kernel running, the number of
kernels, the sum of the idle gaps between consecutive kernels and their histogram -- over the last `steps` steps of the trace.  Answers
"is the chain of a stream kernel-bound or gap-bound".       usage: python tools/stream_busy.py results.db [skip_fraction]"""
import sqlite3, sys
from collections import defaultdict

def short(n):
    import re
    n = re.sub(r"^void ", "", n).replace("dgsct::", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    return n[:60]


def main(path, skip=0.4):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    scol = "stream_id" if "stream_id" in cols else qcol
    rows = c.execute(f"select start, end, {scol}, name from kernels order by start").fetchall()
    # steps end with the optimizer's multi_tensor kernels: the window runs from the end of one step to the end of the last one
    adam = [r for r in rows if "multi_tensor_apply" in r[3]]
    ends = []
    for r in adam:
        if not ends or r[0] - ends[-1] > 5e6: ends.append(r[1])
        else: ends[-1] = r[1]
    nsteps = max(1, int(len(ends) * (1 - skip)))
    lo, hi = ends[-nsteps - 1], ends[-1]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    print(f"{nsteps} steps, {(hi - lo) / 1e6 / nsteps:.2f} ms per step under the tracer")
    wall = (rows[-1][1] - rows[0][0]) / 1e6
    per = defaultdict(list)
    for s, e, q, n in rows:
        per[q].append((s, e, n))
    print(f"window {wall:.2f} ms, {len(rows)} kernels, columns: {scol}")
    print(f"{'stream':>8} {'kernels':>8} {'busy ms':>9} {'busy %':>7} {'gaps<5us':>9} {'5-15us':>8} {'15-50us':>8} {'>50us':>7} {'gap ms (<50us)':>15}")
    for q, ks in sorted(per.items(), key=lambda kv: -len(kv[1])):
        ks.sort()
        busy = 0; cur_e = None; gaps = []
        for s, e, n in ks:
            if cur_e is None: cur_e = e; busy += e - s; continue
            if s >= cur_e: gaps.append(s - cur_e); busy += e - s; cur_e = e
            elif e > cur_e: busy += e - cur_e; cur_e = e
        h = [sum(1 for g in gaps if g < 5e3), sum(1 for g in gaps if 5e3 <= g < 15e3), sum(1 for g in gaps if 15e3 <= g < 50e3), sum(1 for g in gaps if g >= 50e3)]
        small = sum(g for g in gaps if g < 50e3) / 1e6
        print(f"{str(q):>8} {len(ks):8d} {busy/1e6:9.2f} {100*busy/1e6/wall:7.1f} {h[0]:9d} {h[1]:8d} {h[2]:8d} {h[3]:7d} {small:15.2f}")
        ctx = defaultdict(lambda: [0, 0])
        prev = None; cur_e = None
        for s_, e, n in ks:
            if cur_e is not None and s_ - cur_e >= 30e3:
                k = (short(prev), short(n)); ctx[k][0] += 1; ctx[k][1] += s_ - cur_e
            if cur_e is None or e > cur_e: cur_e = e; prev = n
        for (a, b), (cnt, tot) in sorted(ctx.items(), key=lambda kv: -kv[1][1])[:12]:
            print(f"            gap >= 30 us x{cnt:4d} {tot/1e6:7.2f} ms   {a}  ->  {b}")
    # whole device: union of all kernels
    ev = sorted((s, e) for s, e, _, _ in rows)
    busy = 0; cur_e = ev[0][0]
    for s, e in ev:
        if s >= cur_e: busy += e - s; cur_e = e
        elif e > cur_e: busy += e - cur_e; cur_e = e
    print(f"device: some kernel running {busy/1e6:.2f} ms = {100*busy/1e6/wall:.1f} % of the window")

if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4)

"""Wall-clock attribution of a multi-stream rocprofv3 trace: sweep the kernel intervals, split every instant equally
among the kernels active at that instant, report the share per kernel family plus idle time and mean concurrency.
usage: python tools/rocpd_wallshare.py results.db [skip_fraction]   (skips the first fraction of the trace: warm-up)"""
import re
import sqlite3
import sys
from collections import defaultdict


def fam(name):
    name = re.sub(r"^void ", "", name).replace("dgsct::", "")
    if "gemm_kernel" in name:
        m = re.search(r"gemm_kernel<(\d+), (\w+), (\w+), (\d+), (\d+), (\d+), (\d+)>", name)
        return "gemm " + ("%sx%s" % (int(m.group(4)) * int(m.group(6)) * 32, int(m.group(5)) * int(m.group(7)) * 32) if m else "?")
    return re.sub(r"[<(].*", "", name)[:40]


def main(path, skip=0.4):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo = t0 + (t1 - t0) * skip
    ev = []
    for i, (n, s, e) in enumerate(rows):
        if e <= lo:
            continue
        s = max(s, lo)
        ev.append((s, 1, i)); ev.append((e, -1, i))
    ev.sort()
    active = set()
    share = defaultdict(float); solo = defaultdict(float)
    idle = 0.0; conc_t = 0.0
    prev = lo
    for t, d, i in ev:
        dt = t - prev
        if dt > 0:
            if active:
                w = dt / len(active)
                for j in active:
                    share[fam(rows[j][0])] += w
                if len(active) == 1:
                    solo[fam(rows[next(iter(active))][0])] += dt
                conc_t += dt * len(active)
            else:
                idle += dt
        prev = t
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
    wall = t1 - lo
    print(f"# window {wall/1e6:.2f} ms  idle {idle/1e6:.2f} ms ({100*idle/wall:.1f}%)  mean concurrency when busy {conc_t/(wall-idle):.2f}")
    print(f"{'share_ms':>9} {'%wall':>6} {'solo_ms':>8}  family")
    for k, v in sorted(share.items(), key=lambda kv: -kv[1])[:32]:
        print(f"{v/1e6:9.2f} {100*v/wall:6.2f} {solo[k]/1e6:8.2f}  {k}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4)

"""Per-hardware-queue utilisation of a bench.py kernel trace: which stream is the critical resource, and which kernels fill it.
usage: rocprofv3 --kernel-trace -d DIR -o p -- python bench.py ...; python tools/queue_util.py DIR/.../p_results.db [skip_fraction]"""
import re
import sqlite3
import sys
from collections import defaultdict


def fam(name):
    name = re.sub(r"^void ", "", name).replace("dgsct::", "")
    if "gemm_kernel" in name:
        m = re.search(r"gemm_kernel<(\d+), (\w+), (\w+), (\d+), (\d+), (\d+), (\d+)", name)
        return "gemm %dx%d %s%s" % (int(m.group(4)) * int(m.group(6)) * 32, int(m.group(5)) * int(m.group(7)) * 32,
                                    "K" if m.group(2) == "true" else "M", "K" if m.group(3) == "true" else "M") if m else "gemm"
    return re.sub(r"[<(].*", "", name)[:36]


def main(path, skip=0.6):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    n = len(rows)
    rows = rows[int(n * skip):]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    wall = t1 - t0
    byq = defaultdict(list)
    for r in rows:
        byq[r[3]].append(r)
    print(f"# window {wall/1e6:.1f} ms, {len(rows)} dispatches")
    for q, L in sorted(byq.items()):
        busy = sum(e - s for _, s, e, _ in L)
        gaps = [L[i + 1][1] - L[i][2] for i in range(len(L) - 1)]
        small = sum(g for g in gaps if 0 < g < 20000)
        big = sum(g for g in gaps if g >= 20000)
        print(f"queue {q}: {len(L):6d} kernels, busy {100*busy/wall:5.1f} % of the window, gaps < 20 us {100*small/wall:5.1f} %, "
              f"gaps >= 20 us {100*big/wall:5.1f} % ({sum(1 for g in gaps if g >= 20000)} of them)")
        f = defaultdict(lambda: [0, 0])
        for nme, s, e, _ in L:
            a = f[fam(nme)]; a[0] += 1; a[1] += e - s
        top = sorted(f.items(), key=lambda kv: -kv[1][1])[:12]
        print("     " + "; ".join(f"{k} {100*v[1]/wall:.1f}% ({v[0]})" for k, v in top))
        # where the long gaps sit: (kernel before, kernel after) -> total idle time
        gp = defaultdict(lambda: [0, 0])
        for i in range(len(L) - 1):
            g = L[i + 1][1] - L[i][2]
            if g >= 20000:
                a = gp[(fam(L[i][0]), fam(L[i + 1][0]))]; a[0] += 1; a[1] += g
        for k, v in sorted(gp.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"       gap after {k[0]} before {k[1]}: {v[0]} x {v[1]/v[0]/1e3:.0f} us = {100*v[1]/wall:.1f} % of the window")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.6)

"""From a rocprofv3 --kernel-trace db of tools/trace_stage.py: do the two adapters of a pair overlap?
Calls are cut at their first kernel (forward: the hipMemsetAsync fill, backward: zero2_k) on the two adapter queues.
usage: python tools/pair_overlap.py <db> [verbose]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
def nm(n):
    n = re.sub(r"^void ", "", n).replace("dgsct::", "").replace("(anonymous namespace)::", ""); return re.sub(r"[<(].*", "", n)
rows = [(nm(n), s, e, q) for n, s, e, q in rows]
# the two adapter queues = the queues that run zero2_k
mainq = sorted({q for n, s, e, q in rows if n == "zero2_k"})
calls = []      # (queue, kind, start, end, nkernels)
cur = {}
for n, s, e, q in rows:
    if q not in mainq: continue
    first = n == "zero2_k" or "fillBufferAligned" in n
    if first:
        if q in cur: calls.append(cur[q])
        cur[q] = [q, "bwd" if n == "zero2_k" else "fwd", s, e, 1]
    elif q in cur and not n.startswith("at::") and "elementwise" not in n and "multi_tensor" not in n:
        cur[q][3] = max(cur[q][3], e); cur[q][4] += 1
for q in cur: calls.append(cur[q])
calls.sort(key=lambda x: x[2])
# last step: take the last third of the calls
k = len(calls) // 3 if len(calls) >= 12 else 0
calls = calls[-k:] if k else calls
t0 = calls[0][2]
print(f"queues {mainq}; {len(calls)} calls")
print(f"{'kind':4s} {'queue':>5s} {'start us':>10s} {'dur us':>8s} {'kernels':>7s}")
for q, kind, s, e, nk in calls:
    print(f"{kind:4s} {q:5d} {(s-t0)/1e3:10.1f} {(e-s)/1e3:8.1f} {nk:7d}")
# pairs: consecutive calls of the same kind on different queues
i = 0; tot = {"fwd": [0, 0.0, 0.0, 0.0], "bwd": [0, 0.0, 0.0, 0.0]}
while i + 1 < len(calls):
    a, b = calls[i], calls[i + 1]
    if a[1] == b[1] and a[0] != b[0]:
        ov = max(0, min(a[3], b[3]) - max(a[2], b[2])); span = max(a[3], b[3]) - min(a[2], b[2])
        t = tot[a[1]]; t[0] += 1; t[1] += (b[2] - a[2]) / 1e3; t[2] += ov / 1e3; t[3] += span / 1e3
        i += 2
    else:
        i += 1
for kind, (n, stg, ov, span) in tot.items():
    if n: print(f"{kind}: {n} pairs, second call starts {stg/n:.0f} us after the first, overlap {ov/n:.0f} us, pair span {span/n:.0f} us")

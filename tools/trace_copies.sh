cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/cp; timeout 600 rocprofv3 --kernel-trace -d /tmp/cp -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/cp.log 2>&1
tail -2 /tmp/cp.log | cut -c1-200
python - <<'PY'
import sqlite3, glob, re, collections
db = glob.glob("/tmp/cp/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, queue_id, grid_x from kernels order by start").fetchall()
print(len(rows), "dispatches")
byq = collections.defaultdict(list)
for r in rows: byq[r[3]].append(r)
ctx = collections.Counter()
def short(n): return re.sub(r"[<(].*", "", n.replace("void ", "").replace("dgsct::", ""))[:40]
for q, L in byq.items():
    for i, r in enumerate(L):
        if "copyBuffer" in r[0]:
            prev = short(L[i-1][0]) if i else "-"; nxt = short(L[i+1][0]) if i+1 < len(L) else "-"
            ctx[(q, prev, nxt, r[4])] += 1
for k, v in ctx.most_common(25): print(v, k)
PY

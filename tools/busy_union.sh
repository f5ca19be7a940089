# Union-busy time and concurrency histogram of the GPU over the timed steps of bench.py (kernel trace).
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/bu; timeout 600 rocprofv3 --kernel-trace -d /tmp/bu -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/bu.log 2>&1
python - <<'PY'
import sqlite3, glob, collections
db = glob.glob("/tmp/bu/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select start, end, queue_id, name from kernels order by start").fetchall()
# take the last 40 % of dispatches (steady-state timed steps)
n = len(rows); rows = rows[int(n * 0.6):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for s, e, q, _ in rows: ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = collections.Counter(); cur = 0; last = t0
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
wall = t1 - t0
print("window %.1f ms, %d dispatches, queues %s" % (wall / 1e6, len(rows), sorted(set(r[2] for r in rows))))
for k in sorted(hist): print("  %d kernels running: %5.1f %%" % (k, 100.0 * hist[k] / wall))
print("  sum of kernel durations / wall = %.2f" % (sum(r[1] - r[0] for r in rows) / wall))
PY

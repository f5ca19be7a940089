# every kernel (library and torch) of one traced bench step, in start order: gpurun_out/step_timeline.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/stl && rocprofv3 --kernel-trace -d /tmp/stl -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/b.log 2>&1
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/step_timeline.txt
import re, sqlite3, glob
c = sqlite3.connect(glob.glob('/tmp/stl/**/*.db', recursive=True)[0])
rows = c.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
# last step: find the last 'multi_tensor_apply' (Adam) and take the ~4200 kernels before it
idx = [i for i, r in enumerate(rows) if 'multi_tensor_apply' in r[0]]
end = idx[-1] + 8
steps = [i for i in idx]
# previous Adam block end
prev = max(i for i in idx if i < idx[-1] - 500)
seg = rows[prev + 1:end]
t0 = seg[0][1]
for n, s, e, q, gx, wx in seg:
    nm = re.sub(r"^void ", "", n).replace("dgsct::", ""); nm = re.sub(r"\(.*", "", nm)[:60]
    print(f"{(s-t0)/1e3:10.1f} +{(e-s)/1e3:7.1f} q{q} wg={gx//max(wx,1):<6d} {nm}")
PY

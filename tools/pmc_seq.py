"""Per-dispatch HBM-side traffic of ONE adapter forward+backward: joins the FETCH_SIZE and WRITE_SIZE passes of
tools/pmc_stack.sh by dispatch order (second iteration only).  usage: python tools/pmc_seq.py gpurun_out/pmc N_C_No_Co"""
import os, sqlite3, sys


def load(db):
    c = sqlite3.connect(db)
    q = ("select P.dispatch_id, P.name, sum(P.counter_value), max(P.duration), K.grid_size_x, K.grid_size_y, K.grid_size_z, "
         "K.workgroup_size_x, K.workgroup_size_y, K.workgroup_size_z from pmc_events P join rocpd_kernel_dispatch K on "
         "K.dispatch_id = P.dispatch_id group by P.dispatch_id order by P.dispatch_id")
    return [r for r in c.execute(q).fetchall() if "dgsct" in r[1] or "rocclr" in r[1]]


root, shape = sys.argv[1], sys.argv[2]
f = load(os.path.join(root, shape + "_FETCH_SIZE", "p_results.db"))
w = load(os.path.join(root, shape + "_WRITE_SIZE", "p_results.db"))
assert len(f) == len(w), (len(f), len(w))
h = len(f) // 2
tf = tw = 0
for a, b in zip(f[h:], w[h:]):
    fm, wm = a[2] * 2 * 1024 / 1e6, b[2] * 1024 / 1e6      # FETCH_SIZE in KiB, x2 (gfx950 correction); WRITE_SIZE in KiB
    tf += fm; tw += wm
    nm = a[1].replace("void ", "").replace("dgsct::", "")
    print("%8.1f us  fetch %8.1f MB  write %8.1f MB  grid=(%d,%d,%d)  %s" % (a[3] / 1e3, fm, wm, a[4] // a[7], a[5] // max(1, a[8]), a[6] // max(1, a[9]), nm[:60]))
print("# total fetch %.1f MB  write %.1f MB" % (tf, tw))

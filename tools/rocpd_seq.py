"""Print the kernel dispatch sequence (second half = the steady-state iteration) of a rocpd db."""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
rows = [r for r in rows if "dgsct" in r[0] or "rocclr" in r[0]]
half = rows[len(rows) // 2:]
tot = sum(r[2] - r[1] for r in half)
print(f"# {len(half)} dispatches, {tot/1e3:.1f} us kernel time, wall {(half[-1][2]-half[0][1])/1e3:.1f} us")
for name, s, e, gx, gy, gz in half:
    nm = re.sub(r"^void ", "", name).replace("dgsct::", "").replace("(anonymous namespace)::", "")
    nm = re.sub(r"\(.*", "", nm)
    print(f"{(e-s)/1e3:9.1f} us  grid=({gx//256 if gx>=256 else gx},{gy},{gz})  {nm}")

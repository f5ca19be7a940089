import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols=[r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
print("#cols",cols)
qc = "queue_id" if "queue_id" in cols else "stream_id"
rows = c.execute(f"select name, start, end, grid_x, grid_y, grid_z, {qc} from kernels order by start").fetchall()
rows = [r for r in rows if "dgsct" in r[0] or "rocclr" in r[0]]
half = rows[len(rows) // 2:]
t0=half[0][1]; prev_end={}
busy=0; last=t0
# union busy time
iv=sorted((r[1],r[2]) for r in half); cur_s,cur_e=iv[0]; 
for s,e in iv[1:]:
    if s>cur_e: busy+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print(f"# wall {(half[-1][2]-t0)/1e3:.1f} us, union busy {busy/1e3:.1f} us, sum {sum(r[2]-r[1] for r in half)/1e3:.1f}")
for name, s, e, gx, gy, gz, q in half:
    nm = re.sub(r"^void ", "", name).replace("dgsct::", "").replace("(anonymous namespace)::", ""); nm = re.sub(r"\(.*", "", nm)
    gap = (s-prev_end.get(q,s))/1e3; prev_end[q]=e
    print(f"{(s-t0)/1e3:9.1f} +{(e-s)/1e3:7.1f} us gap{gap:7.1f} q={q} grid=({gx//256 if gx>=256 else gx},{gy},{gz}) {nm[:50]}")

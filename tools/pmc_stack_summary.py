"""Scale the per-shape PMC passes (tools/pmc_stack.sh) to one bench step of the AVE Swin-V2-B stack and write
profiles/r01_pmc_traffic.json (+ a text table).  FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B for wide
coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE is uncalibrated and used raw.  Units: KB -> bytes."""
import json, os, re, sqlite3, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC = os.path.join(ROOT, "gpurun_out", "pmc")
# (N, C, No, Co) -> adapters of that shape per step (layers x {p1,p2})
COUNT = {(2304,128,4096,96): 4, (4096,96,2304,128): 4, (576,256,1024,192): 4, (1024,192,576,256): 4,
         (144,512,256,384): 12, (256,384,144,512): 12, (36,1024,64,768): 4, (64,768,36,1024): 4}

def fam(name):
    if "gemm_kernel" in name: return "gemm_kernel<*>"
    name = re.sub(r"^void ", "", name).replace("dgsct::", "")
    return re.sub(r"[<(].*", "", name)

tot = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0.0])   # fetch_B, write_B, launches, ns
for shp, cnt in COUNT.items():
    for ctr, idx in (("FETCH_SIZE", 0), ("WRITE_SIZE", 1)):
        db = os.path.join(PMC, "%d_%d_%d_%d_%s" % (*shp, ctr), "p_results.db")
        c = sqlite3.connect(db)
        rows = c.execute("select name, counter_value, duration from pmc_events where counter_name=? order by dispatch_id", (ctr,)).fetchall()
        rows = [r for r in rows if "dgsct" in r[0] or "rocclr" in r[0]]
        half = rows[len(rows) // 2:]
        for name, val, dur in half:
            a = tot[fam(name)]
            a[idx] += val * 1024 * (2 if idx == 0 else 1) * cnt
            if idx == 0:
                a[2] += cnt; a[3] += dur * cnt
lines = [f"{'launches/step':>13} {'fetch_GB(x2)':>13} {'write_GB':>9} {'MB/launch':>10}  family   (one bench step, B=16, Swin-V2-B shapes)"]
out = {}
for k, (f, w, n, ns) in sorted(tot.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    lines.append(f"{int(n):13d} {f/1e9:13.2f} {w/1e9:9.2f} {(f+w)/n/1e6:10.2f}  {k}")
    out[k] = dict(launches_per_step=int(n), fetch_bytes=f, write_bytes=w, bytes_per_launch=(f + w) / n)
print("\n".join(lines))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"), "w"), indent=1)
open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.txt"), "w").write("\n".join(lines) + "\n")

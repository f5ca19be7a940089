"""gemm8 timing experiments on the five deep remap products (DGSCT_GEMM8_DBG bits; results of bits != 0 are garbage):
usage: python tools/gemm8_dbg.py   (spawns one child per setting)"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import gemm_bench as gb
    print("RESULT " + json.dumps([gb.run(s, iters=20) for s in gb.SHAPES[:5]]))
    sys.exit(0)
settings = [int(x) for x in (sys.argv[1:] or ["0", "8", "1", "2", "3", "6"])]
rows = {}
for d in settings:
    env = dict(os.environ); env["DGSCT_GEMM8_DBG"] = str(d)
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    rows[d] = json.loads(line[0][7:]) if line else None
    if rows[d] is None: print(r.stderr[-2000:])
names = ["fwd audio KK 4096x(96x160)x2304", "fwd visual KM 2304x(96x160)x4096", "bwd MM 4096x(96x160)x2304", "dWn audio KM two-level", "dWn visual KK two-level"]
print("DBG bits (1 no DMA, 2 no fragment reads, 4 no MFMA, 8 DMA issued a whole k-tile early)")
print("shape".ljust(36) + "".join(f"{d:>9d}" for d in rows))
for i, n in enumerate(names):
    print(n.ljust(36) + "".join(f"{(rows[d][i] if rows[d] else float('nan')):9.1f}" for d in rows))

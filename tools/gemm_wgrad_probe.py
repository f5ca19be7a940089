"""Deep-K weight-gradient GEMMs (M, N <= 256, K = all token rows): tile config x split-K sweep (tuning hooks)."""
import os, subprocess, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(128, 128, 368640, 1, 1, 0, 0, 0, 1, 0, 0), (96, 96, 655360, 1, 1, 0, 0, 0, 1, 0, 0), (64, 128, 368640, 1, 1, 0, 0, 0, 1, 0, 0),
          (256, 256, 92160, 1, 1, 0, 0, 0, 1, 0, 0), (512, 512, 23040, 1, 1, 0, 0, 0, 1, 0, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import gemm_bench as gb
    print("RESULT " + json.dumps([gb.run(s, iters=10) for s in SHAPES]))
    sys.exit(0)
rows = {}
for cfg, sk in [("auto", 0), (4, 0), (0, 256), (0, 512), (0, 128), (1, 256), (4, 128), (4, 512)]:
    env = dict(os.environ)
    if cfg != "auto": env["DGSCT_GEMM_CFG"] = str(cfg)
    if sk: env["DGSCT_GEMM_SPLITK"] = str(sk)
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    rows[(cfg, sk)] = json.loads(line[0][7:]) if line else None
print("shape".ljust(28) + "".join(f"{str(k):>13}" for k in rows))
for i, s in enumerate(SHAPES):
    print(str(s[:3]).ljust(28) + "".join(f"{(rows[k][i] if rows[k] else float('nan')):13.1f}" for k in rows))

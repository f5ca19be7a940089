"""Diagnostic: run-to-run spread of the scalar `gate` gradient (fp32, avs_s4 stage-0 audio shape) against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_configs_gpu as T
vals = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    r = T.run_case(4096, 96, 2304, 192, BT=5, dtype=torch.float32, flavour="avs_s4")
    g, go = r["grads"]["gate"]
    vals.append(g.item())
    worst = max((T.rel_err(a, b.reshape(-1)), k) for k, (a, b) in r["grads"].items())
    print(i, "gate gpu %.6f oracle %.6f" % (g.item(), go.item()), "worst", worst, flush=True)

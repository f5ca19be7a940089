"""Do the two adapters of a pair overlap on the GPU?  Event pairs around every library call on its own stream (no tracer: under
rocprofv3 the late stages turn host-bound and the answer changes), one eager step of the bench stack.
usage: python tools/call_overlap.py [BT]"""
import os, sys, time
os.environ.setdefault("DGSCT_WHATIF", "1")       # this tool may set the what-if switches (dgsct_test_tune "skip": results garbage, timing real)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgsct_amd
from dgsct_amd import _lib
import bench
dev = torch.device("cuda:0")
stages, stack = bench.build_stack("swinv2_base", torch.bfloat16, dev, concurrent="--serial" not in sys.argv)
stack.train()
BT = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 160
feats, cots, mcots = bench.make_inputs(stages, BT, torch.bfloat16, dev, 1)
params = [p for p in stack.parameters()]
def step():
    outs, maps = stack(feats)
    tensors = [t for pair in outs for t in pair] + [maps[0], maps[1]]
    grads = [g for pair in cots for g in pair] + [mcots[0], mcots[1]]
    torch.autograd.backward(tensors, grads)
    for p in params: p.grad = None
    for fv, fa in feats: fv.grad = None; fa.grad = None
for _ in range(3): step()
torch.cuda.synchronize()
lib = _lib.default_lib()
path = os.environ.setdefault("DGSCT_CALL_PROF", "/tmp/dgsct_callprof.txt")


def measure(nsteps=3):
    """average pair statistics over nsteps steps: {(kind, Nmin): [pairs, first, second, stagger, overlap, span]} (us), wall ms"""
    agg, walls = {}, []
    for _ in range(nsteps):
        lib.test_tune("callprof", 1)
        t0 = time.perf_counter(); step(); torch.cuda.synchronize(); walls.append(time.perf_counter() - t0)
        lib.test_tune("callprof", 0); lib.test_tune("callprof", 2)
        calls = []
        for ln in open(path):
            k, N, C_, s_, a, b = ln.split(); calls.append((k, int(N), int(C_), s_, float(a), float(b)))
        calls.sort(key=lambda c: c[4])
        used = [False] * len(calls)
        for i, c in enumerate(calls):
            if used[i]: continue
            j = next((j for j in range(i + 1, min(i + 4, len(calls))) if not used[j] and calls[j][0] == c[0] and calls[j][3] != c[3]), None)
            if j is None: continue
            d = calls[j]; used[i] = used[j] = True
            ov = max(0.0, min(c[5], d[5]) - max(c[4], d[4])); span = max(c[5], d[5]) - min(c[4], d[4])
            e = agg.setdefault((c[0], min(c[1], d[1])), [0, 0.0, 0.0, 0.0, 0.0, 0.0])
            e[0] += 1; e[1] += c[5] - c[4]; e[2] += d[5] - d[4]; e[3] += d[4] - c[4]; e[4] += ov; e[5] += span
    return agg, 1e3 * sum(walls) / len(walls)


def show(agg, wall, title):
    print(f"--- {title}: step wall {wall:.2f} ms (with the event pairs)")
    print(f"{'kind':4s} {'N(min)':>6s} {'pairs':>5s} {'first us':>9s} {'second us':>9s} {'stagger us':>10s} {'overlap us':>10s} {'span us':>8s}")
    tot = 0.0
    for (k, N), (n, a, b, st, ov, sp) in sorted(agg.items(), key=lambda x: (x[0][0], -x[0][1])):
        print(f"{k:4s} {N:6d} {n:5d} {a/n:9.0f} {b/n:9.0f} {st/n:10.0f} {ov/n:10.0f} {sp/n:8.0f}")
        tot += sp / (n / (4 if N != 144 else 12)) if n else 0.0
    return


if os.environ.get("SKIPMINC"):           # what-if switches ("skip", plan.cpp) only for adapters at least this wide
    lib.test_tune("skipminc", int(os.environ["SKIPMINC"]))
if os.environ.get("SKIPMAXC"):
    lib.test_tune("skipmaxc", int(os.environ["SKIPMAXC"]))
base, wall = measure()
show(base, wall, "default")
# A/B inside one process (two boxes differ by 2 %): each named switch off, then the default again
for key in [k for k in os.environ.get("AB", "").split(",") if k]:
    key, _, offv = key.partition("=")            # "vq1fuse=3": the value standing for "off" in this comparison
    old = lib.test_tune(key, int(offv) if offv else 0)
    off, w_off = measure()
    lib.test_tune(key, old)
    on, w_on = measure()
    print(f"=== {key}: off {w_off:.2f} ms  on {w_on:.2f} ms   pair span off -> on (us):")
    for kk in sorted(on, key=lambda x: (x[0], -x[1])):
        print(f"    {kk[0]} N>={kk[1]:5d}: {off[kk][5]/off[kk][0]:7.0f} -> {on[kk][5]/on[kk][0]:7.0f}")

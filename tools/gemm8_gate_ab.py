"""Which launch shapes gain on gemm8 when its size gates are lifted?  Joins two per-launch GEMM logs (DGSCT_PROF_DUMP, serial pass of
bench.py) -- default gates vs DGSCT_GEMM8=2 -- by shape and lists the shapes whose engine changed.   usage: gemm8_gate_ab.py a.csv b.csv"""
import csv, sys, collections
def load(p):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        key = tuple(r[k] for k in ("M", "N", "K", "KB", "batch", "ak", "bk", "atomic"))
        a = agg.setdefault(key, [0, 0.0, set()])
        a[0] += 1; a[1] += float(r["ms"]); a[2].add(r["cfg"])
    return agg
a, b = load(sys.argv[1]), load(sys.argv[2])
print(f"{'calls':>5} {'default us':>10} {'gemm8=2 us':>10} {'delta ms':>9}  cfg -> cfg   M N K KB batch ak bk atomic")
tot = 0.0
rows = []
for k, (n, ms, cf) in a.items():
    if k not in b: continue
    n2, ms2, cf2 = b[k]
    if cf == cf2: continue
    rows.append((ms2 - ms, n, ms / n * 1e3, ms2 / n2 * 1e3, cf, cf2, k))
for d, n, u1, u2, cf, cf2, k in sorted(rows):
    tot += d
    print(f"{n:5d} {u1:10.1f} {u2:10.1f} {d:9.3f}  {','.join(sorted(cf))} -> {','.join(sorted(cf2))}   " + " ".join(k))
print(f"sum over changed shapes: {tot:.3f} ms over the profiled passes")

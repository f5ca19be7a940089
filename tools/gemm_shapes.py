"""Aggregate the per-launch GEMM log (DGSCT_PROF_DUMP=path python bench.py --serial ...) by shape."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    key = tuple(r[k] for k in ("M", "N", "K", "KB", "batch", "splitk", "cfg", "ak", "bk", "atomic", "wide"))
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["flops"]); a[3] += float(r["bytes"])
tot = sum(a[1] for a in agg.values())
print(f"# {len(rows)} launches, {tot:.2f} ms")
print(f"{'calls':>5} {'tot_ms':>8} {'avg_us':>8} {'TF/s':>7} {'GB/s':>7}  M N K KB batch splitk cfg ak bk atomic wide")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{a[0]:5d} {a[1]:8.3f} {a[1]/a[0]*1e3:8.1f} {a[2]/a[1]/1e9:7.1f} {a[3]/a[1]/1e6:7.0f}  " + " ".join(key))

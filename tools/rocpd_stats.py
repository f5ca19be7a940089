"""Summarise a rocprofv3 rocpd sqlite file: per-kernel calls / total / avg, like --stats CSV.
usage: python tools/rocpd_stats.py results.db [top_n]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("dgsct::", "").replace("(anonymous namespace)::", "")
    return name if len(name) <= 110 else name[:107] + "..."


def main(path, top=40, gemm_json=None, steps=None):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    # dispatches before the library's first kernel are the bench's set-up (flatten_parameters: ~2900 small copyBuffer launches), not steps
    first = c.execute("select min(start) from kernels where name like '%dgsct%'").fetchone()[0]
    setup = c.execute("select count(*) from kernels where start < ?", (first,)).fetchone()[0] if first else 0
    print(f"# {path}: {sum(r[1] for r in rows)} kernel dispatches, {total/1e6:.3f} ms total GPU kernel time")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'min_us':>8} {'max_us':>9} {'%':>6}  name")
    for name, n, tot, mn, mx in rows[:top]:
        print(f"{n:7d} {tot/1e6:10.3f} {tot/n/1e3:9.2f} {mn/1e3:8.2f} {mx/1e3:9.2f} {100*tot/total:6.2f}  {short(name)}")
    # family roll-up
    fam = {}
    for name, n, tot, mn, mx in rows:
        k = "gemm_kernel<*>" if ("gemm_kernel" in name or "gemm8_kernel" in name or "gemm_fx_kernel" in name or "wgrad_bt_k" in name) else ("torch/other" if "dgsct" not in name else re.sub(r"<.*", "", short(name)))
        a = fam.setdefault(k, [0, 0])
        a[0] += n; a[1] += tot
    if gemm_json and steps:
        import json
        g = fam.get("gemm_kernel<*>", [0, 0])
        sk = fam.get("gemm_skinny_k", [0, 0])
        json.dump({"gemm_ms_per_step": round(g[1] / 1e6 / steps, 3), "launches_per_step": round(g[0] / steps, 1),
                   "skinny_ms_per_step": round(sk[1] / 1e6 / steps, 3), "skinny_launches_per_step": round(sk[0] / steps, 1),
                   "kernel_ms_per_step_all": round(total / 1e6 / steps, 2), "dispatches_per_step": round((sum(r[1] for r in rows) - setup) / steps, 1),
                   "setup_dispatches_excluded": setup,
                   "steps_traced": steps,
                   "source": "rocprofv3 --kernel-trace of `python bench.py --steps 5 --warmup 2 --no-roofline --no-cpu-baseline` "
                             "(7 steps under the timed two-stream schedule); gemm = gemm_kernel<*> + gemm8_kernel<*> + gemm_fx_kernel<*> + wgrad_bt_k"}, open(gemm_json, "w"))
    print("# by family")
    for k, (n, tot) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:7d} {tot/1e6:10.3f} {tot/n/1e3:9.2f} {'':8} {'':9} {100*tot/total:6.2f}  {k}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40, sys.argv[3] if len(sys.argv) > 3 else None,
         int(sys.argv[4]) if len(sys.argv) > 4 else None)

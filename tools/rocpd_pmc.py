"""Per-kernel PMC sums from a rocprofv3 --pmc rocpd sqlite file.
usage: python tools/rocpd_pmc.py results.db [name-regex]"""
import re
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".", re.I)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
# rocpd layout: pmc_events(event_id -> dispatch), info_pmc(name), kernel_dispatch, info_kernel_symbol
q = """
select ks.kernel_name, p.name, sum(e.value), count(distinct d.id)
from rocpd_pmc_event e
join rocpd_info_pmc p on p.id = e.pmc_id
join rocpd_kernel_dispatch d on d.event_id = e.event_id
join rocpd_info_kernel_symbol ks on ks.id = d.kernel_id
group by ks.kernel_name, p.name
"""
try:
    rows = c.execute(q).fetchall()
except Exception as ex:      # schema differs between rocprofv3 builds: show what is there
    print("query failed:", ex)
    for t in tabs:
        if "pmc" in t or "kernel" in t:
            print(t, [r[1] for r in c.execute(f"pragma table_info({t})")])
    sys.exit(1)
tab = defaultdict(dict)
cnt = {}
for k, n, v, d in rows:
    if pat.search(k):
        k2 = re.sub(r"\(.*", "", re.sub(r"^void ", "", k.replace("(anonymous namespace)::", ""))).replace("dgsct::", "")
        tab[k2][n] = v
        cnt[k2] = d
names = sorted({n for v in tab.values() for n in v})
print("kernel".ljust(44), "disp", " ".join(n[-18:].rjust(18) for n in names))
for k, v in sorted(tab.items()):
    print(k[:44].ljust(44), str(cnt[k]).rjust(4), " ".join(f"{v.get(n, 0):18.4g}" for n in names))

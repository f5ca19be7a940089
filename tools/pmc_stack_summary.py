"""Scale the per-shape PMC passes (tools/pmc_stack.sh) to one bench step of the AVE Swin-V2-B stack:
profiles/<tag>_pmc_traffic.{json,txt} (HBM-side bytes per kernel family and per step) and profiles/<tag>_sq_counters.txt
(MFMA-busy share, LDS bank-conflict share, issue-stall share per kernel family).
FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B for wide coalesced reads, MI355X_MICROARCH.md HBM section);
WRITE_SIZE is uncalibrated and used raw.  Both are in KB.  Every per-shape run executes TWO identical iterations
(tools/trace_adapter.py): sums are halved.
usage: python tools/pmc_stack_summary.py <tag> [bench_json]"""
import collections, json, os, re, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC = os.path.join(ROOT, "gpurun_out", "pmc")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
COUNT = {(2304, 128, 4096, 96): 4, (4096, 96, 2304, 128): 4, (576, 256, 1024, 192): 4, (1024, 192, 576, 256): 4,
         (144, 512, 256, 384): 12, (256, 384, 144, 512): 12, (36, 1024, 64, 768): 4, (64, 768, 36, 1024): 4}


def fam(name):
    if "gemm_kernel" in name or "gemm8_kernel" in name or "gemm_fx_kernel" in name or "wgrad_bt_k" in name:      # both MFMA GEMM kernels (4-wave tiled engine, 8-wave deep-product kernel)
        return "gemm_kernel<*>"
    name = re.sub(r"^void ", "", name).replace("dgsct::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"[<(].*", "", name)


def rows_of(shp, grp):
    """one row per (dispatch, counter): hardware counters come as one record per XCD / shader engine instance"""
    db = os.path.join(PMC, "%d_%d_%d_%d_%s" % (*shp, grp), "p_results.db")
    c = sqlite3.connect(db)
    rows = c.execute("select name, counter_name, sum(counter_value), max(duration) from pmc_events group by dispatch_id, counter_name").fetchall()
    return [r for r in rows if "dgsct" in r[0] or "rocclr" in r[0]]


tot = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0.0])      # fetch_B, write_B, launches, ns
sq = collections.defaultdict(lambda: collections.defaultdict(float))
for shp, cnt in COUNT.items():
    for grp, idx in (("FETCH_SIZE", 0), ("WRITE_SIZE", 1)):
        for name, cn, val, dur in rows_of(shp, grp):
            a = tot[fam(name)]
            a[idx] += val * 1024 * (2 if idx == 0 else 1) * cnt / 2
            if idx == 0:
                a[2] += cnt / 2; a[3] += dur * cnt / 2
    try:
        for name, cn, val, dur in rows_of(shp, "SQ"):
            sq[fam(name)][cn] += val * cnt / 2
            if cn == "SQ_WAVE_CYCLES":
                sq[fam(name)]["_ns"] += dur * cnt / 2
    except Exception as e:
        print("no SQ pass for", shp, e)
lines = [f"{'launches/step':>13} {'fetch_GB(x2)':>13} {'write_GB':>9} {'MB/launch':>10} {'ms(serial)':>11}  family   (one bench step, B=16, Swin-V2-B shapes, 48 adapter calls)"]
out = {}
F = W = NL = 0.0
for k, (f, w, n, ns) in sorted(tot.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    lines.append(f"{int(round(n)):13d} {f/1e9:13.2f} {w/1e9:9.2f} {(f+w)/max(n,1)/1e6:10.2f} {ns/1e6:11.2f}  {k}")
    out[k] = dict(launches_per_step=int(round(n)), fetch_bytes=f, write_bytes=w, bytes_per_launch=(f + w) / max(n, 1), serial_ms=ns / 1e6)
    F += f; W += w; NL += n
lines.append(f"{int(round(NL)):13d} {F/1e9:13.2f} {W/1e9:9.2f} {'':10} {'':11}  TOTAL (adapter kernels; torch-side Adam / residual adds excluded)")
out["_total"] = dict(launches_per_step=int(round(NL)), fetch_bytes=F, write_bytes=W, bytes=F + W)
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    rl = b.get("roofline") or {}
    out["_bench"] = dict(ms_per_step=b["ms_per_step"], gemm_launches_per_step=rl.get("launches_per_step"))
    lines.append(f"bench: {b['ms_per_step']} ms/step, {rl.get('launches_per_step')} GEMM launches/step (PMC mix: {out.get('gemm_kernel<*>', {}).get('launches_per_step')});"
                 f" HBM-side traffic / step time = {(F + W) / 1e9 / (b['ms_per_step'] * 1e-3):.0f} GB/s")
print("\n".join(lines))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.txt"), "w").write("\n".join(lines) + "\n")
if sq:
    l2 = [f"{'family':28s} {'ms/step':>8} {'MFMA busy %':>12} {'LDS conflict %':>15} {'issue stall %':>14}   MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / "
          "(kernel time x 2.4 GHz x 1024 SIMDs) (a lower bound: the chip clocks below 2.4 GHz under load); LDS conflict = SQ_LDS_BANK_CONFLICT / "
          "SQ_LDS_IDX_ACTIVE; issue stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.  Per-shape passes scaled to one step."]
    for k, v in sorted(sq.items(), key=lambda kv: -kv[1].get("_ns", 0))[:16]:
        ns = v.get("_ns", 0) or 1
        mf = 100 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (ns * 2.4 * 1024)
        lc = 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / (v.get("SQ_LDS_IDX_ACTIVE", 0) or 1)
        st = 100 * v.get("SQ_WAIT_INST_ANY", 0) / (v.get("SQ_WAVE_CYCLES", 0) or 1)
        l2.append(f"{k:28s} {ns/1e6:8.2f} {mf:12.1f} {lc:15.1f} {st:14.1f}")
    print("\n".join(l2))
    open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters.txt"), "w").write("\n".join(l2) + "\n")

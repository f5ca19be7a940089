"""Static check for serialised loads: compile the kernel sources to gfx950 assembly and list, per kernel, the ratio of
`s_waitcnt vmcnt(0)` to global loads.  A load inside a (lane-dependent) conditional compiles to branch + load + vmcnt(0), so
a kernel whose loads could be in flight together shows a ratio near 1.  Static counts: cold paths count too.
usage: python tools/isa_waits.py [file.hip ...]      (default: the row / strip / attention kernels; gemm.hip takes minutes)"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dg-sct_amd", "csrc")
files = sys.argv[1:] or [os.path.join(SRC, f) for f in ("prims_hip.hip", "prims_strip.hip", "prims_proj.hip", "attn.hip", "attn2.hip", "temporal.hip")]
rows = []
with tempfile.TemporaryDirectory() as tmp:
    procs = []
    for f in files:
        out = os.path.join(tmp, os.path.basename(f) + ".s")
        procs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc", "-S", "--cuda-device-only", "--offload-arch=gfx950", "-O3", "-std=c++17",
                                             "-I" + SRC, "-I" + os.path.join(ROOT, "include"), f, "-o", out], stderr=subprocess.DEVNULL)))
    for out, p in procs:
        p.wait()
        cur = None
        for l in open(out):
            m = re.match(r"^(_Z\w+):", l)
            if m:
                if cur: rows.append((cur, loads, w0, wn))
                cur, loads, w0, wn = m.group(1), 0, 0, 0
                continue
            if cur is None: continue
            if re.search(r"\b(global_load|buffer_load|flat_load)", l): loads += 1
            w = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
            if w: w0, wn = (w0 + 1, wn) if w.group(1) == "0" else (w0, wn + 1)
        if cur: rows.append((cur, loads, w0, wn))
rows = sorted((r for r in rows if r[1] >= 8), key=lambda r: -(r[2] / r[1]))
for name, loads, w0, wn in rows[:40]:
    dm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:100]
    print("%5.2f vmcnt(0)/load   loads %4d  vmcnt(0) %4d  vmcnt(n) %4d   %s" % (w0 / loads, loads, w0, wn, dm))

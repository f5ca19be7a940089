"""Static check of the generated gfx950 ISA of csrc/gemm8.hip (hipcc -S, no GPU needed).

gemm8.hip issues the fragment reads of its transpose-read variants as inline asm and orders them by hand (hipcc would
otherwise drain the LDS-DMA queue in front of every ds_read_b64_tr_b16).  The compiler does not know that the outputs
of those asm statements arrive later, so three properties are checked on what it actually emitted, per kernel:
  1. no scratch traffic (a spill reload is a VMEM load: `s_waitcnt vmcnt(0)` in the loop, and copies of in-flight data);
  2. inside the k-loop every `s_waitcnt vmcnt` sits in an inline-asm block (ours, counted) -- none made by the compiler;
  3. between an asm `ds_read*` and the next asm `s_waitcnt lgkmcnt(0)` no instruction names the registers being loaded;
  4. (round 5, the kernels with the cross-tile pipelined k-loop: last template argument `true`) the interleave itself: the loop has ONE
     s_barrier; behind it come the LDS-DMA of the tile after next and the first fragment reads of the NEXT tile, and only then the last
     MFMA group of the current tile -- i.e. that read flies under MFMAs instead of sitting exposed behind the barrier.
usage: python tools/check_gemm8_isa.py  -> exit 0 / 1, prints one line per kernel."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dg-sct_amd", "csrc", "gemm8.hip")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def isa():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-S",
                        "-o", "-", SRC], capture_output=True, text=True, check=True)
    return r.stdout


def kernels(text):
    cur, name = None, None
    for line in text.splitlines():
        m = re.match(r"^(_ZN5dgsct12gemm8_kernel\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                yield name, cur
                cur = None


def check(name, lines):
    errs = []
    if any(re.search(r"\bscratch_(load|store)", l) for l in lines):
        errs.append("scratch traffic (register spill)")
    # k-loop = from the 'Loop Header' label to the last backward branch to it
    hdr = next((i for i, l in enumerate(lines) if "Loop Header" in l and l.startswith(".LBB")), None)
    if hdr is None:
        return errs + ["no loop found"]
    label = lines[hdr].split(":")[0]
    # the loop may be entered through a preheader block placed before it: take every backward branch to blocks at/above hdr
    # (the latch may be an unconditional s_branch: the round-5 pipelined loop ends in `if (it + 1 < nk) landed(0)` + a plain jump back)
    end = max((i for i, l in enumerate(lines) if re.search(r"s_(c)?branch\w*\s+" + re.escape(label) + r"\b", l)), default=None)
    if end is None or end < hdr:
        cands = [i for i, l in enumerate(lines) if i > hdr and re.search(r"s_cbranch\w*\s+\.LBB\d+_\d+", l)
                 and any(lines[j].startswith(l.split()[-1] + ":") for j in range(0, i))]
        end = max(cands) if cands else len(lines) - 1
    if name.endswith("ELb1EEEvNS_2G8E"):                       # PIPE variant: the interleave (item 4 of the header)
        body = [(i, lines[i].strip()) for i in range(hdr, end + 1)]
        bars = [i for i, l in body if l.startswith("s_barrier")]
        if len(bars) != 1:
            errs.append(f"pipelined k-loop: expected exactly one s_barrier, found {len(bars)}")
        else:
            after = [(i, l) for i, l in body if i > bars[0]]
            dma = [i for i, l in after if l.startswith("global_load_lds")]
            rd = [i for i, l in after if l.startswith("ds_read")]
            mf = [i for i, l in after if l.startswith("v_mfma")]
            if not dma or not rd or not mf or not (min(rd) < min(mf)) or not (min(dma) < min(mf)):
                errs.append("pipelined k-loop: behind the barrier the DMA issue and the next tile's first fragment reads must precede the last MFMA group")
    in_asm = False
    pending = {}            # register -> line of the asm ds_read that loads it
    for i in range(hdr, end + 1):
        l = lines[i].strip()
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not l or l.startswith(";") or l.startswith("."):
            continue
        if "s_waitcnt" in l and "vmcnt" in l and not in_asm:
            errs.append(f"compiler-made vmcnt wait in the k-loop: '{l}'")
        if in_asm and l.startswith("ds_read"):
            dst, addr = l.split(",")[0], ",".join(l.split(",")[1:])
            bad = regs(addr) & set(pending)
            if bad:
                errs.append(f"in-flight register used as an address: '{l}'")
            for r in regs(dst):
                pending[r] = i
            continue
        if in_asm and "s_waitcnt" in l and "lgkmcnt(0)" in l:
            pending.clear()
            continue
        if pending and not in_asm:
            hit = regs(l) & set(pending)
            if hit:
                errs.append(f"instruction touches registers still being loaded by an asm ds_read {sorted(hit)[:4]}: '{l}'")
    return errs


def main():
    bad = 0
    n = 0
    for name, lines in kernels(isa()):
        n += 1
        errs = check(name, lines)
        print(("FAIL " if errs else "ok   ") + name)
        for e in errs[:6]:
            print("     ", e)
        bad += bool(errs)
    if n == 0:
        print("no gemm8 kernels found in the ISA")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

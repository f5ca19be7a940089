"""Kernel sequence of single adapter calls in a rocprofv3 kernel trace (rocpd sqlite): for stream `sid`, the kernels between the first
kernel of a forward call (the memset of `saved`'s accumulators) / backward call (zero2_k) and the next such marker, for the call
indices given (in trace order within the LAST step).      usage: python tools/call_sequence.py results.db sid idx [idx ...]"""
import re, sqlite3, sys

def short(n):
    n = re.sub(r"^void ", "", n).replace("dgsct::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n)[:64]

def main(path, sid, idxs):
    c = sqlite3.connect(path)
    rows = c.execute("select start, end, stream_id, name, grid_x, workgroup_x from kernels order by start").fetchall()
    adam = [r for r in rows if "multi_tensor_apply" in r[3]]
    ends = []
    for r in adam:
        if not ends or r[0] - ends[-1] > 20e6: ends.append(r[1])
        else: ends[-1] = r[1]
    lo, hi = ends[-2] - (ends[-1] - ends[-2]) * 0.6, ends[-1]          # (the forward of the last step starts before the previous Adam group's end is seen on this stream)
    ks = [r for r in rows if r[2] == sid and lo <= r[0] <= hi]
    calls = []
    for r in ks:
        first_fwd = "fillBufferAligned" in r[3] or "cvt_multi_k" in r[3]
        if (first_fwd and not (calls and len(calls[-1]) <= 2 and "cvt_multi" in short(calls[-1][0][3]))) or "zero2_k" in r[3]: calls.append([])
        if calls: calls[-1].append(r)
    print(f"stream {sid}: {len(calls)} calls in the last step ({(hi-lo)/1e6:.1f} ms under the tracer); {len(ks)} kernels; first: {[short(r[3]) for r in ks[:6]]}")
    for i, cl in enumerate(calls):
        span = (cl[-1][1] - cl[0][0]) / 1e3; busy = sum(r[1] - r[0] for r in cl) / 1e3
        print(f"  call {i:3d}: {len(cl):3d} kernels  span {span:8.1f} us  kernel time {busy:8.1f} us  first {short(cl[0][3])}")
    for i in idxs:
        cl = calls[i]
        print(f"--- call {i}")
        prev = None
        for r in cl:
            gap = (r[0] - prev) / 1e3 if prev is not None else 0.0
            print(f"   +{gap:7.1f} gap {(r[1]-r[0])/1e3:8.1f} us  grid {r[4]//max(r[5],1):6d} x {r[5]:4d}  {short(r[3])}")
            prev = r[1]

if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), [int(v) for v in sys.argv[3:]])

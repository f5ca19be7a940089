"""What does a cross-stream dependency cost?  Ping-pong of tiny kernels between two streams through event waits, against the same
kernels chained on ONE stream.  Streams: torch's default + pool streams, and the library's priority streams (the ones AdapterStack uses)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgsct_amd
from dgsct_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.default_lib()
x = torch.zeros(256, device=dev)


def chain(sa, sb, hops=400):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sa):
        torch.cuda._sleep(60_000_000)          # ~25-30 ms: the whole chain is enqueued before the GPU starts on it
        e0.record()
    cur = sa
    for i in range(hops):
        nxt = sb if cur is sa else sa
        if nxt is not cur:
            nxt.wait_stream(cur)
        with torch.cuda.stream(nxt):
            x.add_(1.0)
        cur = nxt
    with torch.cuda.stream(cur):
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / hops, (time.perf_counter() - t0) * 1e6 / hops


main = torch.cuda.current_stream(dev)
pool = torch.cuda.Stream(dev)
side = ops.side_stream(lib, dev)
aux = ops.priority_stream(lib, dev, +1)
for name, a, b in (("one stream (no sync)", main, main), ("default <-> torch pool stream", main, pool), ("default <-> side (high priority)", main, side),
                   ("default <-> aux (low priority)", main, aux), ("side <-> aux", side, aux)):
    chain(a, b, 50)
    g, h = chain(a, b)
    print(f"{name:36s}: {g:6.1f} us per hop on the GPU timeline, {h:6.1f} us per hop of host time")

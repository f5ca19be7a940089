"""Which HIP streams share a hardware queue?  Two streams on one queue serialise: run a spin kernel on each of a pair
and compare the pair's wall time with a single spin.  Prints the priority range and a collision matrix."""
import ctypes as C
import time

import torch

hip = C.CDLL("libamdhip64.so")


def mk(prio):
    s = C.c_void_p()
    r = hip.hipStreamCreateWithPriority(C.byref(s), 1, prio)       # hipStreamNonBlocking
    assert r == 0, r
    return torch.cuda.ExternalStream(s.value)


def pair_time(a, b, cycles):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
    if b is not None:
        with torch.cuda.stream(b):
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    torch.cuda.init()
    lo, hi = C.c_int(), C.c_int()
    hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
    print("priority range: least", lo.value, "greatest", hi.value)
    x = torch.zeros(1, device="cuda")
    cycles = 20_000_000
    null = torch.cuda.default_stream()
    names, streams = ["null"], [null]
    for p in (0, 0, 0, 0, 0, 0):
        names.append(f"n{len(names)}"); streams.append(mk(0))
    for i in range(5):
        names.append(f"L{i}"); streams.append(mk(lo.value))
    for i in range(5):
        names.append(f"H{i}"); streams.append(mk(hi.value))
    tp = torch.cuda.Stream(); names.append("torchpool"); streams.append(tp)
    th = torch.cuda.Stream(priority=-1); names.append("torchhigh"); streams.append(th)
    single = min(pair_time(s, None, cycles) for s in streams[:3])
    print(f"single spin {single * 1e3:.2f} ms")
    print("      " + " ".join(f"{n[:5]:>5}" for n in names))
    for i, a in enumerate(streams):
        row = []
        for j, b in enumerate(streams):
            if j <= i:
                row.append("    .")
                continue
            t = pair_time(a, b, cycles)
            row.append("    X" if t > 1.6 * single else "    -")
        print(f"{names[i][:5]:>5} " + " ".join(row))


if __name__ == "__main__":
    main()

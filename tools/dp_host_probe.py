"""Where does the extra host time of the data-parallel step come from?  Times fwd+bwd host enqueue of the bench stack
in the states bench.py goes through: flat parameters, process group, parameter broadcast, gradient all-reduce."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dgsct_amd  # noqa: E402
from dgsct_amd import GradAllReducer  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
    if os.environ.get("PROBE_MAIN_LOW", "0") == "1":
        from dgsct_amd import ops
        torch.cuda.set_stream(ops.priority_stream(dgsct_amd.default_lib(), dev, ops.COMPUTE_PRIORITY_CLASS))
    import torch.distributed as dist
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    early = os.environ.get("PROBE_EARLY_INIT", "0") == "1"
    if early:
        dgsct_amd.init_process_group(dev)
        t = torch.ones(1 << 20, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
    stages, stack = bench.build_stack("swinv2_base", torch.bfloat16, dev, concurrent=True)
    stack.train()
    feats, cots, mcots = bench.make_inputs(stages, 160, torch.bfloat16, dev, seed=1)
    params = [p for p in stack.parameters()]
    red = [None]

    def fwd_bwd():
        outs, maps = stack(feats)
        torch.autograd.backward([t for pr in outs for t in pr] + [maps[0], maps[1]], [g for pr in cots for g in pr] + list(mcots))
        if red[0] is not None:
            red[0].finish()
        for p in params:
            p.grad = None

    def measure(tag, prof=False):
        for _ in range(2):
            fwd_bwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); h = 0.0
        for _ in range(5):
            a = time.perf_counter(); fwd_bwd(); h += time.perf_counter() - a
        torch.cuda.synchronize()
        print(f"{tag:34s} host {h / 5 * 1e3:7.2f} ms  wall {(time.perf_counter() - t0) / 5 * 1e3:7.2f} ms", flush=True)
        if prof:
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(3):
                fwd_bwd()
            pr.disable()
            torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats("tottime").print_stats(12)

    measure("flat params, pg early" if early else "flat params, no process group")
    if not early:
        dgsct_amd.init_process_group(dev)
    measure("after init_process_group")
    for p in stack.parameters():
        dist.broadcast(p.data, 0)
    torch.cuda.synchronize()
    measure("after broadcast of params")
    red[0] = GradAllReducer(GradAllReducer.stage_buckets(stack), overlap=False, force=True)
    measure("with all-reduce (no overlap)", prof=True)
    red[0] = GradAllReducer(GradAllReducer.stage_buckets(stack), overlap=True, force=True)
    measure("with all-reduce (overlap)")
    for h in red[0]._hooks:
        h.remove()
    red[0] = None
    measure("reducer removed")
    dist.destroy_process_group()
    measure("after destroy")


if __name__ == "__main__":
    main()

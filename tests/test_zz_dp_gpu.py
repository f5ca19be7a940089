"""GPU: the data-parallel gradient exchange over RCCL ("nccl" backend) on ONE device -- a 1-rank process group runs the
same code path as N ranks (grouped in-place ncclAvg all-reduce of the flat gradient buffers, world-size scaling), so the
collective plumbing is exercised on the real backend; the N = 2 arithmetic is covered by the gloo tests on CPU.
(Named zz: runs last, RCCL initialisation changes the stream -> hardware-queue placement of the process.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from helpers import grad_close_fp32, load_golden

import dgsct_amd  # noqa: F401
from dgsct_amd import AdapterStack, GradAllReducer, init_process_group
from dgsct_amd.stack import default_opt

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("overlap", [False, True])
def test_rccl_single_rank_allreduce_keeps_the_gradients(overlap):
    if dist.is_initialized():
        dist.destroy_process_group()
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    init_process_group(DEV)
    try:
        fx = load_golden("stack_2stage")
        st = AdapterStack(fx["stages"], opt=default_opt(num_tokens=4), concurrent=True)
        st.load_state_dict(fx["state0"])
        st = st.to(DEV).flatten_parameters().train()
        red = GradAllReducer(GradAllReducer.stage_buckets(st), overlap=overlap, force=True)
        assert red.active and red.world == 1
        for rep in range(2):
            for p in st.parameters():
                p.grad = None
            if rep:
                st.load_state_dict(fx["state0"])
            feats = [(a.to(DEV), b.to(DEV)) for a, b in fx["feats"]]
            outs, maps = st(feats)
            torch.autograd.backward([t for pr in outs for t in pr] + [maps[0], maps[1]],
                                    [g.to(DEV) for pr in fx["cots"] for g in pr] + [fx["mcots"][0].to(DEV), fx["mcots"][1].to(DEV)])
            red.finish()
            torch.cuda.synchronize()
            n = 0
            for name, m in st.named_modules():
                if hasattr(m, "flat_param"):
                    assert m.flat_param.grad._base is None           # reduced in place, no staging copy
                    for pn, (off, cnt, shape) in m._flat_layout.items():
                        ref = fx["grads"].get(name + "." + pn)
                        if ref is not None:
                            # (the no-floor fp32 metric of the parity tests; the stack fixture's gradients come from the un-pinned
                            #  reference, so a ReLU flip may move a row of a weight gradient: grad_close_fp32 allows two)
                            assert grad_close_fp32(m.flat_param.grad[off:off + cnt].view(shape), ref, tol=2e-3, name=pn), name + "." + pn
                            n += 1
            assert n == len(fx["grads"])
    finally:
        dist.destroy_process_group()


def test_bench_contract_last_stdout_line_is_the_json_line():
    """The driver reads ONE JSON line from rank 0.  RCCL prints a version banner through C stdio, which on a pipe used to come
    out at exit, after the line; bench.py drains it first.  Runs the data-parallel path at one rank (--force-dp)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-roofline", "--force-dp", "--batch", "2"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
